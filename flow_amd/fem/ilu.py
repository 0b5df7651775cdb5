# -*- coding: utf-8 -*-
'''
Host-side plan of the multicolour ILU(0) preconditioner (K11): graph colouring
of a scalar CSR pattern, the permuted (colour-major) CSR the factor lives in,
its split into strictly-lower / strictly-upper sweep streams (sliced ELL, one
wavefront per slice of 64 rows), and the map from factor entries back to the
operator's value plane.

Why multicolour: a triangular solve on a 2-D mesh matrix in natural ordering has
only O(sqrt(N)) rows per dependency level (hard part 3).
Ordering the rows by colour (independent sets) makes every colour one fully
parallel launch: L holds the couplings to lower colours, U those to higher
colours, and each sweep streams its triangle exactly once (12 B per entry, fully
coalesced, no LDS and no barrier in the kernel).  The price is a somewhat weaker
factorisation than natural-order ILU(0).  Replaces the role of the sparse LU in
the reference's Newton and heat solves (pressure_correction.py:224-254,
heat.py:117-121) as the north star prescribes (BiCGStab + ILU(0)).

Setup only (numpy, once per pattern); factorisation and sweeps are HIP kernels
(flow_ilu0_factor / flow_ilu0_solve in include/flow_hip.h).
'''
import ctypes
import os

import numpy

from .. import _hip
from .. import device


# passes of iterated greedy behind the first-fit colouring (0: none).  OFF:
# measured on the ~1 M-DoF channels (round 6), 9 -> 7 / 8 colours cost more
# GMRES applications (the factorisation in the recoloured order is weaker:
# 15.2 -> 17.9 structured, 25.2 -> 26.4 graded) than the saved launches buy
# (4.60 -> 4.86 / 14.3 -> 14.5 ms per step).
COLOUR_ROUNDS = int(os.environ.get('FLOW_AMD_COLOUR_ROUNDS', '0'))


def colour_graph(rowptr, cols, rounds=None):
    '''First-fit greedy colouring in mesh order (flow_color_greedy_host, a host
    routine of the library -- no GPU needed), then optionally `rounds` passes
    of Culberson's iterated greedy (flow_color_iterate_host: first fit again,
    class by class -- never more colours, usually fewer: P2 triangulations 9
    -> 7, P1 6 -> 5; every colour less is two dependent launches less per
    application -- and a weaker factorisation: see COLOUR_ROUNDS).  Within a
    class the rows stay in mesh order, which is what keeps the gathers of the
    sweeps inside a few cache lines per wavefront.
    Returns (colour per vertex, number of colours <= 63).'''
    rowptr = numpy.ascontiguousarray(rowptr, dtype=numpy.int32)
    cols = numpy.ascontiguousarray(cols, dtype=numpy.int32)
    n = len(rowptr) - 1
    colour = numpy.empty(n, dtype=numpy.int32)
    nc = ctypes.c_int(0)
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib = _hip.load_library()
    _hip.check(lib.flow_color_greedy_host(
        n, as_p(rowptr), as_p(cols), as_p(colour), ctypes.byref(nc)
        ))
    rounds = COLOUR_ROUNDS if rounds is None else int(rounds)
    if rounds > 0:
        _hip.check(lib.flow_color_iterate_host(
            n, as_p(rowptr), as_p(cols), as_p(colour), ctypes.byref(nc), rounds))
    return colour, nc.value


def _pad(a, n=2):
    return numpy.concatenate([a, numpy.zeros(n, dtype=a.dtype)])


SLICE = 64     # rows per slice of the sliced-ELL sweep streams = one wavefront


class IluPlan(object):
    '''Colour-major permuted pattern of a scalar layout and its L / U sweep
    streams in sliced-ELL form: within a colour the rows are ordered by their
    number of lower / higher coloured neighbours, cut into slices of 64 rows
    (one wavefront), and every slice is stored column-major and padded to its
    longest row -- lane i of the wavefront reads entry k of ITS row at
    slice_offset + 64 k + i, fully coalesced, and sums its row in registers.
    Pad entries carry value 0 and column 0 (a finished row of colour 0).'''

    def __init__(self, layout, rows=None):
        '''rows = (r0, r1): the plan of the diagonal block of those rows (the
        block-Jacobi ILU(0) of a strip, flow_amd/parallel.py): local numbering
        row - r0, couplings to columns outside the block dropped; the factor
        still reads its entries from the FULL value planes of the layout's
        pattern (src_pos points there).'''
        rowptr = layout.pattern('rowptr').astype(numpy.int64)
        cols = layout.pattern('cols').astype(numpy.int64)
        n = layout.N
        nnz = layout.nnz
        src_index = None
        if rows is not None:
            r0, r1 = rows
            k0, k1 = int(rowptr[r0]), int(rowptr[r1])
            seg = cols[k0:k1]
            keep = (seg >= r0) & (seg < r1)
            row_of = numpy.repeat(numpy.arange(r0, r1), numpy.diff(rowptr[r0:r1 + 1]))
            src_index = (numpy.nonzero(keep)[0] + k0).astype(numpy.int64)
            cols = seg[keep] - r0
            n = r1 - r0
            rowptr = numpy.zeros(n + 1, dtype=numpy.int64)
            numpy.cumsum(numpy.bincount(row_of[keep] - r0, minlength=n),
                         out=rowptr[1:])
            nnz = len(cols)
        colour, nc = colour_graph(rowptr, cols)
        rows = numpy.repeat(numpy.arange(n, dtype=numpy.int64),
                            numpy.diff(rowptr))
        lower = (colour[cols] < colour[rows]).astype(numpy.int64)
        upper = (colour[cols] > colour[rows]).astype(numpy.int64)
        starts = rowptr[:-1]
        l_len = numpy.add.reduceat(lower, starts)
        u_len = numpy.add.reduceat(upper, starts)
        # colour-major; inside a colour by (|L row|, |U row|), ties in mesh order
        old_of_new = numpy.lexsort((numpy.arange(n), u_len, l_len, colour))
        new_of_old = numpy.empty(n, dtype=numpy.int64)
        new_of_old[old_of_new] = numpy.arange(n)
        colour_ptr = numpy.zeros(nc + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(colour, minlength=nc), out=colour_ptr[1:])
        key = new_of_old[rows] * n + new_of_old[cols]
        order = numpy.argsort(key, kind='stable')
        skey = key[order]
        p_rows = skey // n
        p_cols = skey % n
        p_rowptr = numpy.zeros(n + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(p_rows, minlength=n), out=p_rowptr[1:])
        is_l = p_cols < p_rows
        is_u = p_cols > p_rows
        diag = numpy.nonzero(p_cols == p_rows)[0]
        assert len(diag) == n

        # slices: every colour starts a new slice
        p_colour = colour[old_of_new].astype(numpy.int64)
        nsl_c = (numpy.diff(colour_ptr) + SLICE - 1) // SLICE
        slptr = numpy.zeros(nc + 1, dtype=numpy.int64)
        numpy.cumsum(nsl_c, out=slptr[1:])
        nsl = int(slptr[-1])
        in_colour = numpy.arange(n) - colour_ptr[p_colour]
        slice_of_row = slptr[p_colour] + in_colour // SLICE
        lane_of_row = in_colour % SLICE
        sl_row = numpy.zeros(nsl, dtype=numpy.int64)
        sl_row[slice_of_row[lane_of_row == 0]] = \
            numpy.nonzero(lane_of_row == 0)[0]

        def stream(mask):
            pos = numpy.nonzero(mask)[0]
            r_of = p_rows[pos]
            length = numpy.bincount(r_of, minlength=n)
            rp = numpy.zeros(n + 1, dtype=numpy.int64)
            numpy.cumsum(length, out=rp[1:])
            width = numpy.zeros(nsl, dtype=numpy.int64)
            numpy.maximum.at(width, slice_of_row, length)
            sl_off = numpy.zeros(nsl + 1, dtype=numpy.int64)
            numpy.cumsum(width * SLICE, out=sl_off[1:])
            total = int(sl_off[-1])
            assert total < 2**31
            k = numpy.arange(len(pos)) - rp[r_of]
            dest = sl_off[slice_of_row[r_of]] + k * SLICE + lane_of_row[r_of]
            s_cols = numpy.zeros(total, dtype=numpy.int32)
            s_pos = numpy.full(total, -1, dtype=numpy.int32)
            s_cols[dest] = p_cols[pos]
            s_pos[dest] = pos
            return sl_off.astype(numpy.int32), s_cols, s_pos, total

        l_off, l_cols, l_pos, total_l = stream(is_l)
        u_off, u_cols, u_pos, total_u = stream(is_u)
        self.n = n
        self.nnz = nnz
        self.nnz_l = total_l
        self.nnz_u = total_u
        self.fill_l = float(is_l.sum()) / max(total_l, 1)
        self.fill_u = float(is_u.sum()) / max(total_u, 1)
        self.ncolours = nc
        self.colour = colour
        self.colour_ptr = numpy.ascontiguousarray(colour_ptr, dtype=numpy.int32)
        self.slptr = numpy.ascontiguousarray(slptr, dtype=numpy.int32)
        self.host = {
            'rowptr': p_rowptr.astype(numpy.int32),
            'cols': p_cols.astype(numpy.int32),
            'diag': diag.astype(numpy.int32),
            'src_pos': (order if src_index is None
                        else src_index[order]).astype(numpy.int32),
            'old_of_new': old_of_new.astype(numpy.int32),
            'new_of_old': new_of_old.astype(numpy.int32),
            'sl_row': sl_row.astype(numpy.int32),
            'l_sl_off': l_off, 'l_cols': _pad(l_cols), 'l_pos': _pad(l_pos),
            'u_sl_off': u_off, 'u_cols': _pad(u_cols), 'u_pos': _pad(u_pos),
            }
        self._dev = {k: device.to_device(v) for k, v in self.host.items()}
        d = self._dev
        # factor buffer per block: [combined nnz][L stream][U stream][1/diag],
        # every segment starting 16-byte aligned
        ev = lambda m: m + (m & 1)
        self.off_l = ev(nnz)
        self.off_u = self.off_l + ev(total_l + 2)
        self.off_d = self.off_u + ev(total_u + 2)
        self.lu_size = self.off_d + ev(n)
        self.struct = _hip.IluPlanS(
            n, nnz, nc, total_l, total_u,
            self.off_l, self.off_u, self.off_d, self.lu_size,
            int(numpy.diff(rowptr).max()), nsl,
            self.colour_ptr.ctypes.data_as(ctypes.c_void_p),
            self.slptr.ctypes.data_as(ctypes.c_void_p),
            _hip.i32(d['rowptr'], n + 1), _hip.i32(d['cols'], nnz),
            _hip.i32(d['diag'], n), _hip.i32(d['src_pos'], nnz),
            _hip.i32(d['old_of_new'], n), _hip.i32(d['new_of_old'], n),
            _hip.i32(d['sl_row'], nsl),
            _hip.i32(d['l_sl_off'], nsl + 1), _hip.i32(d['l_cols'], total_l),
            _hip.i32(d['l_pos'], total_l),
            _hip.i32(d['u_sl_off'], nsl + 1), _hip.i32(d['u_cols'], total_u),
            _hip.i32(d['u_pos'], total_u),
            )


def plan_for(layout):
    if 'ilu_plan' not in layout._dev:
        layout._dev['ilu_plan'] = IluPlan(layout)
    return layout._dev['ilu_plan']


class Ilu0(object):
    '''ILU(0) factors of the diagonal blocks of a Matrix: one factor for a
    scalar operator, two (the (0,0) and (1,1) blocks) for block operators --
    the couplings between the velocity components are left to the Krylov
    method.  `refactor(A)` re-uses the buffers.  `packed`: the sweeps read the
    factors rounded to fp32, the blocks interleaved (half the bytes per
    application, fp64 arithmetic; include/flow_hip.h: flow_ilu.packed);
    `single_vector` (with packed): the sweep vector in fp32 too -- for the
    flexible GMRES only (flow_ilu.single_vector).'''

    def __init__(self, A, plan=None, packed=False, single_vector=False):
        import torch
        self.plan = plan if plan is not None else plan_for(A.layout)
        self.planes = {0: [0], 1: [0, 1], 2: [0, 3]}[A.kind]
        nb = len(self.planes)
        self.lu = device.zeros(nb * self.plan.lu_size)
        self.struct = _hip.IluS(
            ctypes.pointer(self.plan.struct), nb,
            _hip.f64(self.lu, nb * self.plan.lu_size), None,
            int(bool(single_vector and packed)),
            )
        self.packed = None
        if packed:
            self.packed = torch.zeros(
                (self.plan.nnz_l + self.plan.nnz_u) * nb + 4,
                dtype=torch.float32, device=self.lu.device)
            assert self.packed.data_ptr() % 16 == 0
        self.refactor(A)

    def refactor(self, A):
        lib = _hip.lib()
        nb = len(self.planes)
        _hip.check(lib.flow_ilu0_factor(
            ctypes.byref(self.plan.struct), nb,
            _hip.f64(A.plane(self.planes[0]), self.plan.nnz),
            _hip.f64(A.plane(self.planes[-1]), self.plan.nnz),
            _hip.f64(self.lu, nb * self.plan.lu_size), _hip.stream()
            ))
        if self.packed is not None:
            single = self.struct.single_vector
            self.struct.packed, self.struct.single_vector = None, 0
            _hip.check(lib.flow_ilu0_pack(
                ctypes.byref(self.struct), self.packed.data_ptr(),
                _hip.stream()
                ))
            self.struct.packed = self.packed.data_ptr()
            self.struct.single_vector = single
        return self

    def factor_values(self, k=0):
        '''Combined L\\U values of block k in the permuted CSR (tests).'''
        size = self.plan.lu_size
        return device.to_host(
            self.lu[k * size:k * size + self.plan.nnz]).numpy()

    def solve(self, r, z):
        '''z = blockdiag(LU)^-1 r (testing / direct use).'''
        lib = _hip.lib()
        n = self.plan.n
        nb = len(self.planes)
        work = device.empty(nb * n)
        _hip.check(lib.flow_ilu0_solve(
            ctypes.byref(self.struct), _hip.f64(r, nb * n), _hip.f64(z, nb * n),
            _hip.f64(work, nb * n), _hip.stream()
            ))
        return z
