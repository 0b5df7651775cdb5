# -*- coding: utf-8 -*-
'''
Host-side plan of the multicolour ILU(0) preconditioner (K11): graph colouring
of a scalar CSR pattern, the permuted (colour-major) CSR the factor lives in,
its split into strictly-lower / strictly-upper streams with CSR-stream row
blocks per colour, and the map from factor entries back to the operator's value
plane.

Why multicolour: a triangular solve on a 2-D mesh matrix in natural ordering has
only O(sqrt(N)) rows per dependency level (hard part 3).
Ordering the rows by colour (independent sets) makes every colour one fully
parallel launch: L holds the couplings to lower colours, U those to higher
colours, and each sweep streams its triangle exactly once (12 B per entry) with
the same LDS-tiled kernel structure as the SpMV.  The price is a somewhat weaker
factorisation than natural-order ILU(0).  Replaces the role of the sparse LU in
the reference's Newton and heat solves (pressure_correction.py:224-254,
heat.py:117-121) as the north star prescribes (BiCGStab + ILU(0)).

Setup only (numpy, once per pattern); factorisation and sweeps are HIP kernels
(flow_ilu0_factor / flow_ilu0_solve in include/flow_hip.h).
'''
import ctypes

import numpy

from .space import csr_stream_rowblocks
from .. import _hip
from .. import device


def colour_graph(rowptr, cols, seed=0):
    '''Parallel greedy colouring: Jones-Plassmann rounds (vertices whose random
    priority beats all uncoloured neighbours form an independent set) where
    every selected vertex takes the SMALLEST colour its coloured neighbours do
    not use.  Returns (colour per vertex, number of colours <= 63).'''
    rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
    cols = numpy.asarray(cols, dtype=numpy.int64)
    n = len(rowptr) - 1
    rows = numpy.repeat(numpy.arange(n, dtype=numpy.int64), numpy.diff(rowptr))
    offdiag = cols != rows
    prio = numpy.random.RandomState(seed).permutation(n).astype(numpy.int64)
    colour = numpy.full(n, -1, dtype=numpy.int64)
    starts = rowptr[:-1]
    one = numpy.uint64(1)
    while True:
        active = colour < 0
        if not active.any():
            break
        pr = numpy.where(active, prio, -1)
        nb = numpy.where(offdiag, pr[cols], -1)
        mx = numpy.maximum.reduceat(nb, starts)
        sel = active & (prio > mx)
        assert sel.any()
        # colours used by the neighbours, as a bit mask per vertex
        cn = colour[cols]
        bits = numpy.where(
            (cn >= 0) & offdiag,
            numpy.left_shift(one, numpy.maximum(cn, 0).astype(numpy.uint64)),
            numpy.uint64(0)
            )
        used = numpy.bitwise_or.reduceat(bits, starts)[sel]
        # lowest zero bit of `used`
        low = (~used) & (used + one)
        c = numpy.round(numpy.log2(low.astype(numpy.float64))).astype(numpy.int64)
        assert (c < 63).all(), 'more than 63 colours'
        colour[sel] = c
    return colour.astype(numpy.int32), int(colour.max()) + 1


def _pad(a, n=2):
    return numpy.concatenate([a, numpy.zeros(n, dtype=a.dtype)])


class IluPlan(object):
    '''Colour-major permuted pattern of a scalar layout and its L / U streams.'''

    def __init__(self, layout):
        rowptr = layout.pattern('rowptr').astype(numpy.int64)
        cols = layout.pattern('cols').astype(numpy.int64)
        n = layout.N
        nnz = layout.nnz
        colour, nc = colour_graph(rowptr, cols)
        old_of_new = numpy.argsort(colour, kind='stable')
        new_of_old = numpy.empty(n, dtype=numpy.int64)
        new_of_old[old_of_new] = numpy.arange(n)
        colour_ptr = numpy.zeros(nc + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(colour, minlength=nc), out=colour_ptr[1:])
        rows = numpy.repeat(numpy.arange(n, dtype=numpy.int64),
                            numpy.diff(rowptr))
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

        def stream(mask):
            pos = numpy.nonzero(mask)[0]
            rp = numpy.zeros(n + 1, dtype=numpy.int64)
            numpy.cumsum(numpy.bincount(p_rows[pos], minlength=n), out=rp[1:])
            blocks = [numpy.zeros(1, dtype=numpy.int64)]
            bptr = [0]
            for c in range(nc):
                a, b = int(colour_ptr[c]), int(colour_ptr[c + 1])
                rb = csr_stream_rowblocks(rp[a:b + 1] - rp[a]) + a
                # consecutive colours share the boundary row
                blocks.append(rb[1:])
                bptr.append(bptr[-1] + len(rb) - 1)
            return (rp.astype(numpy.int32), p_cols[pos].astype(numpy.int32),
                    pos.astype(numpy.int32),
                    numpy.concatenate(blocks).astype(numpy.int32),
                    numpy.ascontiguousarray(bptr, dtype=numpy.int32))

        l_rp, l_cols, l_pos, l_rb, l_rbptr = stream(is_l)
        u_rp, u_cols, u_pos, u_rb, u_rbptr = stream(is_u)
        self.n = n
        self.nnz = nnz
        self.nnz_l = len(l_cols)
        self.nnz_u = len(u_cols)
        self.ncolours = nc
        self.colour = colour
        self.colour_ptr = numpy.ascontiguousarray(colour_ptr, dtype=numpy.int32)
        self.l_rbptr = l_rbptr
        self.u_rbptr = u_rbptr
        self.host = {
            'rowptr': p_rowptr.astype(numpy.int32),
            'cols': p_cols.astype(numpy.int32),
            'diag': diag.astype(numpy.int32),
            'src_pos': order.astype(numpy.int32),
            'old_of_new': old_of_new.astype(numpy.int32),
            'l_rowptr': l_rp, 'l_cols': _pad(l_cols), 'l_pos': l_pos,
            'l_rowblocks': l_rb,
            'u_rowptr': u_rp, 'u_cols': _pad(u_cols), 'u_pos': u_pos,
            'u_rowblocks': u_rb,
            }
        self._dev = {k: device.to_device(v) for k, v in self.host.items()}
        d = self._dev
        # factor buffer per block: [combined nnz][L stream][U stream][1/diag],
        # every segment starting 16-byte aligned
        ev = lambda m: m + (m & 1)
        self.off_l = ev(nnz)
        self.off_u = self.off_l + ev(self.nnz_l + 2)
        self.off_d = self.off_u + ev(self.nnz_u + 2)
        self.lu_size = self.off_d + ev(n)
        self.struct = _hip.IluPlanS(
            n, nnz, nc, self.nnz_l, self.nnz_u,
            self.off_l, self.off_u, self.off_d, self.lu_size,
            int(numpy.diff(rowptr).max()),
            self.colour_ptr.ctypes.data_as(ctypes.c_void_p),
            self.l_rbptr.ctypes.data_as(ctypes.c_void_p),
            self.u_rbptr.ctypes.data_as(ctypes.c_void_p),
            _hip.i32(d['rowptr'], n + 1), _hip.i32(d['cols'], nnz),
            _hip.i32(d['diag'], n), _hip.i32(d['src_pos'], nnz),
            _hip.i32(d['old_of_new'], n),
            _hip.i32(d['l_rowptr'], n + 1), _hip.i32(d['l_cols'], self.nnz_l),
            _hip.i32(d['l_pos'], self.nnz_l), _hip.i32(d['l_rowblocks']),
            _hip.i32(d['u_rowptr'], n + 1), _hip.i32(d['u_cols'], self.nnz_u),
            _hip.i32(d['u_pos'], self.nnz_u), _hip.i32(d['u_rowblocks']),
            )


def plan_for(layout):
    if 'ilu_plan' not in layout._dev:
        layout._dev['ilu_plan'] = IluPlan(layout)
    return layout._dev['ilu_plan']


class Ilu0(object):
    '''ILU(0) factors of the diagonal blocks of a Matrix: one factor for a
    scalar operator, two (the (0,0) and (1,1) blocks) for block operators --
    the couplings between the velocity components are left to the Krylov
    method.  `refactor(A)` re-uses the buffers.'''

    def __init__(self, A):
        self.plan = plan_for(A.layout)
        self.planes = {0: [0], 1: [0, 1], 2: [0, 3]}[A.kind]
        self.lu = device.zeros(len(self.planes) * self.plan.lu_size)
        self.struct = _hip.IluS(
            ctypes.pointer(self.plan.struct), len(self.planes),
            _hip.f64(self.lu, len(self.planes) * self.plan.lu_size),
            )
        self.refactor(A)

    def refactor(self, A):
        lib = _hip.lib()
        assert plan_for(A.layout) is self.plan
        nb = len(self.planes)
        _hip.check(lib.flow_ilu0_factor(
            ctypes.byref(self.plan.struct), nb,
            _hip.f64(A.plane(self.planes[0]), self.plan.nnz),
            _hip.f64(A.plane(self.planes[-1]), self.plan.nnz),
            _hip.f64(self.lu, nb * self.plan.lu_size), _hip.stream()
            ))
        return self

    def factor_values(self, k=0):
        '''Combined L\\U values of block k in the permuted CSR (tests).'''
        size = self.plan.lu_size
        return self.lu[k * size:k * size + self.plan.nnz].cpu().numpy()

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
