# -*- coding: utf-8 -*-
'''
Host-side plan of the multicolour ILU(0) preconditioner (K11): graph colouring
of a scalar CSR pattern, the permuted (colour-major) CSR the factor lives in, and
the map from its entries back to the operator's value plane.

Why multicolour: a triangular solve on a 2-D mesh matrix in natural ordering has
only O(sqrt(N)) rows per dependency level (hard part 3).
Ordering the rows by colour (independent sets) makes every colour one fully
parallel kernel launch; L holds the couplings to lower colours, U those to
higher colours.  The price is a somewhat weaker factorisation than natural-order
ILU(0).  Replaces the role of the sparse LU in the reference's Newton and heat
solves (pressure_correction.py:224-254, heat.py:117-121) as the north star
prescribes (BiCGStab + ILU(0)).

Setup only (numpy, once per pattern); factorisation and solves are HIP kernels
(flow_ilu0_factor / flow_ilu0_solve in include/flow_hip.h).
'''
import ctypes

import numpy

from .. import _hip
from .. import device


def colour_graph(rowptr, cols, seed=0):
    '''Jones-Plassmann colouring with random priorities: every round colours the
    vertices whose priority beats all uncoloured neighbours (an independent
    set).  Returns (colour per vertex, number of colours).'''
    rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
    cols = numpy.asarray(cols, dtype=numpy.int64)
    n = len(rowptr) - 1
    rows = numpy.repeat(numpy.arange(n, dtype=numpy.int64), numpy.diff(rowptr))
    offdiag = cols != rows
    prio = numpy.random.RandomState(seed).permutation(n).astype(numpy.int64)
    colour = numpy.full(n, -1, dtype=numpy.int32)
    starts = rowptr[:-1]
    c = 0
    while True:
        active = colour < 0
        if not active.any():
            break
        pr = numpy.where(active, prio, -1)
        nb = numpy.where(offdiag, pr[cols], -1)
        mx = numpy.maximum.reduceat(nb, starts)
        sel = active & (prio > mx)
        assert sel.any()
        colour[sel] = c
        c += 1
    return colour, c


class IluPlan(object):
    '''Colour-major permuted pattern of a scalar layout.'''

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
        p_cols = (skey % n).astype(numpy.int32)
        p_rowptr = numpy.zeros(n + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(p_rows, minlength=n), out=p_rowptr[1:])
        diag = numpy.nonzero(p_cols == p_rows)[0]
        assert len(diag) == n
        self.n = n
        self.nnz = nnz
        self.ncolours = nc
        self.colour = colour
        self.colour_ptr = numpy.ascontiguousarray(colour_ptr, dtype=numpy.int32)
        self.host = {
            'rowptr': p_rowptr.astype(numpy.int32), 'cols': p_cols,
            'diag': diag.astype(numpy.int32),
            'src_pos': order.astype(numpy.int32),
            'old_of_new': old_of_new.astype(numpy.int32),
            }
        self._dev = {k: device.to_device(v) for k, v in self.host.items()}
        d = self._dev
        self.struct = _hip.IluPlanS(
            n, nnz, nc,
            self.colour_ptr.ctypes.data_as(ctypes.c_void_p),
            _hip.i32(d['rowptr'], n + 1), _hip.i32(d['cols'], nnz),
            _hip.i32(d['diag'], n), _hip.i32(d['src_pos'], nnz),
            _hip.i32(d['old_of_new'], n),
            )


def plan_for(layout):
    if 'ilu_plan' not in layout._dev:
        layout._dev['ilu_plan'] = IluPlan(layout)
    return layout._dev['ilu_plan']


class Ilu0(object):
    '''ILU(0) factors of the diagonal blocks of a Matrix: one factor for a
    scalar operator, two (the (0,0) and (1,1) blocks) for block operators --
    the couplings between the velocity components are left to the Krylov
    method.'''

    def __init__(self, A):
        lib = _hip.lib()
        self.plan = plan_for(A.layout)
        self.A = A
        planes = {0: [0], 1: [0, 1], 2: [0, 3]}[A.kind]
        nnz = A.layout.nnz
        self.lu = device.empty(len(planes) * nnz)
        for k, p in enumerate(planes):
            _hip.check(lib.flow_ilu0_factor(
                ctypes.byref(self.plan.struct), _hip.f64(A.plane(p), nnz),
                _hip.f64(self.lu[k * nnz:(k + 1) * nnz], nnz), _hip.stream()
                ))
        self.struct = _hip.IluS(
            ctypes.pointer(self.plan.struct), len(planes),
            _hip.f64(self.lu, len(planes) * nnz),
            )

    def solve(self, r, z):
        '''z = (LU)^-1 r per component block (testing / direct use).'''
        lib = _hip.lib()
        n = self.plan.n
        nb = self.struct.nblocks
        work = device.empty(n)
        for k in range(nb):
            _hip.check(lib.flow_ilu0_solve(
                ctypes.byref(self.plan.struct),
                _hip.f64(self.lu[k * self.plan.nnz:(k + 1) * self.plan.nnz]),
                _hip.f64(r[k * n:(k + 1) * n], n),
                _hip.f64(z[k * n:(k + 1) * n], n), _hip.f64(work, n),
                _hip.stream()
                ))
        return z
