# -*- coding: utf-8 -*-
'''
Host side of the two-level cycle with ILU(0) smoothing (`flow_tl` in
include/flow_hip.h, glue kernels in flow_amd/csrc/tl_kernels.hip): the
preconditioner of the Newton systems (reference: sparse LU,
flow/navier_stokes/pressure_correction.py:224-254) and of the heat system
(flow/heat.py:117-121) where the Chebyshev-smoothed cycle of flow_amd/fem/pmg.py
is rejected by its acceptance test -- cell Peclet numbers beyond ~3 at
CFL-sized steps.  Same two levels (the diagonal blocks of the assembled P2
operator, the P1 discretisation of the same operator on the same mesh), same
transfer tables; a multicolour ILU(0) application (flow_amd/fem/ilu.py) before
(`pre`) and / or after (`post`) the coarse correction, `coarse_sweeps` of them
on the P1 level.  The Newton solver runs 0 / 1 / 1 (coarse correction first:
the fewest launches per application), the heat solver 1 / 1 / 2.

tools/smoother_lab.py (CPU, the oracle's matrices; flexible GMRES applications
to 1e-8, pre + post sweeps): structured channel at cell Peclet 3.5 27 -> 13,
graded unstructured channel 48 -> 17, heat system at cell CFL 6 / 14 / 40 56 /
80 / 146 -> 19 / 23 / 35 (two coarse sweeps: 14 / 17 / 28).  On the GPU
(profiles/tlilu_r06.txt): graded 0.97 M-DoF channel 52.6 -> 25.2 applications,
16.3 -> 13.5 ms per step; config 4's heat solve 60-90 -> 14-19 iterations.
Whether the cycle or its smoother alone runs a solve is decided by measured
convergence rate per time (navier_stokes/newton_preconditioner.rate_verdict).
'''
import ctypes

import numpy
import torch

from . import ilu as _ilu
from .pmg import Pmg, transfer_tables
from .space import scalar_layout
from .. import _hip
from .. import device


class TwoLevelIlu(object):
    '''The cycle for the space W (degree 2).  `refactor(A, A1)` takes the
    assembled operator on the P2 pattern -- kind 2 (the Newton Jacobian: its
    diagonal blocks are smoothed, the residuals are formed with those blocks
    too) or kind 0 (scalar: the heat system) -- and the P1 discretisation of the
    same operator, Dirichlet rows already identity rows in both.  `front` is
    the flow_ilu a Krylov solver is handed (`ilu=cycle.front`): the fine
    smoother with `cycle` pointing at this flow_tl.'''

    def __init__(self, W, pre=1, post=1, coarse_sweeps=1, scalar=False,
                 packed=True, single_vector=True):
        lay = W.layout
        assert lay.degree == 2, 'the two-level cycle needs a P2 space'
        self.scalar = bool(scalar)
        self.lay = lay
        self.lay1 = scalar_layout(lay.mesh, 1)
        self.rows = self.vrows = None
        self.pre, self.post = int(pre), int(post)
        self.coarse_sweeps = int(coarse_sweeps)
        self.packed, self.single_vector = bool(packed), bool(single_vector)
        nb = 1 if self.scalar else 2
        n, n1 = lay.N, self.lay1.N
        ends, rptr, rsrc = transfer_tables(lay)
        self._keep = dict(
            ends=device.to_device(ends.reshape(-1)),
            rptr=device.to_device(rptr), rsrc=device.to_device(rsrc),
            bc_fine=torch.zeros(2 * n, dtype=torch.uint8, device=device.get()),
            bc_coarse=torch.zeros(2 * n1, dtype=torch.uint8,
                                  device=device.get()),
            work=device.zeros(nb * (4 * n + 5 * n1) + 2),
            )
        self._nrsrc = len(rsrc)
        self.fine = self.coarse = None
        self.front = None
        self._bc_key = None
        self.rscale = None

    # the Dirichlet masks of both levels and the P1 level's dof list: the
    # p-multigrid's (the methods only touch what this class has too)
    coarse_bc_dofs = Pmg.coarse_bc_dofs
    set_bcs = Pmg.set_bcs

    @staticmethod
    def plans(lay):
        '''The ILU plans of both levels (colouring, sweep streams: host work of
        seconds at a million rows) -- built here so that a driver can do it
        before its time loop (`prepare`), not inside the first step that
        needs them.'''
        return _ilu.plan_for(lay), _ilu.plan_for(scalar_layout(lay.mesh, 1))

    @staticmethod
    def _block_operator(A):
        lay = A.layout
        if A.kind == 0:
            return A.operator()
        # the two diagonal blocks of the 2 x 2 operator as a block-diagonal one
        op = _hip.Operator()
        src = A.operator()
        op.kind, op.n, op.nnz = 1, lay.N, lay.nnz
        rb = lay.dev('rowblocks')
        op.nblocks = rb.numel() - 1
        op.rowptr, op.cols = src.rowptr, src.cols
        op.rowblocks = _hip.i32(rb, None, 'rowblocks')
        op.vals[0], op.vals[1] = src.vals[0], src.vals[3]
        return op

    def refactor(self, A, A1, rscale=None):
        '''rscale (scalar systems): the cycle is the right preconditioner of
        the ROW-SCALED system diag(rscale)^-1 ... -- A is the scaled operator,
        A1 the P1 operator in its natural scaling, and the fine residual is
        multiplied by rscale (= the diagonal the rows were divided by) before
        it is restricted.'''
        assert A.layout is self.lay and A1.layout is self.lay1
        assert A.kind == (0 if self.scalar else 2) and A1.kind == A.kind
        sv = self.single_vector and self.packed
        if self.fine is None:
            self.fine = _ilu.Ilu0(A, packed=self.packed, single_vector=sv)
            self.coarse = _ilu.Ilu0(A1, packed=self.packed, single_vector=sv)
        else:
            self.fine.refactor(A)
            self.coarse.refactor(A1)
        self._ops = (self._block_operator(A), self._block_operator(A1), A, A1)
        self.rscale = rscale
        k = self._keep
        nb = 1 if self.scalar else 2
        n, n1 = self.lay.N, self.lay1.N
        s = _hip.TlS()
        s.fine = ctypes.pointer(self.fine.struct)
        s.coarse = ctypes.pointer(self.coarse.struct)
        s.fine_op = ctypes.addressof(self._ops[0])
        s.coarse_op = ctypes.addressof(self._ops[1])
        s.pre, s.post, s.coarse_sweeps = self.pre, self.post, self.coarse_sweeps
        s.ends = _hip.i32(k['ends'], 2 * n, 'ends').value
        s.rptr = _hip.i32(k['rptr'], n1 + 1, 'rptr').value
        s.rsrc = _hip.i32(k['rsrc'], self._nrsrc, 'rsrc').value
        s.bc_fine = _hip.u8(k['bc_fine'], nb * n).value
        s.bc_coarse = _hip.u8(k['bc_coarse'], nb * n1).value
        s.rscale = _hip.f64(rscale, nb * n, 'rscale').value \
            if rscale is not None else None
        s.work = _hip.f64(k['work'], nb * (4 * n + 5 * n1)).value
        assert k['work'].data_ptr() % 16 == 0
        self.struct = s
        # what the solvers take: the fine smoother's flow_ilu with the cycle
        f = self.fine.struct
        self.front_struct = _hip.IluS(f.plan, f.nblocks, f.lu, f.packed,
                                      f.single_vector, ctypes.addressof(s))
        self.front = _Front(self)
        return self

    def apply(self, r, z):
        '''z = M^-1 r (tests / the acceptance probe).'''
        n2 = (1 if self.scalar else 2) * self.lay.N
        _hip.check(_hip.lib().flow_tl_apply(
            ctypes.byref(self.struct), _hip.f64(r, n2, 'r'), _hip.f64(z, n2, 'z'),
            _hip.stream()))
        return z


class _Front(object):
    '''Quacks like an ilu.Ilu0 for ops.krylov_solve (`.struct`, `.plan`).'''

    def __init__(self, cycle):
        self.cycle = cycle
        self.struct = cycle.front_struct
        self.plan = cycle.fine.plan
        self.planes = cycle.fine.planes
