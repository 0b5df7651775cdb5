# -*- coding: utf-8 -*-
#
'''
Convection-diffusion (heat) operator in the "alpha M + beta F" form an external
ODE stepper drives; same interface as the reference's flow/heat.py:

    Heat(V, conv, kappa, rho, cp, bcs, source, supg_stabilization=False)
        .V .bcs .M .A .b
        .eval_alpha_M_beta_F(alpha, beta, u, t)
        .solve_alpha_M_beta_F(alpha, beta, b, t)

All operators are assembled by the HIP kernels of libflow_hip.so (K13/K14):
  M  u*v*dx with the 'vertex' quadrature scheme (reference :39-45): a lumped,
     diagonal matrix; for P2 the edge rows are zero (the vertex rule only sees
     the vertex basis functions) -- reproduced on purpose;
  A  matrix of f = -kappa grad(u).grad(v/(rho cp)) - (conv.grad u) v
     (reference :54-58) [+ SUPG terms, :60-86];
  b  rhs(f) = -int source v (UFL's rhs() negates; reference :88).
The reference solves with sparse LU (:117-121); here BiCGStab + ILU(0) on the
row-equilibrated system (GMRES(30) first, BiCGStab as the second try), started
from the solution of the previous solve on the space (a Banach sweep of the
Boussinesq driver solves nearly the system of the sweep before: 100 -> 60
iterations at 1.3 M rows, tools/boussinesq_time.py; the start vector only).

On the strips of flow_amd.parallel (several GPUs) every rank assembles A over
ITS cells for ITS rows, evaluates on its rows, and the solve is the sharded
GMRES with the rank's own ILU(0) of its diagonal block (flow_shard_gmres_solve
with a scalar operator); fields come in and go out valid on owned + ghost rows.
'''
import ctypes

import torch

from .fem import ops
from .fem.bcs import collect
from .fem.function import (
    Constant, Function, Vector, as_cell_coefficient,
    )
from . import _hip
from . import device
from . import parallel
from . import stabilization

solver_parameters = {'rtol': 1.0e-13, 'maxit': 2000, 'check_every': 10,
                     # 'pmg' (P2 spaces on one GPU; else 'ilu0'): the two-level
                     # p-multigrid cycle of flow_amd/fem/pmg.py in its scalar
                     # mode -- Chebyshev smoothing on the P2 operator, the P1
                     # discretisation of the same operator (assembled by the P1
                     # instance of the heat kernel) as the coarse level, treated
                     # with as many Chebyshev steps as its condition number
                     # asks for -- where one cycle contracts (`pmg_accept`),
                     # else the multicolour ILU(0); | 'ilu0' | 'jacobi'
                     'preconditioner': 'pmg', 'pmg_accept': 0.7,
                     'pmg_maxit': 30, 'pmg_retry': 8,
                     'pmg': {'pre': 1, 'post': 2, 'coarse_steps': 6,
                             'ratio_fine': 8.0, 'ratio_coarse': 12.0,
                             'coarse_auto': True, 'coarse_max': 48},
                     # what a rejected Chebyshev cycle is replaced with (P2, one
                     # GPU): 'tlilu' = the same two levels smoothed with ILU(0)
                     # sweeps (flow_amd/fem/tlilu.py; tools/smoother_lab.py
                     # --heat: 56 / 80 / 146 -> 14 / 17 / 28 GMRES iterations
                     # at cell CFL 6 / 14 / 40 with two sweeps on the P1 level)
                     # | 'ilu0' = the bare multicolour ILU(0)
                     'fallback': 'tlilu',
                     'tlilu': {'pre': 1, 'post': 1, 'coarse_sweeps': 2},
                     'tl_select': 'rate', 'tl_probe_smooth': 4,
                     'tl_probe_sweeps': 3, 'tl_accept': 0.9,
                     # 'previous': the solve starts from the solution of the
                     # previous solve on the space; 'zero': no history
                     'start': 'previous'}
last_solve_info = {}


def prepare(V):
    '''Host-side setup of the solve's fallback preconditioner for the space V
    (ILU(0) plans of the P2 and P1 levels, transfer tables, workspaces): a
    driver calls this before its time loop so that the first solve that needs
    the fallback does not pay ~1 s of colouring inside a time step.'''
    lay = V.layout
    if lay.degree != 2 or not device.on_gpu() or parallel.active():
        return
    from .fem.tlilu import TwoLevelIlu
    TwoLevelIlu.plans(lay)
    if solver_parameters.get('fallback') == 'tlilu' and \
            lay._dev.get('heat_tl') is None:
        lay._dev['heat_tl'] = TwoLevelIlu(
            V, scalar=True, packed=True, single_vector=False,
            **solver_parameters.get('tlilu', {}))
    return


def _data(v):
    if isinstance(v, Function):
        return v.data
    if isinstance(v, Vector):
        return v.data
    if hasattr(v, 'vec'):            # vec[:]
        return v.vec.data
    return v


class Heat(object):
    '''The semi-discrete heat equation  M u' = A u + b  behind the two calls an
    implicit ODE stepper makes: evaluate and solve  alpha M u + beta F(u, t),
    F(u, t) = A u + b  (interface of the reference's flow/heat.py:12-122).'''
    def __init__(
            self, V, conv, kappa, rho, cp, bcs, source,
            supg_stabilization=False
            ):
        lib = _hip.lib()
        self.V = V
        self.bcs = bcs
        assert V.dim == 1
        mesh = V.mesh()
        lay = V.layout
        nc = mesh.num_cells()
        rho_cp = float(rho) * float(cp)
        kappa = float(kappa)
        # (what the P1 coarse level of the solver's preconditioner is
        # assembled from, on demand)
        self._form = (conv, kappa, rho_cp, bool(supg_stabilization))
        self._coarse = None

        lumped = ops.assemble_scalar_matrix(lay, ops.LUMPED_MASS)
        self.A = ops.Matrix(lay, 0)
        msupg = ops.value_plane(lay) if supg_stabilization else None
        status = device.zeros(1, dtype=torch.int32)
        if supg_stabilization:
            assert conv is not None
            # kept for interface parity with the reference (:75-77)
            self.tau = stabilization.supg(
                mesh, conv, kappa, V.ufl_element().degree()
                )
        W = conv.function_space()
        assert W.dim == 2 and W.mesh() is mesh
        buf = ops.scratch(mesh, 2 * lay.nloc**2 * nc)
        # (on the strips: the rank's cells, the nonzeros of its rows)
        strips = parallel.active()
        mesh_s = parallel.mesh_view(mesh) if strips else ops.mesh_struct(mesh)
        q_s = parallel.view(lay).space if strips else ops.space_struct(lay)
        w_s = parallel.view(W.layout).space if strips \
            else ops.space_struct(W.layout)
        _hip.check(lib.flow_assemble_heat(
            ctypes.byref(mesh_s), ctypes.byref(q_s), ctypes.byref(w_s),
            _hip.f64(conv.data, W.size()), kappa, rho_cp,
            int(bool(supg_stabilization)), _hip.f64(buf),
            _hip.f64(self.A.vals), _hip.f64(msupg), None, _hip.i32(status),
            _hip.stream()
            ))
        if supg_stabilization:
            if int(device.to_host(status).item()) != 0:
                # the reference's C++ Expression throws (stabilization.py:132-140)
                raise RuntimeError('SUPG stabilization: tau > 1e3')
            ops.axpby(1.0, lumped.vals, 1.0, msupg)      # M_lumped + M_supg
            self.M = ops.Matrix(lay, 0, msupg)
        else:
            self.M = lumped

        # b = rhs(f) = - int source v  [- int (source / rho_cp) tau conv.grad(v)
        # with SUPG: the source term of R2, reference :79-86]
        if isinstance(source, (int, float)):
            source = Constant(source)
        zero_source = isinstance(source, Constant) and \
            float(source.values()[0]) == 0.0
        if zero_source:
            self.b = Vector(device.zeros(V.N))
        else:
            self.b = Vector(-ops.assemble_source(V, source))
            if supg_stabilization:
                coef = as_cell_coefficient(source, mesh, 1)
                assert coef.nl in (1, 3, 6), \
                    'SUPG source: Constant or Expression of degree <= 2'
                cs, keep = ops.coef_struct(coef, mesh, lay.degree)
                bs = device.zeros(V.N)
                _hip.check(lib.flow_assemble_heat_supg_source(
                    ctypes.byref(mesh_s), ctypes.byref(q_s), ctypes.byref(w_s),
                    _hip.f64(conv.data, W.size()), kappa, rho_cp,
                    ctypes.byref(cs), _hip.f64(buf), _hip.f64(bs),
                    _hip.i32(status), _hip.stream()
                    ))
                del keep
                ops.axpby(-1.0, bs, 1.0, self.b.data)
        return

    def eval_alpha_M_beta_F(self, alpha, beta, u, t):
        '''alpha M u + beta (A u + b) as a Vector; `t` is not used: the
        operators do not depend on time (reference :92-101).'''
        uvec = _data(u)
        alpha = float(alpha)
        beta = float(beta)
        n = self.V.N
        if parallel.active():
            # the rank's rows (u valid on owned + ghost rows; zeros elsewhere)
            out = device.zeros(n)
            tmp = device.zeros(n)
            v = parallel.view(self.V.layout)
            for mat, dst in ((self.M, out), (self.A, tmp)):
                _hip.check(_hip.lib().flow_operator_apply(
                    ctypes.byref(v.operator(mat)), _hip.f64(uvec, n, 'u'),
                    _hip.f64(dst, n), _hip.stream()))
            ops.axpby(1.0, self.b.data, 1.0, tmp)
            ops.axpby(beta, tmp, alpha, out)
            return Vector(out)
        out = device.empty(n)
        tmp = device.empty(n)
        self.M.apply(uvec, out)                 # M u
        self.A.apply(uvec, tmp)                 # A u
        ops.axpby(1.0, self.b.data, 1.0, tmp)   # A u + b
        ops.axpby(beta, tmp, alpha, out)        # alpha M u + beta (A u + b)
        return Vector(out)

    def solve_alpha_M_beta_F(self, alpha, beta, b, t):
        '''u with  alpha M u + beta (A u + b) = b_in  under the boundary
        conditions (reference :103-122; see the module docstring for the
        solver that stands in for its sparse LU).'''
        lib = _hip.lib()
        lay = self.V.layout
        st = _hip.stream()
        A = ops.Matrix(lay, 0)
        ops.axpby(float(alpha), self.M.vals, 0.0, A.vals)
        ops.axpby(float(beta), self.A.vals, 1.0, A.vals)
        # The reference computes right_hand_side = -beta*self.b + b but then
        # solves with the raw `b` (reference :109-121); identical when
        # self.b = 0.  Kept.
        bvec = _data(b)
        dofs, vals = collect(self.bcs, self.V.size())
        nbc = len(dofs)
        if nbc > 0:
            bc_dofs = device.to_device(dofs)
            bc_vals = device.to_device(vals)
            # bc.apply(A, b): identity rows, b[d] = g (in place, as dolfin)
            _hip.check(lib.flow_bc_identity_rows(
                ctypes.byref(A.operator()), _hip.f64(A.vals),
                _hip.i32(lay.dev('diag_idx')), nbc, _hip.i32(bc_dofs), st
                ))
            _hip.check(lib.flow_bc_set_values(
                nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(bvec), st
                ))
        # Row equilibration: Dirichlet rows carry temperatures (~300) in b, the
        # lumped mass rows O(h^2) entries, the zero-mass edge rows of P2 only
        # beta*A (~1e-9): a residual test on the raw rows would not see an
        # error on the small ones.  The reference's LU is insensitive to that;
        # the Krylov solve gets diag(A)^-1 A x = diag(A)^-1 b, where a relative
        # residual is a relative error to within the conditioning of a
        # diagonally scaled M-matrix-like operator.
        pmg_ready = None
        rejected = lay._dev.get('heat_pmg_rejected', 0)
        if rejected > 0:
            lay._dev['heat_pmg_rejected'] = rejected - 1
        if solver_parameters.get('preconditioner') == 'pmg' \
                and lay.degree == 2 and not parallel.active() \
                and rejected <= 0:
            # (packed from the UNSCALED operator: the rediscretised P1 level
            # goes with the finite-element scaling of the rows)
            pmg_ready = self._pmg(A, float(alpha), float(beta), dofs)
        dinv = A.diag_inv()
        if parallel.active():
            # (rows the rank does not own are empty here: keep them finite)
            dinv[~torch.isfinite(dinv)] = 1.0
        _hip.check(lib.flow_scale_rows(
            lay.N, _hip.i32(lay.dev('rowptr')), _hip.f64(dinv, lay.N),
            _hip.f64(A.vals, lay.nnz), st
            ))
        bvec = ops.vmul(dinv, bvec)
        u = Function(self.V)
        # Start vector: the solution of the previous solve on this space (the
        # reference's LU has no history; a Banach sweep of the Boussinesq
        # driver solves nearly the system of the sweep before, a time loop
        # nearly the one of the step before) with this call's Dirichlet values
        # -- a start vector only: the solve runs to the same rtol, a start that
        # leaves a larger residual than zero is dropped by the GMRES itself.
        # solver_parameters['start'] = 'zero': as a direct solve.
        par = solver_parameters
        start = lay._dev.get('heat_start') \
            if par.get('start', 'previous') == 'previous' else None
        warm = start is not None and start.numel() == u.data.numel()
        if warm:
            ops.copy(u.data, start)
            if nbc > 0:
                _hip.check(lib.flow_bc_set_values(
                    nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(u.data),
                    st))
        if parallel.active():
            # block-Jacobi ILU(0) GMRES on the strips; the ghost rows of the
            # solution from their owners
            pre = parallel.local_ilu(A)
            info = parallel.gmres(A, pre, bvec, u.data, rtol=par['rtol'],
                                  atol=0.0, maxit=par['maxit'], restart=30,
                                  x_is_zero=not warm, verify=True)
            parallel.halo(u.data, lay, 1)
            last_solve_info['heat'] = info
            self._remember(lay, u.data, warm, start)
            return u
        pre = None
        kind = par.get('preconditioner', 'ilu0')
        pmg = None
        if kind == 'pmg':
            pmg = pmg_ready
            kind = 'pmg' if pmg is not None else 'ilu0'
        if pmg is not None:
            # one application of the error propagation on a fixed full-
            # spectrum vector: the cycle is used where it contracts.
            # Chebyshev smoothing assumes a spectrum near the real axis; the
            # skew convection term puts imaginary parts ~ |conv| dt / h there:
            # beyond CFL ~ 3 the cycle amplifies -- first locally (a plume),
            # where this probe does not see it yet: a solve that has not
            # converged after `pmg_maxit` iterations goes on with the ILU(0);
            # a rejection stands for the next `pmg_retry` solves
            c = self._contraction(pmg, A, dofs)
            last_solve_info['heat_pmg_contraction'] = c
            last_solve_info['heat_pmg_coarse_steps'] = pmg.struct.coarse_steps
            if not c < par.get('pmg_accept', 0.8):
                pmg, kind = None, 'ilu0'
                lay._dev['heat_pmg_rejected'] = int(par.get('pmg_retry', 8))
        if kind == 'ilu0' and par.get('fallback', 'ilu0') == 'tlilu' \
                and par.get('preconditioner') == 'pmg' and lay.degree == 2:
            kind = 'tlilu'
        if kind == 'tlilu':
            # the two-level cycle with ILU(0) smoothing on the scaled system,
            # or -- by the self-test below -- its fine smoother alone
            pre = self._tl_or_bare(A, float(alpha), float(beta), dofs, dinv)
            kind = 'tlilu' if hasattr(pre, 'cycle') else 'ilu0'
        elif kind == 'ilu0':
            # The zero-mass edge rows and the skew convection make the diagonal
            # a poor preconditioner; the reference solves with LU (:116-121).
            pre = self._bare_ilu(A)
        last_solve_info['heat_preconditioner'] = kind
        # GMRES(30) with the cycle or the ILU(0) (minimises the residual
        # monotonically; BiCGStab stagnates on very coarse meshes, where most
        # rows are zero-mass edge rows), BiCGStab + ILU(0) as the second try
        try:
            info = ops.krylov_solve(
                'gmres', A, bvec, u.data, rtol=par['rtol'], atol=0.0,
                maxit=par['maxit'] if pmg is None
                else min(par['maxit'], int(par.get('pmg_maxit', 30))),
                ilu=pre, pmg=pmg, restart=30, x_is_zero=not warm,
                dinv='jacobi' if pre is None and pmg is None else None
                )
        except _hip.NotConverged:
            if pmg is not None:
                # (the cycle passed the probe and GMRES stalled all the same:
                # the fallback goes on from the iterate it left)
                if par.get('fallback', 'ilu0') == 'tlilu':
                    pre = self._tl_or_bare(A, float(alpha), float(beta), dofs,
                                           dinv)
                    last_solve_info['heat_preconditioner'] = \
                        'tlilu' if hasattr(pre, 'cycle') else 'ilu0'
                else:
                    pre = self._bare_ilu(A)
                    last_solve_info['heat_preconditioner'] = 'ilu0'
                lay._dev['heat_pmg_rejected'] = int(par.get('pmg_retry', 8))
                if not bool(torch.isfinite(u.data).all()):
                    ops.fill(u.data, 0.0)
                try:
                    info = ops.krylov_solve(
                        'gmres', A, bvec, u.data, rtol=par['rtol'], atol=0.0,
                        maxit=par['maxit'], ilu=pre, restart=30,
                        x_is_zero=False)
                    last_solve_info['heat'] = info
                    self._remember(lay, u.data, warm, start)
                    return u
                except _hip.NotConverged:
                    pass
            ops.fill(u.data, 0.0)
            if hasattr(pre, 'cycle'):
                # (the two-level ILU cycle stalled: its smoother alone, and no
                # cycle for the next solves either)
                pre = pre.cycle.fine
                lay._dev['heat_tl_choice'] = {
                    'left': int(par.get('pmg_retry', 8)), 'cycle': False}
                last_solve_info['heat_preconditioner'] = 'ilu0'
            info = ops.krylov_solve(
                'bicgstab', A, bvec, u.data, rtol=par['rtol'], atol=0.0,
                maxit=par['maxit'], check_every=par['check_every'], ilu=pre
                )
        last_solve_info['heat'] = info
        self._remember(lay, u.data, warm, start)
        return u

    # -- the p-multigrid preconditioner of the solve ---------------------------
    def _coarse_operators(self, lay1):
        '''M and A of the same form on the P1 space of the mesh (the P1
        instance of the heat kernel), once per Heat object.'''
        if self._coarse is None:
            conv, kappa, rho_cp, supg = self._form
            lib = _hip.lib()
            mesh = self.V.mesh()
            W = conv.function_space()
            nc = mesh.num_cells()
            M1 = ops.assemble_scalar_matrix(lay1, ops.LUMPED_MASS)
            A1 = ops.Matrix(lay1, 0)
            msupg = ops.value_plane(lay1) if supg else None
            status = device.zeros(1, dtype=torch.int32)
            buf = ops.scratch(mesh, 2 * lay1.nloc**2 * nc)
            _hip.check(lib.flow_assemble_heat(
                ctypes.byref(ops.mesh_struct(mesh)),
                ctypes.byref(ops.space_struct(lay1)),
                ctypes.byref(ops.space_struct(W.layout)),
                _hip.f64(conv.data, W.size()), kappa, rho_cp, int(supg),
                _hip.f64(buf), _hip.f64(A1.vals), _hip.f64(msupg), None,
                _hip.i32(status), _hip.stream()))
            if supg:
                ops.axpby(1.0, M1.vals, 1.0, msupg)
                M1 = ops.Matrix(lay1, 0, msupg)
            self._coarse = (M1, A1)
        return self._coarse

    def _pmg(self, S, alpha, beta, bc_dofs_host):
        '''The cycle for S = alpha M + beta A (unscaled, Dirichlet rows already
        identity rows) -- as the right preconditioner of the ROW-SCALED system
        the Krylov method sees: M^-1 r~ = cycle(D r~), i.e. the cycle is fed
        the scaled residual as its scaled residual (the fine level's own
        row scaling in its first kernel is switched off).'''
        import numpy
        from .fem.pmg import Pmg
        lib = _hip.lib()
        lay = self.V.layout
        held = lay._dev.get('heat_pmg')
        if held is None:
            held = lay._dev['heat_pmg'] = Pmg(
                self.V, scalar=True, **solver_parameters.get('pmg', {}))
        pmg = held
        lay1 = pmg.lay1
        M1, A1 = self._coarse_operators(lay1)
        S1 = lay._dev.get('heat_pmg_S1')
        if S1 is None:
            S1 = lay._dev['heat_pmg_S1'] = ops.Matrix(lay1, 0)
        ops.axpby(alpha, M1.vals, 0.0, S1.vals)
        ops.axpby(beta, A1.vals, 1.0, S1.vals)
        bc1_host, bc1 = pmg.set_bcs(numpy.asarray(bc_dofs_host))
        if len(bc1_host):
            _hip.check(lib.flow_bc_identity_rows(
                ctypes.byref(S1.operator()), _hip.f64(S1.vals),
                _hip.i32(lay1.dev('diag_idx')), len(bc1_host), _hip.i32(bc1),
                _hip.stream()))
        # share of the mass term in the diagonal of the P1 operator (free rows)
        di = lay1.dev('diag_idx').long()
        share = alpha * M1.vals[di] / S1.vals[di]
        if len(bc1_host):
            share[bc1.long()] = 1.0
        share = float(device.to_host(share.min()))
        if not (share > 0.0):
            return None       # (not a mass term plus a positive rest)
        pmg.refactor(S, S1, mass_share=share)
        pmg.fine.dinv.fill_(1.0)
        return pmg

    def _bare_ilu(self, A):
        '''The multicolour ILU(0) of the scaled system, in buffers that live
        with the space (refactored per solve).'''
        from .fem import ilu
        lay = self.V.layout
        held = lay._dev.get('heat_ilu')
        if held is None:
            held = lay._dev['heat_ilu'] = ilu.Ilu0(A)
        else:
            held.refactor(A)
        return held

    def _tl_or_bare(self, S, alpha, beta, bc_dofs_host, dinv):
        '''The two-level ILU cycle -- where it converges faster per unit of
        time than its fine smoother alone (rate_verdict of
        navier_stokes/newton_preconditioner.py: a few applications of both as
        stationary iterations on the fixed probe vector, timed; at the first
        use and every `pmg_retry` solves after.  On an under-resolved mesh --
        cell Peclet numbers in the hundreds -- the P1 rediscretisation is no
        coarse problem any more) --, else that smoother, the bare ILU(0), from
        the same factors.'''
        lay = self.V.layout
        tl = self._tl(S, alpha, beta, bc_dofs_host, dinv)
        choice = lay._dev.setdefault('heat_tl_choice', {'left': 0, 'cycle': True})
        select = solver_parameters.get('tl_select', 'rate')
        if select != 'rate':
            choice['cycle'], choice['left'] = select == 'cycle', 1
        if choice['left'] <= 0:
            from .navier_stokes.newton_preconditioner import rate_verdict
            import numpy
            n = lay.N
            hold = lay._dev.setdefault('heat_pmg_probe', {})
            key = hash(numpy.asarray(bc_dofs_host).tobytes())
            if hold.get('key') != key:
                v = numpy.random.RandomState(7).standard_normal(n)
                v[bc_dofs_host] = 0.0
                hold.update(key=key, v=device.to_device(v), w=device.empty(n),
                            z=device.empty(n))
            v, w, z = hold['v'], hold['w'], hold['z']
            choice['cycle'], probe = rate_verdict(
                S.apply, tl.apply, tl.fine.solve, v, w, z,
                smooth=int(solver_parameters.get('tl_probe_smooth', 4)),
                sweeps=int(solver_parameters.get('tl_probe_sweeps', 3)),
                accept=float(solver_parameters.get('tl_accept', 0.9)))
            choice['left'] = int(solver_parameters.get('pmg_retry', 8))
            last_solve_info['heat_tl_contraction'] = probe
        choice['left'] -= 1
        return tl.front if choice['cycle'] else tl.fine

    def _tl(self, S, alpha, beta, bc_dofs_host, dinv):
        '''The two-level ILU cycle for the ROW-SCALED system S = diag(dinv)
        (alpha M + beta A) (Dirichlet rows identity rows): ILU(0) of S on the
        fine level (ILU(0) does not see a row scaling), the P1 discretisation
        of the same operator in its finite-element scaling as the coarse
        level, the fine residual scaled back (1 / dinv) before it is
        restricted.'''
        import numpy
        from .fem.tlilu import TwoLevelIlu
        lib = _hip.lib()
        lay = self.V.layout
        held = lay._dev.get('heat_tl')
        if held is None:
            held = lay._dev['heat_tl'] = TwoLevelIlu(
                self.V, scalar=True, packed=True, single_vector=False,
                **solver_parameters.get('tlilu', {}))
        lay1 = held.lay1
        M1, A1 = self._coarse_operators(lay1)
        S1 = lay._dev.get('heat_pmg_S1')
        if S1 is None:
            S1 = lay._dev['heat_pmg_S1'] = ops.Matrix(lay1, 0)
        ops.axpby(alpha, M1.vals, 0.0, S1.vals)
        ops.axpby(beta, A1.vals, 1.0, S1.vals)
        bc1_host, bc1 = held.set_bcs(numpy.asarray(bc_dofs_host))
        if len(bc1_host):
            _hip.check(lib.flow_bc_identity_rows(
                ctypes.byref(S1.operator()), _hip.f64(S1.vals),
                _hip.i32(lay1.dev('diag_idx')), len(bc1_host), _hip.i32(bc1),
                _hip.stream()))
        held.refactor(S, S1, rscale=torch.reciprocal(dinv))
        return held

    def _contraction(self, pmg, A, bc_dofs_host):
        import numpy
        lay = self.V.layout
        n = lay.N
        hold = lay._dev.setdefault('heat_pmg_probe', {})
        key = hash(numpy.asarray(bc_dofs_host).tobytes())
        if hold.get('key') != key:
            v = numpy.random.RandomState(7).standard_normal(n)
            v[bc_dofs_host] = 0.0
            hold.update(key=key, v=device.to_device(v), w=device.empty(n),
                        z=device.empty(n))
        v, w, z = hold['v'], hold['w'], hold['z']
        from .navier_stokes.newton_preconditioner import power_probe
        return power_probe(A.apply, pmg.apply, v, w, z, sweeps=1)

    @staticmethod
    def _remember(lay, x, warm, start):
        if solver_parameters.get('start', 'previous') != 'previous':
            return
        if warm:
            ops.copy(start, x)
        else:
            lay._dev['heat_start'] = _hip.clone(x)
