# -*- coding: utf-8 -*-
#
'''
Steady Stokes solve used to bootstrap the Karman run; same interface as the
reference's `flow.stokes.solve(WP, bcs, mu, f, verbose=True, tol=1e-13,
max_iter=500)` (flow/stokes.py:13-148), which returns deep-copied `(u, p)`.

The reference assembles the mixed Taylor-Hood system
    a = mu (grad u, grad v) - (p, div v) - (q, div u),   L = (f, v)
with `assemble_system(a, L, bcs)` and hands it to GMRES preconditioned with
BoomerAMG on  mu (grad u, grad v) - p q  (:40-60).

Default here (solver_parameters['method'] = 'minres'): the same symmetric
saddle-point system, Dirichlet values eliminated symmetrically like
`assemble_system`, solved by preconditioned MINRES with the same block-diagonal
preconditioner -- the viscous block replaced by ONE multigrid cycle per
component (`ViscousCycle`: Chebyshev smoothing on P2, smoothed-aggregation
V-cycle on the P1 Galerkin operator; where the reference runs BoomerAMG), the
pressure block by the scaled lumped pressure mass matrix.  One product with the
system and one preconditioner application per iteration, no inner solves.  On
the 10 M-DoF Karman channel at tol 1e-13: 130 iterations, 0.4 s of iterations
behind 3.6 s of host-side hierarchy setup (round 2, with one Jacobi +
aggregate-coarse-space application on the viscous block instead: 8328
iterations, 16 s).

'schur': the same discrete system solved by CG on its pressure Schur complement
    S = B K^-1 B^T        (K = mu * vector Laplacian with the Dirichlet rows
                           eliminated, B = -(q, div u) on the free dofs)
preconditioned with the scaled lumped pressure mass matrix (S is spectrally
equivalent to M_p / mu -- the same fact the reference's `- p*q` preconditioner
block uses).  Every application of S is a velocity solve: CG with the Jacobi +
aggregate-coarse-space preconditioner of the pressure-Poisson solver.  All
pieces run on the HIP path: stiffness / mass assembly, the two coupling kernels
(flow_assemble_pressure_rhs with p0 = 0 for B, flow_assemble_div_adjoint for
B^T), SpMV, CG.  Non-convergence raises RuntimeError as
'error_on_nonconvergence' does (:139).
'''
import ctypes

import numpy
import torch

from .fem import ops
from .fem.bcs import collect
from .fem.function import Function, scalar_value
from .fem.space import MixedFunctionSpace
from .message import info
from . import _hip
from . import device

last_solve_info = {}
# 'multigrid': smoothed-aggregation V-cycle for the velocity solves inside the
# Schur complement instead of Jacobi + aggregate coarse space.  Measured on the
# 2 M-DoF cavity (tools/run_configs.py stokes): 23 instead of 590 CG iterations
# per velocity solve (1.7 s of solves instead of 4.5 s), but 11 s of host-side
# hierarchy setup (scipy triple products on the P2 stiffness matrix) against
# 0.3 s -- a one-shot solve is faster with the two-level scheme (5.5 s vs
# 14.5 s in total); it pays when the same operator is solved with many times.
# 'viscous_cycle' (method 'minres'): the preconditioner of the viscous block --
# 'pmg' = ViscousCycle (Chebyshev on P2 + smoothed-aggregation V-cycle on the
# P1 Galerkin operator), 'two_level' = one Jacobi + aggregate-coarse-space
# application (round 2; 8328 MINRES iterations on the 10 M-DoF channel where
# the cycle needs a few hundred)
solver_parameters = {'multigrid': False, 'method': 'minres',
                     'coarse_size': 4096, 'viscous_cycle': 'pmg'}


class ViscousCycle(object):
    """Preconditioner of ONE component of the viscous block mu (grad u, grad v)
    with its Dirichlet rows and columns eliminated -- what the reference hands
    to BoomerAMG (flow/stokes.py:53-60, `hypre_amg` on mu inner(grad u, grad v)):
    a symmetric two-level p-multigrid cycle

        x = cheb(r);  x += P V(P^T (r - K x));  x += cheb(r - K x)

    P2 level: `steps` Chebyshev steps for D^-1 K on [lam/ratio, 1.1 lam] (lam by
    the power method) before and after; coarse level: the P1 discretisation of
    the same form on the same mesh -- exactly the Galerkin operator P^T K P,
    P1 being a subspace of P2 --, treated with one smoothed-aggregation V(1,1)
    cycle (flow_amd/fem/multigrid.py, as the pressure solver: its hierarchy is
    built on a matrix with 1/4 of the rows and 1/7 of the nonzeros of K, seconds
    where a hierarchy on K itself takes a minute of host-side triple products).
    Fixed, linear, symmetric positive definite (pre- and post-smoother are the
    same polynomial in D^-1 K, the V-cycle is symmetric): fit for MINRES.
    Everything is fp64 CSR-stream products of the library."""

    def __init__(self, Kc, isbc, mu, steps=2, ratio=8.0):
        import scipy.sparse as sp
        from .fem.multigrid import Multigrid, CsrOperator
        from .fem.pmg import transfer_tables
        from .fem.space import scalar_layout
        lay = Kc.layout
        assert lay.degree == 2
        lay1 = scalar_layout(lay.mesh, 1)
        n, n1 = lay.N, lay1.N
        self.K, self.n, self.n1 = Kc, n, n1
        self.dinv = Kc.diag_inv()
        self.steps = steps
        # P1 level: mu * stiffness, Dirichlet vertices eliminated
        isbc1 = isbc[lay.vertex_dofs]
        K1 = ops.assemble_scalar_matrix(lay1, ops.STIFFNESS)
        K1c = ops.symmetric_bc_matrix(
            K1, device.to_device(isbc1.astype(numpy.uint8)))
        ops.axpby(mu, K1c.vals, 0.0, K1c.vals)
        if isbc1.any():
            K1c.vals[lay1.dev('diag_idx').long()[device.to_device(isbc1)]] = 1.0
        self.K1c = K1c
        # (coarsest level <= 1500 rows: its dense inverse is formed on the host
        # -- 0.9 s for the 2.7 k rows the default threshold leaves on a 2 M-DoF
        # cavity, more than the whole MINRES iteration takes)
        self.mg = Multigrid(K1c, isbc1, singular=not isbc1.any(), coarsest=1500)
        self.usable = self.mg.nlevels >= 2
        if not self.usable:
            return
        # transfers between the free dofs of the two levels
        ends, _rptr, _rsrc = transfer_tables(lay)
        rows = numpy.repeat(numpy.arange(n), 2)
        P = sp.csr_matrix((numpy.full(2 * n, 0.5), (rows, ends.ravel())),
                          shape=(n, n1))
        P = sp.diags((~isbc).astype(float)).dot(P).dot(
            sp.diags((~isbc1).astype(float))).tocsr()
        P.eliminate_zeros()
        self.P = CsrOperator(P)
        self.R = CsrOperator(P.T.tocsr())
        self._vec = [device.empty(n) for _ in range(4)]
        self._c = [device.empty(n1) for _ in range(2)]
        # spectral radius of D^-1 K: power method
        v = self._vec[0]
        ops.copy(v, device.to_device(
            numpy.random.RandomState(5).standard_normal(n)))
        lam = 1.0
        for _ in range(20):
            Kc.apply(v, self._vec[1])
            ops.vmul(self._vec[1], self.dinv, v)
            lam = ops.vector_norm(v)
            ops.axpby(0.0, v, 1.0 / lam, v)
        self.lam = lam
        hi, lo = 1.1 * lam, lam / ratio
        self.theta, self.delta = 0.5 * (hi + lo), 0.5 * (hi - lo)

    def _apply_csr(self, op, x, y):
        _hip.check(_hip.lib().flow_operator_apply(
            ctypes.byref(op.op), _hip.f64(x, op.shape[1]),
            _hip.f64(y, op.shape[0]), _hip.stream()))

    def _cheb(self, res, x, first):
        """`steps` Chebyshev steps for K e = res, added to x (first: x = 0 on
        entry and is overwritten).  res is overwritten."""
        d, t = self._vec[2], self._vec[3]
        sigma = self.theta / self.delta
        rho = 1.0 / sigma
        ops.vmul(res, self.dinv, d, a=1.0 / self.theta)
        for k in range(self.steps):
            if first and k == 0:
                ops.copy(x, d)
            else:
                ops.axpby(1.0, d, 1.0, x)
            if k + 1 < self.steps:
                self.K.apply(d, t)
                ops.axpby(-1.0, t, 1.0, res)
                rn = 1.0 / (2.0 * sigma - rho)
                ops.vmul(res, self.dinv, t, a=2.0 * rn / self.delta)
                ops.axpby(1.0, t, rn * rho, d)
                rho = rn

    def apply(self, r, z):
        """z = B r (r is left untouched)."""
        res, t = self._vec[0], self._vec[1]
        rc, xc = self._c
        ops.copy(res, r)
        self._cheb(res, z, True)
        self.K.apply(z, t)
        ops.copy(res, r)
        ops.axpby(-1.0, t, 1.0, res)
        self._apply_csr(self.R, res, rc)
        self.mg.apply(rc, xc)
        self._apply_csr(self.P, xc, t)
        ops.axpby(1.0, t, 1.0, z)
        self.K.apply(z, t)
        ops.copy(res, r)
        ops.axpby(-1.0, t, 1.0, res)
        self._cheb(res, z, False)
        return z


def solve(
        WP,
        bcs,
        mu,
        f,
        verbose=True,
        tol=1.0e-13,
        max_iter=500
        ):
    from .fem.multigrid import Multigrid
    lib = _hip.lib()
    multigrid = solver_parameters['multigrid']
    mu = scalar_value(mu)
    assert mu > 0.0
    assert isinstance(WP, MixedFunctionSpace)
    W, P = WP.sub(0), WP.sub(1)
    assert W.dim == 2 and P.dim == 1 and P.degree == 1
    mesh = W.mesh()
    lay, play = W.layout, P.layout
    nc = mesh.num_cells()
    n, n2, npr = W.N, W.size(), P.N
    st = _hip.stream()
    ms = ops.mesh_struct(mesh)
    ws = ops.space_struct(lay)
    ps = ops.space_struct(play)
    buf = ops.scratch(mesh, 2 * lay.nloc * nc)

    # Dirichlet data, split into velocity and pressure conditions
    def target(bc):
        # a component condition (W.sub(0)) lives on its parent vector space
        return getattr(bc, 'space', None) or bc.function_space()
    u_bcs = [bc for bc in bcs if target(bc).dim == 2]
    p_bcs = [bc for bc in bcs if target(bc).dim == 1]
    ud, uv = collect(u_bcs, n2)
    pd, pv = collect(p_bcs, npr) if p_bcs else (numpy.zeros(0, numpy.int32),
                                                numpy.zeros(0))
    umask = numpy.ones(n2)
    umask[ud] = 0.0
    pmask = numpy.ones(npr)
    pmask[pd] = 0.0
    umask_d = device.to_device(umask)
    pmask_d = device.to_device(pmask)
    ug = numpy.zeros(n2)
    ug[ud] = uv
    pg = numpy.zeros(npr)
    pg[pd] = pv
    ug_d = device.to_device(ug)
    pg_d = device.to_device(pg)

    # K = mu * stiffness per component, Dirichlet rows/columns eliminated
    K = ops.assemble_scalar_matrix(lay, ops.STIFFNESS)
    planes = []
    masks = []
    for comp in range(2):
        isbc = (umask[comp * n:(comp + 1) * n] == 0.0)
        Kc = ops.symmetric_bc_matrix(
            K, device.to_device(isbc.astype(numpy.uint8))
            )
        # scale the free part by mu (identity rows stay 1)
        ops.axpby(mu, Kc.vals, 0.0, Kc.vals)
        Kc.vals[lay.dev('diag_idx').long()[device.to_device(isbc)]] = 1.0
        planes.append(Kc)
        masks.append(isbc)
    method = solver_parameters.get('method', 'minres')
    # MINRES: the p-multigrid cycle on the viscous block (P2 spaces; systems too
    # small to coarsen fall back to the two-level scheme)
    cycles = None
    if method == 'minres' and lay.degree == 2 and \
            solver_parameters.get('viscous_cycle', 'pmg') == 'pmg':
        made = {}
        cycles = []
        for comp in range(2):
            key = masks[comp].tobytes()
            if key not in made:
                made[key] = ViscousCycle(planes[comp], masks[comp], mu)
            cycles.append(made[key])
        if not all(c.usable for c in cycles):
            cycles = None
    # preconditioner of the velocity solves of the other paths: the smoothed-
    # aggregation V-cycle (aggregates of ~3x3 dofs: half a mesh width per dof
    # for P2) or the two-level scheme
    coarse = []
    hierarchies = {}
    for comp in range(2 if cycles is None else 0):
        Kc, isbc = planes[comp], masks[comp]
        mg = None
        if multigrid:
            # both components usually carry Dirichlet data on the same dofs:
            # one hierarchy then serves both (same operator)
            mkey = isbc.tobytes()
            if mkey not in hierarchies:
                hierarchies[mkey] = Multigrid(
                    Kc, isbc, singular=not isbc.any(), s=3.0 / lay.degree)
            mg = hierarchies[mkey]
        if mg is not None and mg.nlevels >= 2:
            coarse.append((None, mg))
        else:
            coarse.append((ops.CoarseSpace(
                Kc, isbc, singular=not isbc.any(),
                target_nc=solver_parameters.get('coarse_size', 4096)
                ), None))
    dinvs = [Kc.diag_inv() for Kc in planes]
    Kfull = ops.Matrix(lay, 1, torch.cat([K.vals, K.vals]))

    inner = {'its': 0, 'solves': 0}
    inner_tol = min(1.0e-12, 1.0e-2 * tol)

    def solve_K(rhs, out):
        '''out = K_bc^-1 rhs, component by component (rhs holds the Dirichlet
        values on the eliminated rows).'''
        for comp in range(2):
            sl = slice(comp * n, (comp + 1) * n)
            x = out[sl]
            sol = ops.krylov_solve(
                'cg', planes[comp], rhs[sl], x, rtol=inner_tol, atol=1.0e-300,
                maxit=20000, dinv=dinvs[comp],
                check_every=2 if coarse[comp][1] is not None else 10,
                coarse=coarse[comp][0], mg=coarse[comp][1],
                tag='stokes_velocity_%d' % comp
                )
            inner['its'] += sol.iterations
            inner['solves'] += 1
        return out

    def apply_B(u, out):
        '''out = -(q, div u), rows of pressure Dirichlet dofs zeroed.'''
        zero_p = device.zeros(npr)
        _hip.check(lib.flow_assemble_pressure_rhs(
            ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(ps),
            _hip.f64(u, n2), _hip.f64(zero_p, npr), 1.0, 0.0, 0,
            _hip.f64(buf), _hip.f64(out, npr), st
            ))
        ops.vmul(out, pmask_d, out)
        return out

    def apply_Bt(p, out):
        '''out = -(p, div v), rows of velocity Dirichlet dofs zeroed.'''
        _hip.check(lib.flow_assemble_div_adjoint(
            ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(ps),
            _hip.f64(p, npr), _hip.f64(buf), _hip.f64(out, n2), st
            ))
        ops.vmul(out, umask_d, out)
        return out

    # right-hand side of the velocity block: (f, v) - K u_g - B^T p_g on the
    # free rows, the Dirichlet values on the eliminated rows
    F = ops.assemble_source(W, f)
    tmp_u = device.empty(n2)
    Kfull.apply(ug_d, tmp_u)
    ops.axpby(-mu, tmp_u, 1.0, F)
    apply_Bt_raw = device.empty(n2)
    _hip.check(lib.flow_assemble_div_adjoint(
        ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(ps),
        _hip.f64(pg_d, npr), _hip.f64(buf), _hip.f64(apply_Bt_raw, n2), st
        ))
    ops.axpby(-1.0, apply_Bt_raw, 1.0, F)
    ops.vmul(F, umask_d, F)
    ops.axpby(1.0, ug_d, 1.0, F)

    # preconditioner: mu / lumped pressure mass (S ~ M_p / mu)
    Mp = ops.assemble_scalar_matrix(play, ops.MASS)
    one = torch.ones(npr, dtype=torch.float64, device=device.get())
    lumped = device.empty(npr)
    Mp.apply(one, lumped)
    # minv = mu / lumped, zero on the pressure Dirichlet rows
    minv = device.to_device(mu * pmask / device.to_host(lumped).numpy())

    if method == 'minres':
        return _minres(
            W, P, mu, tol, max_iter, verbose, planes, dinvs, coarse, F, ug_d,
            pg_d, umask_d, pmask_d, minv, apply_B, apply_Bt, cycles
            )

    # u(p_f) = K^-1 (F - B^T p_f);  continuity on the free pressure rows:
    #   B u(p_f) = 0   <=>   S p_f = B u(0)
    u_hat = device.zeros(n2)
    solve_K(F, u_hat)
    rhs = device.empty(npr)
    apply_B(u_hat, rhs)


    def apply_S(p, out):
        t = device.empty(n2)
        apply_Bt(p, t)
        y = device.zeros(n2)
        solve_K(t, y)
        return apply_B(y, out)

    # preconditioned CG on S p_f = rhs (host loop; every operation is a
    # library call)
    pf = device.zeros(npr)
    r = device.empty(npr)
    ops.copy(r, rhs)
    z = ops.vmul(r, minv)
    d = device.empty(npr)
    ops.copy(d, z)
    rz = ops.dot(r, z)
    bnorm = numpy.sqrt(ops.dot(rhs, rhs))
    Sd = device.empty(npr)
    its = 0
    res = bnorm
    while bnorm > 0.0 and res > tol * bnorm:
        if its >= max_iter:
            raise _hip.NotConverged(
                'Stokes Schur-complement CG did not converge in %d iterations '
                '(|r|/|b| = %.3e > %.3e)' % (its, res / bnorm, tol)
                )
        apply_S(d, Sd)
        alpha = rz / ops.dot(d, Sd)
        ops.axpby(alpha, d, 1.0, pf)
        ops.axpby(-alpha, Sd, 1.0, r)
        res = numpy.sqrt(ops.dot(r, r))
        ops.vmul(r, minv, z)
        rz_new = ops.dot(r, z)
        ops.axpby(1.0, z, rz_new / rz, d)
        rz = rz_new
        its += 1
        if verbose:
            info('Stokes CG %d: |r|/|b| = %.3e' % (its, res / bnorm))

    # velocity of the converged pressure, assemble the outputs
    t = device.empty(n2)
    apply_Bt(pf, t)
    ops.axpby(-1.0, t, 1.0, F)          # F - B^T p_f (Dirichlet rows keep u_g)
    u = Function(W)
    solve_K(F, u.data)
    p = Function(P)
    ops.vmul(pf, pmask_d, p.data)
    ops.axpby(1.0, pg_d, 1.0, p.data)
    last_solve_info.update(
        outer_iterations=its, inner_iterations=inner['its'],
        inner_solves=inner['solves'], residual=res / max(bnorm, 1e-300)
        )
    return u, p


def _minres(W, P, mu, tol, max_iter, verbose, planes, dinvs, coarse, F, ug_d,
            pg_d, umask_d, pmask_d, minv, apply_B, apply_Bt, cycles=None):
    '''Preconditioned MINRES (Paige & Saunders; the form of Elman, Silvester &
    Wathen, Alg. 4.1) on the symmetric system

        [ K    Bm^T ] [u]   [F                    ]
        [ Bm   Ip   ] [p] = [-pm B(u_g) + (1-pm) p_g]

    K: mu * stiffness per component with the Dirichlet rows AND columns
    eliminated (identity rows; F carries the lifted boundary values), Bm =
    pm B um the divergence coupling between the free rows, Ip the identity on
    the pressure Dirichlet rows -- `assemble_system(a, L, bcs)` of the
    reference (flow/stokes.py:40-42).  Preconditioner: blockdiag(ViscousCycle
    of K per component -- or, cycles = None, one application of the two-level
    scheme --, lumped pressure mass / mu) (:53-60, with one multigrid cycle
    where the reference runs BoomerAMG).  Converged when the preconditioned residual
    has fallen by `tol` (relative_tolerance, absolute 0, :127-131).'''
    lib = _hip.lib()
    n, n2, npr = W.N, W.size(), P.N
    st = _hip.stream()
    lay = W.layout
    if cycles is None:
        for c in coarse:
            assert c[0] is not None, \
                "'minres' without the cycle uses the two-level preconditioner"
        cwork = device.empty(2 * max(c[0].struct.lda for c in coarse) + 2)

    class Vec(object):
        def __init__(self):
            self.u = device.zeros(n2)
            self.p = device.zeros(npr)

    def dot(a, b):
        return ops.dot(a.u, b.u) + ops.dot(a.p, b.p)

    def axpby(alpha, x, beta, y):
        ops.axpby(alpha, x.u, beta, y.u)
        ops.axpby(alpha, x.p, beta, y.p)

    def copy(dst, src):
        ops.copy(dst.u, src.u)
        ops.copy(dst.p, src.p)

    tmp_u = device.empty(n2)
    tmp_p = device.empty(npr)
    one_minus_pm = device.zeros(npr) + 1.0
    ops.axpby(-1.0, pmask_d, 1.0, one_minus_pm)

    def apply_A(x, y):
        for comp in range(2):
            sl = slice(comp * n, (comp + 1) * n)
            planes[comp].apply(x.u[sl], y.u[sl])
        # + um B^T (pm p)
        ops.vmul(x.p, pmask_d, tmp_p)
        apply_Bt(tmp_p, tmp_u)
        ops.axpby(1.0, tmp_u, 1.0, y.u)
        # pm B (um u) + (1 - pm) p
        ops.vmul(x.u, umask_d, tmp_u)
        apply_B(tmp_u, y.p)
        ops.vmul(x.p, one_minus_pm, tmp_p)
        ops.axpby(1.0, tmp_p, 1.0, y.p)

    def precondition(v, z):
        for comp in range(2):
            sl = slice(comp * n, (comp + 1) * n)
            if cycles is not None:
                cycles[comp].apply(v.u[sl], z.u[sl])
                continue
            _hip.check(lib.flow_two_level_apply(
                ctypes.byref(coarse[comp][0].struct),
                _hip.f64(dinvs[comp], n), _hip.f64(v.u[sl].contiguous(), n),
                _hip.f64(z.u[sl], n), _hip.f64(cwork), st
                ))
        # pressure block: mu / lumped mass on the free rows, identity elsewhere
        ops.vmul(v.p, minv, z.p)
        ops.vmul(v.p, one_minus_pm, tmp_p)
        ops.axpby(1.0, tmp_p, 1.0, z.p)

    # right-hand side
    b = Vec()
    ops.copy(b.u, F)
    apply_B(ug_d, b.p)                      # pm B u_g
    ops.axpby(1.0, pg_d, -1.0, b.p)         # -pm B u_g + p_g (p_g: masked rows)
    x = Vec()
    v_old, v, v_new = Vec(), Vec(), Vec()
    z, z_new = Vec(), Vec()
    w_old, w, w_new = Vec(), Vec(), Vec()
    Az = Vec()
    copy(v, b)                              # x0 = 0
    precondition(v, z)
    gamma = numpy.sqrt(max(dot(z, v), 0.0))
    gamma0 = gamma
    eta = gamma
    s_old = s_cur = 0.0
    c_old = c_cur = 1.0
    gamma_old = 1.0
    its = 0
    res = gamma
    while gamma0 > 0.0 and abs(res) > tol * gamma0:
        if its >= max_iter:
            raise _hip.NotConverged(
                'Stokes MINRES did not converge in %d iterations '
                '(|r|/|r0| = %.3e > %.3e)' % (its, abs(res) / gamma0, tol)
                )
        axpby(0.0, z, 1.0 / gamma, z)                  # z_j /= gamma_j
        apply_A(z, Az)
        delta = dot(Az, z)
        # v_{j+1} = A z_j - delta/gamma v_j - gamma/gamma_old v_{j-1}
        copy(v_new, Az)
        axpby(-delta / gamma, v, 1.0, v_new)
        if its > 0:
            axpby(-gamma / gamma_old, v_old, 1.0, v_new)
        precondition(v_new, z_new)
        gamma_new = numpy.sqrt(max(dot(z_new, v_new), 0.0))
        a0 = c_cur * delta - c_old * s_cur * gamma
        a1 = numpy.sqrt(a0 * a0 + gamma_new * gamma_new)
        a2 = s_cur * delta + c_old * c_cur * gamma
        a3 = s_old * gamma
        c_new, s_new = a0 / a1, gamma_new / a1
        # w_{j+1} = (z_j - a3 w_{j-1} - a2 w_j) / a1
        copy(w_new, z)
        axpby(-a3, w_old, 1.0, w_new)
        axpby(-a2, w, 1.0, w_new)
        axpby(0.0, w_new, 1.0 / a1, w_new)
        axpby(c_new * eta, w_new, 1.0, x)
        eta = -s_new * eta
        res = eta
        # shift
        v_old, v, v_new = v, v_new, v_old
        z, z_new = z_new, z
        w_old, w, w_new = w, w_new, w_old
        gamma_old, gamma = gamma, gamma_new
        s_old, s_cur = s_cur, s_new
        c_old, c_cur = c_cur, c_new
        its += 1
        if verbose and its % 50 == 0:
            info('Stokes MINRES %d: |r|/|r0| = %.3e' % (its, abs(res) / gamma0))
        if gamma == 0.0:
            break
    u = Function(W)
    ops.copy(u.data, x.u)
    p = Function(P)
    ops.copy(p.data, x.p)
    last_solve_info.clear()
    last_solve_info.update(
        outer_iterations=its, inner_iterations=0, inner_solves=0,
        residual=abs(res) / max(gamma0, 1e-300), method='minres',
        viscous_cycle='pmg' if cycles is not None else 'two_level'
        )
    return u, p
