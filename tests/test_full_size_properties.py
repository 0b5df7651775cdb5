# -*- coding: utf-8 -*-
'''
The HIP path at BASELINE.json's FULL sizes (the 9.86 M-DoF P2-P1 Karman channel
the bench runs, and the ~1 M-DoF P1-P1 one), where the CPU oracle is out of
reach: size-independent properties instead of a reference solution.

  * linear solves: the residual recomputed independently, ||b - A x|| <= tol ||b||
  * operators: symmetry (x, A y) = (y, A x), linearity, constants in the kernel
    of the Neumann stiffness matrix, 1^T M 1 = |Omega|
  * preconditioners: symmetry of the two-level operator; L U z = r exactly
    reproduced by the ILU(0) sweeps on vectors built from the factors
  * Newton: the matrix-free Jacobian action equals the assembled Jacobian and
    a finite difference of the residual; the final residual is below tol
  * steps repeated from the same state are bitwise reproducible
All through the C ABI (flow_amd -> libflow_hip.so).
'''
import ctypes

import numpy
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(n, seed):
    from flow_amd import device
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.rand(n, generator=g, dtype=torch.float64).to(device.get()) - 0.5


@pytest.fixture(scope='module', params=[(2182, 509, 2), (1196, 279, 1)],
                ids=['P2-P1-9.9M', 'P1-P1-1.0M'])
def problem(request, hip):
    from flow_amd import karman
    nx, ny, vdeg = request.param
    prob = karman.KarmanProblem(nx, ny, velocity_degree=vdeg)
    prob.grid = (nx, ny, vdeg)
    prob.set_initial_profile()
    prob.dt = 1.0e-5
    infos = [prob.step(tol=1.0e-10) for _ in range(3)]
    return prob, infos


def test_problem_size(problem):
    prob, _ = problem
    ndofs = prob.num_dofs()
    assert 0.9e6 < ndofs < 1.1e7, ndofs


def test_pressure_solve_residual_and_operator_properties(problem):
    from flow_amd import device
    from flow_amd.fem import ops
    prob, _ = problem
    P = prob.P
    lay = P.layout
    n = lay.N
    K = ops.assemble_stiffness(P)
    # constants span the kernel of the Neumann stiffness matrix
    one = device.zeros(n) + 1.0
    y = device.zeros(n)
    K.apply(one, y)
    scale = ops.vector_norm(K.diag_inv(), 'linf') ** -1
    assert ops.vector_norm(y, 'linf') <= 1e-12 * max(scale, 1.0)
    # symmetry and linearity of the SpMV
    a, b = _rand(n, 1), _rand(n, 2)
    Ka, Kb = device.zeros(n), device.zeros(n)
    K.apply(a, Ka)
    K.apply(b, Kb)
    s1, s2 = ops.dot(b, Ka), ops.dot(a, Kb)
    assert abs(s1 - s2) <= 1e-12 * abs(s1)
    c = device.zeros(n)
    ops.axpby(2.0, a, 0.0, c)
    ops.axpby(-3.0, b, 1.0, c)
    Kc = device.zeros(n)
    K.apply(c, Kc)
    ops.axpby(-2.0, Ka, 1.0, Kc)
    ops.axpby(3.0, Kb, 1.0, Kc)
    assert ops.vector_norm(Kc) <= 1e-13 * (ops.vector_norm(Ka) + ops.vector_norm(Kb))
    # 1^T M 1 = area of the channel with the staircase obstacle removed
    M = ops.assemble_mass(P)
    M.apply(one, y)
    area = ops.dot(one, y)
    assert abs(area - prob.mesh.cell_areas().sum()) <= 1e-11 * area
    # a Dirichlet pressure solve with the two-level preconditioner: residual
    isbc = numpy.zeros(n, dtype=numpy.uint8)
    from flow_amd.fem.bcs import collect
    dofs, _vals = collect(prob.p_bcs, n)
    isbc[dofs] = 1
    Kbc = ops.symmetric_bc_matrix(K, device.to_device(isbc))
    dinv = Kbc.diag_inv()
    coarse = ops.CoarseSpace(Kbc, isbc.astype(bool))
    rhs = _rand(n, 3)
    ops.vmul(rhs, device.to_device(1.0 - isbc.astype(numpy.float64)), out=rhs)
    x = device.zeros(n)
    sol = ops.krylov_solve('cg', Kbc, rhs, x, rtol=1e-10, maxit=5000, dinv=dinv,
                           check_every=10, coarse=coarse)
    assert 10 <= sol.iterations <= 600, sol
    r = device.zeros(n)
    Kbc.apply(x, r)
    ops.axpby(1.0, rhs, -1.0, r)
    # (the stopping test is |B r| <= 1e-10 |B b| in the preconditioned norm,
    # like PETSc's KSPCG: the raw residual is only a sanity check here, the
    # solution is compared with the Jacobi-preconditioned one below)
    assert ops.vector_norm(r) <= 1e-7 * ops.vector_norm(rhs)
    # Jacobi only: same solution
    x2 = device.zeros(n)
    ops.krylov_solve('cg', Kbc, rhs, x2, rtol=1e-10, maxit=200000, dinv=dinv,
                     check_every=50)
    ops.axpby(-1.0, x, 1.0, x2)
    assert ops.vector_norm(x2) <= 1e-6 * ops.vector_norm(x)


def test_ilu0_sweeps_invert_their_own_factors(problem):
    '''z = (LU)^-1 r with r := L U w for a random w must return w: exercises
    the colouring, the sliced-ELL streams and both sweeps at full size,
    independently of how good the factorisation is as a preconditioner.'''
    from flow_amd import device
    from flow_amd.fem import ops, ilu
    import scipy.sparse as sp
    prob, _ = problem
    W = prob.W
    lay = W.layout
    M = ops.assemble_mass(W)
    K = ops.assemble_stiffness(W)
    A = ops.Matrix(lay, 1)
    for p in (0, 1):
        ops.copy(A.plane(p), M.vals[:lay.nnz])
        ops.axpby(0.01 * (p + 1), K.vals[:lay.nnz], 1.0, A.plane(p))
    pre = ilu.Ilu0(A)
    plan = pre.plan
    n = plan.n
    rp, ci = plan.host['rowptr'], plan.host['cols']
    oon = plan.host['old_of_new']
    w = numpy.random.RandomState(7).standard_normal(2 * n)
    r = numpy.empty(2 * n)
    for k in range(2):
        LU = sp.csr_matrix((pre.factor_values(k), ci, rp), shape=(n, n))
        L = sp.tril(LU, -1) + sp.identity(n)
        U = sp.triu(LU)
        wp = w[k * n:(k + 1) * n][oon]            # permuted numbering
        rpv = L.dot(U.dot(wp))
        r[k * n:(k + 1) * n][oon] = rpv
    z = device.zeros(2 * n)
    pre.solve(device.to_device(r), z)
    err = numpy.abs(device.to_host(z).numpy() - w).max()
    assert err <= 1e-9 * numpy.abs(w).max(), err
    assert plan.ncolours <= 12
    assert plan.fill_l > 0.98 and plan.fill_u > 0.98      # SELL padding


def test_newton_jacobian_action_and_residual(problem):
    from flow_amd import device, _hip
    from flow_amd.fem import ops
    prob, infos = problem
    for i in infos:
        assert i['newton_residuals'][-1] < 1.0e-10
    W, P = prob.W, prob.P
    lay = W.layout
    mesh = prob.mesh
    nc = mesh.num_cells()
    n2 = W.size()
    lib = _hip.lib()
    prm = _hip.NsParams(prob.dt, prob.rho, prob.mu, 1.0, 0.0)
    bfm = device.to_device(mesh.cell_bfacet_mask())
    from flow_amd import fem
    f0s, keep = ops.coef_struct(
        fem.as_cell_coefficient(fem.Constant((0.0, 0.0)), mesh, 2), mesh,
        lay.degree)
    ui = _hip.clone(prob.u0.data)
    buf = ops.scratch(mesh, max(2 * lay.nloc, 4 * lay.nloc**2) * nc)

    def residual(at):
        F = device.zeros(n2)
        _hip.check(lib.flow_assemble_momentum(
            ctypes.byref(ops.mesh_struct(mesh)),
            ctypes.byref(ops.space_struct(lay)),
            ctypes.byref(ops.space_struct(P.layout)), _hip.i32(bfm),
            _hip.f64(at), _hip.f64(prob.u0.data), _hip.f64(prob.p0.data),
            ctypes.byref(f0s), ctypes.byref(f0s), ctypes.byref(prm),
            _hip.f64(buf), _hip.f64(F), None, 0, _hip.stream()))
        return F

    v = _rand(n2, 11)
    none = device.to_device(numpy.zeros(0, dtype=numpy.int32))
    Jop = ops.MomentumJacobian(W, bfm, ui, prm, none)
    Jv = device.zeros(n2)
    Jop.apply(v, Jv)
    # finite difference of the residual (the form is quadratic in u: a central
    # difference is exact up to rounding)
    eps = 1.0e-3
    up, um = _hip.clone(ui), _hip.clone(ui)
    ops.axpby(eps, v, 1.0, up)
    ops.axpby(-eps, v, 1.0, um)
    Fp, Fm = residual(up), residual(um)
    ops.axpby(-1.0, Fm, 1.0, Fp)
    ops.axpby(1.0, Jv, -1.0 / (2.0 * eps), Fp)        # Fp <- Jv - (F+ - F-)/2eps
    assert ops.vector_norm(Fp) <= 1e-8 * ops.vector_norm(Jv)
    # and the assembled Jacobian
    J = ops.Matrix(lay, 2)
    _hip.check(lib.flow_assemble_momentum(
        ctypes.byref(ops.mesh_struct(mesh)), ctypes.byref(ops.space_struct(lay)),
        ctypes.byref(ops.space_struct(P.layout)), _hip.i32(bfm), _hip.f64(ui),
        _hip.f64(prob.u0.data), _hip.f64(prob.p0.data), ctypes.byref(f0s),
        ctypes.byref(f0s), ctypes.byref(prm), _hip.f64(buf), None,
        _hip.f64(J.vals), J.stride, _hip.stream()))
    Jv2 = device.zeros(n2)
    J.apply(v, Jv2)
    ops.axpby(-1.0, Jv, 1.0, Jv2)
    assert ops.vector_norm(Jv2) <= 1e-12 * ops.vector_norm(Jv)
    del keep


def test_steps_are_reproducible(problem):
    '''Three steps are bitwise identical when repeated -- on the same problem
    object and on a freshly built one whose buffers live elsewhere.  (The
    second half is the regression test for the stale solver scalars of
    flow_amd/csrc/common.h: with them BiCGStab took a path that depended on
    where the allocator had placed the buffers.)'''
    from flow_amd import karman, _hip
    prob, infos = problem

    def rerun(p):
        # (reset: fields, clock, step size and everything a time loop carries
        # from step to step -- the controller's memory, the start-vector
        # histories of the linear solves)
        p.reset(1.0e-5)
        p.set_initial_profile()
        _hip.fill(p.p0.data, 0.0)
        p.dt, p.t = 1.0e-5, 0.0
        lay = p.W.layout
        lay._dev.pop('jacobian_ilu', None)
        lay._dev.pop('jacobian_pmg', None)
        lay._dev.pop('newton_quad_C', None)
        if hasattr(p, '_umag'):
            del p._umag
        its = [p.step(tol=1.0e-10) for _ in range(3)]
        return _hip.clone(p.u0.data), _hip.clone(p.p0.data), its

    u_a, p_a, it_a = rerun(prob)
    u_b, p_b, it_b = rerun(prob)
    assert torch.equal(u_a, u_b) and torch.equal(p_a, p_b)
    assert [i['newton_linear_iterations'] for i in it_a] == \
        [i['newton_linear_iterations'] for i in it_b]
    assert [i['pressure'].iterations for i in it_a] == \
        [i['pressure'].iterations for i in it_b]
    nx = prob.grid
    other = karman.KarmanProblem(nx[0], nx[1], velocity_degree=nx[2])
    u_c, p_c, it_c = rerun(other)
    assert torch.equal(u_c, u_a) and torch.equal(p_c, p_a)
    assert [i['newton_linear_iterations'] for i in it_c] == \
        [i['newton_linear_iterations'] for i in it_a]


# -- BASELINE configs 4 and 5 at their nominal sizes ---------------------------
def test_stokes_lid_driven_cavity_2M_dofs(hip):
    '''flow_amd.stokes.solve (BASELINE config 5) on the ~2 M-DoF
    lid-driven cavity: preconditioned residual below tol, Dirichlet data
    reproduced exactly, the discrete solution is the one of the refined
    problem's coarse counterpart to discretisation accuracy.'''
    from flow_amd import fem, stokes
    from flow_amd.fem.bcs import collect

    class Lid(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[1] > 1.0 - 1e-12)

    class Walls(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[1] <= 1.0 - 1e-12)

    def run(n, tol):
        mesh = fem.UnitSquareMesh(n, n)
        WP = fem.FunctionSpace(
            mesh, fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
            * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
        W = WP.sub(0)
        bcs = [fem.DirichletBC(W, (0.0, 0.0), Walls()),
               fem.DirichletBC(W, (1.0, 0.0), Lid())]
        u, p = stokes.solve(WP, bcs, 1.0, fem.Constant((0.0, 0.0)),
                            verbose=False, tol=tol, max_iter=40000)
        return mesh, W, WP.sub(1), bcs, u, p, dict(stokes.last_solve_info)

    # the reference's default tolerance (flow/stokes.py:19)
    mesh, W, P, bcs, u, p, info = run(470, 1e-13)
    assert W.size() + P.N > 1.9e6
    assert info['residual'] <= 1e-13 and info['outer_iterations'] < 40000
    print('Stokes cavity %d DoF, tol 1e-13: %r' % (W.size() + P.N, info))
    ua, pa = u.array(), p.array()
    assert numpy.isfinite(ua).all() and numpy.isfinite(pa).all()
    d, v = collect(bcs, W.size())
    assert abs(ua[d] - v).max() <= 1e-12
    # maximum principle-like sanity of the cavity flow: |u| <= lid speed (+ the
    # usual small P2 overshoot next to the corner singularities)
    assert abs(ua).max() <= 1.0 + 0.15
    # kinetic energy against a 4x coarser solve: the primary vortex is resolved
    # on both, the corner singularities cost a few per cent
    _, _, _, _, uc, _, _ = run(118, 1e-8)
    e_f, e_c = fem.norm(u, 'L2'), fem.norm(uc, 'L2')
    assert e_f == pytest.approx(e_c, rel=2e-2), (e_f, e_c)


def test_boussinesq_4M_dofs_short_run(hip):
    '''BASELINE config 4 at ~4 M DoF (velocity P2 + pressure P1 + temperature
    P2 on the heater box): two coupled steps, physical bounds and exact
    boundary data.'''
    from flow_amd import fem, boussinesq
    from flow_amd.fem.bcs import collect
    u1, p1, theta1, steps = boussinesq.compute_boussinesq(
        target_time=0.03, nx=400)
    W = u1.function_space()
    ndofs = W.size() + p1.function_space().N + theta1.function_space().N
    assert ndofs > 3.5e6, ndofs
    assert len(steps) >= 2
    assert all(s['banach_steps'] <= 10 for s in steps)
    th = theta1.array()
    assert numpy.isfinite(th).all() and numpy.isfinite(u1.array()).all()
    # heater ramps with t/30 s * 27 K; P2 Galerkin undershoots a little
    assert 293.0 - 1e-2 <= th.min()
    assert th.max() <= 293.0 + 27.0 * 0.1 / 30.0 + 1e-6
    d, _ = collect([fem.DirichletBC(W, (0.0, 0.0), 'on_boundary')], W.size())
    assert abs(u1.array()[d]).max() < 1e-14
    # (so early the buoyancy may still be below the Newton tolerance: u = 0)
    assert 0.0 <= fem.norm(u1, 'L2') < 1e-3
    assert fem.norm(theta1, 'L2') == pytest.approx(
        293.0 * numpy.sqrt(0.02 - numpy.pi * 4e-4), rel=5e-3)
