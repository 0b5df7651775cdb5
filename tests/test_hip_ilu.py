# -*- coding: utf-8 -*-
'''
K11: multicolour ILU(0) (flow_amd/fem/ilu.py + flow_amd/csrc/ilu_kernels.hip).
GPU tests: the factor equals a textbook IKJ ILU(0) of the colour-permuted
matrix computed on the host, the sweeps equal host triangular solves, and
BiCGStab + ILU(0) solves a convection-dominated heat system on which
BiCGStab + Jacobi does not converge (the reference uses LU there:
"The Krylov solver doesn't converge", flow/heat.py:116).  CPU test: colouring.
'''
import numpy
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from flow_amd import fem
from flow_amd.fem import ilu

import cases


def _ilu0_reference(A):
    '''IKJ ILU(0) on a scipy CSR with sorted indices (host, loops).'''
    A = A.tocsr().copy()
    A.sort_indices()
    ip, ix, v = A.indptr, A.indices, A.data.copy()
    n = A.shape[0]
    diag = numpy.array([ip[i] + numpy.searchsorted(ix[ip[i]:ip[i + 1]], i)
                        for i in range(n)])
    for i in range(n):
        for p in range(ip[i], diag[i]):
            k = ix[p]
            v[p] /= v[diag[k]]
            pos = {ix[q]: q for q in range(diag[k] + 1, ip[k + 1])}
            for t in range(p + 1, ip[i + 1]):
                q = pos.get(ix[t])
                if q is not None:
                    v[t] -= v[p] * v[q]
    return sp.csr_matrix((v, ix, ip), shape=A.shape), diag


def test_colouring_is_proper():
    for mesh in (fem.UnitSquareMesh(7, 5, 'crossed'), fem.karman_channel(30, 8)):
        for deg in (1, 2):
            lay = fem.FunctionSpace(mesh, 'CG', deg).layout
            rp, ci = lay.pattern('rowptr'), lay.pattern('cols')
            colour, nc = ilu.colour_graph(rp, ci)
            assert colour.min() == 0 and colour.max() == nc - 1
            rows = numpy.repeat(numpy.arange(lay.N), numpy.diff(rp))
            off = rows != ci
            assert (colour[rows[off]] != colour[ci[off]]).all()
            assert nc <= 40
            # iterated greedy (optional): still proper, never more colours
            c2, nc2 = ilu.colour_graph(rp, ci, rounds=6)
            assert nc2 <= nc and c2.min() == 0 and c2.max() == nc2 - 1
            assert (c2[rows[off]] != c2[ci[off]]).all()


@pytest.mark.gpu
def test_factor_and_sweeps_match_host_ilu0(hip):
    from flow_amd import device
    from flow_amd.fem import ops
    rng = numpy.random.RandomState(0)
    mesh = fem.karman_channel(20, 6)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    M = ops.assemble_mass(V)
    K = ops.assemble_stiffness(V)
    pert = device.to_device(0.1 * rng.standard_normal(M.vals.numel())) \
        * float(M.vals.abs().max())
    A = ops.Matrix(lay, 0, (M.vals + 0.002 * K.vals + pert * (M.vals != 0)))
    pre = ilu.Ilu0(A)
    plan = pre.plan
    # host reference on the permuted matrix
    P = sp.csr_matrix(
        (numpy.ones(lay.N), (numpy.arange(lay.N), plan.host['old_of_new'])),
        shape=(lay.N, lay.N))
    Ap = (P @ A.to_scipy() @ P.T).tocsr()
    Ap.sort_indices()
    assert numpy.array_equal(Ap.indices, plan.host['cols'])
    LUref, diag = _ilu0_reference(Ap)
    lu = pre.factor_values()
    assert abs(lu - LUref.data).max() < 1e-12 * abs(LUref.data).max()
    # sweeps
    r = rng.standard_normal(lay.N)
    z = device.zeros(lay.N)
    pre.solve(device.to_device(r), z)
    L = sp.tril(LUref, -1) + sp.identity(lay.N)
    U = sp.triu(LUref)
    rp = r[plan.host['old_of_new']]
    zp = spla.spsolve_triangular(
        U.tocsr(), spla.spsolve_triangular(L.tocsr(), rp, lower=True),
        lower=False)
    zref = numpy.empty(lay.N)
    zref[plan.host['old_of_new']] = zp
    assert cases.rel_l2(z.cpu().numpy(), zref) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('kind', [0, 1])
def test_packed_sweeps_apply_the_rounded_factors(hip, kind):
    '''flow_ilu.packed: the sweeps read the factors rounded to fp32 (blocks
    interleaved) and compute in fp64 -- the result is the EXACT inverse of the
    rounded factors (a fixed linear operator), to fp64 round-off, and close to
    that of the unrounded ones.'''
    from flow_amd import device
    from flow_amd.fem import ops
    import torch
    rng = numpy.random.RandomState(3)
    mesh = fem.karman_channel(24, 7)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    n = lay.N
    M = ops.assemble_mass(V)
    K = ops.assemble_stiffness(V)
    scale = float(M.vals.abs().max())
    planes = []
    for _ in range(kind + 1):
        pert = device.to_device(0.1 * rng.standard_normal(M.vals.numel())) * scale
        planes.append(M.vals + 0.002 * K.vals + pert * (M.vals != 0))
    A = ops.Matrix(lay, kind, planes[0] if kind == 0 else torch.cat(planes))
    full = ilu.Ilu0(A)
    packed = ilu.Ilu0(A, packed=True)
    plan = packed.plan
    nb = kind + 1
    r = rng.standard_normal(nb * n)
    z64 = device.zeros(nb * n)
    z32 = device.zeros(nb * n)
    full.solve(device.to_device(r), z64)
    packed.solve(device.to_device(r), z32)
    z64 = device.to_host(z64).numpy()
    z32 = device.to_host(z32).numpy()
    ip, ix = plan.host['rowptr'], plan.host['cols']
    diag = plan.host['diag']
    perm = plan.host['old_of_new']
    for b in range(nb):
        lu = packed.factor_values(b)
        rounded = lu.astype(numpy.float32).astype(numpy.float64)
        rounded[diag] = 1.0 / (1.0 / lu[diag])      # pivots stay fp64 (1/d)
        LU = sp.csr_matrix((rounded, ix, ip), shape=(n, n))
        L = sp.tril(LU, -1) + sp.identity(n)
        U = sp.triu(LU)
        rp = r[b * n:(b + 1) * n][perm]
        zp = spla.spsolve_triangular(
            U.tocsr(), spla.spsolve_triangular(L.tocsr(), rp, lower=True),
            lower=False)
        zref = numpy.empty(n)
        zref[perm] = zp
        zb = z32[b * n:(b + 1) * n]
        assert abs(zb - zref).max() < 1e-11 * abs(zref).max()
        d = abs(zb - z64[b * n:(b + 1) * n]).max() / abs(zref).max()
        assert 0.0 < d < 1e-5
    # flow_ilu.single_vector: the sweep vector in fp32 too (fp64 row sums) --
    # the same preconditioner to fp32 accuracy, for the flexible GMRES
    single = ilu.Ilu0(A, packed=True, single_vector=True)
    zs = device.zeros(nb * n)
    single.solve(device.to_device(r), zs)
    zs = device.to_host(zs).numpy()
    d = abs(zs - z32).max() / abs(z32).max()
    assert 0.0 < d < 2e-5, d


@pytest.mark.gpu
def test_bicgstab_ilu0_on_convection_dominated_heat(hip):
    '''The regime where Jacobi fails (tests/test_hip_heat.py keeps to the
    diffusion-resolved one): cell Peclet number ~ 600.'''
    from flow_amd import heat, time_steppers
    from flow_amd.fem.bcs import collect
    from oracle import fem_oracle as orc
    import test_hip_heat as th
    kappa, rho, cp = 0.6, 998.0, 4182.0
    mesh, Q, W, conv, Qo, Wo, bcs = th._setup(2, 2, 1.0e-2)
    H = heat.Heat(Q, conv, kappa, rho, cp, bcs, fem.Constant(0.0))
    Mo, Ao, bo = orc.heat_operators(Qo, Wo, conv.array(), kappa, rho, cp, 0.0,
                                    False)
    rng = numpy.random.RandomState(0)
    u = fem.Function(Q)
    u.set_array(293.0 + rng.standard_normal(Q.N))
    dt = 0.5
    u1 = time_steppers.ImplicitEuler(H).step(u, 0.0, dt)
    info = heat.last_solve_info['heat']
    # (cell Peclet ~600: the self-test of the heat solve's fallback keeps the
    # bare ILU(0) -- the P1 rediscretisation of a 2 x 2 mesh is no coarse level)
    assert 'ilu0' in info.method and info.iterations < 2000, info
    assert heat.last_solve_info['heat_preconditioner'] == 'ilu0'
    dofs, vals = collect(bcs, Q.N)
    ref1 = orc.heat_solve(Mo, Ao, 1.0, -dt, Mo.dot(u.array()), dofs, vals)
    assert cases.rel_l2(u1.array(), ref1) < 1e-6


@pytest.mark.gpu
def test_newton_with_ilu0_preconditioner(hip):
    '''ILU(0) of the two diagonal Jacobian blocks as the BiCGStab
    preconditioner of the tentative-velocity Newton solve: same converged
    step as with Jacobi.'''
    import flow_amd.navier_stokes as navsto
    mesh = fem.karman_channel(30, 10)
    case = cases.Case(mesh, vdeg=2, dt=0.05, bc_kind='channel', rho=1.5,
                      mu=0.05, seed=3)
    default = navsto.solver_parameters['newton']['preconditioner']
    try:
        navsto.solver_parameters['newton']['preconditioner'] = 'jacobi'
        u1j, p1j, uij = case.product_step('rotational')
        its_j = sum(navsto.last_step_info['newton_linear_iterations'])
        navsto.solver_parameters['newton']['preconditioner'] = 'ilu0'
        u1i, p1i, uii = case.product_step('rotational')
        its_i = sum(navsto.last_step_info['newton_linear_iterations'])
    finally:
        navsto.solver_parameters['newton']['preconditioner'] = default
    # both runs stop at the same Newton tolerance: agreement to solver accuracy
    assert cases.rel_l2(uii, uij) < 1e-7
    assert cases.rel_l2(u1i, u1j) < 1e-7
    assert its_i < its_j, (its_i, its_j)


@pytest.mark.gpu
def test_newton_gmres_and_bicgstab_agree(hip):
    '''The Newton systems solved with GMRES (default) or BiCGStab, both with
    the ILU(0) preconditioner: the same converged step, GMRES with no more
    operator applications.'''
    import flow_amd.navier_stokes as navsto
    mesh = fem.karman_channel(30, 10)
    case = cases.Case(mesh, vdeg=2, dt=0.05, bc_kind='channel', rho=1.5,
                      mu=0.05, seed=3)
    npar = navsto.solver_parameters['newton']
    default = npar['linear_solver']
    out = {}
    try:
        for solver in ('bicgstab', 'gmres'):
            npar['linear_solver'] = solver
            u1, p1, ui = case.product_step('rotational')
            out[solver] = (u1, p1, ui, sum(
                navsto.last_step_info['newton_linear_applications']))
    finally:
        npar['linear_solver'] = default
    assert default == 'gmres'
    for k in range(3):
        assert cases.rel_l2(out['gmres'][k], out['bicgstab'][k]) < 1e-7
    assert out['gmres'][3] <= out['bicgstab'][3], (out['gmres'][3],
                                                  out['bicgstab'][3])


@pytest.mark.gpu
def test_start_vectors_do_not_change_the_solution(hip):
    '''Mode 'fast' starts its solves from extrapolated fields (Newton start
    candidates, pressure, velocity correction, CFL projection) and solves the
    Newton systems less tightly.  Those are start vectors and tolerances
    only: the same run in mode 'parity' (none of them) ends at the same
    fields to the accuracy the Newton tolerance leaves, step sizes included.'''
    import flow_amd.navier_stokes as navsto
    from flow_amd import karman

    def run(tricks):
        navsto.set_mode('fast' if tricks else 'parity')
        try:
            prob = karman.KarmanProblem(150, 35)
            prob.extrapolate_projection = tricks
            prob.set_initial_profile()
            starts = set()
            for _ in range(26):
                info = prob.step()
                starts.add(info.get('initial_guess', 'u0'))
            return (prob.u0.vector().get_local(), prob.p0.vector().get_local(),
                    prob.t, starts)
        finally:
            navsto.set_mode('parity')
    u_a, p_a, t_a, starts = run(True)
    u_b, p_b, t_b, _ = run(False)
    assert starts - {'u0'}, starts          # the candidates were exercised
    # (the step sizes come from a projection solved to 1e-7, so the two runs
    # drift apart by ~1e-6 in dt per step; the fields are compared at those
    # slightly different times)
    diffs = (abs(t_a - t_b) / t_b, cases.rel_l2(u_a, u_b), cases.rel_l2(p_a, p_b))
    assert diffs[0] < 5e-5 and diffs[1] < 5e-5 and diffs[2] < 3e-4, diffs
    print('start vectors on/off: dt, u, p differences', diffs)


@pytest.mark.gpu
def test_prepare_builds_the_caches_and_leaves_no_trace_in_the_fields(hip):
    '''KarmanProblem.prepare() (bench.py: one throw-away step before the timed
    windows) creates the cached structures -- the preconditioner of the Newton
    systems among them -- and
    resets fields and clock; the steps that follow agree with those of a problem
    that was never prepared to the solvers' tolerance (caches and
    preconditioners only change Krylov paths).'''
    from flow_amd import karman
    a = karman.KarmanProblem(120, 28, velocity_degree=2)
    b = karman.KarmanProblem(120, 28, velocity_degree=2)
    a.prepare()
    lay = a.W.layout
    assert lay._dev['jacobian_pmg'].stale
    assert a.t == 0.0 and a.dt == 1.0e-5 and a.history == []
    assert float(a.u0.data.abs().max()) == 0.0
    assert float(a.p0.data.abs().max()) == 0.0
    for prob in (a, b):
        prob.set_initial_profile()
        prob.dt = 2.0e-3
        for _ in range(3):
            prob.step(adapt=False)
    ua, ub = a.u0.data.cpu().numpy(), b.u0.data.cpu().numpy()
    pa, pb = a.p0.data.cpu().numpy(), b.p0.data.cpu().numpy()
    assert cases.rel_l2(ua, ub) < 1e-9
    assert cases.rel_l2(pa, pb) < 1e-7
