# -*- coding: utf-8 -*-
'''
HIP path vs the CPU oracle at sizes where the kernels actually tile.

The step comparisons of tests/test_hip_parity.py run on a few hundred cells
(3-5 CSR-stream tiles per launch).  Here:

  * LIVE oracle, whole Rotational steps (backward Euler and Crank-Nicolson) of
    the Karman channel problem (reference tests/test_karman_vortex_street.py:
    geometry, conditions, parameters) on the body-fitted 100 x 23 and 160 x 37
    channels (21 k / 53 k DoF: seconds of sparse LU on the GPU box), one heat
    solve and one Stokes solve of ~50 k DoF, and a 12-step trajectory with the
    start vectors of the time loop on, against the oracle stepping the same 12
    steps (reference pressure_correction.py:468-518, harness
    tests/test_navier_stokes.py:232-376);
  * OFFLINE goldens at size (tests/golden/ns_large_*.npz, written by
    `tests/golden/make_golden.py --large` in the build container: minutes of
    sparse LU): BASELINE config 2 (P1-P1 1196 x 279, 0.99 M DoF) and a
    Taylor-Hood channel of 0.75 M DoF -- thousands of tiles per launch, so the
    XCD tile map, the 16-bit column offsets, the packed fp16 streams, 3+-level
    V-cycles and the p-multigrid at the workload's cell Peclet number are all
    under the comparison.  The fixtures hold every 87th dof and the norms of
    the oracle's fields (every 261st at 2.5 M DoF); the inputs are analytic (tests/large_cases.py) and
    rebuilt here, checked against the fixture's fingerprint.

Tolerance: 1e-7 relative l2 (north star: 1e-6), Krylov tolerances 1e-13 as in
the other parity tests.  GPU only.
'''
import os

import numpy
import pytest

from flow_amd import fem
from flow_amd.fem import reference
from flow_amd.fem.bcs import collect
from oracle import fem_oracle as orc

import cases
import large_cases

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _newton_history_matches(got, want):
    '''The Newton path is the oracle's: the same number of iterations, the
    same residual norms where they are above round-off.'''
    assert len(got) == len(want), (got, want)
    for g, w in zip(got, want):
        if w > 1e-11:
            assert abs(g - w) <= 1e-4 * w, (got, want)


@pytest.mark.parametrize('nx,ny', [(100, 23), (160, 37)])
@pytest.mark.parametrize('method', ['backward euler', 'crank-nicolson'])
def test_karman_step_against_the_live_oracle(hip, nx, ny, method):
    import flow_amd.navier_stokes as navsto
    case = large_cases.KarmanStepCase(nx, ny)
    info = {}
    u1o, p1o, uio = case.oracle_step(method, info=info)
    u1, p1, ui = case.product_step(method)
    assert cases.rel_l2(ui, uio) < 1e-7
    assert cases.rel_l2(p1, p1o) < 1e-7
    assert cases.rel_l2(u1, u1o) < 1e-7
    _newton_history_matches(navsto.last_step_info['newton_residuals'],
                            info['newton_history'])
    assert len(info['newton_history']) >= 3      # a step that does real work


@pytest.mark.parametrize('name', sorted(large_cases.LARGE))
@pytest.mark.parametrize('method', ['backward euler', 'crank-nicolson'])
def test_karman_step_against_the_offline_oracle(hip, name, method):
    import flow_amd.navier_stokes as navsto
    path = os.path.join(GOLDEN, 'ns_large_%s.npz' % name)
    if not os.path.exists(path):
        pytest.skip('fixture %s not generated' % os.path.basename(path))
    gold = numpy.load(path)
    case = large_cases.KarmanStepCase(**large_cases.LARGE[name])
    # the generators still produce what the fixture was computed from
    # (sums over ~1e6 terms with cancellation: the order of summation differs
    # between the hosts' numpy builds)
    fp, fpg = case.fingerprint(), gold['fingerprint']
    assert numpy.allclose(fp, fpg, rtol=1e-9, atol=1e-12), (fp, fpg)
    stride = int(gold['stride'])
    key = method.replace(' ', '_').replace('-', '_')
    u1, p1, ui = case.product_step(method)
    for fname, field, ncomp in (('ui', ui, 2), ('p1', p1, 1), ('u1', u1, 2)):
        sample, l2, linf = large_cases.summary(field, ncomp, stride)
        gs = gold['%s_%s_sample' % (key, fname)]
        gl2 = gold['%s_%s_l2' % (key, fname)]
        glinf = gold['%s_%s_linf' % (key, fname)]
        # the sampled dofs: relative l2 distance of the samples, scaled to the
        # field (a sample of every 87th dof of a field of norm |f| has norm
        # ~ |f| / sqrt(87)); and no sampled dof further off than 1e-6 of the
        # field's maximum
        err = numpy.linalg.norm(sample - gs) / numpy.linalg.norm(gs)
        assert err < 1e-7, (name, method, fname, err)
        assert abs(sample - gs).max() < 1e-6 * glinf.max(), (name, fname)
        # all dofs: the norms per component
        assert numpy.allclose(l2, gl2, rtol=1e-8, atol=1e-9 * gl2.max()), \
            (name, fname, l2, gl2)
        assert numpy.allclose(linf, glinf, rtol=1e-7,
                              atol=1e-8 * glinf.max()), (name, fname)
    _newton_history_matches(navsto.last_step_info['newton_residuals'],
                            gold[key + '_newton_history'])
    print('%s, %s: %d DoF, Newton residuals %s, pressure %r, correction %r'
          % (name, method, case.num_dofs(),
             ' '.join('%.2e' % r for r in
                      navsto.last_step_info['newton_residuals']),
             navsto.last_step_info['pressure'],
             navsto.last_step_info['correction']))


def _compare_with_fixture(gold, fields, what):
    stride = int(gold['stride'])
    for fname, field, ncomp in fields:
        sample, l2, linf = large_cases.summary(field, ncomp, stride)
        gs, gl2, glinf = (gold[fname + '_sample'], gold[fname + '_l2'],
                          gold[fname + '_linf'])
        err = numpy.linalg.norm(sample - gs) / numpy.linalg.norm(gs)
        assert err < 1e-7, (what, fname, err)
        assert abs(sample - gs).max() < 1e-6 * glinf.max(), (what, fname)
        assert numpy.allclose(l2, gl2, rtol=1e-8, atol=1e-9 * gl2.max()), \
            (what, fname, l2, gl2)
        assert numpy.allclose(linf, glinf, rtol=1e-7,
                              atol=1e-8 * glinf.max()), (what, fname)


@pytest.mark.parametrize('name', sorted(large_cases.LARGE_BOUSSINESQ))
def test_boussinesq_sweep_against_the_offline_oracle(hip, name):
    '''BASELINE config 4 above toy size: one coupled sweep (heat with the old
    velocity, then Rotational.step with the buoyancy; reference
    tests/test_boussinesq.py:213-253) against the oracle's, computed in the
    build container (tests/golden/make_golden.py --large): temperature excess,
    velocity and mean-free pressure to 1e-7, the Newton residuals to the
    digits the conditioning leaves.'''
    import flow_amd.navier_stokes as navsto
    path = os.path.join(GOLDEN, 'bq_large_%s.npz' % name)
    if not os.path.exists(path):
        pytest.skip('fixture %s not generated' % os.path.basename(path))
    gold = numpy.load(path)
    case = large_cases.BoussinesqSweepCase(**large_cases.LARGE_BOUSSINESQ[name])
    fp, fpg = case.fingerprint(), gold['fingerprint']
    assert numpy.allclose(fp, fpg, rtol=1e-9, atol=1e-12), (fp, fpg)
    theta, u, p = case.product_sweep()
    p = cases.mean_free(p, case.pressure_mass())
    _compare_with_fixture(gold, (('theta', theta - 293.0, 1), ('u', u, 2),
                                 ('p', p, 1)), name)
    _newton_history_matches(navsto.last_step_info['newton_residuals'],
                            gold['newton_history'])
    print('%s: %d DoF, Newton residuals %s, pressure %r' % (
        name, case.num_dofs(),
        ' '.join('%.2e' % r for r in navsto.last_step_info['newton_residuals']),
        navsto.last_step_info['pressure']))


@pytest.mark.parametrize('name', sorted(large_cases.LARGE_STOKES))
def test_stokes_against_the_offline_oracle(hip, name):
    '''The Stokes solver (BASELINE config 5's; reference flow/stokes.py:13-148)
    above toy size against the oracle's sparse LU of the saddle-point system.'''
    path = os.path.join(GOLDEN, 'stokes_large_%s.npz' % name)
    if not os.path.exists(path):
        pytest.skip('fixture %s not generated' % os.path.basename(path))
    gold = numpy.load(path)
    case = large_cases.StokesChannelCase(**large_cases.LARGE_STOKES[name])
    fp, fpg = case.fingerprint(), gold['fingerprint']
    assert numpy.allclose(fp, fpg, rtol=1e-9, atol=1e-12), (fp, fpg)
    u, p = case.product_solve()
    _compare_with_fixture(gold, (('u', u, 2), ('p', p, 1)), name)


def test_twelve_steps_with_start_vectors_against_the_oracle(hip):
    '''A time loop at 53 k DoF: the oracle and the product each step their own
    trajectory 12 times from the same state (fixed step size, the start
    vectors of the product's solves extrapolated from its previous steps:
    the default of mode 'parity'), compared after every step.'''
    import flow_amd.navier_stokes as navsto
    case = large_cases.KarmanStepCase(160, 37)
    assert navsto.solver_parameters['newton']['linear_start'] == 'extrapolated'
    assert navsto.solver_parameters['pressure']['start'] == 'extrapolated'
    navsto.forget_history(case.W)
    uo, po = case.u0, case.p0
    up, pp = case.u0, case.p0
    counts = []
    for k in range(12):
        info = {}
        uo, po, _ = case.oracle_step(u0=uo, p0=po, info=info)
        up, pp, _ = case.product_step(u0=up, p0=pp)
        du, dp = cases.rel_l2(up, uo), cases.rel_l2(pp, po)
        assert du < 1e-7 and dp < 1e-7, (k, du, dp)
        assert len(navsto.last_step_info['newton_residuals']) == \
            len(info['newton_history']), k
        counts.append((sum(navsto.last_step_info['newton_linear_applications']),
                       navsto.last_step_info['pressure'].iterations,
                       navsto.last_step_info['correction'].iterations))
    st = case.W.layout._dev['start_vector_state']
    assert [t.level for t in st.trajectories] == [12]
    # ... and the start vectors were worth something
    early = sum(sum(c) for c in counts[1:4])
    late = sum(sum(c) for c in counts[9:12])
    print('iterations per step (GMRES applications, pressure CG, corrections):',
          counts, 'last du %.1e dp %.1e' % (du, dp))
    assert late < early


class _Hot(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] < 1e-12)


class _Cool(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] > 0.2 - 1e-12)


@pytest.mark.parametrize('supg', [False, True])
def test_heat_solve_at_50k_dofs(hip, supg):
    '''Heat operators (sampled rows) and one implicit Euler step on the
    body-fitted heater box, scalar P2, 51 k DoF (reference flow/heat.py:20-122;
    tests/test_hip_heat.py runs this on 6 cells per side).'''
    from flow_amd import heat, time_steppers
    kappa, rho, cp = 0.6, 998.0, 4182.0
    mesh = fem.heater_box(80, fitted=True)
    Q = fem.FunctionSpace(mesh, 'Lagrange', 2)
    W = fem.VectorFunctionSpace(mesh, 'Lagrange', 2)
    conv = fem.Function(W)
    x = W.layout.dof_coords
    conv.set_array(2.0e-4 * numpy.concatenate([
        -(x[:, 1] - 0.1) * (1.0 + x[:, 0]),
        (x[:, 0] - 0.05) * (1.0 + x[:, 1]**2)]))
    Qo = orc.Space(mesh.points, mesh.cell_vertices, Q.layout.cell_dofs, 2, Q.N)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    bcs = [fem.DirichletBC(Q, 320.0, _Hot()), fem.DirichletBC(Q, 293.0, _Cool())]
    H = heat.Heat(Q, conv, kappa, rho, cp, bcs, fem.Constant(0.0),
                  supg_stabilization=supg)
    Mo, Ao, bo = orc.heat_operators(Qo, Wo, conv.array(), kappa, rho, cp, 0.0,
                                    supg)
    assert Q.N > 45000
    Ah, Mh = H.A.to_scipy(), H.M.to_scipy()
    assert abs(Ah - Ao).max() < 1e-11 * abs(Ao).max()
    assert abs(Mh - Mo).max() < 1e-11 * abs(Mo).max()
    # a smooth temperature field (a warm plume above the heater)
    xq = Q.layout.dof_coords
    u = fem.Function(Q)
    u.set_array(293.0 + 20.0 * numpy.exp(
        -((xq[:, 0] - 0.05)**2 + (xq[:, 1] - 0.09)**2) / 0.02**2))
    dt = 0.5
    u1 = time_steppers.ImplicitEuler(H).step(u, 0.0, dt)
    dofs, vals = collect(bcs, Q.N)
    ref1 = orc.heat_solve(Mo, Ao, 1.0, -dt, Mo.dot(u.array()), dofs, vals)
    # (compared on the excess over 293 K, not on the absolute temperature)
    err = numpy.linalg.norm(u1.array() - ref1) / numpy.linalg.norm(ref1 - 293.0)
    assert err < 1e-7, err


def test_stokes_solve_at_50k_dofs(hip):
    '''flow_amd.stokes.solve (MINRES + block preconditioner) against the
    oracle's direct solve for the Karman driver's bootstrap (reference
    tests/test_karman_vortex_street.py:171-179, flow/stokes.py:13-148) on the
    160 x 37 body-fitted channel.'''
    from flow_amd import stokes, karman
    mesh = fem.karman_channel(160, 37, fitted=True)
    WP = fem.FunctionSpace(
        mesh,
        fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
        * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
    W, P = WP.sub(0), WP.sub(1)
    prof = '%e * (%e - x[1]) * (x[1] - %e) / %e' % (
        karman.ENTRANCE_VELOCITY, karman.Y1, karman.Y0,
        (0.5 * (karman.Y1 - karman.Y0))**2)
    inflow = fem.Expression(prof, degree=2)
    u_bcs = [
        fem.DirichletBC(W, (0.0, 0.0), karman.UpperBoundary()),
        fem.DirichletBC(W, (0.0, 0.0), karman.LowerBoundary()),
        fem.DirichletBC(W, (0.0, 0.0), karman.ObstacleBoundary()),
        fem.DirichletBC(W.sub(0), inflow, karman.LeftBoundary()),
        fem.DirichletBC(W.sub(0), inflow, karman.RightBoundary()),
        ]
    p_bcs = [fem.DirichletBC(P, 0.0, karman.RightBoundary())]
    force = fem.Expression(lambda x: large_cases.force(x, 0.3), degree=2)
    u0, p0 = stokes.solve(WP, u_bcs + p_bcs, 0.002, force, verbose=False,
                          tol=1.0e-13, max_iter=10000)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    Po = orc.Space(mesh.points, mesh.cell_vertices, P.layout.cell_dofs, 1, P.N)
    X = fem.cell_lattice_points(mesh, 2)
    nc, nl = X.shape[:2]
    v = force.eval(X.reshape(-1, 2).T)
    lat = (reference.lattice(2), numpy.ascontiguousarray(
        v.reshape(2, nc, nl).transpose(1, 2, 0)))
    uo, po = orc.stokes_solve(Wo, Po, lat, 0.002, collect(u_bcs, W.size()),
                              collect(p_bcs, P.N))
    assert W.size() + P.N > 50000
    assert cases.rel_l2(u0.array(), uo) < 1e-7
    assert cases.rel_l2(p0.array(), po) < 1e-7


def test_a_step_of_the_developed_street_against_the_oracle(hip):
    '''The regime the headline value is measured in: a Karman problem at the
    driver's viscosity (53 k DoF: under-resolved, cell Peclet ~12 -- the
    Newton systems run on the ILU fallback) stepped by the product from the
    Stokes start until the wake sheds, then ONE step from that state by the
    product and by the oracle (zero forcing, the controller's step size).'''
    import flow_amd.navier_stokes as navsto
    from flow_amd import karman
    prob = karman.KarmanProblem(160, 37)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    its = []
    while prob.t < 75.0 and len(its) < 1500:
        its.append(len(prob.step()['newton_residuals']) - 1)
    u = prob.u0.array().copy()
    p = prob.p0.array().copy()
    n = prob.W.layout.N
    # the wake is unsteady: a cross-stream velocity behind the cylinder of the
    # order of the inflow's, and steps that need more than one Newton iteration
    x = prob.W.layout.dof_coords
    wake = (x[:, 0] > 0.16) & (x[:, 0] < 0.4) & (abs(x[:, 1] - 0.01) < 0.01)
    assert abs(u[n:][wake]).max() > 0.2 * karman.ENTRANCE_VELOCITY, \
        abs(u[n:][wake]).max()
    assert max(its[-50:]) >= 2, its[-50:]
    case = large_cases.KarmanStepCase(160, 37, dt=prob.dt)
    zero = fem.Expression(lambda xx: numpy.zeros((2, xx.shape[1])), degree=2)
    case.f0 = case.f1 = zero
    info = {}
    u1o, p1o, uio = case.oracle_step(u0=u, p0=p, info=info)
    u1, p1, ui = case.product_step(u0=u, p0=p)
    errs = (cases.rel_l2(ui, uio), cases.rel_l2(p1, p1o), cases.rel_l2(u1, u1o))
    print('street state at t = %.1f (dt %.3f, %d steps): rel-L2 vs the oracle ui '
          '%.1e p1 %.1e u1 %.1e; Newton %s' % (
              prob.t, prob.dt, len(its), errs[0], errs[1], errs[2],
              ' '.join('%.2e' % r for r in info['newton_history'])))
    assert max(errs) < 1e-7, errs
    _newton_history_matches(navsto.last_step_info['newton_residuals'],
                            info['newton_history'])
    assert len(info['newton_history']) >= 3
