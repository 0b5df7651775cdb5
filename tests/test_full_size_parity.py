# -*- coding: utf-8 -*-
'''
Parity of ONE time step at BASELINE.json's full size (the 9.86 M-DoF P2-P1
Karman channel the bench runs), where the CPU oracle is out of reach.

The reference solves F1(ui) = 0 with Newton from ui = u0, exact (LU) steps,
stopped at the first iterate with ||F||_2 < 1e-10
(flow/navier_stokes/pressure_correction.py:204-254).  At this size that
tolerance is loose (||F|| = 1e-10 corresponds to ~3e-4 relative in the
velocity), so the iterate the reference accepts is a specific point: the
yardstick here is the same Newton path with linear systems solved 1e4 times
tighter than the default (residual 1e-9 of the Newton tolerance) and the
pressure / correction systems solved to 1e-13 instead of 1e-10.  The default
('parity') mode must reproduce that step to the north-star tolerance of 1e-6
relative l2 in u AND p -- asserted with a margin -- on a start-up step
(dt ~ 4e-5, the flow still impulsive) and on a CFL-sized one (dt ~ 1e-2).
'''
import numpy
import pytest

pytestmark = pytest.mark.gpu

NORTH_STAR = 1.0e-6


def _rel(a, b):
    return float(numpy.linalg.norm(a - b) / numpy.linalg.norm(b))


@pytest.fixture(scope='module')
def problem(hip):
    from flow_amd import karman
    import flow_amd.navier_stokes as navsto
    assert navsto.solver_parameters['mode'] == 'parity'
    prob = karman.KarmanProblem(2182, 509, velocity_degree=2)
    prob.set_initial_profile()
    prob.dt = 1.0e-5
    return prob


def _one_step(prob, state, tol, **newton):
    import flow_amd.navier_stokes as navsto
    npar = navsto.solver_parameters['newton']
    saved = dict(npar)
    u_s, p_s, dt, t = state
    prob.u0.vector().set_local(u_s)
    prob.p0.vector().set_local(p_s)
    prob.dt, prob.t = dt, t
    try:
        npar.update(newton)
        info = prob.step(tol=tol, adapt=False)
    finally:
        npar.clear()
        npar.update(saved)
    return (prob.u0.vector().get_local().copy(),
            prob.p0.vector().get_local().copy(), info)


@pytest.mark.parametrize('warm', [2, 14], ids=['start-up', 'cfl-sized'])
def test_default_step_matches_exact_newton_step(problem, warm):
    import flow_amd.navier_stokes as navsto
    prob = problem
    while getattr(prob, 'warm_steps', 0) < warm:
        prob.step()
        prob.warm_steps = getattr(prob, 'warm_steps', 0) + 1
    state = (prob.u0.vector().get_local().copy(),
             prob.p0.vector().get_local().copy(), prob.dt, prob.t)
    # yardstick: the reference's Newton path with (almost) exact solves
    u_y, p_y, info_y = _one_step(prob, state, 1.0e-13,
                                 linear_atol_factor=1.0e-9)
    # the default mode
    assert navsto.solver_parameters['mode'] == 'parity'
    assert navsto.solver_parameters['newton']['initial_guess'] == 'previous'
    u_d, p_d, info_d = _one_step(prob, state, 1.0e-10)
    res_y, res_d = info_y['newton_residuals'], info_d['newton_residuals']
    # a Newton SOLVE: at least one correction, from u0, the same number of
    # iterations as the exact path, stopped by the reference's test
    assert len(res_d) >= 2 and len(res_d) == len(res_y), (res_d, res_y)
    assert res_d[-1] < 1.0e-10 <= res_d[-2]
    assert abs(res_d[0] - res_y[0]) <= 1e-12 * res_y[0]      # same start
    du, dp = _rel(u_d, u_y), _rel(p_d, p_y)
    print('%d steps in, dt %.2e: default vs exact Newton step: du %.2e dp %.2e'
          ' (GMRES applications %r vs %r)'
          % (warm, state[2], du, dp, info_d['newton_linear_applications'],
             info_y['newton_linear_applications']))
    assert du < 0.3 * NORTH_STAR and dp < 0.3 * NORTH_STAR, (du, dp)
    # mode 'fast' for comparison: legal by every stopping test, but a
    # different Newton iterate (reported, and bounded loosely)
    navsto.set_mode('fast')
    try:
        u_f, p_f, info_f = _one_step(prob, state, 1.0e-10)
    finally:
        navsto.set_mode('parity')
    print('   mode fast: du %.2e dp %.2e' % (_rel(u_f, u_y), _rel(p_f, p_y)))
    assert info_f['newton_residuals'][-1] < 1.0e-10
    assert _rel(u_f, u_y) < 1.0e-3 and _rel(p_f, p_y) < 1.0e-2
    # leave the problem on the default path for the next regime
    prob.u0.vector().set_local(state[0])
    prob.p0.vector().set_local(state[1])
    prob.dt, prob.t = state[2], state[3]


def test_step_of_the_bench_window_matches_exact_newton_step(problem):
    '''The regime the headline number is measured in: the flow started from
    the Stokes solution (tests/test_karman_vortex_street.py:171-179 of the
    reference) with the step size at its CFL plateau (dt ~ 0.024): one Newton
    iteration of 17-18 GMRES applications per step.  Same comparison as
    above.'''
    import flow_amd.navier_stokes as navsto
    prob = problem
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    for _ in range(20):
        prob.step()
    assert prob.dt > 1.0e-2
    state = (prob.u0.vector().get_local().copy(),
             prob.p0.vector().get_local().copy(), prob.dt, prob.t)
    u_y, p_y, info_y = _one_step(prob, state, 1.0e-13,
                                 linear_atol_factor=1.0e-9)
    assert navsto.solver_parameters['mode'] == 'parity'
    u_d, p_d, info_d = _one_step(prob, state, 1.0e-10)
    res_y, res_d = info_y['newton_residuals'], info_d['newton_residuals']
    assert len(res_d) >= 2 and len(res_d) == len(res_y), (res_d, res_y)
    assert res_d[-1] < 1.0e-10 <= res_d[-2]
    du, dp = _rel(u_d, u_y), _rel(p_d, p_y)
    print('Stokes start, dt %.2e: default vs exact Newton step: du %.2e dp %.2e'
          ' (GMRES applications %r vs %r)'
          % (state[2], du, dp, info_d['newton_linear_applications'],
             info_y['newton_linear_applications']))
    assert du < 0.3 * NORTH_STAR and dp < 0.3 * NORTH_STAR, (du, dp)


def test_repeated_solves_do_not_drift(problem):
    '''The linear solvers decide convergence on the device and freeze the
    solution there: a solve asked for more iterations than it needs (the
    host's first read-back is placed from the previous call's count) returns
    the same solution.'''
    from flow_amd import device
    from flow_amd.fem import ops
    prob = problem
    W = prob.W
    M = ops.assemble_mass(W.collapse())
    dinv = M.diag_inv()
    n = W.layout.N
    b = device.to_device(numpy.sin(numpy.arange(n, dtype=float)))
    x1 = device.zeros(n)
    s1 = ops.krylov_solve('cg', M, b, x1, 1e-10, maxit=1000, dinv=dinv,
                          check_every=2)
    x2 = device.zeros(n)
    s2 = ops.krylov_solve('cg', M, b, x2, 1e-10, maxit=1000, dinv=dinv,
                          check_every=2, first_check=s1.iterations + 40)
    assert s2.iterations == s1.iterations, (s1, s2)
    ops.axpby(-1.0, x1, 1.0, x2)
    assert ops.vector_norm(x2, 'linf') == 0.0
