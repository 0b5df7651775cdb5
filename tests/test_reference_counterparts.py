# -*- coding: utf-8 -*-
'''
Counterparts of the reference's own tests, run through the drop-in API on the
HIP path (GPU only):

  * temporal-order tests of Chorin / IPCS / Rotational on manufactured
    solutions (reference tests/test_navier_stokes.py:379-446), same harness
    steps: project the exact data, one step per (mesh, dt), errornorm of the
    velocity, pressure error after shifting by the mean error (:347-360);
  * sealed box: a hydrostatic state stays at rest, ||u||_inf < 1e-13 after two
    IPCS steps (reference tests/test_sealed_box.py:56-143);
  * Karman channel smoke run: two steps must not raise (reference
    tests/test_karman_vortex_street.py:56,289 -- it asserts nothing either).
'''
import numpy
import pytest

from flow_amd import fem, karman
import flow_amd.navier_stokes as navsto

import mms

pytestmark = pytest.mark.gpu


class _TopBottom(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & ((x[1] < 1e-12) | (x[1] > 1.0 - 1e-12))


class _Sides(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & ((x[0] < 1e-12) | (x[0] > 1.0 - 1e-12))


class _Right(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[0] > 1.0 - 1e-12)


def compute_time_errors(problem, method, mesh_sizes, Dt, bc='all'):
    errors = {'u': numpy.empty((len(mesh_sizes), len(Dt))),
              'p': numpy.empty((len(mesh_sizes), len(Dt)))}
    (x0, y0), (x1, y1) = problem.domain
    for k, n in enumerate(mesh_sizes):
        mesh = fem.RectangleMesh(fem.Point(x0, y0), fem.Point(x1, y1), n, n,
                                 problem.diagonal)
        W = fem.VectorFunctionSpace(mesh, 'CG', 2)
        P = fem.FunctionSpace(mesh, 'CG', 1)
        one = fem.interpolate(fem.Constant(1.0), P)
        mesh_area = fem.integral(one)
        for j, dt in enumerate(Dt):
            sol_u = fem.Expression(lambda x, t: problem.u(x, t),
                                   degree=problem.u_degree, t=0.0)
            sol_p = fem.Expression(lambda x, t: problem.p(x, t),
                                   degree=problem.p_degree, t=0.0)
            rhs0 = fem.Expression(lambda x, t: problem.f(x, t),
                                  degree=problem.f_degree, t=0.0)
            rhs1 = fem.Expression(lambda x, t: problem.f(x, t),
                                  degree=problem.f_degree, t=dt)
            sol_u.t = -dt
            u_1 = fem.project(sol_u, W)
            sol_u.t = 0.0
            u0 = fem.project(sol_u, W)
            p0 = fem.project(sol_p, P)
            sol_u.t = dt
            u_bcs = [fem.DirichletBC(W, sol_u, 'on_boundary')]
            p_bcs = []
            if bc == 'channel':
                # the conditions of the Karman driver on the unit square
                # (tests/test_karman_vortex_street.py:190-203): mms.channel()
                sol_ux = fem.Expression(lambda x: problem.u(x, dt)[0],
                                        degree=problem.u_degree)
                u_bcs = [fem.DirichletBC(W, (0.0, 0.0), _TopBottom()),
                         fem.DirichletBC(W.sub(0), sol_ux, _Sides())]
                p_bcs = [fem.DirichletBC(P, 0.0, _Right())]
            u1, p1 = method.step(
                fem.Constant(dt),
                {-1: u_1, 0: u0}, p0,
                u_bcs=u_bcs, p_bcs=p_bcs,
                rho=fem.Constant(problem.rho), mu=fem.Constant(problem.mu),
                f={0: rhs0, 1: rhs1},
                verbose=False,
                tol=1.0e-10
                )
            sol_p.t = dt
            errors['u'][k][j] = fem.errornorm(sol_u, u1)
            # shift p1 by the mean error: the pressure is determined up to a
            # constant
            alpha = (fem.integral(fem.project(sol_p, P)) - fem.integral(p1)) \
                / mesh_area
            p1.vector()[:] += alpha
            errors['p'][k][j] = fem.errornorm(sol_p, p1)
    return errors


def assert_time_order(problem, method, mesh_sizes, Dt, bc='all'):
    errors = compute_time_errors(problem, method, mesh_sizes, Dt, bc)
    orders = {
        key: numpy.array([
            numpy.log(row[:-1] / row[1:]) / numpy.log(
                numpy.array(Dt[:-1]) / numpy.array(Dt[1:]))
            for row in val])
        for key, val in errors.items()
        }
    assert (orders['u'][:, 0] > method.order['velocity'] - 0.1).all(), orders
    assert (orders['p'][:, 0] > method.order['pressure'] - 0.1).all(), orders


@pytest.mark.parametrize('problem', [mms.flat, mms.guermond1, mms.guermond2])
def test_chorin(hip, problem):
    assert_time_order(problem(), navsto.Chorin(), [16, 32], [1.0e-3, 0.5e-3])


def test_ipcs(hip):
    assert_time_order(mms.guermond2(),
                      navsto.IPCS(time_step_method='backward euler'),
                      [8, 16, 32], [0.5**k for k in range(2)])


def test_rotational(hip):
    assert_time_order(mms.guermond1(),
                      navsto.Rotational(time_step_method='backward euler'),
                      [32, 64], [1.0e-2, 0.5e-2])


@pytest.mark.parametrize('scheme', ['ipcs', 'rotational'])
def test_orders_with_free_boundary_rows(hip, scheme):
    '''The HIP path on mms.channel(): exterior-facet terms on free rows,
    component-wise velocity conditions, Dirichlet pressure branch (the oracle's
    pin of the same name: tests/test_oracle_pinning.py).'''
    method = navsto.IPCS() if scheme == 'ipcs' else navsto.Rotational()
    assert_time_order(mms.channel(), method, [8], [0.1, 0.05, 0.025],
                      bc='channel')


def test_sealed_box(hip, num_steps=2):
    mesh = fem.heater_box(12)
    W = fem.VectorFunctionSpace(mesh, 'Lagrange', 2)
    P = fem.FunctionSpace(mesh, 'Lagrange', 1)
    u_bcs = [fem.DirichletBC(W, (0.0, 0.0), 'on_boundary')]
    p_bcs = []
    mu = 1.0e-3          # water at 293 K (the reference reads `materials`)
    rho = 998.2
    g = -9.81
    u0 = fem.project(fem.Constant([0, 0]), W)
    p0 = fem.project(fem.Expression('g * x[1]', degree=1, g=g), P)
    stepper = navsto.IPCS()
    dt = 1.0e-2
    for _ in range(num_steps):
        u1, p1 = stepper.step(
            fem.Constant(dt),
            {0: u0}, p0,
            u_bcs, p_bcs,
            fem.Constant(rho), fem.Constant(mu),
            f={0: fem.Constant((0.0, g)), 1: fem.Constant((0.0, g))},
            verbose=False,
            tol=1.0e-10
            )
        u0.assign(u1)
        p0.assign(p1)
    unorm = fem.project_magnitude(u0).vector().norm('linf')
    assert unorm < 1.0e-13
    # pure Neumann pressure: the constant the singular system leaves open does
    # not accumulate -- every solve starts from p0 minus its mean and CG stays
    # in the space orthogonal to the constants (sum p = 0 up to what the
    # V-cycle leaks), while the pressure itself is O(g * height)
    pa = p0.array()
    assert abs(pa.sum()) <= 1.0e-6 * len(pa) * abs(pa).max(), pa.sum()
    assert abs(pa).max() > 0.5


def test_karman(hip, num_steps=2):
    prob = karman.KarmanProblem(60, 14)
    prob.set_initial_profile()
    print('Reynolds number:  %e' % prob.reynolds())
    for _ in range(num_steps):
        info = prob.step(tol=1.0e-10)
        assert numpy.isfinite(info['unorm']) and info['unorm'] > 0.0
    assert prob.dt > 1.0e-5           # the controller opened the step size
    assert numpy.isfinite(prob.u0.array()).all()
    # Dirichlet data hold on the final field
    from flow_amd.fem.bcs import collect
    d, v = collect(prob.u_bcs, prob.W.size())
    assert abs(prob.u0.array()[d] - v).max() < 1e-12
