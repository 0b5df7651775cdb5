# -*- coding: utf-8 -*-
'''
Stokes bootstrap (reference flow/stokes.py).

CPU: the oracle's restatement (`oracle.fem_oracle.stokes_solve`) is pinned by
the reference's own known-answer test -- spatial convergence orders > 1.9 for
velocity and pressure on the manufactured solution `Guermond1`, meshes
n = [8, 16], `UnitSquareMesh(n, n, 'left/right')`, Dirichlet data for u AND p on
the whole boundary (reference tests/test_stokes.py:68-118, 121-158).

GPU: `flow_amd.stokes.solve` (Schur-complement CG on the HIP path) against the
oracle on the same inputs, the order test through the drop-in API, and a
Karman-type set of conditions (component-wise velocity data, no pressure data).
'''
import numpy
import pytest
import sympy

from flow_amd import fem
from flow_amd.fem import reference
from flow_amd.fem.bcs import collect
from oracle import fem_oracle as orc

import cases


class Guermond1(object):
    '''u = pi (2 sin(pi y) cos(pi y) sin^2(pi x), -2 sin(pi x) cos(pi x)
    sin^2(pi y)), p = cos(pi x) sin(pi y), mu = 1; f = -mu Lap u + grad p.'''
    mu = 1.0

    def __init__(self):
        X, Y = sympy.symbols('X Y')
        pi = sympy.pi
        u = (+pi * 2 * sympy.sin(pi * Y) * sympy.cos(pi * Y) * sympy.sin(pi * X)**2,
             -pi * 2 * sympy.sin(pi * X) * sympy.cos(pi * X) * sympy.sin(pi * Y)**2)
        p = sympy.cos(pi * X) * sympy.sin(pi * Y)
        assert sympy.simplify(sympy.diff(u[0], X) + sympy.diff(u[1], Y)) == 0
        f = [-self.mu * (sympy.diff(c, X, 2) + sympy.diff(c, Y, 2))
             + sympy.diff(p, v) for c, v in zip(u, (X, Y))]
        lam = lambda e: sympy.lambdify((X, Y), e, 'numpy')
        self._u = [lam(c) for c in u]
        self._p = lam(p)
        self._f = [lam(c) for c in f]

    def u(self, x):
        return numpy.array([numpy.broadcast_to(c(x[0], x[1]), x[0].shape)
                            for c in self._u])

    def p(self, x):
        return numpy.broadcast_to(self._p(x[0], x[1]), x[0].shape)[None, :]

    def f(self, x):
        return numpy.array([numpy.broadcast_to(c(x[0], x[1]), x[0].shape)
                            for c in self._f])


def _oracle_case(problem, n):
    mesh = fem.UnitSquareMesh(n, n, 'left/right')
    W = fem.VectorFunctionSpace(mesh, 'CG', 2)
    P = fem.FunctionSpace(mesh, 'CG', 1)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    Po = orc.Space(mesh.points, mesh.cell_vertices, P.layout.cell_dofs, 1, P.N)
    u_bc = collect([fem.DirichletBC(
        W, fem.Expression(problem.u, degree=5), 'on_boundary')], W.size())
    p_bc = collect([fem.DirichletBC(
        P, fem.Expression(problem.p, degree=5), 'on_boundary')], P.N)
    X = fem.cell_lattice_points(mesh, 5)
    nc, nl = X.shape[:2]
    pts = X.reshape(-1, 2).T
    lat = reference.lattice(5)

    def lattice(fun):
        v = fun(pts)
        return lat, numpy.ascontiguousarray(
            v.reshape(v.shape[0], nc, nl).transpose(1, 2, 0))
    return mesh, W, P, Wo, Po, u_bc, p_bc, lattice


def test_oracle_stokes_orders():
    problem = Guermond1()
    hmax, eu, ep = [], [], []
    for n in (8, 16):
        mesh, W, P, Wo, Po, u_bc, p_bc, lattice = _oracle_case(problem, n)
        u, p = orc.stokes_solve(Wo, Po, lattice(problem.f), problem.mu, u_bc,
                                p_bc)
        lu, lp = lattice(problem.u), lattice(problem.p)
        eu.append(orc.l2_error(Wo, u, lu[0], lu[1], dim=2))
        ep.append(orc.l2_error(Po, p, lp[0], lp[1], dim=1))
        hmax.append(mesh.hmax())
    u_order = orc.order_of_convergence(hmax, eu)[0]
    p_order = orc.order_of_convergence(hmax, ep)[0]
    assert u_order > 1.9, (u_order, eu)
    assert p_order > 1.9, (p_order, ep)


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['minres', 'schur'])
def test_stokes_matches_oracle(hip, method):
    from flow_amd import stokes
    stokes.solver_parameters['method'] = method
    problem = Guermond1()
    mesh, W, P, Wo, Po, u_bc, p_bc, lattice = _oracle_case(problem, 8)
    uo, po = orc.stokes_solve(Wo, Po, lattice(problem.f), problem.mu, u_bc, p_bc)
    WP = fem.FunctionSpace(
        mesh,
        fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
        * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
    bcs = [fem.DirichletBC(WP.sub(0), fem.Expression(problem.u, degree=5),
                           'on_boundary'),
           fem.DirichletBC(WP.sub(1), fem.Expression(problem.p, degree=5),
                           'on_boundary')]
    try:
        u, p = stokes.solve(WP, bcs, problem.mu,
                            fem.Expression(problem.f, degree=5), verbose=False,
                            tol=1.0e-12, max_iter=5000)
    finally:
        stokes.solver_parameters['method'] = 'minres'
    assert cases.rel_l2(u.array(), uo) < 1e-8
    assert cases.rel_l2(p.array(), po) < 1e-7
    # (MINRES with one two-level cycle per block; the Schur-complement CG
    # with nested velocity solves)
    assert stokes.last_solve_info['outer_iterations'] < \
        (1000 if method == 'minres' else 100)


@pytest.mark.gpu
def test_stokes_order(hip):
    '''Counterpart of the reference's tests/test_stokes.py:102-118.'''
    from flow_amd import stokes
    problem = Guermond1()
    rows = []
    for n in (8, 16):
        mesh = fem.UnitSquareMesh(n, n, 'left/right')
        W_el = fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
        P_el = fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1)
        WP = fem.FunctionSpace(mesh, W_el * P_el)
        u_sol = fem.Expression(problem.u, degree=5)
        p_sol = fem.Expression(problem.p, degree=5)
        f = fem.Expression(problem.f, degree=5)
        u_bcs = fem.DirichletBC(WP.sub(0), u_sol, 'on_boundary')
        p_bcs = fem.DirichletBC(WP.sub(1), p_sol, 'on_boundary')
        u_approx, p_approx = stokes.solve(
            WP, bcs=[u_bcs, p_bcs], mu=problem.mu, f=f, verbose=True,
            tol=1.0e-12, max_iter=5000)
        rows.append((mesh.hmax(), fem.errornorm(u_sol, u_approx),
                     fem.errornorm(p_sol, p_approx)))
    hmax, u_errors, p_errors = numpy.array(rows).T
    u_order = numpy.log(u_errors[0] / u_errors[1]) / numpy.log(hmax[0] / hmax[1])
    p_order = numpy.log(p_errors[0] / p_errors[1]) / numpy.log(hmax[0] / hmax[1])
    assert u_order > 1.9
    assert p_order > 1.9


@pytest.mark.gpu
def test_stokes_karman_conditions(hip):
    '''Component-wise velocity data, no pressure data (the bootstrap call of
    the Karman driver, reference tests/test_karman_vortex_street.py:171-179).'''
    from flow_amd import stokes, karman
    prob = karman.KarmanProblem(40, 10)
    mesh = prob.mesh
    WP = fem.FunctionSpace(
        mesh,
        fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
        * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
    W, P = WP.sub(0), WP.sub(1)
    u_bcs = [
        fem.DirichletBC(W, (0.0, 0.0), karman.UpperBoundary()),
        fem.DirichletBC(W, (0.0, 0.0), karman.LowerBoundary()),
        fem.DirichletBC(W, (0.0, 0.0), karman.ObstacleBoundary()),
        fem.DirichletBC(W.sub(0), prob.inflow, karman.LeftBoundary()),
        fem.DirichletBC(W.sub(0), prob.outflow, karman.RightBoundary()),
        ]
    # p = 0 at the outlet fixes the pressure constant (the option the
    # reference keeps commented out at :160-162) so that the direct solve of
    # the oracle is well posed
    p_bcs = [fem.DirichletBC(P, 0.0, karman.RightBoundary())]
    u0, p0 = stokes.solve(WP, u_bcs + p_bcs, 0.002, fem.Constant((0.0, 0.0)),
                          verbose=False, tol=1.0e-11, max_iter=10000)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    Po = orc.Space(mesh.points, mesh.cell_vertices, P.layout.cell_dofs, 1, P.N)
    lat0 = (reference.lattice(0), numpy.zeros((mesh.num_cells(), 1, 2)))
    uo, po = orc.stokes_solve(Wo, Po, lat0, 0.002, collect(u_bcs, W.size()),
                              collect(p_bcs, P.N))
    assert cases.rel_l2(u0.array(), uo) < 1e-6
    assert cases.rel_l2(p0.array(), po) < 1e-6
    # bootstrapping the Navier-Stokes run with it works
    prob.u0.assign(u0)
    prob.p0.assign(p0)
    info = prob.step()
    assert numpy.isfinite(info['unorm'])
