# -*- coding: utf-8 -*-
'''
Glue between the product's host front end (meshes, dof maps, Dirichlet dof
search, Expression lattices: plain data) and the CPU oracle (helper, not a
test).  Mirrors the structure of the reference harness
`compute_time_errors` (tests/test_navier_stokes.py:232-376).
'''
import numpy

from flow_amd import fem
from flow_amd.fem import reference
from flow_amd.fem.bcs import collect
from oracle import fem_oracle as orc


def oracle_space(mesh, degree):
    lay = fem.FunctionSpace(mesh, 'CG', degree).layout
    return orc.Space(mesh.points, mesh.cell_vertices, lay.cell_dofs, degree,
                     lay.N)


def lattice_values(mesh, k, fun):
    '''(lattice_pts, values (Nc, nl, dim)) of fun(x) -> (dim, n).'''
    X = fem.cell_lattice_points(mesh, k)
    nc, nl = X.shape[:2]
    vals = fun(X.reshape(-1, 2).T)
    dim = vals.shape[0]
    return reference.lattice(k), numpy.ascontiguousarray(
        vals.reshape(dim, nc, nl).transpose(1, 2, 0)
        )


def make_mesh(problem, n):
    (x0, y0), (x1, y1) = problem.domain
    return fem.RectangleMesh(
        fem.Point(x0, y0), fem.Point(x1, y1), n, n, problem.diagonal
        )


def channel_bcs(problem, mesh, dt):
    '''Component-wise conditions of the Karman driver on the problem's box:
    no-slip top and bottom, x-velocity on the left and right sides, p = 0 on
    the right (tests/test_karman_vortex_street.py:190-203).'''
    (x0, y0), (x1, y1) = problem.domain
    eps = 1e-12

    class TopBottom(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & ((x[1] < y0 + eps) | (x[1] > y1 - eps))

    class Sides(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & ((x[0] < x0 + eps) | (x[0] > x1 - eps))

    class Right(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[0] > x1 - eps)

    Wv = fem.VectorFunctionSpace(mesh, 'CG', 2)
    Pv = fem.FunctionSpace(mesh, 'CG', 1)
    ux = fem.Expression(lambda x: problem.u(x, dt)[0], degree=problem.u_degree)
    u_bcs = [fem.DirichletBC(Wv, (0.0, 0.0), TopBottom()),
             fem.DirichletBC(Wv.sub(0), ux, Sides())]
    p_bcs = [fem.DirichletBC(Pv, 0.0, Right())]
    return collect(u_bcs, Wv.size()), collect(p_bcs, Pv.N)


def initial_data(problem, mesh, dt):
    '''Inputs of one step from exact data: L2-projected u0, p0
    (tests/test_navier_stokes.py:290-308), boundary data at t = dt (:305),
    forcing at t = 0 and t = dt (:310-311).'''
    W = oracle_space(mesh, 2)
    P = oracle_space(mesh, 1)
    lat_u = lattice_values(mesh, problem.u_degree, lambda x: problem.u(x, 0.0))
    lat_p = lattice_values(mesh, problem.p_degree, lambda x: problem.p(x, 0.0))
    u0 = orc.l2_project(W, lat_u[0], lat_u[1], dim=2)
    p0 = orc.l2_project(P, lat_p[0], lat_p[1], dim=1)
    Wv = fem.VectorFunctionSpace(mesh, 'CG', 2)
    bc = fem.DirichletBC(
        Wv, fem.Expression(lambda x: problem.u(x, dt), degree=problem.u_degree),
        'on_boundary'
        )
    u_bc = collect([bc], Wv.size())
    f0 = lattice_values(mesh, problem.f_degree, lambda x: problem.f(x, 0.0))
    f1 = lattice_values(mesh, problem.f_degree, lambda x: problem.f(x, dt))
    return W, P, u0, p0, u_bc, f0, f1


def errors_after_step(problem, mesh, W, P, u1, p1, dt):
    '''Velocity L2 error and pressure L2 error after shifting p1 by the mean
    error (tests/test_navier_stokes.py:333-360).'''
    k = 5
    lat_u = lattice_values(mesh, k, lambda x: problem.u(x, dt))
    lat_p = lattice_values(mesh, k, lambda x: problem.p(x, dt))
    err_u = orc.l2_error(W, u1, lat_u[0], lat_u[1], dim=2)
    Mp = orc.mass_matrix(P)
    one = numpy.ones(P.N)
    area = one.dot(Mp.dot(one))
    int_exact = orc.load_vector(P, lat_p[0], lat_p[1]).sum()
    alpha = (int_exact - one.dot(Mp.dot(p1))) / area
    err_p = orc.l2_error(P, p1 + alpha, lat_p[0], lat_p[1], dim=1)
    return err_u, err_p


def oracle_time_errors(problem, scheme, method, mesh_sizes, Dt, bc='all'):
    '''bc 'all': velocity data on the whole boundary, Neumann pressure (the
    reference harness); 'channel': channel_bcs above.'''
    errors = {
        'u': numpy.empty((len(mesh_sizes), len(Dt))),
        'p': numpy.empty((len(mesh_sizes), len(Dt))),
        }
    for k, n in enumerate(mesh_sizes):
        mesh = make_mesh(problem, n)
        for j, dt in enumerate(Dt):
            W, P, u0, p0, u_bc, f0, f1 = initial_data(problem, mesh, dt)
            p_bc = None
            if bc == 'channel':
                u_bc, p_bc = channel_bcs(problem, mesh, dt)
            u1, p1, _ = orc.step(
                W, P, u0, p0, f0, f1, u_bc, p_bc, problem.rho, problem.mu, dt,
                scheme=scheme, method=method
                )
            errors['u'][k][j], errors['p'][k][j] = errors_after_step(
                problem, mesh, W, P, u1, p1, dt
                )
    return errors


def orders(Dt, errors):
    return {
        key: numpy.array([orc.order_of_convergence(Dt, row) for row in val])
        for key, val in errors.items()
        }
