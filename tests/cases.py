# -*- coding: utf-8 -*-
'''
Seeded Navier-Stokes step cases shared by the parity tests, the golden-fixture
generator and smoke() (helper, not a test).  A case is plain data: mesh arrays,
dof tables, input fields, boundary data, forcing lattices, parameters.
'''
import numpy

from flow_amd import fem
from flow_amd.fem import reference
from flow_amd.fem.bcs import collect

import mms


class LeftRight(fem.SubDomain):
    def __init__(self, x0, x1):
        self.x0, self.x1 = x0, x1

    def inside(self, x, on_boundary):
        return on_boundary & ((x[0] < self.x0 + 1e-12) | (x[0] > self.x1 - 1e-12))


class TopBottom(fem.SubDomain):
    def __init__(self, y0, y1):
        self.y0, self.y1 = y0, y1

    def inside(self, x, on_boundary):
        return on_boundary & ((x[1] < self.y0 + 1e-12) | (x[1] > self.y1 - 1e-12))


class Right(fem.SubDomain):
    def __init__(self, x1):
        self.x1 = x1

    def inside(self, x, on_boundary):
        return on_boundary & (x[0] > self.x1 - 1e-12)


class Case(object):
    '''One pressure-correction step on a given mesh.

    bc_kind 'all': velocity Dirichlet on the whole boundary, Neumann pressure
    (the setting of tests/test_navier_stokes.py); 'channel': no-slip top/bottom
    walls, x-component only on the left/right sides (component-wise conditions,
    so the exterior-facet terms act on free rows) and p = 0 on the right
    (the setting of tests/test_karman_vortex_street.py:190-203).
    '''
    def __init__(self, mesh, vdeg=2, problem=None, dt=0.05, bc_kind='all',
                 rho=1.0, mu=1.0, f_degree=2, seed=0):
        self.mesh = mesh
        self.vdeg = vdeg
        self.dt = dt
        self.rho = rho
        self.mu = mu
        self.bc_kind = bc_kind
        self.problem = problem if problem is not None else mms.guermond2()
        pb = self.problem
        self.W = fem.VectorFunctionSpace(mesh, 'CG', vdeg)
        self.P = fem.FunctionSpace(mesh, 'CG', 1)
        rng = numpy.random.RandomState(seed)
        xw = self.W.layout.dof_coords.T
        xp = self.P.layout.dof_coords.T
        # nodal interpolants of the exact fields plus a smooth-ish perturbation
        self.u0 = pb.u(xw, 0.0).reshape(-1) \
            + 0.05 * rng.standard_normal(2 * self.W.N)
        self.p0 = pb.p(xp, 0.0).reshape(-1) \
            + 0.05 * rng.standard_normal(self.P.N)
        self.f_degree = f_degree
        self.f0 = fem.Expression(lambda x: pb.f(x, 0.0), degree=f_degree)
        self.f1 = fem.Expression(lambda x: pb.f(x, dt), degree=f_degree)
        uex = fem.Expression(lambda x: pb.u(x, dt), degree=2)
        if bc_kind == 'all':
            self.u_bcs = [fem.DirichletBC(self.W, uex, 'on_boundary')]
            self.p_bcs = []
        else:
            (x0, y0), (x1, y1) = self._bbox()
            ux = fem.Expression(lambda x: pb.u(x, dt)[0], degree=2)
            self.u_bcs = [
                fem.DirichletBC(self.W, (0.0, 0.0), TopBottom(y0, y1)),
                fem.DirichletBC(self.W.sub(0), ux, LeftRight(x0, x1)),
                ]
            self.p_bcs = [fem.DirichletBC(self.P, 0.0, Right(x1))]

    def _bbox(self):
        p = self.mesh.points
        return (p[:, 0].min(), p[:, 1].min()), (p[:, 0].max(), p[:, 1].max())

    # -- oracle side ----------------------------------------------------------
    def oracle_spaces(self):
        from oracle import fem_oracle as orc
        m = self.mesh
        W = orc.Space(m.points, m.cell_vertices, self.W.layout.cell_dofs,
                      self.vdeg, self.W.N)
        P = orc.Space(m.points, m.cell_vertices, self.P.layout.cell_dofs, 1,
                      self.P.N)
        return W, P

    def lattice(self, expr):
        X = fem.cell_lattice_points(self.mesh, expr.degree)
        nc, nl = X.shape[:2]
        vals = expr.eval(X.reshape(-1, 2).T)
        return reference.lattice(expr.degree), numpy.ascontiguousarray(
            vals.reshape(vals.shape[0], nc, nl).transpose(1, 2, 0)
            )

    def bc_data(self):
        u_bc = collect(self.u_bcs, self.W.size())
        p_bc = collect(self.p_bcs, self.P.size()) if self.p_bcs else None
        return u_bc, p_bc

    def oracle_step(self, scheme, method='backward euler'):
        from oracle import fem_oracle as orc
        W, P = self.oracle_spaces()
        u_bc, p_bc = self.bc_data()
        return orc.step(
            W, P, self.u0, self.p0, self.lattice(self.f0), self.lattice(self.f1),
            u_bc, p_bc, self.rho, self.mu, self.dt, scheme=scheme, method=method
            )

    # -- product side ---------------------------------------------------------
    def product_step(self, scheme, method='backward euler', tol=1.0e-13):
        import flow_amd.navier_stokes as navsto
        u0 = fem.Function(self.W)
        u0.set_array(self.u0)
        p0 = fem.Function(self.P)
        p0.set_array(self.p0)
        stepper = {
            'chorin': navsto.Chorin,
            'ipcs': lambda: navsto.IPCS(method),
            'rotational': lambda: navsto.Rotational(method),
            }[scheme]()
        u1, p1 = stepper.step(
            fem.Constant(self.dt), {0: u0}, p0, self.u_bcs, self.p_bcs,
            fem.Constant(self.rho), fem.Constant(self.mu),
            f={0: self.f0, 1: self.f1}, verbose=False, tol=tol
            )
        ui = navsto.last_step_info['tentative_velocity']
        return u1.array(), p1.array(), ui.array()


def rel_l2(a, b):
    return numpy.linalg.norm(a - b) / max(numpy.linalg.norm(b), 1e-300)


def mean_free(p, mass):
    one = numpy.ones(len(p))
    return p - one.dot(mass.dot(p)) / one.dot(mass.dot(one))
