# -*- coding: utf-8 -*-
'''
One pressure-correction step of the Karman channel problem at sizes where the
kernels tile (helper, not a test): the inputs are analytic -- a function of
the generator's arguments alone -- so that the build container (oracle,
minutes to half an hour of sparse LU) and the GPU box (product) work on the
same data without a field travelling between them.  Shared by
tests/golden/make_golden.py --large, tests/test_large_parity.py and the live
oracle comparisons at ~50 k DoF.

Setting = the reference driver's (tests/test_karman_vortex_street.py): channel
[0, 0.6] x [-0.07, 0.07] with the cylinder (:18-23, :35-38), no-slip walls and
obstacle, x-velocity prescribed at both ends, p = 0 at the outlet (:128-145,
:190-203), mu = 0.002 (:167), water density, Rotational scheme (:186), a
CFL-sized step (:272).  The state the step starts from is a smooth perturbed
channel flow (not a solution of anything: the step has to do real work in all
three sub-steps, and the Newton iteration takes more than one iteration).
'''
import numpy

from flow_amd import fem
from flow_amd import karman
from flow_amd.fem import reference
from flow_amd.fem.bcs import collect


CX, CY, RADIUS = 0.1, 0.01, 0.02      # tests/test_karman_vortex_street.py:35-38


def _profile(y):
    return karman.ENTRANCE_VELOCITY * (karman.Y1 - y) * (y - karman.Y0) / (
        0.5 * (karman.Y1 - karman.Y0))**2


def velocity(x, t=0.0):
    '''(2, n): perturbed channel flow that vanishes on the cylinder.'''
    X, Y = x[0], x[1]
    r = numpy.sqrt((X - CX)**2 + (Y - CY)**2)
    mask = 1.0 - numpy.exp(-(numpy.maximum(r - RADIUS, 0.0) / 0.012)**2)
    s = (karman.Y1 - Y) * (Y - karman.Y0) / 0.07**2
    ux = _profile(Y) * (1.0 + 0.3 * numpy.sin(31.0 * X + 2.0 * t)
                        * numpy.cos(45.0 * Y))
    uy = 0.004 * numpy.sin(52.0 * X + 1.0 + 3.0 * t) * s
    return numpy.stack([mask * ux, mask * uy])


def pressure(x):
    X, Y = x[0], x[1]
    return 0.05 * (karman.X1 - X) + 0.01 * numpy.sin(10.0 * X) * numpy.cos(20.0 * Y)


def force(x, t):
    '''A small smooth body force (the driver's is zero, :234-237; a non-zero
    one puts the source assembly under the comparison as well).'''
    X, Y = x[0], x[1]
    return numpy.stack([
        0.02 * numpy.sin(9.0 * X + t) * numpy.cos(14.0 * Y),
        -0.03 * numpy.cos(7.0 * X) * numpy.sin(11.0 * Y + 2.0 * t)])


class KarmanStepCase(object):
    def __init__(self, nx=None, ny=None, vdeg=2, dt=None, mu=0.002,
                 rho=karman.RHO_WATER_293K, fitted=True, mesh=None):
        self.args = dict(nx=nx, ny=ny, vdeg=vdeg, mu=mu, rho=rho, fitted=fitted)
        if mesh is None:
            mesh = fem.karman_channel(nx, ny, fitted=fitted)
        self.mesh = mesh
        self.vdeg = vdeg
        self.mu, self.rho = mu, rho
        self.W = W = fem.VectorFunctionSpace(mesh, 'Lagrange', vdeg)
        self.P = P = fem.FunctionSpace(mesh, 'Lagrange', 1)
        # CFL-sized like the controller's (hmax / max|u|), three digits
        if dt is None:
            dt = float('%.3g' % (mesh.hmax() / 0.0125))
        self.dt = dt
        prof = '%e * (%e - x[1]) * (x[1] - %e) / %e' % (
            karman.ENTRANCE_VELOCITY, karman.Y1, karman.Y0,
            (0.5 * (karman.Y1 - karman.Y0))**2)
        self.inflow = fem.Expression(prof, degree=2)
        self.u_bcs = [
            fem.DirichletBC(W, (0.0, 0.0), karman.UpperBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), karman.LowerBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), karman.ObstacleBoundary()),
            fem.DirichletBC(W.sub(0), self.inflow, karman.LeftBoundary()),
            fem.DirichletBC(W.sub(0), self.inflow, karman.RightBoundary()),
            ]
        self.p_bcs = [fem.DirichletBC(P, 0.0, karman.RightBoundary())]
        self.u0 = velocity(W.layout.dof_coords.T).reshape(-1)
        self.p0 = pressure(P.layout.dof_coords.T)
        self.f0 = fem.Expression(lambda x: force(x, 0.0), degree=2)
        self.f1 = fem.Expression(lambda x: force(x, dt), degree=2)

    def num_dofs(self):
        return self.W.size() + self.P.size()

    def fingerprint(self):
        '''Numbers that change if the generators (mesh, numbering, inputs)
        ever do: stored with a fixture, checked before it is used.'''
        m = self.mesh
        w = numpy.cos(numpy.arange(len(self.u0)) * 0.37)
        return numpy.array([
            m.num_vertices(), m.num_cells(), self.W.N, self.P.N,
            m.points.sum(), numpy.dot(w, self.u0),
            numpy.dot(w[:len(self.p0)], self.p0), self.dt])

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
            vals.reshape(vals.shape[0], nc, nl).transpose(1, 2, 0))

    def bc_data(self):
        return (collect(self.u_bcs, self.W.size()),
                collect(self.p_bcs, self.P.size()))

    def oracle_step(self, method='backward euler', u0=None, p0=None, info=None):
        from oracle import fem_oracle as orc
        W, P = self.oracle_spaces()
        u_bc, p_bc = self.bc_data()
        return orc.step(
            W, P, self.u0 if u0 is None else u0, self.p0 if p0 is None else p0,
            self.lattice(self.f0), self.lattice(self.f1), u_bc, p_bc,
            self.rho, self.mu, self.dt, scheme='rotational', method=method,
            info=info)

    # -- product side ---------------------------------------------------------
    def product_step(self, method='backward euler', tol=1.0e-13, u0=None,
                     p0=None):
        import flow_amd.navier_stokes as navsto
        U0 = fem.Function(self.W)
        U0.set_array(self.u0 if u0 is None else u0)
        P0 = fem.Function(self.P)
        P0.set_array(self.p0 if p0 is None else p0)
        u1, p1 = navsto.Rotational(method).step(
            fem.Constant(self.dt), {0: U0}, P0, self.u_bcs, self.p_bcs,
            fem.Constant(self.rho), fem.Constant(self.mu),
            f={0: self.f0, 1: self.f1}, verbose=False, tol=tol)
        ui = navsto.last_step_info['tentative_velocity']
        return u1.array(), p1.array(), ui.array()


# the configurations of tests/golden/ns_large_*.npz
LARGE = {
    # BASELINE config 2: the 1196 x 279 P1-P1 channel, 0.99 M DoF
    'c2_p1p1': dict(nx=1196, ny=279, vdeg=1),
    # a C3-shaped Taylor-Hood channel large enough that every launch of the
    # step has > 512 CSR-stream tiles (pressure matrix: 84 k rows, 0.58 M
    # nonzeros; P2 scalar pattern: 0.34 M rows, 3.8 M nonzeros), 0.76 M DoF
    'p2p1_600x140': dict(nx=600, ny=140, vdeg=2),
    # a quarter of the headline workload (2182 x 509): 2.5 M DoF, four
    # multigrid levels under the pressure solve
    'p2p1_1091x255': dict(nx=1091, ny=255, vdeg=2),
    }
STRIDE = 87          # every 87th dof of each field is stored (a fixture
                     # says which stride it was written with)


def summary(field, ncomp, stride=STRIDE):
    '''What a fixture keeps of a field: a strided sample, the l2 and max norm
    per component.'''
    f = numpy.asarray(field).reshape(ncomp, -1)
    return (f[:, ::stride].copy(),
            numpy.sqrt((f**2).sum(axis=1)), abs(f).max(axis=1))
