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
                 rho=karman.RHO_WATER_293K, fitted=True, mesh=None,
                 linear='lu'):
        '''linear: how the ORACLE solves its Newton and mass systems -- one
        sparse LU ('lu') or, beyond what SuperLU factors whole, the same
        solution through the LUs of the diagonal blocks ('block',
        fem_oracle.solve_blockwise).'''
        self.args = dict(nx=nx, ny=ny, vdeg=vdeg, mu=mu, rho=rho, fitted=fitted)
        self.linear = linear
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

    def oracle_step(self, method='backward euler', u0=None, p0=None, info=None,
                    linear=None):
        from oracle import fem_oracle as orc
        linear = self.linear if linear is None else linear
        W, P = self.oracle_spaces()
        u_bc, p_bc = self.bc_data()
        return orc.step(
            W, P, self.u0 if u0 is None else u0, self.p0 if p0 is None else p0,
            self.lattice(self.f0), self.lattice(self.f1), u_bc, p_bc,
            self.rho, self.mu, self.dt, scheme='rotational', method=method,
            info=info, linear=linear)

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
    # half of the headline workload: 4.9 M DoF.  SuperLU gives up on the
    # coupled Newton matrix here ("not enough memory to perform factorization"
    # after 25 minutes, with 40 GB still free: the index range of its
    # factors); the oracle reaches the same discrete solution through the LUs
    # of the two diagonal blocks (fem_oracle.solve_blockwise, `linear='block'`)
    'p2p1_1543x360': dict(nx=1543, ny=360, vdeg=2, linear='block'),
    # THE headline workload (BASELINE config 3 on one GPU): 9.87 M DoF, the
    # same block-wise oracle solve with the block LUs held in fp32 (they only
    # precondition; the solution is the fp64 one to a true residual of 1e-14):
    # what fits the build container's 62 GB
    'p2p1_2182x509': dict(nx=2182, ny=509, vdeg=2, linear='block32'),
    }
STRIDE = 87          # every 87th dof of each field is stored (a fixture
                     # says which stride it was written with)


def summary(field, ncomp, stride=STRIDE):
    '''What a fixture keeps of a field: a strided sample, the l2 and max norm
    per component.'''
    f = numpy.asarray(field).reshape(ncomp, -1)
    return (f[:, ::stride].copy(),
            numpy.sqrt((f**2).sum(axis=1)), abs(f).max(axis=1))


class BoussinesqSweepCase(object):
    '''BASELINE config 4 at a size where the kernels tile: ONE fixed-point
    sweep of a Boussinesq time step (reference tests/test_boussinesq.py:213-253
    -- implicit Euler on Heat with the old velocity, then Rotational.step with
    the buoyancy rho(theta) g) on the body-fitted heater box, from an analytic
    perturbed state (a warm plume above the heater, a weak swirl that vanishes
    on the walls), as tests/test_boussinesq_counterpart.py runs it on 12 cells
    per side.'''
    def __init__(self, n, dt=0.05, t=12.0):
        from flow_amd import boussinesq
        self.args = dict(n=n, dt=dt, t=t)
        self.mesh = mesh = fem.heater_box(n, fitted=True)
        self.pb = pb = boussinesq.HeaterBox(mesh)
        self.dt, self.t = dt, t
        xq = pb.Q.layout.dof_coords
        xw = pb.W.layout.dof_coords
        self.theta0 = 293.0 + 8.0 * numpy.exp(
            -((xq[:, 0] - 0.05)**2 + (xq[:, 1] - 0.09)**2) / 4e-4)
        bump = numpy.sin(numpy.pi * xw[:, 0] / 0.1) \
            * numpy.sin(numpy.pi * xw[:, 1] / 0.2)
        u = 1.0e-3 * numpy.concatenate([-(xw[:, 1] - 0.1) * bump,
                                        (xw[:, 0] - 0.05) * bump])
        self.u_bc = collect(pb.no_slip, pb.W.size())
        u[self.u_bc[0]] = self.u_bc[1]
        self.u0 = u

    def num_dofs(self):
        return self.pb.W.size() + self.pb.P.size() + self.pb.Q.size()

    def fingerprint(self):
        m = self.mesh
        w = numpy.cos(numpy.arange(len(self.u0)) * 0.37)
        return numpy.array([
            m.num_vertices(), m.num_cells(), self.pb.W.N, self.pb.P.N,
            m.points.sum(), numpy.dot(w, self.u0),
            numpy.dot(w[:len(self.theta0)], self.theta0 - 293.0), self.dt])

    def product_sweep(self, flow_tol=1.0e-13):
        '''(theta, u, p) of flow_amd.boussinesq.CoupledStep.sweep.'''
        from flow_amd import boussinesq
        u0, p0, theta0 = self.pb.state_of_rest()
        theta0.set_array(self.theta0)
        u0.set_array(self.u0)
        self.p_rest = p0.array().copy()
        step = boussinesq.CoupledStep(self.pb, u0, p0, theta0, self.t, self.dt)
        step.flow_tol = flow_tol
        dist = step.sweep()
        assert all(numpy.isfinite(dist)) and step.sweeps == 1
        return step.theta.array(), step.u.array(), step.p.array()

    def oracle_sweep(self, info=None):
        from oracle import fem_oracle as orc
        pb, mesh = self.pb, self.mesh
        Qo = orc.Space(mesh.points, mesh.cell_vertices, pb.Q.layout.cell_dofs, 2,
                       pb.Q.N)
        Wo = orc.Space(mesh.points, mesh.cell_vertices, pb.W.layout.cell_dofs, 2,
                       pb.W.N)
        Po = orc.Space(mesh.points, mesh.cell_vertices, pb.P.layout.cell_dofs, 1,
                       pb.P.N)
        M, A, _b = orc.heat_operators(Qo, Wo, self.u0, pb.kappa, pb.rho_room,
                                      pb.cp, 0.0, False)
        d_t, v_t = collect(pb.temperature_bcs(self.t), pb.Q.size())
        theta = orc.heat_solve(M.tocsr(), A.tocsr(), 1.0, -self.dt,
                               M.dot(self.theta0), d_t, v_t)
        dens = pb.rho(self.theta0)[pb.Q.layout.cell_dofs]
        f = numpy.stack([numpy.zeros_like(dens), dens * pb.gravity], axis=2)
        lat = (reference.lattice(2), f)
        # (hydrostatic pressure of the state of rest: linear, so its P1
        # projection is its nodal interpolant)
        p_rest = pb.rho_room * pb.gravity * pb.P.layout.dof_coords[:, 1]
        u, p, _ui = orc.step(Wo, Po, self.u0, p_rest, lat, lat, self.u_bc,
                             None, pb.rho_room, pb.mu, self.dt,
                             scheme='rotational', info=info)
        return theta, u, p

    def pressure_mass(self):
        '''P1 mass matrix (scipy): the Neumann pressure is compared mean-free.'''
        from oracle import fem_oracle as orc
        pb, mesh = self.pb, self.mesh
        return orc.mass_matrix(orc.Space(
            mesh.points, mesh.cell_vertices, pb.P.layout.cell_dofs, 1, pb.P.N))


class StokesChannelCase(object):
    '''The Karman driver's Stokes bootstrap (reference
    tests/test_karman_vortex_street.py:171-179, flow/stokes.py:13-148) on a
    body-fitted channel with the analytic body force above: flow_amd.stokes
    (MINRES + block preconditioner) against the oracle's sparse LU of the
    saddle-point matrix -- BASELINE config 5's solver at a size where the
    kernels tile.'''
    def __init__(self, nx, ny, mu=0.002):
        self.args = dict(nx=nx, ny=ny, mu=mu)
        self.mesh = mesh = fem.karman_channel(nx, ny, fitted=True)
        self.mu = mu
        self.WP = fem.FunctionSpace(
            mesh,
            fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
            * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
        self.W, self.P = W, P = self.WP.sub(0), self.WP.sub(1)
        prof = '%e * (%e - x[1]) * (x[1] - %e) / %e' % (
            karman.ENTRANCE_VELOCITY, karman.Y1, karman.Y0,
            (0.5 * (karman.Y1 - karman.Y0))**2)
        inflow = fem.Expression(prof, degree=2)
        self.u_bcs = [
            fem.DirichletBC(W, (0.0, 0.0), karman.UpperBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), karman.LowerBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), karman.ObstacleBoundary()),
            fem.DirichletBC(W.sub(0), inflow, karman.LeftBoundary()),
            fem.DirichletBC(W.sub(0), inflow, karman.RightBoundary()),
            ]
        self.p_bcs = [fem.DirichletBC(P, 0.0, karman.RightBoundary())]
        self.force = fem.Expression(lambda x: force(x, 0.3), degree=2)

    def num_dofs(self):
        return self.W.size() + self.P.N

    def fingerprint(self):
        m = self.mesh
        return numpy.array([m.num_vertices(), m.num_cells(), self.W.N, self.P.N,
                            m.points.sum(), self.mu])

    def product_solve(self, tol=1.0e-13):
        from flow_amd import stokes
        u, p = stokes.solve(self.WP, self.u_bcs + self.p_bcs, self.mu, self.force,
                            verbose=False, tol=tol, max_iter=20000)
        return u.array(), p.array()

    def oracle_solve(self):
        from oracle import fem_oracle as orc
        m = self.mesh
        Wo = orc.Space(m.points, m.cell_vertices, self.W.layout.cell_dofs, 2,
                       self.W.N)
        Po = orc.Space(m.points, m.cell_vertices, self.P.layout.cell_dofs, 1,
                       self.P.N)
        X = fem.cell_lattice_points(m, 2)
        nc, nl = X.shape[:2]
        v = self.force.eval(X.reshape(-1, 2).T)
        lat = (reference.lattice(2), numpy.ascontiguousarray(
            v.reshape(2, nc, nl).transpose(1, 2, 0)))
        return orc.stokes_solve(Wo, Po, lat, self.mu,
                                collect(self.u_bcs, self.W.size()),
                                collect(self.p_bcs, self.P.N))


# the configurations of tests/golden/bq_large_*.npz / stokes_large_*.npz
LARGE_BOUSSINESQ = {
    # 0.39 M DoF (velocity 2 x 0.12 M, pressure 30 k, temperature 0.12 M)
    'box_120': dict(n=120),
    # BASELINE config 4's nominal size: 4.2 M DoF
    'box_400': dict(n=400),
    }
LARGE_STOKES = {
    # 0.34 M DoF
    'channel_400x93': dict(nx=400, ny=93),
    # BASELINE config 5's nominal size: 2.0 M DoF
    'channel_980x229': dict(nx=980, ny=229),
    }

