# -*- coding: utf-8 -*-
'''
Karman-vortex-street channel problem: the caller of the hot path used by
bench.py and the tests.  Counterpart of the reference driver
tests/test_karman_vortex_street.py:56-289 with the same geometry (:18-23,
:35-38), boundary conditions (:128-145, :190-203), parameters (mu = 0.002 :167,
dt0 = 1e-5 :210, dt_max = 1 :211, tol = 1e-10 :239) and step-size controller
(:262-286).  gmsh is not available: the mesh is the structured channel of
fem.karman_channel pulled onto the cylinder (body-fitted: the hole's boundary
vertices lie on the circle; `fitted=False` gives the staircase obstacle of
round 1, whose one-cell notches grow a spurious velocity spike in long runs).
The run
starts from the Stokes solution like the reference's (`set_initial_stokes`,
:171-179) or impulsively from the inflow profile (`set_initial_profile`).
'''
from __future__ import print_function

from . import fem
from . import navier_stokes
from . import parallel

X0, X1 = 0.0, 0.6
Y0, Y1 = -0.07, 0.07
ENTRANCE_VELOCITY = 0.01
MESH_EPS = 1.0e-12
RHO_WATER_293K = 998.2     # materials.water.density(T=293.0) is not available


class LeftBoundary(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[0] < X0 + MESH_EPS)


class RightBoundary(fem.SubDomain):
    def __init__(self, x1=X1):
        fem.SubDomain.__init__(self)
        self.x1 = x1

    def inside(self, x, on_boundary):
        return on_boundary & (x[0] > self.x1 - MESH_EPS)


class LowerBoundary(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] < Y0 + MESH_EPS)


class UpperBoundary(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] > Y1 - MESH_EPS)


class ObstacleBoundary(fem.SubDomain):
    def __init__(self, x1=X1):
        fem.SubDomain.__init__(self)
        self.x1 = x1

    def inside(self, x, on_boundary):
        return (
            on_boundary
            & (X0 + MESH_EPS < x[0]) & (x[0] < self.x1 - MESH_EPS)
            & (Y0 + MESH_EPS < x[1]) & (x[1] < Y1 - MESH_EPS)
            )


class KarmanProblem(object):
    def __init__(self, nx=None, ny=None, velocity_degree=2, mu=0.002,
                 rho=RHO_WATER_293K, scheme='rotational', fitted=True,
                 mesh=None, length=X1):
        # body-fitted obstacle (fem/mesh.py: rectangle_with_fitted_hole); the
        # staircase variant (fitted=False) leaves one-cell notches in which,
        # at the controller's step size, a node-scale velocity spike grows
        # once the wake becomes unsteady (profiles/NOTES.md section 5).
        # mesh: any triangulation of the channel instead (a file name, or a
        # Mesh: the reference's driver reads the one gmsh made,
        # tests/test_karman_vortex_street.py:26-53); renumbered along the
        # channel unless it already is (Mesh.reordered)
        if mesh is not None:
            if isinstance(mesh, str):
                mesh = fem.Mesh(mesh)
            elif mesh.bandwidth() > 8 * int(mesh.num_vertices()**0.5) + 64:
                mesh = mesh.reordered()
            self.mesh = mesh
        else:
            # length: the channel [0, length] x [-0.07, 0.07] (the
            # reference's is 0.6 long; a weak-scaling run keeps nx / length,
            # i.e. the cells, and lengthens the channel with the ranks)
            self.mesh = fem.karman_channel(nx, ny, fitted=fitted) \
                if length == X1 else fem.karman_channel(
                    nx, ny, fitted=fitted, length=length)
        self.length = length
        self.W = fem.VectorFunctionSpace(self.mesh, 'Lagrange', velocity_degree)
        self.P = fem.FunctionSpace(self.mesh, 'Lagrange', 1)
        self.mu = mu
        self.rho = rho
        profile = '%e * (%e - x[1]) * (x[1] - %e) / %e' % (
            ENTRANCE_VELOCITY, Y1, Y0, (0.5 * (Y1 - Y0))**2
            )
        self.inflow = fem.Expression(profile, degree=2)
        self.outflow = fem.Expression(profile, degree=2)
        W, P = self.W, self.P
        self.u_bcs = [
            fem.DirichletBC(W, (0.0, 0.0), UpperBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), LowerBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), ObstacleBoundary(length)),
            fem.DirichletBC(W.sub(0), self.inflow, LeftBoundary()),
            fem.DirichletBC(W.sub(0), self.outflow, RightBoundary(length)),
            ]
        self.p_bcs = [fem.DirichletBC(P, 0.0, RightBoundary(length))]
        self.stepper = {
            'chorin': navier_stokes.Chorin,
            'ipcs': navier_stokes.IPCS,
            'rotational': navier_stokes.Rotational,
            }[scheme]()
        self.u0 = fem.Function(W)
        self.p0 = fem.Function(P)
        self.u0.rename('velocity', 'velocity')
        self.p0.rename('pressure', 'pressure')
        self.dt = 1.0e-5
        self.dt_max = 1.0
        self.t = 0.0
        self.hmax = self.mesh.hmax()
        self.history = []
        self._umag_hist = []
        self._umag_start = fem.ops.StartChooser()
        self.extrapolate_projection = True
        return

    def reset(self, dt=1.0e-5):
        '''Back to the state before the first step (fields, clock, step size,
        the controller's memory, and what the steps of mode 'fast' remember).'''
        fem.ops.fill(self.u0.data, 0.0)
        fem.ops.fill(self.p0.data, 0.0)
        self.dt = dt
        self.t = 0.0
        self.history = []
        self._umag_hist = []
        self._umag_start = fem.ops.StartChooser()
        self.W.layout._dev.pop('step_history', None)
        self.W.layout._dev.pop('newton_quad_C', None)
        navier_stokes.forget_history(self.W)
        return

    def prepare(self):
        '''One throw-away step from the inflow profile at a CFL-sized dt: builds
        everything that is built once per (mesh, conditions) and then cached --
        operators, multigrid hierarchy, the ILU(0) plan (host-side colouring)
        and a first set of factors, boundary data, workspaces -- so that a
        timed window never contains one-off setup (a run from the Stokes start
        would otherwise meet its first Newton iteration, and with it seconds
        of plan building, some steps into the window).  State and clock are
        reset afterwards; what stays are caches and preconditioners.'''
        self.reset(1.0e-2)
        self.set_initial_profile()
        self.step()
        self.reset()
        # the factors of that step belong to another state and dt: the first
        # real Newton iteration computes its own (a few ms, no plan building)
        for name in ('jacobian_ilu', 'jacobian_pmg', 'jacobian_tl',
                     'jacobian_ilu_strip', 'jacobian_pmg_strip'):
            pre = self.W.layout._dev.get(name)
            if pre is not None:
                pre.stale = True
        return

    def snapshot(self):
        '''Fields, clock and step size (device copies).'''
        from . import _hip
        return dict(u=_hip.clone(self.u0.data), p=_hip.clone(self.p0.data),
                    dt=self.dt, t=self.t)

    def restore(self, snap):
        '''Back to a snapshot; the controller's memory and what the steps of
        mode 'fast' remember start afresh.'''
        self.reset(snap['dt'])
        fem.ops.copy(self.u0.data, snap['u'])
        fem.ops.copy(self.p0.data, snap['p'])
        self.t = snap['t']
        return

    def settle(self, rel=0.01, hold=3, max_steps=80, tol=1.0e-10):
        '''Step until the CFL controller has brought the step size to its
        plateau: `hold` consecutive steps on which dt moved by less than `rel`.
        (From dt0 = 1e-5 the controller at most doubles dt per step: ~12 steps
        of ramp in which the Newton systems are mass-dominated and the start
        vector often passes the stopping test as it is -- not the regime a run
        lives in.)  Returns the number of steps taken.'''
        calm = 0
        for k in range(max_steps):
            before = self.dt
            self.step(tol=tol)
            calm = calm + 1 if abs(self.dt - before) <= rel * before else 0
            if calm >= hold:
                return k + 1
        return max_steps

    def num_dofs(self):
        return self.W.size() + self.P.size()

    def set_initial_profile(self):
        '''Start from the inflow profile (x-velocity), zero pressure.'''
        prof = fem.Expression(
            (self.inflow.cppcode, '0.0'), degree=2
            )
        self.u0.assign(fem.interpolate(prof, self.W))
        return

    def set_initial_stokes(self, tol=1.0e-13, max_iter=10000):
        '''Start from the Stokes solution, as the reference driver does
        (tests/test_karman_vortex_street.py:171-179: the velocity conditions
        only -- its pressure condition list is empty at that point --,
        mu = 0.002, f = 0, tol = 1e-13).'''
        from . import stokes
        mesh = self.mesh
        WP = fem.FunctionSpace(
            mesh,
            fem.VectorElement('Lagrange', mesh.ufl_cell(), self.W.degree)
            * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
        W = WP.sub(0)
        bcs = [
            fem.DirichletBC(W, (0.0, 0.0), UpperBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), LowerBoundary()),
            fem.DirichletBC(W, (0.0, 0.0), ObstacleBoundary()),
            fem.DirichletBC(W.sub(0), self.inflow, LeftBoundary()),
            fem.DirichletBC(W.sub(0), self.outflow, RightBoundary()),
            ]
        u, p = stokes.solve(WP, bcs, fem.Constant(self.mu),
                            f=fem.Constant((0.0, 0.0)), verbose=False,
                            tol=tol, max_iter=max_iter)
        fem.ops.copy(self.u0.data, u.data)
        fem.ops.copy(self.p0.data, p.data)
        self.stokes_info = dict(stokes.last_solve_info)
        return

    def reynolds(self):
        return ENTRANCE_VELOCITY * 0.04 * self.rho / self.mu

    def step(self, tol=1.0e-10, adapt=True):
        '''One pass of the reference's time loop body (:219-286).'''
        u1, p1 = self.stepper.step(
            fem.Constant(self.dt),
            {0: self.u0}, self.p0,
            self.u_bcs, self.p_bcs,
            fem.Constant(self.rho), fem.Constant(self.mu),
            f={0: fem.Constant((0.0, 0.0)), 1: fem.Constant((0.0, 0.0))},
            verbose=False,
            tol=tol
            )
        self.u0.assign(u1)
        self.p0.assign(p1)
        info = dict(navier_stokes.last_step_info)
        info.pop('tentative_velocity', None)
        info['dt'] = self.dt
        info['t'] = self.t
        if adapt:
            # CFL-like step-size control on ||project(|u|)||_inf (:262-286)
            # (mass solve to 1e-7, started from the previous step's projection:
            # the value only steers dt, which inherits that relative accuracy)
            # ... extrapolated in time through the last two or three: linearly,
            # or quadratically once the step size has settled)
            hist = self._umag_hist
            guess = hist[0][0] if hist else None
            mode = None
            if self.extrapolate_projection and len(hist) >= 2:
                c, a = self.dt, hist[0][1]
                guess = fem.Function(hist[0][0].function_space())
                guess.assign(hist[0][0])
                mode = 1
                if len(hist) >= 3 and 0.7 <= c / a <= 1.5 and \
                        0.7 <= a / hist[1][1] <= 1.5:
                    mode = self._umag_start.pick()
                if mode == 2:
                    b = hist[1][1]
                    w0 = (c + a) * (c + a + b) / (a * (a + b))
                    w1 = -c * (c + a + b) / (a * b)
                    w2 = c * (c + a) / ((a + b) * b)
                    fem.ops.axpby(w1, hist[1][0].data, w0, guess.data)
                    fem.ops.axpby(w2, hist[2][0].data, 1.0, guess.data)
                else:
                    r = c / a
                    fem.ops.axpby(-r, hist[1][0].data, 1.0 + r, guess.data)
            umag = fem.project_magnitude(self.u0, tol=1.0e-7, initial_guess=guess)
            # (projection, length of the step that led to it), newest first
            self._umag_hist = [(umag, self.dt)] + hist[:2]
            info['projection_iterations'] = umag.solve_info.iterations
            if mode is not None:
                self._umag_start.report(mode, umag.solve_info.iterations)
            if parallel.active():
                unorm = parallel.norm_linf(
                    umag.data, umag.function_space().layout)
            else:
                unorm = umag.vector().norm('linf')
            target_dt = 1.0 * self.hmax / unorm
            alpha = 0.5
            self.dt = min(
                self.dt_max,
                # at most double the step size from step to step
                self.dt * min(2.0, 1.0 + alpha * (target_dt - self.dt) / self.dt)
                )
            info['unorm'] = unorm
            info['target_dt'] = target_dt
        self.t += self.dt
        self.history.append(info)
        return info
