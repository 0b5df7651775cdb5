# -*- coding: utf-8 -*-
'''
Boussinesq natural convection in a sealed box with a circular heater -- the
coupled caller of the hot path (flow_amd.navier_stokes + flow_amd.heat),
BASELINE config 4.

What the reference's driver (tests/test_boussinesq.py:100-367) does, expressed
as three small objects instead of one script:

  HeaterBox        the problem data: geometry and spaces, water properties at
                   room temperature (:106-110), boundary sets (:27-64), the
                   heater ramp (:172-176), the state of rest with hydrostatic
                   pressure (:140-158);
  CoupledStep      ONE fixed-point sweep of a time step: temperature with the
                   previous sweep's velocity (implicit Euler on Heat, :213-229),
                   then velocity / pressure with the buoyancy of the previous
                   sweep's temperature (Rotational.step, :231-253), and the
                   distance between consecutive sweeps (:266-281);
  FixedPointStepper
                   the time loop: sweeps until that distance is below
                   `Coupling.tolerance`, step-size policy on the number of
                   sweeps it took.

The numbers of the policy are the reference's (`Coupling`, each with its line);
gmsh, `materials` and `parabolic` are replaced by fem.heater_box,
flow_amd.materials and flow_amd.time_steppers.
'''
from __future__ import print_function

from . import fem
from . import heat
from . import materials
from . import navier_stokes
from . import parallel
from . import time_steppers
from .message import Message, info


class Coupling(object):
    '''Policy constants of the reference driver (tests/test_boussinesq.py).'''
    tolerance = 1.0e-1        # both sweep distances below this: converged (:202)
    max_sweeps = 10           # more than that: the step is redone ... (:204)
    stall_factor = 0.25       # ... at a quarter of the step size (:208-211)
    failure_factor = 0.5      # Navier-Stokes RuntimeError: half of it (:254-264)
    target_sweeps = 5         # the controller aims at this many sweeps (:205)
    relaxation = 0.5          # dt moves half way to its target ... (:351)
    max_growth = 2.0          # ... and at most doubles per step (:354-358)
    dt_max = 1.0              # (:113)


class _Interior(fem.SubDomain):
    '''Boundary facets strictly inside the bounding box: the heater.'''
    def __init__(self, box):
        fem.SubDomain.__init__(self)
        self.box = box

    def inside(self, x, on_boundary):
        (x0, y0), (x1, y1) = self.box
        eps = 1.0e-10
        return on_boundary & (x[0] > x0 + eps) & (x[0] < x1 - eps) \
            & (x[1] > y0 + eps) & (x[1] < y1 - eps)


class _Walls(fem.SubDomain):
    '''Boundary facets on the bounding box: the outer walls.'''
    def __init__(self, box):
        fem.SubDomain.__init__(self)
        self.box = box

    def inside(self, x, on_boundary):
        (x0, y0), (x1, y1) = self.box
        eps = 1.0e-10
        return on_boundary & ((x[0] < x0 + eps) | (x[0] > x1 - eps)
                              | (x[1] < y0 + eps) | (x[1] > y1 - eps))


class HeaterBox(object):
    '''Water in the box [0, 0.1] x [0, 0.2] around a circular heater (centre
    (0.05, 0.05), radius 0.02), all properties but the density taken at room
    temperature (tests/test_boussinesq.py:106-110: "to avoid nonlinearity").'''
    box = ((0.0, 0.0), (0.1, 0.2))
    room = 293.0                 # K (:105)
    heater_max = 320.0           # K (:115)
    ramp_time = 30.0             # s: the heater reaches heater_max then (:172)
    gravity = -9.81              # m/s^2 along y (:123)

    def __init__(self, mesh, supg=False):
        self.mesh = mesh
        self.supg = supg
        self.W = fem.VectorFunctionSpace(mesh, 'Lagrange', 2)
        self.P = fem.FunctionSpace(mesh, 'Lagrange', 1)
        self.Q = fem.FunctionSpace(mesh, 'Lagrange', 2)
        self.rho = materials.density
        self.rho_room = materials.density(self.room)
        self.mu = materials.dynamic_viscosity(self.room)
        self.cp = materials.specific_heat_capacity(self.room)
        self.kappa = materials.thermal_conductivity(self.room)
        self.g = fem.Constant((0.0, self.gravity))
        self.heater = _Interior(self.box)
        self.walls = _Walls(self.box)
        self.no_slip = [fem.DirichletBC(self.W, (0.0, 0.0), 'on_boundary')]

    def prepare(self):
        '''What is built once per mesh on the HOST and would otherwise be built
        inside the first time step that needs it: the ILU(0) plans (graph
        colouring, sweep streams: ~1 s at a million rows) of the P2 and P1
        levels the heat and Newton solves fall back to once the plume is too
        fast for the Chebyshev cycle (flow_amd/fem/tlilu.py).'''
        from . import device
        if device.on_gpu() and not parallel.active():
            # (the strips build their own block plans where they need them)
            from .fem.tlilu import TwoLevelIlu
            heat.prepare(self.Q)
            TwoLevelIlu.plans(self.W.layout)
        return self

    def heater_temperature(self, t):
        '''Linear ramp from room temperature to heater_max in ramp_time.'''
        return self.room + min(1.0, t / self.ramp_time) * (
            self.heater_max - self.room)

    def temperature_bcs(self, t):
        return [fem.DirichletBC(self.Q, self.heater_temperature(t), self.heater),
                fem.DirichletBC(self.Q, self.room, self.walls)]

    def state_of_rest(self):
        '''(u, p, theta): no flow, room temperature, hydrostatic pressure.'''
        theta = fem.project(fem.Constant(self.room), self.Q)
        u = fem.project(fem.Constant([0, 0]), self.W)
        p = fem.project(
            fem.Expression('c * x[1]', degree=1, c=self.rho_room * self.gravity),
            self.P)
        for f, name in ((u, 'velocity'), (p, 'pressure'), (theta, 'temperature')):
            f.rename(name, name)
        return u, p, theta


class CoupledStep(object):
    '''One time step t -> t + dt from (u0, p0, theta0): `sweep` advances both
    fields once, each with the other's values of the PREVIOUS sweep, and
    returns how far the new pair is from the previous one.'''

    # tolerance handed to Rotational.step (the reference driver's, :252)
    flow_tol = 1.0e-10

    def __init__(self, problem, u0, p0, theta0, t, dt):
        self.pb = problem
        self.u0, self.p0, self.theta0 = u0, p0, theta0
        self.t, self.dt = t, dt
        # the "previous sweep" starts at the old time level
        self.u = self._copy(u0)
        self.theta = self._copy(theta0)
        self.p = None
        self.sweeps = 0

    @staticmethod
    def _copy(f):
        out = fem.Function(f.function_space())
        out.assign(f)
        return out

    def temperature(self):
        '''Implicit Euler on the heat equation convected by the previous
        sweep's velocity; the heater value is the one at time t.'''
        pb = self.pb
        operator = heat.Heat(
            pb.Q, self.u, pb.kappa, pb.rho_room, pb.cp,
            pb.temperature_bcs(self.t), fem.Constant(0.0),
            supg_stabilization=pb.supg)
        return time_steppers.ImplicitEuler(operator).step(
            self.theta0, self.t, self.dt)

    def flow(self):
        '''Rotational pressure correction with the buoyancy rho(theta) g of the
        previous sweep's temperature at both time levels.  RuntimeError (Newton
        or Krylov non-convergence) is the caller's to handle.'''
        pb = self.pb
        buoyancy = fem.NodalExpression(pb.rho, [self.theta]) * pb.g
        u, p = navier_stokes.Rotational().step(
            fem.Constant(self.dt), {0: self.u0}, self.p0, pb.no_slip, [],
            pb.rho_room, fem.Constant(pb.mu), f={0: buoyancy, 1: buoyancy},
            verbose=False, tol=self.flow_tol)
        # (on the strips of flow_amd.parallel the fields stay valid on the
        # rank's owned + ghost rows: the heat operator is assembled and solved
        # on the strips too)
        return u, p

    def sweep(self):
        '''-> (|u - u_prev|, |theta - theta_prev|), max norms: of the
        projected |ux| + |uy| (:268-273) and of the nodal values (:275-277).'''
        self.sweeps += 1
        theta = self.temperature()
        u, p = self.flow()
        du = self._copy(u)
        fem.ops.axpby(-1.0, self.u.data, 1.0, du.data)
        umag = fem.project_magnitude(du, mode=1)
        dth = self._copy(theta)
        fem.ops.axpby(-1.0, self.theta.data, 1.0, dth.data)
        if parallel.active():
            # (maxima over the owned rows of all ranks: the same numbers, and
            # with them the same decisions, everywhere)
            u_dist = parallel.norm_linf(umag.data,
                                        umag.function_space().layout)
            theta_dist = parallel.norm_linf(dth.data, self.pb.Q.layout)
        else:
            u_dist = umag.vector().norm('linf')
            theta_dist = dth.vector().norm('linf')
        self.u, self.p, self.theta = u, p, theta
        return u_dist, theta_dist


class FixedPointStepper(object):
    '''The time loop: per step, sweeps of CoupledStep until both distances are
    below Coupling.tolerance; a step that needs more than Coupling.max_sweeps,
    or whose flow solve raises RuntimeError, is redone with a smaller dt; after
    an accepted step dt moves towards the value that would have needed
    Coupling.target_sweeps.'''

    def __init__(self, problem, dt0, policy=Coupling):
        self.pb = problem
        self.policy = policy
        if hasattr(problem, 'prepare'):
            problem.prepare()
        self.u, self.p, self.theta = problem.state_of_rest()
        self.t = 0.0
        self.dt = dt0
        self.log = []

    def advance(self):
        '''One accepted time step (retrying with smaller dt as needed).'''
        pol = self.policy
        while True:
            step = CoupledStep(self.pb, self.u, self.p, self.theta, self.t,
                               self.dt)
            verdict = self._iterate(step)
            if verdict == 'converged':
                break
            shrink = pol.stall_factor if verdict == 'stalled' \
                else pol.failure_factor
            info('%s: dt %e -> %e, step redone' % (
                {'stalled': 'fixed-point iteration not converged after %d '
                            'sweeps' % pol.max_sweeps,
                 'failed': 'flow solver did not converge'}[verdict],
                self.dt, shrink * self.dt))
            self.dt *= shrink
        self.u.assign(step.u)
        self.p.assign(step.p)
        self.theta.assign(step.theta)
        self.log.append({'t': self.t, 'dt': self.dt,
                         'banach_steps': step.sweeps,
                         'heater_temp': self.pb.heater_temperature(self.t)})
        # the next step size: what would have taken target_sweeps, approached
        # half way, at most doubled, capped
        wanted = self.dt * pol.target_sweeps / float(step.sweeps)
        factor = 1.0 + pol.relaxation * (wanted - self.dt) / self.dt
        self.dt = min(pol.dt_max, self.dt * min(pol.max_growth, factor))
        self.t += self.dt
        return step

    def _iterate(self, step):
        pol = self.policy
        with Message('time %e -> %e' % (step.t, step.t + step.dt)):
            while step.sweeps < pol.max_sweeps:
                try:
                    u_dist, theta_dist = step.sweep()
                except RuntimeError:
                    return 'failed'
                info('sweep %d: velocity moved %e, temperature %e' % (
                    step.sweeps, u_dist, theta_dist))
                if u_dist < pol.tolerance and theta_dist < pol.tolerance:
                    return 'converged'
        return 'stalled'

    def run(self, target_time):
        '''Steps while t < target_time (+ DOLFIN_EPS, :167).'''
        last = None
        while self.t < target_time + fem.DOLFIN_EPS:
            last = self.advance()
        return last


def compute_boussinesq(target_time, nx=16, supg=False, verbose=False,
                       dt0=1.0e-2, mesh=None):
    '''The reference driver's entry point (tests/test_boussinesq.py:100):
    -> (u, p, theta, per-step log).  mesh: default the structured heater box
    with nx cells across, body-fitted at the heater from 12 cells on (the
    staircase variant below that); fem.heater_box_coarse() is the counterpart
    of the reference's `lcar = 0.1` gmsh mesh (:84-97).'''
    if mesh is None:
        mesh = fem.heater_box(nx, fitted=nx >= 12)
    stepper = FixedPointStepper(HeaterBox(mesh, supg=supg), dt0)
    stepper.run(target_time)
    return stepper.u, stepper.p, stepper.theta, stepper.log
