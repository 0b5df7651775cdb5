# -*- coding: utf-8 -*-
#
'''
Incremental pressure-correction schemes for the incompressible Navier--Stokes
equations

        rho (u' + u.nabla(u)) = - nabla(p) + mu Delta(u) + f,
        div(u) = 0,

on MI355X.  Same interface and the same three sub-steps as the reference
(flow/navier_stokes/pressure_correction.py): `Chorin`, `IPCS`, `Rotational`
with `.step(dt, u, p0, u_bcs, p_bcs, rho, mu, f, verbose, tol)` and the class
attribute `.order` (reference :521-617).  Where the reference hands UFL forms
to dolfin/PETSc, this module calls the HIP kernels of libflow_hip.so:

  tentative velocity   Newton on F1 (reference :147-255): residual (K5) and the
                       exact Jacobian, applied cell by cell (matrix-free) and
                       assembled (K6) only for the lagged ILU(0) factors;
                       Dirichlet rows (K7); GMRES (or BiCGStab) + ILU(0) instead
                       of sparse LU;
  pressure             P1 Poisson (reference :258-433): cached stiffness matrix
                       (K1), right-hand side kernel (K2), CG + a smoothed-
                       aggregation V-cycle instead of CG + BoomerAMG; Dirichlet
                       branch with symmetric elimination (CG from p0), Neumann
                       branch without null-space handling, from p0 minus its
                       mean (sum zero like the reference's x0 = 0);
  velocity correction  vector mass system (reference :436-465): cached mass
                       matrix (K3), right-hand side kernel (K4), CG + Jacobi.

The Krylov solvers differ from the reference's (LU, CG + hypre), so parity
means "same converged discrete solution"; iteration counts are reported in
`last_step_info`.

Two modes (`set_mode`, `solver_parameters['mode']`):

  'parity' (default)  follows the reference's Newton path: start from u0
      (:220), every Newton step an (almost) exact one -- linear residual below
      1e-6 of the Newton tolerance --, stop at the first iterate with
      ||F||_2 < 1e-10 (:230-236).  The tolerance 1e-10 is loose at 10 M DoF
      (||F|| = 1e-10 is a relative 3e-4 in the velocity there), so the iterate
      the reference stops at is a specific point, and only this path reproduces
      it: measured on the 10 M-DoF workload (tools/parity_single_step.py,
      tests/test_full_size_parity.py) the step agrees with one whose linear
      systems are solved 1e3 times tighter to < 1e-7 relative l2 in u and p
      -- in the start-up steps, at CFL-sized steps and from the Stokes start
      at dt = 0.024 (the bench's window: the pressure is the sensitive field
      there, 7e-7 with the factor 1e-5, hence 1e-6).
      Nothing is carried from one call to the next except preconditioners.
  'fast'  the Newton iteration may start from the previous step's tentative
      velocity (or its extrapolation) when that leaves the smaller residual,
      its linear systems are solved to 0.02 of the tolerance (below the
      quadratic remainder an exact step leaves), and the pressure / correction
      solves start from extrapolations in time.  Every stopping test is the
      reference's, but the Newton iterate it accepts is a different one: ~1e-5
      to 1e-4 relative from the reference's at 10 M DoF (same table).
'''
from __future__ import print_function

import ctypes
import time

import numpy

from ..fem import ops
from ..fem.bcs import collect
from ..fem.function import Function, as_cell_coefficient, scalar_value
from ..message import Message, info
from .. import _hip
from .. import device
from .. import parallel
from . import start_vectors
from .start_vectors import (                                # noqa: F401
    extrapolation_weights, forget_history,
    extrapolated_increment as _extrapolated_increment,
    remember_increment as _remember_increment,
    newton_key as _newton_key,
    )
from .newton_preconditioner import (
    age as _age, contraction as _contraction, choose_cycle as _choose_cycle,
    contraction_on_strips as _contraction_on_strips,
    coarse_jacobian as _coarse_jacobian, mass_share as _mass_share,
    )

__all__ = ['Chorin', 'IPCS', 'Rotational', 'solver_parameters',
           'last_step_info', 'set_mode', 'forget_history']

# Jacobi-preconditioned Krylov needs more iterations than the reference's AMG
# (maxit 100/1000, reference :335,422,460): limits are scaled up, everything
# else (rtol = tol, atol = 0, error on non-convergence) is kept.
solver_parameters = {
    'mode': 'parity',
    # Newton systems: 'linear_solver' 'gmres' (GMRES(gmres_restart): one
    # Jacobian action + one preconditioner application per iteration) or
    # 'bicgstab' (two each); 'preconditioner' 'ilu0' (default: multicolour
    # ILU(0) of the two diagonal blocks, refactored when dt has moved by more
    # than `ilu_lag` or the factors have gone stale; 3-4x fewer and far more
    # regular iterations than Jacobi at CFL-sized steps) or 'jacobi';
    # linear residual <= max(linear_atol_factor * tol, forcing * ||F||)
    # gmres_restart 10: the convergence of the ILU-preconditioned systems is
    # uniform (a factor ~5 per iteration), a cycle of 10 loses nothing against
    # 20 (17-18 applications per solve either way) while the Gram-Schmidt
    # sweeps over the basis -- 2(j+1) vectors of 69 MB in iteration j -- halve.
    'newton': {'maximum_iterations': 10, 'linear_maxit': 5000,
               'linear_solver': 'gmres', 'gmres_restart': 10,
               'linear_rtol': 1.0e-13, 'linear_atol_factor': 1.0e-6,
               # optional ('linear_remainder_fraction' > 0): once the
               # quadratic model of the Newton steps is known (||F_{k+1}|| ~
               # C ||F_k||^2, observed on every step), that fraction of the
               # remainder the step is going to leave anyway, at most
               # linear_atol_cap * tol.  Measured (tools/linear_tol_check.py,
               # eighth-size proxy, remainder 1.3e-12, fraction 1e-3): 4.6
               # instead of 6.6 GMRES applications per step, 20 settled steps
               # within 2.4e-9 / 3.6e-9 (u / p) of the run with 1e-6 * tol --
               # and nothing on the 10 M-DoF mesh (remainder 4e-14: the rule
               # asks for 1e-16 there as well), while the natural-convection
               # start (velocities ~0) then differs by 6e-7 between one GPU
               # and two strips.  Off.
               'linear_remainder_fraction': 0.0, 'linear_atol_cap': 1.0e-4,
               'forcing': 0.0, 'check_every': 1, 'restart': 400,
               # 'pmg' (P2 velocity spaces; falls back to 'ilu0' otherwise):
               # two-level p-multigrid with Chebyshev smoothing, CSR-stream
               # products only (flow_amd/fem/pmg.py) -- 14-15 applications per
               # solve where the multicolour ILU(0) needs 33
               'preconditioner': 'pmg', 'ilu_lag': 8.0,
               # (the cycle is rebuilt after `pmg_refresh` solves where they
               # average `pmg_refresh_min` applications or more, and not
               # within `pmg_min_solves` of a rebuild because of one long
               # solve: newton_preconditioner.age)
               'pmg_refresh': 50, 'pmg_min_solves': 3, 'pmg_refresh_min': 6.0,
               # (Chebyshev steps before / after the coarse correction and
               # on the P1 level; intervals [lam_max / ratio, 1.1 lam_max].
               # r4, with the start vectors extrapolated in time -- the
               # solves are short, what counts is the price of an
               # application --: one fine product less and two cheap coarse
               # ones more, tools/pmg_param_sweep.sh: 9.15 against 10.15
               # ms/step with 2 / 2 / 4 and ratios 8 / 8)
               # coarse_auto: where the P1 level is not mass-dominated (dt >>
               # h^2 / nu: the Boussinesq driver's steps) its Chebyshev
               # treatment grows with the level's condition number, up to
               # coarse_max steps (Pmg.refactor; at CFL-sized steps the rule
               # asks for fewer than `coarse_steps` and changes nothing)
               'pmg': {'pre': 1, 'post': 2, 'coarse_steps': 6,
                       'ratio_fine': 5.0, 'ratio_coarse': 12.0,
                       'coarse_auto': True, 'coarse_max': 48},
               # ... used where one cycle contracts a full-spectrum vector by
               # at least this factor (else: ILU(0)), and a GMRES that has not
               # converged after `pmg_maxit` applications is redone with ILU(0)
               'pmg_accept': 0.8, 'pmg_maxit': 150,
               # ... what a rejected cycle is replaced with on one GPU: 'tlilu'
               # = the same two levels with ILU(0) sweeps as smoothers
               # (flow_amd/fem/tlilu.py: one before and one after the coarse
               # correction, `coarse_sweeps` on the P1 level) or 'ilu0' = the
               # bare multicolour ILU(0) (what the strips and P1 velocity
               # spaces run).  tools/smoother_lab.py: 27 -> 13 applications on
               # the structured channel at cell Peclet 3.5, 48 -> 17 on the
               # graded unstructured one.
               'fallback': 'tlilu',
               # (coarse correction first, then one sweep: 0 / 1 / 1 is the
               # variant with the fewest launches per application -- at the
               # sizes where the Chebyshev cycle is rejected a time step is
               # bound by its ~1300 launches of ~10 us, not by bytes.  Graded
               # channel, 0.97 M DoF: 0/1/1 25.2 applications 13.5 ms per step,
               # 0/1/2 20.8 / 13.4 ms, 1/1/1 22.1 / 17.1 ms; bare ILU(0) 52.6 /
               # 16.3 ms -- profiles/tlilu_r06.txt)
               'tlilu': {'pre': 0, 'post': 1, 'coarse_sweeps': 1},
               # which of the two runs the solves until the next rebuild:
               # 'rate' = the one with the better convergence rate per unit of
               # TIME, both measured at the rebuild
               # (newton_preconditioner.rate_verdict: `tl_probe_sweeps`
               # applications of each as a stationary iteration on the probe
               # vector pre-smoothed by `tl_probe_smooth` sweeps of the bare
               # smoother, timed); 'cycle' / 'bare': no test.  On the
               # structured channel at cell Peclet 3.5 the cycle saves a
               # quarter of the applications (15.2 -> 11.7) at twice the
               # launches each: 'rate' stays with the bare sweeps there (4.4
               # against 5.4 ms per step) and takes the cycle on the graded
               # mesh (16.3 -> 13.5 ms) and in the plume of config 4 (CFL 17,
               # diffusion number 16: ~100 applications per solve against
               # 200-800).
               'tl_select': 'rate', 'tl_probe_smooth': 6, 'tl_probe_sweeps': 4,
               # (the cycle is only taken where it contracts by at least this
               # per application; and a GMRES that has not converged with it
               # after `tl_maxit` applications is redone with the bare sweeps)
               'tl_accept': 0.9, 'tl_maxit': 400,
               # the sweeps read the factors rounded to fp32 (fp64 arithmetic):
               # half the bytes per application, same iteration counts
               'ilu_storage': 'fp32',
               # ... and keep their vector in fp32 as well (fp64 row sums):
               # the GMRES here is flexible (it stores M^-1 V_j and updates x
               # with it), the Arnoldi relation does not care how exactly the
               # preconditioner was applied
               'ilu_vector': 'fp32',
               # 'adaptive_forcing': a Newton iteration that CANNOT be the last
               # one -- the quadratic model of this flow regime, ||F_{k+1}|| ~
               # C ||F_k||^2 with C observed on the previous steps, predicts a
               # remainder above intermediate_margin * tol -- only has its
               # linear system solved to intermediate_fraction of that
               # predicted remainder: its iterate is a linearisation point,
               # and what the loose solve leaves in it the next Newton step
               # removes (Newton's map contracts an error of its argument by
               # C' |u_k - u*| ~ 4e-4 in the developed vortex street).  An
               # iterate that PASSES the Newton test against the prediction
               # is never accepted from a loose solve: that solve is first
               # continued to the tight tolerance (see `finish`).  OFF in mode
               # 'parity' (on in mode 'fast'), by measurement: in the burst
               # phases of the developed street (tools/developed_lab.py, t ~
               # 74: fraction 1e-2 / 1e-3 / 1e-4 / 1e-5 / 3e-6: 21.9 / 20.4 /
               # 18.6 / 19.2 / 20.0 ms per step, off: 22.3 -- the first solve
               # of a step drops from 10 to 3 applications, the second grows
               # from 6 to 9) it pays; over a whole run it does not
               # (tools/long_run.py, 4000 steps to t = 122: 32.1 k GMRES
               # applications, 13.0 ms per step with it against 29.6 k, 12.2
               # ms without).  What a loose solve finds goes into the history
               # the start vectors are extrapolated from; in the calmer phases
               # the tight solves live on starts that are good to 1e-5 (3 + 3
               # applications per step) and the loosely solved increments
               # cost them that (15).  No signal separates the phases without
               # a tight solve from a clean history: the Newton residuals and
               # max|u| do not (long_run: F1 = 2.9e-10 costs 4 + 3 at t = 70,
               # 10 + 7 at t = 74).
               'adaptive_forcing': False, 'matrix_free': True,
               'intermediate_fraction': 1.0e-4, 'intermediate_margin': 1.25,
               'finish_loose_solves': True,
               # start vector of the FIRST Newton iteration's linear solve:
               # 'extrapolated' = the Newton increments of the previous calls,
               # extrapolated linearly in time (a time loop's steps differ
               # little: fewer Krylov iterations to the same tolerance -- the
               # Newton path is untouched: still from u0, still a step solved
               # to linear_atol_factor * tol); 'zero' = nothing carried
               # 'linear_history': what the first system's start is
               # extrapolated FROM -- 'increments' = the first Newton
               # increments of the previous calls; 'total' = their TOTAL
               # increments u0 - ui (converged to the Newton tolerance whatever
               # the forcing of the single solves was: a loosely solved first
               # system cannot spoil this history).  With 'total' the later
               # iterations of a step whose first solve was loose start from
               # [predicted total - what the earlier iterations already
               # found], and such a step does not feed the per-iteration
               # histories.  `forcing_gate`: with 'total', 'adaptive_forcing'
               # only acts in steps whose predecessor's total increment was
               # predicted worse than this (relative l2) -- the bursts of a
               # vortex street (1e-3 .. 6e-2); in its calm phases (1e-5) the
               # tight solves are short anyway.
               'linear_history': 'increments', 'forcing_gate': 1.0e-3,
               'linear_start': 'extrapolated', 'linear_start_points': 5,
               # (cubic least-squares fit through 5 points: one point more than
               # interpolation needs averages the solver noise of the stored
               # increments instead of amplifying it; 0: interpolation)
               'linear_start_degree': 3,
               # 'previous' = always u0, the reference's choice (:204-220);
               # 'best' (mode 'fast'): see _compute_tentative_velocity
               'initial_guess': 'previous', 'guess_retry': 4},
    # 'two_level': Jacobi + aggregate coarse space (stands in for the
    # reference's hypre_amg, :331, :414); False = plain Jacobi
    # 'multigrid': smoothed-aggregation V-cycle (single GPU); else / sharded:
    # the two-level scheme below
    # 'extrapolate' (mode 'fast'): start vectors extrapolated in time
    'pressure': {'maxit': 200000, 'check_every': 10, 'two_level': True,
                 'multigrid': True, 'coarse_size': 4096, 'extrapolate': False,
                 # rows below which the multigrid hierarchy stops coarsening
                 # (dense inverse there)
                 'mg_coarsest': 4200,
                 # aggregates of the hierarchy: 'geometric' (patches of the
                 # dof coordinates) or 'algebraic' (from the matrix alone, as
                 # the reference's BoomerAMG: fem/multigrid.py)
                 'aggregation': 'geometric',
                 # start vector: p0 ('zero') or p0 + the previous increments
                 # extrapolated in time ('extrapolated'; Dirichlet branch)
                 'start': 'extrapolated', 'start_points': 5, 'start_degree': 3},
    # 'method': 'chebyshev' = mixed-precision defect correction with a fixed
    # Chebyshev polynomial of D^-1 M (flow_amd/fem/mass.py: 3-4 corrections of
    # one fp64 + `chebyshev_steps` - 1 fp16 products, no dot products) or 'cg'
    # (Jacobi-CG, 11-14 iterations; what the strips run)
    'correction': {'maxit': 10000, 'check_every': 2, 'extrapolate': False,
                   'method': 'chebyshev', 'chebyshev_steps': 6,
                   # solve for u1 - ui (flow_mass_solve_increment), started from
                   # the previous increments extrapolated in time ('zero': not)
                   'increment': True, 'increment_start': 'extrapolated',
                   'start_points': 5, 'start_degree': 3},
    }

_MODES = {
    'parity': {
        'newton': {'initial_guess': 'previous', 'linear_atol_factor': 1.0e-6,
                   'linear_remainder_fraction': 0.0,
                   'forcing': 0.0, 'adaptive_forcing': False,
                   'intermediate_fraction': 1.0e-4, 'intermediate_margin': 1.25,
                   'finish_loose_solves': True,
                   'linear_start': 'extrapolated'},
        'pressure': {'extrapolate': False, 'start': 'extrapolated'},
        'correction': {'extrapolate': False},
        },
    'fast': {
        'newton': {'initial_guess': 'best', 'linear_atol_factor': 0.02,
                   'linear_remainder_fraction': 0.0,
                   'forcing': 1.0e-4, 'adaptive_forcing': True,
                   'intermediate_fraction': 0.1, 'intermediate_margin': 1.0,
                   # (mode 'fast' accepts an iterate from a loose solve: its
                   # linear tolerance is a fraction of the Newton tolerance
                   # anyway)
                   'finish_loose_solves': False,
                   'linear_start': 'zero'},
        'pressure': {'extrapolate': True, 'start': 'zero'},
        'correction': {'extrapolate': True},
        },
    }


def set_mode(name):
    '''Switch solver_parameters between 'parity' (default: the reference's
    Newton path, nothing carried between calls) and 'fast' (module
    docstring).'''
    for group, values in _MODES[name].items():
        solver_parameters[group].update(values)
    solver_parameters['mode'] = name
    return


def _uses_history():
    if parallel.active():      # the strips run the parity path only
        return False
    return (solver_parameters['newton'].get('initial_guess') == 'best'
            or solver_parameters['pressure'].get('extrapolate', False)
            or solver_parameters['correction'].get('extrapolate', False))


def _uses_start_vectors():
    return (solver_parameters['newton'].get('linear_start') == 'extrapolated'
            or solver_parameters['pressure'].get('start') == 'extrapolated'
            or (solver_parameters['correction'].get('increment_start')
                == 'extrapolated'
                and solver_parameters['correction'].get('method',
                                                         'chebyshev')
                == 'chebyshev'))


def _history(lay):
    '''The previous call's fields, when this call continues its trajectory
    (u[0] IS the velocity it returned) and a setting that uses them is on;
    else None.'''
    hist = lay._dev.get('step_history')
    if hist is None or not _uses_history() or not hist.get('continuing'):
        return None
    return hist

# iteration counts / residuals of the most recent step()
last_step_info = {}

_THETA = {
    'forward euler': (0.0, 1.0),
    'backward euler': (1.0, 0.0),
    'crank-nicolson': (0.5, 0.5),
    }


def _persistent(lay, name, n):
    '''A work vector of the layout that keeps its ADDRESS from call to call
    (uninitialised): the vectors of a solver loop are arguments of its
    launches, and only a loop whose arguments repeat is replayed as a HIP
    graph (csrc/graph_replay.hip) instead of being captured again.'''
    key = ('persistent', name)
    buf = lay._dev.get(key)
    if buf is None or buf.numel() != n:
        buf = lay._dev[key] = device.empty(n)
    return buf


def _zeros(n):
    '''A zeroed work vector; the fill is a kernel of the library on its stream
    (no torch kernels inside a step).'''
    return _hip.fill(device.empty(n), 0.0)


def _bc_arrays(bcs, size):
    '''Sorted unique (dofs, values) on the device.'''
    dofs, vals = collect(bcs, size)
    # `collect` returns the same arrays while the conditions are unchanged:
    # upload once
    hit = _BC_UPLOADS.get(id(vals))
    if hit is not None and hit[0] is vals and hit[1] is dofs:
        return hit[2]
    out = (
        dofs, device.to_device(dofs.astype(numpy.int32)),
        device.to_device(vals.astype(numpy.float64)),
        )
    if len(_BC_UPLOADS) >= 16:
        _BC_UPLOADS.clear()
    _BC_UPLOADS[id(vals)] = (vals, dofs, out)
    return out


_BC_UPLOADS = {}


def _mesh_s(mesh):
    '''flow_mesh of the whole mesh, or of the calling rank's cells when the
    step runs on the strips of flow_amd.parallel.'''
    return parallel.mesh_view(mesh) if parallel.active() \
        else ops.mesh_struct(mesh)


def _space_s(layout):
    return parallel.view(layout).space if parallel.active() \
        else ops.space_struct(layout)


def _bc_mask(dofs, n, comp=None):
    '''uint8 mask of length n for the dofs of component `comp`.'''
    mask = numpy.zeros(n, dtype=numpy.uint8)
    if comp is None:
        mask[dofs] = 1
    else:
        sel = dofs[(dofs >= comp * n) & (dofs < (comp + 1) * n)] - comp * n
        mask[sel] = 1
    return mask


def _as_ilu(kind, pre):
    '''What ops.krylov_solve takes as `ilu`: the factors themselves, or the
    front of the two-level cycle they smooth (flow_amd/fem/tlilu.py).'''
    if kind == 'ilu0':
        return pre
    if kind == 'tlilu':
        return pre.front if getattr(pre, 'use_cycle', True) else pre.fine
    return None


class _Bare(object):
    '''An ILU(0) as the probe of newton_preconditioner.contraction sees a
    preconditioner.'''

    def __init__(self, factors):
        self.factors = factors

    def apply(self, r, z):
        return self.factors.solve(r, z)


def _remainder_tolerance(lay, npar, lin_atol, nrm, tol):
    """The absolute tolerance of a Newton system's linear solve, raised to
    `linear_remainder_fraction` of the remainder C ||F||^2 the quadratic model
    predicts for this Newton step (solver_parameters['newton']), capped."""
    frac = npar.get('linear_remainder_fraction', 0.0)
    quad_c = lay._dev.get('newton_quad_C')
    if not frac or quad_c is None or not numpy.isfinite(quad_c):
        return lin_atol
    predicted = quad_c * nrm * nrm
    return max(lin_atol, min(frac * predicted,
                             npar.get('linear_atol_cap', 1.0e-4) * tol))


def _compute_tentative_velocity(
        u, p0, f, u_bcs, time_step_method, rho, mu, dt, v=None,
        tol=1.0e-10
        ):
    '''Solve F1(ui) = 0 (reference :147-255): F1 scaled with dt/rho,
    time_step_method in {forward euler, backward euler, crank-nicolson},
    Newton from ui = u[0] with the exact Jacobian, at most 10 iterations,
    converged when ||F||_2 < tol (absolute; relative_tolerance 0),
    RuntimeError otherwise.'''
    lib = _hip.lib()
    assert time_step_method in _THETA, time_step_method
    theta_i, theta_e = _THETA[time_step_method]
    alpha = 1.0
    if parallel.active():
        return _tentative_velocity_on_strips(
            u, p0, f, u_bcs, theta_i, theta_e, rho, mu, dt, tol), alpha
    W = u[0].function_space()
    P = p0.function_space()
    mesh = W.mesh()
    lay = W.layout
    nc = mesh.num_cells()
    n2 = W.size()

    # initial guess: previous velocity (reference :204-220) ...
    # (in a buffer of the layout, the same for every call: the iterate's
    # address is an argument of every launch of the Newton-Krylov loop, and a
    # loop whose arguments do not change is replayed as a HIP graph --
    # csrc/graph_replay.hip.  The tentative velocity reported in
    # last_step_info lives until the next call on this space.)
    # (... only where a replay is possible: otherwise the iterate is a vector
    # of its own and `last_step_info['tentative_velocity']` stays what this
    # call computed)
    ui = Function(W, ops.copy(
        _persistent(lay, 'newton_iterate', n2) if _hip.graphs_possible()
        else device.empty(n2), u[0].data))
    # ... or ('initial_guess': 'best'), when this call continues the trajectory
    # of the previous one (u[0] IS the velocity the last step returned), the
    # previous step's TENTATIVE velocity if its residual is smaller (choice (2)
    # of the reference's comment, :204-220: worse than u0 in a transient, but
    # as the flow settles it is almost the solution -- 1-2 Newton iterations
    # instead of 4 on the developed Karman flow), or that tentative velocity
    # extrapolated linearly in time through the one before it (at settled step
    # sizes: initial residual 1.5e-9 -> 2e-10 at CFL-sized steps, 4-5 GMRES
    # iterations instead of 6-8).  The guess only changes the Newton path, not
    # what it converges to.  Which candidate won is remembered; the others are
    # re-tried every few steps.
    hist = _history(lay)
    candidates = ['u0']
    if hist is not None and 'ui' in hist and \
            solver_parameters['newton'].get('initial_guess') == 'best':
        settled = 'ui_prev' in hist and 0.7 <= dt / hist['dt'] <= 1.5
        if hist.get('countdown', 0) > 0 and (
                hist['winner'] != 'ux' or settled):
            hist['countdown'] -= 1
            candidates = [hist['winner']]
        else:
            candidates = ['u0', 'ui']
            # ... or that tentative velocity extrapolated linearly through
            # the one before it (settled step sizes only)
            if settled:
                candidates.append('ux')

    f0 = as_cell_coefficient(f[0], mesh, 2)
    f1 = as_cell_coefficient(f[1], mesh, 2)
    f0s, keep0 = ops.coef_struct(f0, mesh, lay.degree)
    f1s, keep1 = ops.coef_struct(f1, mesh, lay.degree)
    prm = _hip.NsParams(dt, rho, mu, theta_i, theta_e)
    bc_dofs_host, bc_dofs, bc_vals = _bc_arrays(u_bcs, n2)
    nbc = bc_dofs.numel()
    bfmask = mesh._cache.get('bfmask_dev')
    if bfmask is None:
        bfmask = device.to_device(mesh.cell_bfacet_mask())
        mesh._cache['bfmask_dev'] = bfmask

    J = lay._dev.get('jacobian')
    if J is None:
        J = ops.Matrix(lay, 2)
        lay._dev['jacobian'] = J
    F = device.empty(n2)
    dx = device.empty(n2)
    buf = ops.scratch(mesh, max(2 * lay.nloc, 4 * lay.nloc**2) * nc)
    ms = ops.mesh_struct(mesh)
    ws = ops.space_struct(lay)
    ps = ops.space_struct(P.layout)
    st = _hip.stream()
    npar = solver_parameters['newton']

    def assemble(want_f, want_j):
        _hip.check(lib.flow_assemble_momentum(
            ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(ps),
            _hip.i32(bfmask, nc, 'bfmask'), _hip.f64(ui.data, n2),
            _hip.f64(u[0].data, n2), _hip.f64(p0.data, P.size()),
            ctypes.byref(f0s), ctypes.byref(f1s), ctypes.byref(prm),
            _hip.f64(buf), _hip.f64(F, n2) if want_f else None,
            _hip.f64(J.vals, 4 * J.stride) if want_j else None, J.stride, st
            ))

    def residual():
        assemble(True, False)
        _hip.check(lib.flow_bc_residual(
            nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(ui.data),
            _hip.f64(F), st
            ))
        return ops.vector_norm(F)

    def load_candidate(name):
        ops.copy(ui.data, u[0].data if name == 'u0' else hist['ui'])
        if name == 'ux':
            r = dt / hist['dt']
            ops.axpby(r, hist['ui'], 1.0, ui.data)
            ops.axpby(-r, hist['ui_prev'], 1.0, ui.data)

    # pick the start among the candidates by the residual it leaves
    first_nrm = None
    if candidates != ['u0']:
        best = None
        for name in candidates:
            load_candidate(name)
            nrm_c = residual()
            # the extrapolated start has to be clearly better: on a settled
            # flow the two tentative velocities it is built from differ by
            # solver noise only, which the extrapolation amplifies
            score = 3.0 * nrm_c if name == 'ux' else nrm_c
            if best is None or score < best[3]:
                best = (name, nrm_c, _hip.clone(F) if len(candidates) > 1
                        else None, score)
        if len(candidates) > 1:
            hist['winner'] = best[0]
            hist['countdown'] = npar['guess_retry']
            if best[0] != candidates[-1]:        # not the one F belongs to
                load_candidate(best[0])
                ops.copy(F, best[2])
        first_nrm = best[1]
        last_step_info['initial_guess'] = best[0]

    history = []
    linear_its = []
    applications = []
    linear_residuals = []
    last_linear_residual = 0.0
    it = 0
    Jop = None
    finish = None       # set by an iteration whose linear solve was loose
    # 'linear_history': 'total' (see solver_parameters)
    by_total = npar.get('linear_history', 'increments') == 'total' and \
        npar.get('linear_solver', 'gmres') == 'gmres'
    total_pred = found = None
    loose_step = False
    while True:
        if first_nrm is not None:
            nrm, first_nrm = first_nrm, None
        else:
            nrm = residual()
        if nrm < tol and finish is not None and \
                npar.get('finish_loose_solves', True):
            # The quadratic model said this iterate could not pass the Newton
            # test, so its linear system was only solved loosely
            # ('adaptive_forcing') -- and it passes.  An accepted iterate must
            # come from a tight solve (the pressure sees what the linear solve
            # leaves in div u): continue THAT solve to the tight tolerance from
            # where it stopped, and test again.
            info('Newton iteration %d passes against the prediction: its '
                 'linear solve is continued to the tight tolerance' % it)
            nrm, last_linear_residual = finish()
            linear_residuals[-1] = last_linear_residual
            last_step_info['newton_finished_loose_solves'] = \
                last_step_info.get('newton_finished_loose_solves', 0) + 1
        finish = None
        if history and history[-1] > 0.0:
            # quadratic-model constant  ||F_{k+1}|| ~ C ||F_k||^2  of this flow
            # regime (kept across time steps); the part of ||F_{k+1}|| that is
            # the residual of the inexact linear solve is taken out first
            quad = numpy.sqrt(max(nrm**2 - last_linear_residual**2, 0.0))
            lay._dev['newton_quad_C'] = quad / history[-1]**2
        history.append(nrm)
        info('Newton iteration %d: r (abs) = %.3e (tol = %.3e)' % (it, nrm, tol))
        if nrm < tol:
            break
        if it >= npar['maximum_iterations'] or not numpy.isfinite(nrm):
            raise RuntimeError(
                'Newton solver did not converge after %d iterations '
                '(residual history %r)' % (it, history)
                )
        def assemble_jacobian():
            assemble(False, True)
            _hip.check(lib.flow_bc_identity_rows(
                ctypes.byref(J.operator()), _hip.f64(J.vals),
                _hip.i32(lay.dev('diag_idx')), nbc, _hip.i32(bc_dofs), st
                ))

        ops.fill(dx, 0.0)
        dx_is_zero = True
        hkey = _newton_key(it)
        extrapolate = npar.get('linear_start') == 'extrapolated'
        if extrapolate and by_total and it == 0:
            dx_is_zero = not _extrapolated_increment(
                lay, dt, dx, int(npar.get('linear_start_points', 5)),
                key='newton_total', degree=npar.get('linear_start_degree'))
            if not dx_is_zero:
                total_pred = ops.copy(_persistent(lay, 'newton_total_pred', n2),
                                      dx)
        elif extrapolate and by_total and loose_step and total_pred is not None:
            # predicted total increment minus what the iterations before found
            ops.copy(dx, total_pred)
            ops.axpby(-1.0, found, 1.0, dx)
            dx_is_zero = False
        elif hkey is not None and extrapolate:
            dx_is_zero = not _extrapolated_increment(
                lay, dt, dx, int(npar.get('linear_start_points', 5)),
                key=hkey, degree=npar.get('linear_start_degree'))
        pre = None
        kind = npar.get('preconditioner', 'jacobi')
        use_gmres = npar.get('linear_solver', 'gmres') == 'gmres'
        key = (rho, mu, theta_i, nbc, hash(bc_dofs_host.tobytes()))
        # the p-multigrid needs a P2 space and, its application not being
        # exactly linear, the flexible GMRES; and it is only used where its
        # cycle contracts (see _build_preconditioner): a rejection stands until
        # the problem changes or the step size has halved
        if kind == 'pmg':
            rej = lay._dev.get('pmg_rejected')
            if lay.degree != 2 or not use_gmres:
                kind = 'ilu0'
            elif rej is not None and rej[0] == key and dt > 0.5 * rej[1]:
                kind = npar.get('fallback', 'ilu0')
        if kind == 'tlilu' and (lay.degree != 2 or not use_gmres):
            kind = 'ilu0'
        with_ilu = kind in ('ilu0', 'pmg', 'tlilu')
        # matrix-free Newton-Krylov: J(ui) is applied cell by cell (as cheap
        # as the assembled 2x2-block SpMV) and only assembled when the lagged
        # preconditioner has to be rebuilt
        matfree = with_ilu and npar.get('matrix_free', True)
        if not matfree:
            assemble_jacobian()
        if matfree and Jop is None:
            Jop = ops.MomentumJacobian.cached(W, bfmask, ui.data, prm, bc_dofs)
        operator = Jop if matfree else J
        refactored = False

        def build(kind):
            '''The lagged preconditioner of `kind`, rebuilt when the problem
            itself changed, when dt has moved by more than `ilu_lag` since the
            last factorisation (checked at the first Newton iteration of a
            step), or when it has gone stale: a solve needed more than twice
            the iterations the fresh one needed (while the flow spins up at
            tiny dt the matrix is mass-dominated and old factors stay good; at
            CFL-sized steps they do not).  Returns (kind, pre, rebuilt): a
            p-multigrid whose cycle does not contract is replaced by the
            ILU(0).'''
            from ..fem import ilu
            slot = {'ilu0': 'jacobian_ilu', 'pmg': 'jacobian_pmg',
                    'tlilu': 'jacobian_tl'}[kind]
            pre = lay._dev.get(slot)
            if not (pre is None or pre.key != key or pre.stale or (
                    it == 0 and not (1.0 / npar['ilu_lag'] <= dt / pre.dt
                                     <= npar['ilu_lag']))):
                return kind, pre, False
            if matfree:
                assemble_jacobian()
            # (how often that happens: cumulative, reported with every step)
            lay._dev['newton_rebuilds'] = lay._dev.get('newton_rebuilds', 0) + 1
            last_step_info['newton_preconditioner_rebuilds'] = \
                lay._dev['newton_rebuilds']
            if kind == 'pmg':
                if pre is None:
                    from ..fem.pmg import Pmg
                    pre = Pmg(W, **npar.get('pmg', {}))
                    lay._dev[slot] = pre
                J1 = _coarse_jacobian(
                    pre, W, P, ui, p0, f0, f1, prm, bfmask, bc_dofs_host)
                pre.refactor(J, J1, mass_share=_mass_share(pre, J1)
                             if pre.coarse_auto else None)
                last_step_info['pmg_coarse_steps'] = pre.struct.coarse_steps
                pre.dt, pre.key, pre.stale = dt, key, False
                # Chebyshev smoothing assumes a spectrum near the real axis:
                # on an under-resolved convection-dominated problem (cell
                # Peclet number >> 1 with the Galerkin discretisation) the
                # cycle amplifies instead.  One application on a fixed
                # full-spectrum vector tells: |v - M^-1 J v| / |v| is ~0.3
                # where the cycle works and > 1 where it does not.
                pre.contraction = _contraction(pre, operator, lay, bc_dofs_host)
                last_step_info['pmg_contraction'] = pre.contraction
                if not pre.contraction < npar.get('pmg_accept', 0.8):
                    lay._dev['pmg_rejected'] = (key, dt)
                    # (once the rejection lapses the cycle is refactored and
                    # tested again, not taken from the slot as it is)
                    pre.stale = True
                    fallback = npar.get('fallback', 'ilu0')
                    info('p-multigrid rejected (contraction %.2f): %s'
                         % (pre.contraction, fallback))
                    # (J and the P1 level have just been reassembled in the
                    # buffers the fallback's residuals read: it is refactored
                    # from them, not taken from its slot as it is)
                    for other in ('jacobian_ilu', 'jacobian_tl'):
                        if lay._dev.get(other) is not None:
                            lay._dev[other].stale = True
                    return build(fallback)
                return kind, pre, True
            if kind == 'tlilu':
                if pre is None:
                    from ..fem.tlilu import TwoLevelIlu
                    pre = TwoLevelIlu(W, **npar.get('tlilu', {}))
                    lay._dev[slot] = pre
                J1 = _coarse_jacobian(
                    pre, W, P, ui, p0, f0, f1, prm, bfmask, bc_dofs_host)
                pre.refactor(J, J1)
                pre.dt, pre.key, pre.stale = dt, key, False
                # the self-test of every rebuild: is the cycle worth what it
                # costs?  (newton_preconditioner.choose_cycle)
                pre.use_cycle, probe = _choose_cycle(
                    pre, _Bare(pre.fine), operator, lay, bc_dofs_host, npar)
                pre.contraction, pre.contraction_bare = probe[0], probe[1]
                last_step_info['tl_contraction'] = probe
                if not pre.use_cycle:
                    info('two-level ILU cycle: contraction %.2f in %.2f ms, its '
                         'smoother alone %.2f in %.2f ms: bare ILU(0)' % (
                             probe[0], probe[2], probe[1], probe[3]))
                return kind, pre, True
            if pre is None:
                # (the fp32 sweep vector makes the application slightly
                # nonlinear: for the flexible GMRES only, flow_hip.h)
                pre = ilu.Ilu0(
                    J, packed=npar.get('ilu_storage', 'fp32') == 'fp32',
                    single_vector=use_gmres and npar.get('ilu_vector') == 'fp32')
                lay._dev[slot] = pre
            else:
                pre.refactor(J)
            pre.dt, pre.key, pre.stale = dt, key, False
            return kind, pre, True

        if with_ilu:
            kind, pre, refactored = build(kind)
        # Inexact Newton: the linear residual only has to get below what the
        # quadratic term leaves anyway (forcing term 1e-4 ||F||), and below a
        # fraction of the Newton tolerance so that one more step is never
        # needed because of the linear solve.
        lin_atol = max(npar['linear_atol_factor'] * tol, npar['forcing'] * nrm)
        lin_atol = _remainder_tolerance(lay, npar, lin_atol, nrm, tol)
        # Eisenstat-Walker style: when the quadratic model (constant observed
        # on the previous Newton steps) predicts that this step cannot reach
        # the tolerance anyway, the linear residual only has to stay below a
        # tenth of the predicted remainder.
        quad_c = lay._dev.get('newton_quad_C')
        tight_atol = lin_atol
        gate_open = not by_total or lay._dev.get(
            'newton_prediction_error', 1.0) > npar.get('forcing_gate', 1.0e-3)
        if quad_c is not None and npar.get('adaptive_forcing', True) \
                and use_gmres and gate_open:
            predicted = quad_c * nrm * nrm
            if predicted > npar.get('intermediate_margin', 1.0) * tol:
                lin_atol = max(lin_atol, min(
                    npar.get('intermediate_fraction', 0.1) * predicted,
                    1.0e-2 * nrm))
        lin_rtol = max(npar['linear_rtol'], lin_atol / nrm)
        if use_gmres:
            # GMRES(restart): one Jacobian action + one preconditioner
            # application per iteration, the least of the Krylov methods here
            # (how many Arnoldi steps to enqueue before the first read-back:
            # what Newton iteration `it` of the previous call needed -- a
            # scheduling hint, the accepted iterate does not depend on it)
            expected = lay._dev.setdefault('gmres_expected', {})

            def gmres(maxit):
                return ops.krylov_solve(
                    'gmres', operator, F, dx, rtol=lin_rtol, atol=0.0,
                    maxit=maxit, ilu=_as_ilu(kind, pre),
                    pmg=pre if kind == 'pmg' else None,
                    restart=npar['gmres_restart'], x_is_zero=dx_is_zero,
                    dinv='jacobi' if pre is None else None,
                    first_check=expected.get(it, 0),
                    # (the next Newton residual is the check of this solve)
                    verify=False)
            if kind == 'pmg':
                # second line of defence behind the contraction test: a solve
                # that does not get there in `pmg_maxit` applications is redone
                # with the ILU(0)
                try:
                    sol = gmres(min(npar['linear_maxit'],
                                    npar.get('pmg_maxit', 150)))
                except _hip.NotConverged:
                    lay._dev['pmg_rejected'] = (key, dt)
                    pre.stale = True
                    fallback = npar.get('fallback', 'ilu0')
                    info('p-multigrid: GMRES stalled, redone with %s' % fallback)
                    ops.fill(dx, 0.0)
                    dx_is_zero = True
                    kind, pre, refactored = build(fallback)
                    sol = gmres(npar['linear_maxit'])
            elif kind == 'tlilu' and getattr(pre, 'use_cycle', False):
                try:
                    sol = gmres(min(npar['linear_maxit'],
                                    npar.get('tl_maxit', 400)))
                except _hip.NotConverged:
                    info('two-level ILU cycle: GMRES stalled, redone with the '
                         'bare ILU(0)')
                    pre.use_cycle = False
                    ops.fill(dx, 0.0)
                    dx_is_zero = True
                    sol = gmres(npar['linear_maxit'])
            else:
                sol = gmres(npar['linear_maxit'])
            expected[it] = sol.iterations
            # counted like BiCGStab iterations (two applications each) for the
            # staleness test of the lagged factors below
            its = (sol.iterations + 1) // 2
            applications.append(sol.iterations)
        else:
            sol, its = _bicgstab_with_restarts(operator, F, dx, lin_rtol, pre, npar)
            applications.append(2 * its)
        last_linear_residual = sol.residual
        linear_residuals.append(sol.residual)
        linear_its.append(its)
        last_step_info['newton_preconditioner'] = 'ilu0' if (
            kind == 'tlilu' and not getattr(pre, 'use_cycle', True)) else kind
        if pre is not None:
            _age(pre, kind, refactored, its, sol.iterations, npar, it=it)
        if lin_atol > tight_atol:
            loose_step = True
        if by_total:
            # (the sum of this step's increments so far)
            if it == 0:
                found = ops.copy(_persistent(lay, 'newton_found', n2), dx)
            else:
                ops.axpby(1.0, dx, 1.0, found)
        if hkey is not None and npar.get('linear_start') == 'extrapolated' \
                and not (by_total and (it == 0 or loose_step)):
            _remember_increment(lay, dt, dx, key=hkey)
        ops.axpby(-1.0, dx, 1.0, ui.data)
        if use_gmres and lin_atol > tight_atol:
            def finish(operator=operator, pre=pre, kind=kind, hkey=hkey,
                       tight=tight_atol, slot=len(applications) - 1):
                '''Back to this iteration's linearisation point (the operator
                reads ui), the same system again from the increment the loose
                solve found, now to the tight tolerance; returns the Newton
                residual of the corrected iterate and the linear residual.'''
                ops.axpby(1.0, dx, 1.0, ui.data)
                nrm0 = residual()
                sol = ops.krylov_solve(
                    'gmres', operator, F, dx,
                    rtol=max(npar['linear_rtol'], tight / nrm0), atol=0.0,
                    maxit=npar['linear_maxit'],
                    ilu=_as_ilu(kind, pre),
                    pmg=pre if kind == 'pmg' else None,
                    restart=npar['gmres_restart'], x_is_zero=False,
                    dinv='jacobi' if pre is None else None, verify=False)
                applications[slot] += sol.iterations
                if hkey is not None and not by_total and \
                        npar.get('linear_start') == 'extrapolated':
                    _remember_increment(lay, dt, dx, key=hkey)
                ops.axpby(-1.0, dx, 1.0, ui.data)
                return residual(), sol.residual
        it += 1
    del keep0, keep1
    if by_total and it > 0 and npar.get('linear_start') == 'extrapolated':
        # the step's total increment u0 - ui: Newton-converged whatever the
        # single solves were; and how well it had been predicted
        total = _persistent(lay, 'newton_total', n2)
        ops.copy(total, u[0].data)
        ops.axpby(-1.0, ui.data, 1.0, total)
        _remember_increment(lay, dt, total, key='newton_total')
        if npar.get('adaptive_forcing', False):
            if total_pred is not None:
                tn = ops.vector_norm(total)
                ops.axpby(-1.0, total, 1.0, total_pred)
                err = ops.vector_norm(total_pred) / tn if tn > 0.0 else 0.0
            else:
                err = 1.0
            lay._dev['newton_prediction_error'] = err
            last_step_info['newton_prediction_error'] = err
    last_step_info['newton_residuals'] = history
    last_step_info['newton_linear_iterations'] = linear_its
    # operator + preconditioner applications of the linear solves
    last_step_info['newton_linear_applications'] = applications
    # absolute residual each Newton iteration's linear solve stopped at (the
    # last one: always <= linear_atol_factor * tol, see `finish`)
    last_step_info['newton_linear_residuals'] = linear_residuals
    return ui, alpha


def _tentative_velocity_on_strips(u, p0, f, u_bcs, theta_i, theta_e, rho, mu,
                                  dt, tol):
    '''The Newton iteration of _compute_tentative_velocity on the x-strips of
    flow_amd.parallel: the same path (start u0, near-exact steps, stop at
    ||F|| < tol), every rank on its cells and rows.  Residual and Jacobian
    action are evaluated over the rank's cells and gathered for its owned
    rows; the linear systems are solved by GMRES with the rank's own ILU(0)
    (block Jacobi); ||F|| is the norm over all ranks.  Fields are valid on the
    owned + ghost rows on entry and on exit.'''
    lib = _hip.lib()
    W = u[0].function_space()
    P = p0.function_space()
    mesh = W.mesh()
    lay = W.layout
    nc = mesh.num_cells()
    n2 = W.size()
    npar = solver_parameters['newton']
    # (the strips run GMRES + block-Jacobi ILU(0), whatever the single-GPU
    # preconditioner is)
    assert npar.get('preconditioner', 'ilu0') in ('ilu0', 'pmg') and \
        npar.get('linear_solver', 'gmres') == 'gmres', \
        'the strips run GMRES + block-Jacobi ILU(0)'
    ms = parallel.mesh_view(mesh)
    wv = parallel.view(lay)
    pv = parallel.view(P.layout)
    ui = Function(W, _hip.clone(u[0].data))
    f0 = as_cell_coefficient(f[0], mesh, 2)
    f1 = as_cell_coefficient(f[1], mesh, 2)
    f0s, keep0 = ops.coef_struct(f0, mesh, lay.degree)
    f1s, keep1 = ops.coef_struct(f1, mesh, lay.degree)
    prm = _hip.NsParams(dt, rho, mu, theta_i, theta_e)
    bc_dofs_host, bc_dofs, bc_vals = _bc_arrays(u_bcs, n2)
    nbc = bc_dofs.numel()
    bfmask = mesh._cache.get('bfmask_dev')
    if bfmask is None:
        bfmask = device.to_device(mesh.cell_bfacet_mask())
        mesh._cache['bfmask_dev'] = bfmask
    J = lay._dev.get('jacobian')
    if J is None:
        J = ops.Matrix(lay, 2)
        lay._dev['jacobian'] = J
    F = _zeros(n2)
    dx = _zeros(n2)
    buf = ops.scratch(mesh, max(2 * lay.nloc, 4 * lay.nloc**2) * nc)
    st = _hip.stream()

    def assemble(want_f, want_j):
        _hip.check(lib.flow_assemble_momentum(
            ctypes.byref(ms), ctypes.byref(wv.space), ctypes.byref(pv.space),
            _hip.i32(bfmask, nc, 'bfmask'), _hip.f64(ui.data, n2),
            _hip.f64(u[0].data, n2), _hip.f64(p0.data, P.size()),
            ctypes.byref(f0s), ctypes.byref(f1s), ctypes.byref(prm),
            _hip.f64(buf), _hip.f64(F, n2) if want_f else None,
            _hip.f64(J.vals, 4 * J.stride) if want_j else None, J.stride, st
            ))

    history = []
    applications = []
    linear_its = []
    it = 0
    state = {}
    key = (rho, mu, theta_i, nbc, hash(bc_dofs_host.tobytes()),
           parallel.comm().world)
    while True:
        assemble(True, False)
        _hip.check(lib.flow_bc_residual(
            nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(ui.data),
            _hip.f64(F), st
            ))
        nrm = numpy.sqrt(parallel.dot(F, F, lay, 2))
        if history and history[-1] > 0.0:
            # (quadratic-model constant, as on one GPU)
            lay._dev['newton_quad_C'] = nrm / history[-1]**2
        history.append(nrm)
        info('Newton iteration %d: r (abs) = %.3e (tol = %.3e)' % (it, nrm, tol))
        if nrm < tol:
            break
        if it >= npar['maximum_iterations'] or not numpy.isfinite(nrm):
            raise RuntimeError(
                'Newton solver did not converge after %d iterations '
                '(residual history %r)' % (it, history)
                )
        def jacobian_action():
            held = state.get('Jop')
            if held is None:
                held = state['Jop'] = ops.MomentumJacobian.cached(
                    W, bfmask, ui.data, prm, bc_dofs, mesh_s=ms,
                    space_s=wv.space)
            return held

        # lagged block-Jacobi preconditioner -- the rank's own two-level cycle
        # (P2 spaces) or ILU(0) of its diagonal block --, rebuilt by the same
        # rules as on one GPU; `stale` is decided from the iteration count,
        # which is the same on every rank
        kind = 'pmg' if (npar.get('preconditioner') == 'pmg'
                         and lay.degree == 2) else 'ilu0'
        rej = lay._dev.get('pmg_rejected_strip')
        if kind == 'pmg' and rej is not None and rej[0] == key \
                and dt > 0.5 * rej[1]:
            kind = 'ilu0'

        def build(kind):
            slot = 'jacobian_%s_strip' % ('ilu' if kind == 'ilu0' else 'pmg')
            pre = lay._dev.get(slot)
            if not (pre is None or pre.key != key or pre.stale or (
                    it == 0 and not (1.0 / npar['ilu_lag'] <= dt / pre.dt
                                     <= npar['ilu_lag']))):
                return kind, pre, False
            assemble(False, True)
            _hip.check(lib.flow_bc_identity_rows(
                ctypes.byref(J.operator()), _hip.f64(J.vals),
                _hip.i32(lay.dev('diag_idx')), nbc, _hip.i32(bc_dofs), st
                ))
            if kind == 'pmg':
                if pre is None:
                    pre = parallel.local_pmg(W, **npar.get('pmg', {}))
                    lay._dev[slot] = pre
                pre.refactor(J, _coarse_jacobian(
                    pre, W, P, ui, p0, f0, f1, prm, bfmask, bc_dofs_host,
                    mesh_s=ms, space1_s=parallel.view(pre.lay1).space,
                    pspace_s=pv.space))
                pre.dt, pre.key, pre.stale = dt, key, False
                # (the contraction test of the single-GPU path, summed over
                # the ranks: the same verdict everywhere)
                pre.contraction = _contraction_on_strips(
                    pre, jacobian_action(), lay, bc_dofs_host)
                last_step_info['pmg_contraction'] = pre.contraction
                if not pre.contraction < npar.get('pmg_accept', 0.8):
                    lay._dev['pmg_rejected_strip'] = (key, dt)
                    pre.stale = True
                    return build('ilu0')
                return 'pmg', pre, True
            elif pre is None:
                pre = parallel.local_ilu(
                    J, packed=npar.get('ilu_storage', 'fp32') == 'fp32',
                    single_vector=npar.get('ilu_vector') == 'fp32')
                lay._dev[slot] = pre
            else:
                pre.refactor(J)
            pre.dt, pre.key, pre.stale = dt, key, False
            return kind, pre, True

        kind, pre, refactored = build(kind)
        Jop = jacobian_action()
        lin_atol = max(npar['linear_atol_factor'] * tol, npar['forcing'] * nrm)
        lin_atol = _remainder_tolerance(lay, npar, lin_atol, nrm, tol)
        lin_rtol = max(npar['linear_rtol'], lin_atol / nrm)
        ops.fill(dx, 0.0)
        dx_is_zero = True
        hkey = _newton_key(it)
        if hkey is not None and npar.get('linear_start') == 'extrapolated':
            # (every rank keeps the increments of its own rows: the same
            # history length and step sizes everywhere -- the ranks agree on
            # the trajectory a call belongs to, start_vectors._resolve)
            dx_is_zero = not _extrapolated_increment(
                lay, dt, dx, int(npar.get('linear_start_points', 5)),
                key=hkey, degree=npar.get('linear_start_degree'))
        # (the count of the previous call's Newton iteration `it`: the same
        # number on every rank -- they all ran the same solve)
        expected = lay._dev.setdefault('gmres_expected_strip', {})
        def solve(maxit):
            return parallel.gmres(Jop, pre, F, dx, rtol=lin_rtol, atol=0.0,
                                  maxit=maxit, restart=npar['gmres_restart'],
                                  x_is_zero=dx_is_zero,
                                  expected=expected.get(it, 0))
        if kind == 'pmg':
            # (no contraction test on the strips: a GMRES that has not
            # converged after `pmg_maxit` applications -- the count is the same
            # on every rank -- is redone with the block ILU(0) everywhere)
            try:
                sol = solve(min(npar['linear_maxit'], npar.get('pmg_maxit', 150)))
            except _hip.NotConverged:
                lay._dev['pmg_rejected_strip'] = (key, dt)
                pre.stale = True
                ops.fill(dx, 0.0)
                dx_is_zero = True
                kind, pre, refactored = build('ilu0')
                sol = solve(npar['linear_maxit'])
        else:
            sol = solve(npar['linear_maxit'])
        expected[it] = sol.iterations
        its = (sol.iterations + 1) // 2
        applications.append(sol.iterations)
        linear_its.append(its)
        last_step_info['newton_preconditioner'] = kind + ' (block Jacobi)'
        _age(pre, kind, refactored, its, sol.iterations, npar, it=it)
        if hkey is not None and npar.get('linear_start') == 'extrapolated':
            _remember_increment(lay, dt, dx, key=hkey)
        # (dx is zero outside the owned rows)
        ops.axpby(-1.0, dx, 1.0, ui.data)
        parallel.halo(ui.data, lay, 2)
        it += 1
    del keep0, keep1
    last_step_info['newton_residuals'] = history
    last_step_info['newton_linear_iterations'] = linear_its
    last_step_info['newton_linear_applications'] = applications
    return ui


def _bicgstab_with_restarts(A, b, x, rtol, pre, npar):
    '''BiCGStab can stagnate when its bi-orthogonality degrades: restart from
    the current iterate every `restart` iterations (x is updated in place also
    when the solver reports non-convergence).  Returns (info, iterations).'''
    its = 0
    while True:
        chunk = min(npar['restart'], npar['linear_maxit'] - its)
        try:
            sol = ops.krylov_solve(
                'bicgstab', A, b, x, rtol=rtol, atol=0.0, maxit=chunk,
                check_every=npar['check_every'], ilu=pre
                )
            return sol, its + sol.iterations
        except _hip.NotConverged:
            its += chunk
            if its >= npar['linear_maxit']:
                raise


def _pressure_cg(A, dinv, prec, b, x, tol, par, fallback=False):
    '''CG for the pressure system: rtol = tol, atol = 0 (reference :332-335,
    :420-422), preconditioned with the multigrid V-cycle, with Jacobi + the
    aggregate coarse space, or with Jacobi alone (`prec` from
    _preconditioner); row-sharded over the GPUs of the node when
    flow_amd.parallel is enabled and the system is large enough for that to pay
    (parallel.min_rows()).  fallback: what the guarded start x is dropped for
    when it leaves a larger preconditioned residual than zero would (a vector,
    or False: zero at once; ops.krylov_solve `guard`).'''
    coarse, mg = prec
    if parallel.active():
        if mg is not None:
            # the same V-cycle, finest level cut into the ranks' strips
            return parallel.mgcg(A, dinv, mg, b, x, tol, 0.0, par['maxit'],
                                 check_every=2, tag='pressure', guard=fallback)
        # (a system too small to coarsen: plain Jacobi-CG on the strips)
        return parallel.cg(A, dinv, b, x, tol, 0.0, par['maxit'],
                           check_every=par['check_every'], tag='pressure',
                           guard=fallback)
    return ops.krylov_solve(
        'cg', A, b, x, rtol=tol, atol=0.0, maxit=par['maxit'], dinv=dinv,
        check_every=2 if mg is not None else par['check_every'],
        coarse=coarse, mg=mg, tag='pressure', guard=fallback
        )


def _pressure_from(A, dinv, prec, b, x, tol, par, plain, wlay):
    '''_pressure_cg from the start in x; `plain`: the start without the
    extrapolated increment (None: x is that start).  A solve that does not
    converge from an extrapolated start -- the guard has already ruled out
    that the start was worse than zero -- forgets the trajectory's history and
    is redone once from the plain start before the error goes to the caller
    (dolfin: 'error_on_nonconvergence').'''
    try:
        return _pressure_cg(A, dinv, prec, b, x, tol, par,
                            fallback=plain if plain is not None else False)
    except _hip.NotConverged:
        if plain is None:
            raise
        info('pressure: no convergence from the extrapolated start; history '
             'dropped, redone from the plain start')
        start_vectors.drop_current(wlay)
        ops.copy(x, plain)
        return _pressure_cg(A, dinv, prec, b, x, tol, par, fallback=False)


def _preconditioner(lay, key, A, isbc, singular, par):
    '''(coarse, mg): the smoothed-aggregation multigrid hierarchy (default), or
    the two-level aggregate coarse space -- which is what the row-sharded loop
    uses, and the fallback for systems too small to coarsen -- or (None, None)
    for plain Jacobi ('two_level': False).  Built once per (operator, BC set).'''
    if not par.get('two_level', False):
        return None, None
    if par.get('multigrid', True):
        mkey = ('mg', key, par.get('mg_coarsest', 4200),
                par.get('aggregation', 'geometric'))
        if mkey not in lay._dev:
            from ..fem.multigrid import Multigrid
            lay._dev[mkey] = Multigrid(
                A, isbc, singular=singular,
                coarsest=par.get('mg_coarsest', 4200),
                aggregation=par.get('aggregation', 'geometric'))
        if lay._dev[mkey].nlevels >= 2:
            return None, lay._dev[mkey]
    ckey = ('coarse', key, par['coarse_size'])
    if ckey not in lay._dev:
        lay._dev[ckey] = ops.CoarseSpace(
            A, isbc, singular=singular, target_nc=par['coarse_size']
            )
    return lay._dev[ckey], None


def _compute_pressure(
        p0,
        alpha, rho, dt, mu,
        ui,
        p_bcs=None,
        rotational_form=False,
        tol=1.0e-10,
        verbose=True
        ):
    '''Solve the pressure Poisson equation (reference :258-433)

        (grad p1, grad q) = -alpha rho/dt (div ui, q) + (grad p0, grad q)
                            [- mu (grad div ui, grad q)].
    '''
    lib = _hip.lib()
    P = p0.function_space()
    W = ui.function_space()
    mesh = P.mesh()
    lay = P.layout
    nc = mesh.num_cells()
    st = _hip.stream()

    par = solver_parameters['pressure']
    p1 = Function(P, device.empty(P.N))
    start_mode = None
    hist = _history(W.layout) if par.get('extrapolate', False) else None
    # The reference starts its Krylov solve from a fresh (zero) Function
    # (:313).  With Dirichlet conditions the solution is unique and p0 is the
    # natural start of an incremental scheme (same stopping test, a state of
    # rest stays exactly at rest).  Without them the constant the singular
    # system leaves open is the start's: p0 minus its (Euclidean) mean -- sum
    # zero like the reference's zero start, so that no constant accumulates
    # from step to step, and still exact on a state of rest.
    ops.copy(p1.data, p0.data)
    phi_start = None
    plain = None      # the start without the extrapolated increment
    if hist is None and par.get('start') == 'extrapolated':
        # ... plus the pressure increments p1 - p0 of the trajectory's earlier
        # time levels extrapolated in time (start_vectors: a start vector only,
        # guarded by the solver; on the strips every rank extrapolates its own
        # + ghost rows)
        phi_start = lay._dev.get('pressure_phi_scratch')
        if phi_start is None:
            phi_start = lay._dev['pressure_phi_scratch'] = device.empty(P.N)
        if _extrapolated_increment(W.layout, dt, phi_start,
                                   int(par.get('start_points', 5)),
                                   key='pressure_increments', power=1,
                                   degree=par.get('start_degree')):
            plain = _hip.clone(p0.data)
            ops.axpby(1.0, phi_start, 1.0, p1.data)
    if not p_bcs:
        one = lay._dev.get('ones')
        if one is None:
            one = _zeros(P.N)
            ops.fill(one, 1.0)
            lay._dev['ones'] = one
        for vec in (p1.data, plain):
            if vec is None:
                continue
            total = parallel.dot(one, vec, lay) if parallel.active() \
                else ops.dot(one, vec)
            ops.axpby(-total / P.N, one, 1.0, vec)
    if hist is not None and 'p_in' in hist and 0.7 <= dt / hist['dt'] <= 1.5:
        # ... extrapolated through the previous pressures when this call
        # continues the previous step's trajectory at a settled step size
        a, c = hist['dt'], dt
        b = hist.get('dt_prev')
        start_mode = 1
        if 'p_in2' in hist and b and 0.7 <= a / b <= 1.5:
            # (quadratic while the flow evolves, linear once successive
            # pressures differ by solver noise only: ops.StartChooser)
            start_mode = hist.setdefault(
                'p_start', ops.StartChooser()).pick()
        if start_mode == 2:
            # quadratic through the last three pressures (Lagrange
            # weights for the times -(a+b), -a, 0 evaluated at c)
            w0 = (c + a) * (c + a + b) / (a * (a + b))
            w1 = -c * (c + a + b) / (a * b)
            w2 = c * (c + a) / ((a + b) * b)
            ops.axpby(w0 - 1.0, p0.data, 1.0, p1.data)
            ops.axpby(w1, hist['p_in'], 1.0, p1.data)
            ops.axpby(w2, hist['p_in2'], 1.0, p1.data)
        else:
            r = c / a
            ops.axpby(r, p0.data, 1.0, p1.data)
            ops.axpby(-r, hist['p_in'], 1.0, p1.data)
    K = ops.assemble_stiffness(P)
    # (the gather writes every row it owns: all of them on one GPU)
    b = _zeros(P.N) if parallel.active() else device.empty(P.N)
    buf = ops.scratch(mesh, 3 * nc)
    _hip.check(lib.flow_assemble_pressure_rhs(
        ctypes.byref(_mesh_s(mesh)), ctypes.byref(_space_s(W.layout)),
        ctypes.byref(_space_s(lay)), _hip.f64(ui.data, W.size()),
        _hip.f64(p0.data, P.N), alpha * rho / dt, mu, int(rotational_form),
        _hip.f64(buf), _hip.f64(b, P.N), st
        ))
    if p_bcs:
        # 'symmetric': True  =>  assemble_system-style elimination
        # (reference :325-339)
        dofs, bc_dofs, bc_vals = _bc_arrays(p_bcs, P.N)
        key = ('K_bc', dofs.tobytes())
        if key not in lay._dev:
            Kbc = ops.symmetric_bc_matrix(
                K, device.to_device(_bc_mask(dofs, P.N))
                )
            lay._dev[key] = (Kbc, Kbc.diag_inv())
        Kbc, dinv = lay._dev[key]
        xg = _zeros(P.N)
        nbc = bc_dofs.numel()
        _hip.check(lib.flow_bc_set_values(
            nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(xg), st
            ))
        tmp = device.empty(P.N)
        K.apply(xg, tmp)
        ops.axpby(-1.0, tmp, 1.0, b)
        _hip.check(lib.flow_bc_set_values(
            nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(b), st
            ))
        coarse = _preconditioner(lay, key, Kbc, _bc_mask(dofs, P.N) != 0, False,
                               par)
        # the initial guess satisfies the Dirichlet data
        _hip.check(lib.flow_bc_set_values(
            nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(p1.data), st
            ))
        if plain is not None:
            _hip.check(lib.flow_bc_set_values(
                nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(plain), st
                ))
        sol = _pressure_from(Kbc, dinv, coarse, b, p1.data, tol, par, plain,
                             W.layout)
    else:
        # pure Neumann problem: singular but consistent, no null-space
        # handling (reference :340-432); start: see above
        key = ('K_dinv',)
        if key not in lay._dev:
            lay._dev[key] = K.diag_inv()
        coarse = _preconditioner(lay, key, K, None, True, par)
        sol = _pressure_from(K, lay._dev[key], coarse, b, p1.data, tol, par,
                             plain, W.layout)
    if phi_start is not None:
        ops.copy(phi_start, p1.data)
        ops.axpby(-1.0, p0.data, 1.0, phi_start)
        _remember_increment(W.layout, dt, phi_start, key='pressure_increments')
    last_step_info['pressure_starts_dropped'] = getattr(sol, 'starts_dropped', 0)
    if verbose:
        info('pressure: %r' % sol)
    last_step_info['pressure'] = sol
    if start_mode is not None and 'p_start' in hist:
        hist['p_start'].report(start_mode, sol.iterations)
    return p1


def _compute_velocity_correction(
        ui, u, u_bcs, p1, p0, v, mu, rho, dt, rotational_form, tol, verbose
        ):
    '''Velocity correction  (u1, v) = (ui, v) - dt/rho (grad phi, v),
    phi = p1 - p0 [+ mu div ui]  (reference :436-465).'''
    lib = _hip.lib()
    W = u[0].function_space()
    P = p0.function_space()
    mesh = W.mesh()
    lay = W.layout
    nc = mesh.num_cells()
    n = W.N
    n2 = W.size()
    st = _hip.stream()

    par = solver_parameters['correction']
    # With the defect-correction solver the system is solved for the INCREMENT
    # u1 - ui: its right-hand side -dt/rho (grad phi, v) is the defect of the
    # start ui itself, assembled without the (ui, v) term -- one fp64 product
    # and (the increment being ~1e-5 of the field) one correction less
    # (not with an extrapolated start vector -- mode 'fast' --: the right-hand
    # side below is the defect of the start ui and of nothing else)
    increment = (par.get('method', 'chebyshev') == 'chebyshev'
                 and par.get('increment', True)
                 and not par.get('extrapolate', False))
    b = _zeros(n2) if parallel.active() else _persistent(lay, 'correction_rhs', n2)
    buf = ops.scratch(mesh, 2 * lay.nloc * nc)
    _hip.check(lib.flow_assemble_correction_rhs(
        ctypes.byref(_mesh_s(mesh)), ctypes.byref(_space_s(lay)),
        ctypes.byref(_space_s(P.layout)), _hip.f64(ui.data, n2),
        _hip.f64(p1.data, P.N), _hip.f64(p0.data, P.N), dt / rho, mu,
        int(rotational_form) | (2 if increment else 0), _hip.f64(buf),
        _hip.f64(b, n2), st
        ))
    # `solve(a == L, u1, bcs, 'symmetric': True)` eliminates the Dirichlet dofs
    # symmetrically (assemble_system).  Both velocity components share ONE mass
    # matrix, so the system is kept in its identity-row form instead
    # (flow_operator kind 4: (M u)_i = b_i on free rows, u_i = g_i on Dirichlet
    # rows): CG started from a vector that carries the boundary values only
    # ever sees directions that vanish on the Dirichlet dofs, where this
    # operator and the symmetrically eliminated one coincide -- same iterates,
    # same solution, no lifting of the boundary values into b, and the matrix
    # is streamed once per product for the two components.  (||b|| in the
    # stopping test is then the norm before lifting; the two differ by the
    # columns of M at the boundary.)
    M = ops.assemble_mass(W)
    dofs, bc_dofs, bc_vals = _bc_arrays(u_bcs, n2)
    key = ('M_rows', dofs.tobytes())
    if key not in lay._dev:
        free = numpy.ones(n2, dtype=numpy.uint8)
        free[dofs] = 0
        Mrows = ops.Matrix(lay, 4, M.vals, rowmask=device.to_device(free))
        lay._dev[key] = (Mrows, Mrows.diag_inv())
    Mbc, dinv = lay._dev[key]
    nbc = bc_dofs.numel()
    # the tentative velocity is the natural initial guess: u1 - ui = O(dt) ...
    # (increment form on one GPU: the start lives in a buffer of the layout --
    # the solver's loop only sees it and its own workspace, the same addresses
    # every call --, the result base + increment is written to a fresh u1)
    apart = increment and not parallel.active()
    if apart:
        start = ops.copy(_persistent(lay, 'correction_start', n2), ui.data)
        u1 = Function(W, device.empty(n2))
    else:
        u1 = Function(W, _hip.clone(ui.data))
        start = u1.data
    # ... plus, when this call continues the previous one's trajectory, that
    # step's correction u1 - ui scaled with the step sizes (the correction
    # -dt/rho M^-1 grad(phi) varies slowly once the flow has settled: 6 -> 2-3
    # CG iterations on the developed flow, 4 -> 2 at CFL-sized steps).  Not
    # while the step size is still being ramped up: the pressure of an
    # impulsively started flow scales like 1/dt, and the scaled old correction
    # is then a worse start than none (8 instead of 4-6 iterations).  Only a
    # start vector either way.
    hist = _history(lay) \
        if solver_parameters['correction'].get('extrapolate', False) else None
    if hist is not None and 'dt' in hist and 0.7 <= dt / hist['dt'] <= 1.5:
        r = dt / hist['dt']
        ops.axpby(r, hist['u_out'], 1.0, start)
        ops.axpby(-r, hist['ui'], 1.0, start)
    if nbc > 0:
        _hip.check(lib.flow_bc_set_values(
            nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(start), st
            ))
        if increment:
            # (the increment vanishes on the Dirichlet rows: the start carries
            # the boundary values)
            zeros = lay._dev.get(('bc_zeros', nbc))
            if zeros is None:
                zeros = lay._dev[('bc_zeros', nbc)] = _zeros(nbc)
            _hip.check(lib.flow_bc_set_values(
                nbc, _hip.i32(bc_dofs), _hip.f64(zeros), _hip.f64(b), st))
        else:
            _hip.check(lib.flow_bc_set_values(
                nbc, _hip.i32(bc_dofs), _hip.f64(bc_vals), _hip.f64(b), st))
    if parallel.active() and par.get('method', 'chebyshev') != 'chebyshev':
        sol = parallel.cg(Mbc, dinv, b, u1.data, tol, 0.0, par['maxit'],
                          check_every=par['check_every'], tag='correction')
    elif par.get('method', 'chebyshev') == 'chebyshev':
        from ..fem.mass import MassSolver
        solver = MassSolver.cached(Mbc, dinv,
                                   steps=par.get('chebyshev_steps', 6))
        if increment:
            # start of the increment: the previous calls' increments,
            # extrapolated in time (u1 - ui = -dt/rho M^-1 grad(phi) ~ dt^2
            # p_t: smooth from step to step) -- a start vector only
            d0 = None
            if par.get('increment_start') == 'extrapolated':
                d0 = lay._dev.get('correction_d0_scratch')
                if d0 is None:
                    d0 = lay._dev['correction_d0_scratch'] = device.empty(n2)
                if not _extrapolated_increment(
                        lay, dt, d0, int(par.get('start_points', 5)),
                        key='correction_increments', power=2,
                        degree=par.get('start_degree')):
                    d0 = None
                elif nbc > 0:
                    _hip.check(lib.flow_bc_set_values(
                        nbc, _hip.i32(bc_dofs),
                        _hip.f64(lay._dev[('bc_zeros', nbc)]), _hip.f64(d0),
                        st))
            ui_keep = ui.data
            if parallel.active():
                # (on the strips: one collective per correction, the deep halo
                # of the defect; u1 comes back valid on own + ghost rows)
                sol = parallel.mass_solve(
                    solver, b, u1.data, tol, maxit=min(par['maxit'], 100),
                    tag='correction', xbase=u1.data, delta0=d0)
            else:
                sol = solver.solve_increment(
                    b, start, u1.data, tol, maxit=min(par['maxit'], 100),
                    tag='correction', delta0=d0)
            if par.get('increment_start') == 'extrapolated':
                d0 = lay._dev['correction_d0_scratch']
                ops.copy(d0, u1.data)
                ops.axpby(-1.0, ui_keep, 1.0, d0)
                _remember_increment(lay, dt, d0, key='correction_increments')
        elif parallel.active():
            sol = parallel.mass_solve(solver, b, u1.data, tol,
                                      maxit=min(par['maxit'], 100),
                                      tag='correction')
        else:
            sol = solver.solve(b, u1.data, tol, maxit=min(par['maxit'], 100),
                               tag='correction')
    else:
        sol = ops.krylov_solve(
            'cg', Mbc, b, u1.data, rtol=tol, atol=0.0, maxit=par['maxit'],
            dinv=dinv, check_every=par['check_every'], tag='correction'
            )
    if verbose:
        info('velocity correction: %r' % sol)
    last_step_info['correction'] = sol
    return u1


def _step(
        dt,
        u, p0,
        u_bcs, p_bcs,
        rho, mu,
        time_step_method,
        f,
        rotational_form=False,
        verbose=True,
        tol=1.0e-10,
        p_identity=None,
        ):
    '''Incremental pressure correction scheme as described in section 3.4 of
    Guermond, Minev, Shen (2006); reference :468-518.  p_identity: the pressure
    the caller handed in when p0 is not it (Chorin drops it, :545): what
    identifies the trajectory this call continues (start_vectors).'''
    # dt, mu are Constant()s; rho may be a Constant or a plain float
    dt_ = scalar_value(dt)
    mu_ = scalar_value(mu)
    rho_ = scalar_value(rho)
    assert dt_ > 0.0
    assert mu_ > 0.0
    device.check_stream()

    lay = u[0].function_space().layout
    hist = lay._dev.get('step_history') if _uses_history() else None
    if hist is not None and 'u_out' in hist:
        # does this call continue the trajectory of the previous one?
        tmp = _hip.clone(u[0].data)
        ops.axpby(-1.0, hist['u_out'], 1.0, tmp)
        hist['continuing'] = ops.vector_norm(tmp, 'linf') == 0.0
        del tmp

    # the start vectors of the linear solves belong to the trajectory whose
    # last step returned the fields this call is handed (start_vectors)
    tracked = _uses_start_vectors()
    if tracked:
        start_vectors.begin_step(
            lay, u[0].data, (p_identity if p_identity is not None else p0).data)
    else:
        lay._dev.pop('start_vector_state', None)

    t_0 = time.perf_counter()
    with Message('Computing tentative velocity'):
        ui, alpha = _compute_tentative_velocity(
                u, p0, f, u_bcs, time_step_method, rho_, mu_, dt_, None,
                tol=1.0e-10
                )

    t_1 = time.perf_counter()
    with Message('Computing pressure'):
        p1 = _compute_pressure(
                p0,
                alpha, rho_, dt_, mu_,
                ui,
                p_bcs=p_bcs,
                rotational_form=rotational_form,
                tol=tol,
                verbose=verbose
                )

    t_2 = time.perf_counter()
    with Message('Computing velocity correction'):
        u1 = _compute_velocity_correction(
            ui, u, u_bcs, p1, p0, None, mu_, rho_, dt_, rotational_form, tol,
            verbose
            )
    t_3 = time.perf_counter()
    # every sub-step ends with a host read-back of a residual norm, so the host
    # clock brackets the device work
    last_step_info['timings'] = {
        'tentative_s': t_1 - t_0, 'pressure_s': t_2 - t_1,
        'correction_s': t_3 - t_2,
        }
    last_step_info['tentative_velocity'] = ui
    if tracked:
        start_vectors.end_step(lay, u1.data, p1.data)
    if _uses_history():
        hist = lay._dev.setdefault('step_history', {})
        if 'ui' not in hist:
            hist['ui'] = _hip.clone(ui.data)
            hist['u_out'] = _hip.clone(u1.data)
        else:
            if 'ui_prev' not in hist:
                hist['ui_prev'] = _hip.clone(hist['ui'])
            else:
                ops.copy(hist['ui_prev'], hist['ui'])
            ops.copy(hist['ui'], ui.data)
            ops.copy(hist['u_out'], u1.data)
        if 'p_in' not in hist or hist['p_in'].numel() != p0.data.numel():
            hist['p_in'] = _hip.clone(p0.data)
            hist.pop('p_in2', None)
        else:
            if 'p_in2' not in hist:
                hist['p_in2'] = _hip.clone(hist['p_in'])
            else:
                ops.copy(hist['p_in2'], hist['p_in'])
            ops.copy(hist['p_in'], p0.data)
        hist['dt_prev'] = hist.get('dt')
        hist['dt'] = dt_
    return u1, p1


class _PressureCorrection(object):
    '''The three schemes differ in flags only (reference :521-617): Chorin
    drops the old pressure and is backward Euler; IPCS keeps it and takes the
    time discretisation of the momentum equation as an argument; Rotational
    adds the rotational form of the pressure update.'''
    drop_pressure = False
    rotational_form = False

    def __init__(self, time_step_method='backward euler'):
        self.time_step_method = time_step_method

    def step(self, dt, u, p0, u_bcs, p_bcs, rho, mu, f, verbose=True,
             tol=1.0e-10):
        p_in = None
        if self.drop_pressure:
            p_in, p0 = p0, Function(p0.function_space())
        return _step(
            dt, u, p0, u_bcs, p_bcs, rho, mu, self.time_step_method, f,
            rotational_form=self.rotational_form, verbose=verbose, tol=tol,
            p_identity=p_in
            )


class Chorin(_PressureCorrection):
    order = {'velocity': 1.0, 'pressure': 0.5}
    drop_pressure = True

    def __init__(self):
        _PressureCorrection.__init__(self, 'backward euler')


class IPCS(_PressureCorrection):
    order = {'velocity': 2.0, 'pressure': 1.0}


class Rotational(_PressureCorrection):
    order = {'velocity': 2.0, 'pressure': 1.5}
    rotational_form = True
