# Development: how far is ONE time step (same state, same dt) from the step the
# reference's algorithm takes -- Newton from u0 with (almost) exact linear
# solves, stopped at ||F|| < 1e-10 (flow/navier_stokes/pressure_correction.py
# :204-254), pressure and correction converged far below their tolerance?
# Relative l2 over the dof vectors of u1 and p1, for several settings of the
# Newton linear tolerance / start vector, in two regimes: start-up (after 2
# steps) and CFL-sized steps (after 14).
#
#   python tools/parity_single_step.py [nx] [factor ...]
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman, device
import flow_amd.navier_stokes as navsto

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
factors = [float(a) for a in sys.argv[2:]] or [2e-2, 1e-3, 1e-4, 1e-5, 1e-6]
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
npar = navsto.solver_parameters['newton']
defaults = dict(npar)
lay = prob.W.layout


def fast():
    npar.update(defaults)
    npar.update(initial_guess='best', linear_atol_factor=0.02, forcing=1.0e-4,
                adaptive_forcing=True)
    navsto.solver_parameters['pressure']['extrapolate'] = True
    navsto.solver_parameters['correction']['extrapolate'] = True


def exact(factor):
    npar.update(defaults)
    npar.update(initial_guess='previous', linear_atol_factor=factor,
                forcing=0.0, adaptive_forcing=False)
    navsto.solver_parameters['pressure']['extrapolate'] = False
    navsto.solver_parameters['correction']['extrapolate'] = False


def run(tol):
    device.synchronize()
    t0 = time.perf_counter()
    info = prob.step(tol=tol, adapt=False)
    device.synchronize()
    wall = time.perf_counter() - t0
    return (prob.u0.vector().get_local().copy(),
            prob.p0.vector().get_local().copy(), info, wall)


def rel(a, b):
    return numpy.linalg.norm(a - b) / numpy.linalg.norm(b)


done = 0
for warm in (2, 14):
    fast()
    while done < warm:
        prob.step()
        done += 1
    u_s = prob.u0.vector().get_local().copy()
    p_s = prob.p0.vector().get_local().copy()
    dt, t = prob.dt, prob.t
    hist = lay._dev.get('step_history')
    hist_s = {k: (v.clone() if hasattr(v, 'clone') else v)
              for k, v in hist.items()}

    def restore():
        prob.u0.vector().set_local(u_s)
        prob.p0.vector().set_local(p_s)
        prob.dt, prob.t = dt, t
        for k, v in hist_s.items():
            if hasattr(v, 'clone'):
                hist[k].copy_(v)
            else:
                hist[k] = v

    exact(1.0e-9)
    restore()
    ur, pr, info, wall = run(1.0e-13)
    print('after %2d steps, dt %.2e: yardstick (factor 1e-9, tol 1e-13): '
          'applications %r, Newton residuals %r, pressure its %d, corr its %d'
          % (warm, dt, info['newton_linear_applications'],
             ['%.1e' % r for r in info['newton_residuals']],
             info['pressure'].iterations, info['correction'].iterations),
          flush=True)
    # how much of the difference is the pressure / correction tolerance?
    exact(1.0e-9)
    restore()
    u, p, info, wall = run(1.0e-10)
    print('   factor 1e-9, tol 1e-10            : du %.2e dp %.2e  (p its %d, c its %d)'
          % (rel(u, ur), rel(p, pr), info['pressure'].iterations,
             info['correction'].iterations), flush=True)
    for factor in factors:
        for tol in (1.0e-10,):
            exact(factor)
            restore()
            u, p, info, wall = run(tol)
            print('   previous, factor %.0e, tol %.0e: du %.2e dp %.2e  '
                  'applications %r residuals %r  %.1f ms'
                  % (factor, tol, rel(u, ur), rel(p, pr),
                     info['newton_linear_applications'],
                     ['%.1e' % r for r in info['newton_residuals']],
                     1e3 * wall), flush=True)
    fast()
    restore()
    u, p, info, wall = run(1.0e-10)
    print('   fast mode (best start, 0.02)      : du %.2e dp %.2e  '
          'applications %r residuals %r start %s  %.1f ms'
          % (rel(u, ur), rel(p, pr), info['newton_linear_applications'],
             ['%.1e' % r for r in info['newton_residuals']],
             info.get('initial_guess'), 1e3 * wall), flush=True)
    fast()
    restore()
