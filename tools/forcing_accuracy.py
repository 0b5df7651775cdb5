# Development: how far do the fields move when the linear solves of the Newton
# iteration are stopped earlier?  Runs the headline workload with several
# `linear_atol_factor`s and compares u, p after N steps with a tightly
# converged run (relative l2 over the dof vectors).
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman, device
import flow_amd.navier_stokes as navsto

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 2182
ref = None
for factor in (0.001, 0.05, 0.2, 0.5):
    navsto.solver_parameters['newton']['linear_atol_factor'] = factor
    prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
    prob.set_initial_profile()
    tot = 0
    device.synchronize()
    t0 = time.time()
    for k in range(nsteps):
        info = prob.step()
        tot += sum(info['newton_linear_applications'])
    device.synchronize()
    wall = time.time() - t0
    u = prob.u0.vector().get_local()
    p = prob.p0.vector().get_local()
    if ref is None:
        ref = (u, p, prob.t)
        print('factor %.3f: reference run, BiCGStab its %d, t = %.6e, %.1f ms/step' % (
            factor, tot, prob.t, 1e3 * wall / nsteps), flush=True)
        continue
    print('factor %.3f: BiCGStab its %d, %.1f ms/step, t - t_ref = %.1e, '
          'rel l2 diff u %.2e  p %.2e  (last Newton residual %.1e)' % (
              factor, tot, 1e3 * wall / nsteps, prob.t - ref[2],
              numpy.linalg.norm(u - ref[0]) / numpy.linalg.norm(ref[0]),
              numpy.linalg.norm(p - ref[1]) / numpy.linalg.norm(ref[1]),
              info['newton_residuals'][-1]), flush=True)
    # release the step history etc. of this problem before the next one
    del prob
