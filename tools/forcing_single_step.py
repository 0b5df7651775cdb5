# Development: ONE time step at fixed dt from the same state, with several
# linear tolerances in the Newton iteration; fields against a tightly converged
# step (relative l2 over the dof vectors).  Regimes: start-up (after 2 steps)
# and CFL plateau (after 14 steps).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman, fem
import flow_amd.navier_stokes as navsto

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
npar = navsto.solver_parameters['newton']
done = 0
for warm in (2, 14):
    while done < warm:
        prob.step()
        done += 1
    u_s = prob.u0.vector().get_local().copy()
    p_s = prob.p0.vector().get_local().copy()
    dt, t = prob.dt, prob.t
    lay = prob.W.layout
    hist = lay._dev['step_history']
    hist_s = {k: (v.clone() if hasattr(v, 'clone') else v) for k, v in hist.items()}
    quad_s = lay._dev.get('newton_quad_C')

    def restore():
        prob.u0.vector().set_local(u_s)
        prob.p0.vector().set_local(p_s)
        prob.dt, prob.t = dt, t
        for k, v in hist_s.items():
            if hasattr(v, 'clone'):
                hist[k].copy_(v)
            else:
                hist[k] = v
        lay._dev['newton_quad_C'] = quad_s
    ref = None
    for factor in (1.0e-5, 0.05, 0.02, 0.01):
        npar['linear_atol_factor'] = factor
        # the first run is the yardstick: an (almost) exact Newton step
        npar['forcing'] = 1.0e-8 if ref is None else 1.0e-4
        npar['adaptive_forcing'] = ref is not None
        restore()
        info = prob.step(adapt=False)
        u = prob.u0.vector().get_local().copy()
        p = prob.p0.vector().get_local().copy()
        its = sum(info['newton_linear_applications'])
        if ref is None:
            ref = (u, p)
            print('after %2d steps, dt %.2e: tight step: applications %d, Newton residuals %r'
                  % (warm, dt, its, ['%.1e' % r for r in info['newton_residuals']]), flush=True)
            continue
        print('   factor %.2f: applications %d, final residual %.1e, rel l2 diff u %.2e  p %.2e'
              % (factor, its, info['newton_residuals'][-1],
                 numpy.linalg.norm(u - ref[0]) / numpy.linalg.norm(ref[0]),
                 numpy.linalg.norm(p - ref[1]) / numpy.linalg.norm(ref[1])), flush=True)
    npar['linear_atol_factor'] = 0.02
    restore()
