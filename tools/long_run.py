# Development: a longer Karman run (default 200 steps) -- iteration counts, dt and
# |u|_inf every 10 steps; checks that nothing drifts or blows up.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman, device
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:      # e.g. newton.linear_atol_factor=0.1
    import flow_amd.navier_stokes as navsto
    for kv in sys.argv[2:]:
        key, val = kv.split('=')
        grp, name = key.split('.')
        old = navsto.solver_parameters[grp][name]
        navsto.solver_parameters[grp][name] = val if isinstance(old, str) \
            else type(old)(float(val))
        print('set', grp, name, navsto.solver_parameters[grp][name])
NX = int(os.environ.get('NX', '2182'))
prob = karman.KarmanProblem(NX, int(round(NX * 509.0 / 2182.0)), velocity_degree=2)
if os.environ.get('START', 'stokes') == 'stokes':
    prob.set_initial_stokes()
else:
    prob.set_initial_profile()
prob.dt = 1e-5
t0 = time.time()
worst = 0
tot_newton = tot_lin = 0
t20 = None
for k in range(nsteps):
    if k == 20:
        device.synchronize()
        t20 = time.time()
    info = prob.step(tol=1e-10)
    its = sum(info['newton_linear_applications'])
    worst = max(worst, its)
    if k >= 20:
        tot_newton += len(info['newton_linear_iterations'])
        tot_lin += its
    if k % 10 == 9 or k == nsteps - 1:
        device.synchronize()
        print('step %4d  t %.4f  dt %.3e  |u|inf %.4f  newton %d  applications %2d (%s)  cg(p) %3d  cg(corr) %d  wall %.1f s  F %s  apps %s'
              % (k + 1, prob.t, info['dt'], info['unorm'], len(info['newton_linear_iterations']), its,
                 info.get('newton_preconditioner', '-'),
                 info['pressure'].iterations, info['correction'].iterations, time.time() - t0,
                 ' '.join('%.2e' % r for r in info['newton_residuals']),
                 '+'.join(str(a) for a in info['newton_linear_applications'])), flush=True)
assert numpy.isfinite(info['unorm'])
device.synchronize()
if t20 is not None:
    print('steps 20..%d: %.2f ms/step, Newton its %d, linear applications %d' % (
        nsteps, 1e3 * (time.time() - t20) / (nsteps - 20), tot_newton, tot_lin))
print('done: %d steps in %.1f s (%.1f steps/s incl. setup of the first steps), worst application total per step %d'
      % (nsteps, time.time() - t0, nsteps / (time.time() - t0), worst))
