# Development: a longer Karman run (default 200 steps) -- iteration counts, dt and
# |u|_inf every 10 steps; checks that nothing drifts or blows up.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman, device
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prob = karman.KarmanProblem(2182, 509, velocity_degree=2)
prob.set_initial_profile(); prob.dt = 1e-5
t0 = time.time()
worst = 0
for k in range(nsteps):
    info = prob.step(tol=1e-10)
    its = sum(info['newton_linear_iterations'])
    worst = max(worst, its)
    if k % 10 == 9 or k == nsteps - 1:
        device.synchronize()
        print('step %4d  t %.4f  dt %.3e  |u|inf %.4f  newton %d  bicgstab %2d  cg(p) %3d  cg(corr) %d  wall %.1f s'
              % (k + 1, prob.t, info['dt'], info['unorm'], len(info['newton_linear_iterations']), its,
                 info['pressure'].iterations, info['correction'].iterations, time.time() - t0), flush=True)
assert numpy.isfinite(info['unorm'])
print('done: %d steps in %.1f s (%.1f steps/s incl. setup of the first steps), worst BiCGStab total per step %d'
      % (nsteps, time.time() - t0, nsteps / (time.time() - t0), worst))
