# Development: where the velocity maximum of a long Karman run sits (is a late
# growth of |u|_inf physical -- the wake -- or a boundary artefact?).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1091
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
prob = karman.KarmanProblem(nx, max(2, int(round(nx * 509.0 / 2182.0))), velocity_degree=2)
if os.environ.get('START', 'stokes') == 'stokes':
    prob.set_initial_stokes()
else:
    prob.set_initial_profile()
prob.dt = 1e-5
if os.environ.get('DTMAX'):
    prob.dt_max = float(os.environ['DTMAX'])
if os.environ.get('MODE'):
    import flow_amd.navier_stokes as navsto
    navsto.set_mode(os.environ['MODE'])
xy = prob.W.layout.dof_coords
n = prob.W.layout.N
for k in range(nsteps):
    info = prob.step()
    every = int(os.environ.get('EVERY', 25))
    if k % every == every - 1 or k == nsteps - 1:
        u = prob.u0.vector().get_local()
        mag = numpy.hypot(u[:n], u[n:])
        i = int(mag.argmax())
        print('step %4d t %.3f dt %.3e |u|max %.4f at (%.3f, %.3f) u=(%.4f, %.4f) newton %d apps %d  |uy|max %.4f'
              % (k + 1, prob.t, info['dt'], mag[i], xy[i, 0], xy[i, 1], u[i], u[n + i],
                 len(info['newton_linear_iterations']), sum(info['newton_linear_applications']), abs(u[n:]).max()), flush=True)
