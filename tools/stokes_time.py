# Development: time the Stokes bootstrap of the Karman problem.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import karman, stokes, device
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
mgflag = (sys.argv[2] == 'mg') if len(sys.argv) > 2 else False
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-13
stokes.solver_parameters['multigrid'] = mgflag
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
device.synchronize(); t0 = time.time()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
prob.set_initial_stokes(tol=tol, max_iter=60000)
pr.disable()
device.synchronize()
print('nx %d multigrid %s tol %.0e: %.1f s, %r' % (nx, mgflag, tol, time.time() - t0, prob.stokes_info), flush=True)
info = prob.step()
print('first step: newton', info['newton_residuals'], 'pressure', info['pressure'], flush=True)

pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
