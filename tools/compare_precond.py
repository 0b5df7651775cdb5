# Development helper: Jacobi vs ILU(0) for the Newton systems as dt grows.
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import karman
import flow_amd.navier_stokes as navsto
import torch
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1091
for pre in ('jacobi', 'ilu0'):
    navsto.solver_parameters['newton']['preconditioner'] = pre
    prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
    prob.set_initial_profile()
    t0 = time.time()
    for k in range(14):
        info = prob.step()
        if k in (0, 1, 6, 9, 11, 13):
            print(pre, k, 'dt %.1e' % info['dt'], 'newton', len(info['newton_residuals']) - 1,
                  'bicg', info['newton_linear_iterations'], 'tent %.1f ms' % (1e3 * info['timings']['tentative_s']))
    print(pre, 'total', time.time() - t0)
