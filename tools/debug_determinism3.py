# Development: like debug_determinism2 but WITHOUT read-backs between the steps;
# the Newton histories (host scalars the solver returns anyway) are kept.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, device
import flow_amd.navier_stokes as navsto
SIZE = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1196, 279, 1)
for trial in range(6):
    prob = karman.KarmanProblem(SIZE[0], SIZE[1], velocity_degree=SIZE[2])
    prob.set_initial_profile(); prob.dt = 1e-5
    line = []
    for k in range(3):
        info = prob.step(tol=1e-10)
        line.append((info['newton_linear_iterations'], ['%.17e' % r for r in info['newton_residuals']],
                     info['pressure'].iterations, '%.17e' % info['pressure'].residual, '%.17e' % info['unorm']))
    print(trial, line, flush=True)
