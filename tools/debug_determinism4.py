# Development: repeated runs on ONE problem object (same device addresses).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, device, _hip
import flow_amd.navier_stokes as navsto
SIZE = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1196, 279, 1)
keep_ilu = len(sys.argv) > 2 and sys.argv[2] == 'keep_ilu'
prob = karman.KarmanProblem(SIZE[0], SIZE[1], velocity_degree=SIZE[2])
for trial in range(6):
    prob.set_initial_profile(); prob.dt = 1e-5; prob.t = 0.0
    _hip.fill(prob.p0.data, 0.0)
    lay = prob.W.layout
    if not keep_ilu:
        lay._dev.pop('jacobian_ilu', None)
    lay._dev.pop('newton_quad_C', None)
    if hasattr(prob, '_umag'):
        del prob._umag
    line = []
    for k in range(3):
        info = prob.step(tol=1e-10)
        line.append((info['newton_linear_iterations'], ['%.17e' % r for r in info['newton_residuals']][-1],
                     info['pressure'].iterations, '%.17e' % info['unorm']))
    print(trial, line, flush=True)
