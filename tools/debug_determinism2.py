# Development: per-step hashes of the fields over repeated runs in one process.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, device
import flow_amd.navier_stokes as navsto
SIZE = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1196, 279, 1)
h = lambda t: hash(device.to_host(t).numpy().tobytes()) % 100000
for trial in range(5):
    prob = karman.KarmanProblem(SIZE[0], SIZE[1], velocity_degree=SIZE[2])
    prob.set_initial_profile(); prob.dt = 1e-5
    line = []
    for k in range(3):
        info = prob.step(tol=1e-10)
        ui = navsto.last_step_info['tentative_velocity']
        line.append((h(ui.data), h(prob.p0.data), h(prob.u0.data), '%.17g' % prob.dt,
                     info['newton_linear_iterations'], ['%.6e' % r for r in info['newton_residuals']]))
    print(trial, line, flush=True)
