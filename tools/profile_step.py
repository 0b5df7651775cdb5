# Development helper: host-side cProfile of Karman steps at the headline size.
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import karman
import torch
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
prob.step(); prob.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    prob.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue())
