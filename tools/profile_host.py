# Development: host-side (Python) profile of a few time steps.
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import karman
prob = karman.KarmanProblem(2182, 509, velocity_degree=2)
prob.set_initial_profile(); prob.dt = 1e-5
for _ in range(12):
    prob.step(tol=1e-10)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    prob.step(tol=1e-10)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(40)
