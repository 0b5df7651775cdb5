# Development: wall time of every one of the first steps (one-off costs show).
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import karman, device
for extra in (True, False):
    prob = karman.KarmanProblem(2182, 509)
    prob.extrapolate_projection = extra
    prob.set_initial_profile()
    line = []
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        device.synchronize()
        t0 = time.perf_counter()
        info = prob.step()
        device.synchronize()
        t1 = time.perf_counter()
        sub = sum(info['timings'].values())
        line.append('%d: %.1f ms (sub-steps %.1f, projection its %d)' % (
            k, 1e3 * (t1 - t0), 1e3 * sub, info['projection_iterations']))
    print('extrapolate', extra, '\n  ' + '\n  '.join(line), flush=True)
    del prob
