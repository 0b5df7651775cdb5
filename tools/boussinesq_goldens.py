# Development: the reference's Boussinesq test settings (target_time = 1.0,
# lcar = 0.1-like mesh, plain and SUPG) next to its golden norms
# (tests/test_boussinesq.py:84-97).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import fem, boussinesq
GOLD = {False: (3.959158183043053e-06, 40.225818326711604),
        True: (3.9591568082077104e-06, 40.225818361936234)}
for side_points in (1, 3):
    out = {}
    for supg in (False, True):
        mesh = fem.heater_box_coarse(side_points=side_points)
        u1, p1, th1, steps = boussinesq.compute_boussinesq(
            target_time=1.0, supg=supg, mesh=mesh)
        out[supg] = (fem.norm(u1, 'L2'), fem.norm(th1, 'L2'), len(steps))
        print('side points %d, cells %d, supg %s: |u| %.15e (golden %.15e, ratio %.4f)  |theta| %.15e (golden %.15e, rel diff %.2e)  steps %d'
              % (side_points, mesh.num_cells(), supg, out[supg][0], GOLD[supg][0], out[supg][0] / GOLD[supg][0],
                 out[supg][1], GOLD[supg][1], out[supg][1] / GOLD[supg][1] - 1.0, out[supg][2]), flush=True)
    print('   SUPG - plain: u rel %.3e (reference %.3e), theta rel %.3e (reference %.3e)'
          % (out[True][0] / out[False][0] - 1.0, GOLD[True][0] / GOLD[False][0] - 1.0,
             out[True][1] / out[False][1] - 1.0, GOLD[True][1] / GOLD[False][1] - 1.0), flush=True)
