# Development: BASELINE config 5 (lid-driven cavity, ~2 M DoF) -- where the
# time of flow_amd.stokes.solve goes: host profile + wall time of a second,
# warm solve.
import cProfile
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import fem, stokes, device, _hip                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 470
mesh = fem.UnitSquareMesh(n, n)
WP = fem.FunctionSpace(
    mesh, fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
    * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
W, P = WP.sub(0), WP.sub(1)


class Lid(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] > 1.0 - 1e-12)


class Walls(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] <= 1.0 - 1e-12)


bcs = [fem.DirichletBC(W, (0.0, 0.0), Walls()),
       fem.DirichletBC(W, (1.0, 0.0), Lid())]
print('dofs', W.size() + P.N, flush=True)
for rep in range(2):
    device.synchronize()
    n0 = _hip.launch_count()
    t = time.time()
    pr = cProfile.Profile()
    pr.enable()
    u, p = stokes.solve(WP, bcs, 1.0, fem.Constant((0.0, 0.0)), verbose=False,
                        tol=1e-8, max_iter=2000)
    device.synchronize()
    pr.disable()
    print('solve %d: %.2f s, %d launches, %r' % (
        rep, time.time() - t, _hip.launch_count() - n0, stokes.last_solve_info),
        flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
