# Development harness: times one application of the multicolour ILU(0)
# preconditioner (both sweeps, both velocity blocks) and one refactorisation on a
# P2 block operator of the headline workload's kind.
#   python tools/ilu_tune.py [nx ny]
import os, sys, time
import numpy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import fem, device
from flow_amd.fem import ops, ilu
from flow_amd.fem.mesh import RectangleMesh

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 500)
mesh = RectangleMesh((0.0, 0.0), (2.0, 1.0), nx, ny)
V = fem.FunctionSpace(mesh, 'Lagrange', 2)
lay = V.layout
t0 = time.time()
M = ops.assemble_mass(V)
K = ops.assemble_stiffness(V)
# block-diagonal operator (kind 1) with two planes M + 0.01 K
A = ops.Matrix(lay, 1)
for p in (0, 1):
    A.plane(p).copy_(M.vals[:lay.nnz] + 0.01 * K.vals[:lay.nnz])
t1 = time.time()
P = ilu.Ilu0(A)
torch.cuda.synchronize()
t2 = time.time()
plan = P.plan
print('n=%d nnz=%d colours=%d  L entries %d (fill %.3f)  U entries %d (fill %.3f)'
      % (plan.n, plan.nnz, plan.ncolours, plan.nnz_l, plan.fill_l, plan.nnz_u,
         plan.fill_u))
print('assembly %.2f s, plan + first factorisation %.2f s' % (t1 - t0, t2 - t1))
n2 = 2 * plan.n
r = torch.sin(torch.arange(n2, dtype=torch.float64, device=device.get()))
z = torch.zeros_like(r)


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


t = timed(lambda: P.solve(r, z), 50)
B = 2 * (12.0 * (plan.nnz_l + plan.nnz_u) + 8.0 * 5 * plan.n)
print('apply: %.1f us  (%.1f us per colour launch)  %.0f GB/s' % (
    t * 1e6, t * 1e6 / (2 * plan.ncolours), B / t / 1e9))
t = timed(lambda: P.refactor(A), 5)
print('refactor: %.2f ms' % (t * 1e3))
