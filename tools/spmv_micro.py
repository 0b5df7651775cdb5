# Development: replay timings of the three CSR-stream SpMV flavours on the
# headline workload's matrices (pressure K, P2 mass scalar, P2 mass pair).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import fem, device
from flow_amd.fem import ops
sys.path.insert(0, ROOT)
import bench
mesh = fem.karman_channel(2182, 509)
W = fem.VectorFunctionSpace(mesh, 'CG', 2)
P = fem.FunctionSpace(mesh, 'CG', 1)
K = ops.assemble_stiffness(P)
M = ops.assemble_mass(W.collapse())
free = numpy.ones(2 * W.N, dtype=numpy.uint8)
Mp = ops.Matrix(W.layout, 4, M.vals, rowmask=device.to_device(free))
for name, A in (('pressure K', K), ('P2 mass', M), ('P2 mass pair', Mp)):
    n = A.size
    t = bench.measure_spmv_replay(A.apply, n, reps=50)
    lay = A.layout
    B = bench.spmv_bytes(lay.N, lay.nnz) + (16 * lay.N if A.kind == 4 else 0)
    print('%-14s %8.1f us  %6.0f GB/s' % (name, t * 1e6, B / t / 1e9), flush=True)
