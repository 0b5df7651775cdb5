# Development: how many Krylov iterations the P2 mass system needs with the
# multicolour ILU(0) as preconditioner instead of Jacobi (probe for the velocity
# correction solve).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import fem, device
from flow_amd.fem import ops, ilu
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
mesh = fem.karman_channel(nx, max(2, int(round(nx * 509.0 / 2182.0))))
V = fem.FunctionSpace(mesh, 'CG', 2)
M = ops.assemble_mass(V)
n = V.layout.N
rng = numpy.random.RandomState(0)
xs = numpy.sin(3 * V.layout.dof_coords[:, 0]) * numpy.cos(5 * V.layout.dof_coords[:, 1]) + 0.01 * rng.standard_normal(n)
b = device.zeros(n)
M.apply(device.to_device(xs), b)
for packed in (False, True):
    pre = ilu.Ilu0(M, packed=packed)
    x = device.zeros(n)
    info = ops.krylov_solve('gmres', M, b, x, rtol=1e-10, maxit=200, ilu=pre, restart=30, x_is_zero=True)
    print('gmres + ilu0 packed=%s:' % packed, info, flush=True)
x = device.zeros(n)
info = ops.krylov_solve('cg', M, b, x, rtol=1e-10, maxit=500, dinv=M.diag_inv(), check_every=1)
print('cg + jacobi:', info)
x = device.zeros(n)
info = ops.krylov_solve('gmres', M, b, x, rtol=1e-10, maxit=500, dinv='jacobi', restart=30, x_is_zero=True)
print('gmres + jacobi:', info)
