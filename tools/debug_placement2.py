# Development: which operand's PLACEMENT changes the BiCGStab result?
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, fem, device, _hip
from flow_amd.fem import ops

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 700
mesh = fem.UnitSquareMesh(n_side, n_side)
V = fem.VectorFunctionSpace(mesh, 'Lagrange', 1)
lay = V.layout
n2 = V.size()
M = ops.assemble_mass(V); K = ops.assemble_stiffness(V)
A0 = ops.Matrix(lay, 1)
for p in (0, 1):
    ops.copy(A0.plane(p), M.vals[:lay.nnz]); ops.axpby(2e-5, K.vals[:lay.nnz], 1.0, A0.plane(p))
g = torch.Generator().manual_seed(1)
b0 = (torch.rand(n2, generator=g, dtype=torch.float64) - 0.5).to(device.get())
d0 = A0.diag_inv()
h = lambda t: hash(device.to_host(t).numpy().tobytes()) % 1000000
rnd = random.Random(3)
keep = []


def fresh(t):
    keep.append(torch.empty(rnd.randrange(1, 300000) * 8, dtype=torch.uint8, device=device.get()))
    c = _hip.clone(t)
    keep.append(c)
    return c


x_fixed = device.zeros(n2)
for what in ('nothing', 'x', 'b', 'dinv', 'A', 'work'):
    outs = []
    for trial in range(8):
        A, b, d = A0, b0, d0
        x = x_fixed
        _hip.fill(x, 0.0)
        if what == 'x':
            x = fresh(x_fixed)
        if what == 'b':
            b = fresh(b0)
        if what == 'dinv':
            d = fresh(d0)
        if what == 'A':
            A = ops.Matrix(lay, 1, fresh(A0.vals))
        if what == 'work':
            ops._WORK.clear()
            keep.append(torch.empty(rnd.randrange(1, 300000) * 8, dtype=torch.uint8, device=device.get()))
        s = ops.krylov_solve('bicgstab', A, b, x, rtol=1e-9, maxit=5000, dinv=d, check_every=2)
        outs.append((s.iterations, h(x)))
    print('moving %-8s -> %s' % (what, 'same' if all(o == outs[0] for o in outs) else 'DIFFERENT %r' % ([o[0] for o in outs],)), flush=True)
