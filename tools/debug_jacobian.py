# Development helper: finite-difference check of the GPU Jacobian on the Karman
# problem (J v  vs  (F(u+eps v) - F(u-eps v)) / (2 eps)).
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, device, _hip, fem
from flow_amd.fem import ops
from flow_amd.fem.function import as_cell_coefficient
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
for k in range(9):
    prob.step()
lib = _hip.lib()
W, P, mesh = prob.W, prob.P, prob.mesh
lay = W.layout; nc = mesh.num_cells(); n2 = W.size()
f = as_cell_coefficient(fem.Constant((0.0, 0.0)), mesh, 2)
fs, keep = ops.coef_struct(f, mesh, 2)
dt = prob.dt
prm = _hip.NsParams(dt, prob.rho, prob.mu, 1.0, 0.0)
bfmask = device.to_device(mesh.cell_bfacet_mask())
buf = ops.scratch(mesh, 4 * lay.nloc**2 * nc)
J = ops.Matrix(lay, 2)
def assemble(u, F=None, Jm=None):
    _hip.check(lib.flow_assemble_momentum(
        ctypes.byref(ops.mesh_struct(mesh)), ctypes.byref(ops.space_struct(lay)),
        ctypes.byref(ops.space_struct(P.layout)), _hip.i32(bfmask), _hip.f64(u),
        _hip.f64(prob.u0.data), _hip.f64(prob.p0.data), ctypes.byref(fs), ctypes.byref(fs),
        ctypes.byref(prm), _hip.f64(buf), _hip.f64(F) if F is not None else None,
        _hip.f64(Jm.vals) if Jm is not None else None, Jm.stride if Jm is not None else 0, _hip.stream()))
u = prob.u0.data.clone()
torch.manual_seed(0)
v = torch.randn(n2, dtype=torch.float64, device=u.device) * 1e-3
assemble(u, Jm=J)
jv = device.empty(n2); J.apply(v, jv)
for eps in (1.0, 1e-2):
    Fp = device.empty(n2); Fm = device.empty(n2)
    assemble(u + eps * v, F=Fp); assemble(u - eps * v, F=Fm)
    fd = (Fp - Fm) / (2 * eps)
    print('eps', eps, 'rel diff', float((fd - jv).norm() / jv.norm()), 'norms', float(jv.norm()), float(fd.norm()))
# split by plane: which block is off?
n = lay.N
d = (fd - jv)
print('err x rows', float(d[:n].norm()), 'y rows', float(d[n:].norm()))
