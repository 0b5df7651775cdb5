# Development helper: where does F(u - dx) differ from F(u) - J dx ?
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, device, _hip, fem
from flow_amd.fem import ops
from flow_amd.fem.bcs import collect
from flow_amd.fem.function import as_cell_coefficient
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
for k in range(9):
    prob.step()
lib = _hip.lib()
W, P, mesh = prob.W, prob.P, prob.mesh
lay = W.layout; nc = mesh.num_cells(); n2 = W.size(); n = lay.N
f = as_cell_coefficient(fem.Constant((0.0, 0.0)), mesh, 2)
fs, keep = ops.coef_struct(f, mesh, 2)
prm = _hip.NsParams(prob.dt, prob.rho, prob.mu, 1.0, 0.0)
bfmask = device.to_device(mesh.cell_bfacet_mask())
buf = ops.scratch(mesh, 4 * lay.nloc**2 * nc)
def assemble(u, F=None, Jm=None):
    _hip.check(lib.flow_assemble_momentum(
        ctypes.byref(ops.mesh_struct(mesh)), ctypes.byref(ops.space_struct(lay)),
        ctypes.byref(ops.space_struct(P.layout)), _hip.i32(bfmask), _hip.f64(u),
        _hip.f64(prob.u0.data), _hip.f64(prob.p0.data), ctypes.byref(fs), ctypes.byref(fs),
        ctypes.byref(prm), _hip.f64(buf), _hip.f64(F) if F is not None else None,
        _hip.f64(Jm.vals) if Jm is not None else None, Jm.stride if Jm is not None else 0, _hip.stream()))
dofs, vals = collect(prob.u_bcs, n2)
bd = device.to_device(dofs); bv = device.to_device(vals)
ui = prob.u0.data.clone()
F = device.empty(n2); J = ops.Matrix(lay, 2)
assemble(ui, F=F)
_hip.check(lib.flow_bc_residual(len(dofs), _hip.i32(bd), _hip.f64(bv), _hip.f64(ui), _hip.f64(F), _hip.stream()))
print('F0', float(F.norm()), 'bc part', float(F[bd.long()].norm()))
assemble(ui, Jm=J)
Jfull = ops.Matrix(lay, 2, J.vals.clone())
_hip.check(lib.flow_bc_identity_rows(ctypes.byref(J.operator()), _hip.f64(J.vals), _hip.i32(lay.dev('diag_idx')), len(dofs), _hip.i32(bd), _hip.stream()))
dx = device.zeros(n2)
info = ops.krylov_solve('bicgstab', J, F, dx, rtol=1e-13, atol=5e-12, maxit=5000, check_every=5)
print(info)
tmp = device.empty(n2); J.apply(dx, tmp)
print('linear residual', float((F - tmp).norm()))
u1 = ui - dx
F1 = device.empty(n2)
assemble(u1, F=F1)
_hip.check(lib.flow_bc_residual(len(dofs), _hip.i32(bd), _hip.f64(bv), _hip.f64(u1), _hip.f64(F1), _hip.stream()))
print('F1', float(F1.norm()))
d = F1.cpu().numpy()
idx = numpy.argsort(-abs(d))[:10]
xy = numpy.concatenate([lay.dof_coords, lay.dof_coords])
isbc = numpy.zeros(n2, bool); isbc[dofs] = True
for i in idx:
    print(i, i // n, '%.3e' % d[i], xy[i], 'bc' if isbc[i] else '')
print('dx max', float(dx.abs().max()), 'dx at bc max', float(dx[bd.long()].abs().max()))
