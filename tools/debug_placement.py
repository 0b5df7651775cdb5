# Development: does the RESULT of an operation depend on where its operands live?
# Every trial clones the inputs to fresh addresses (with random dummy allocations
# in between) and hashes the outputs.
import os, sys, ctypes, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, fem, device, _hip
from flow_amd.fem import ops, ilu

SIZE = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1196, 279, 1)
prob = karman.KarmanProblem(SIZE[0], SIZE[1], velocity_degree=SIZE[2])
prob.set_initial_profile(); prob.dt = 1e-5
prob.step(tol=1e-10)
h = lambda t: hash(device.to_host(t).numpy().tobytes()) % 1000000
W, P, mesh = prob.W, prob.P, prob.mesh
lay = W.layout
n2 = W.size(); nc = mesh.num_cells()
lib = _hip.lib()
prm = _hip.NsParams(prob.dt, prob.rho, prob.mu, 1.0, 0.0)
bfm = device.to_device(mesh.cell_bfacet_mask())
f0s, keep = ops.coef_struct(fem.as_cell_coefficient(fem.Constant((0.0, 0.0)), mesh, 2), mesh, lay.degree)
g = torch.Generator().manual_seed(1)
v0 = (torch.rand(n2, generator=g, dtype=torch.float64) - 0.5).to(device.get())
M = ops.assemble_mass(W); K = ops.assemble_stiffness(W)
A0 = ops.Matrix(lay, 1)
for p in (0, 1):
    ops.copy(A0.plane(p), M.vals[:lay.nnz]); ops.axpby(1e-3, K.vals[:lay.nnz], 1.0, A0.plane(p))
rnd = random.Random(5)
junk = []
rows = []
for trial in range(8):
    junk.append(torch.empty(rnd.randrange(1, 400000) * 8, dtype=torch.uint8, device=device.get()))
    u0 = _hip.clone(prob.u0.data); p0 = _hip.clone(prob.p0.data); v = _hip.clone(v0)
    junk.append(torch.empty(rnd.randrange(1, 400000) * 8, dtype=torch.uint8, device=device.get()))
    buf = device.empty(max(2 * lay.nloc, 4 * lay.nloc**2) * nc)
    F = device.zeros(n2)
    _hip.check(lib.flow_assemble_momentum(
        ctypes.byref(ops.mesh_struct(mesh)), ctypes.byref(ops.space_struct(lay)),
        ctypes.byref(ops.space_struct(P.layout)), _hip.i32(bfm), _hip.f64(u0), _hip.f64(u0), _hip.f64(p0),
        ctypes.byref(f0s), ctypes.byref(f0s), ctypes.byref(prm), _hip.f64(buf), _hip.f64(F), None, 0, _hip.stream()))
    none = device.to_device(numpy.zeros(0, dtype=numpy.int32))
    Jop = ops.MomentumJacobian(W, bfm, u0, prm, none)
    Jv = device.zeros(n2); Jop.apply(v, Jv)
    A = ops.Matrix(lay, 1, _hip.clone(A0.vals))
    Av = device.zeros(n2); A.apply(v, Av)
    pre = ilu.Ilu0(A)
    z = device.zeros(n2); pre.solve(v, z)
    x = device.zeros(n2)
    s = ops.krylov_solve('bicgstab', A, v, x, rtol=1e-9, maxit=500, ilu=pre, check_every=2)
    xj = device.zeros(n2)
    sj = ops.krylov_solve('bicgstab', A, v, xj, rtol=1e-9, maxit=2000, check_every=2)
    xc = device.zeros(n2)
    sc = ops.krylov_solve('cg', A, v, xc, rtol=1e-9, maxit=2000, check_every=2)
    rows.append(dict(F=h(F), Jv=h(Jv), Av=h(Av), lu=h(pre.lu), z=h(z), bicg_ilu=(s.iterations, h(x)),
                     bicg_jac=(sj.iterations, h(xj)), cg=(sc.iterations, h(xc)),
                     dot=(ops.dot(v, Av), ops.vector_norm(Av))))
for k in rows[0]:
    vals = [r[k] for r in rows]
    print('%-10s %s' % (k, 'same' if all(x == vals[0] for x in vals) else 'PLACEMENT-DEPENDENT %r' % (vals,)), flush=True)
