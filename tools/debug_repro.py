# Development: which operation of a step is not bitwise reproducible when it is
# repeated on the same inputs in one process?
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman, fem, device, _hip
from flow_amd.fem import ops, ilu
import flow_amd.navier_stokes as navsto
from flow_amd.navier_stokes import pressure_correction as pc

SIZE = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1196, 279, 1)
prob = karman.KarmanProblem(SIZE[0], SIZE[1], velocity_degree=SIZE[2])
prob.set_initial_profile(); prob.dt = 1e-5
prob.step(tol=1e-10)
h = lambda t: hash(device.to_host(t).numpy().tobytes())
W, P, mesh = prob.W, prob.P, prob.mesh
lay = W.layout
n2 = W.size(); nc = mesh.num_cells()
lib = _hip.lib()


def report(name, fn, reps=6):
    vals = [fn() for _ in range(reps)]
    print('%-34s %s' % (name, 'reproducible' if len(set(vals)) == 1 else 'DIFFERS: %d distinct of %d' % (len(set(vals)), reps)), flush=True)


# tentative velocity / pressure / correction sub-steps from the same state
def tentative():
    ui, _ = pc._compute_tentative_velocity(
        {0: prob.u0}, prob.p0, {0: fem.Constant((0.0, 0.0)), 1: fem.Constant((0.0, 0.0))},
        prob.u_bcs, 'backward euler', prob.rho, prob.mu, prob.dt, None, tol=1e-10)
    return (h(ui.data), tuple(navsto.last_step_info['newton_linear_iterations']))
report('tentative velocity (Newton)', tentative)
ui, _ = pc._compute_tentative_velocity(
    {0: prob.u0}, prob.p0, {0: fem.Constant((0.0, 0.0)), 1: fem.Constant((0.0, 0.0))},
    prob.u_bcs, 'backward euler', prob.rho, prob.mu, prob.dt, None, tol=1e-10)


def pressure():
    p1 = pc._compute_pressure(prob.p0, 1.0, prob.rho, prob.dt, prob.mu, ui, p_bcs=prob.p_bcs,
                              rotational_form=True, tol=1e-10, verbose=False)
    return h(p1.data)
report('pressure (two-level CG)', pressure)
p1 = pc._compute_pressure(prob.p0, 1.0, prob.rho, prob.dt, prob.mu, ui, p_bcs=prob.p_bcs,
                          rotational_form=True, tol=1e-10, verbose=False)


def correction():
    u1 = pc._compute_velocity_correction(ui, {0: prob.u0}, prob.u_bcs, p1, prob.p0, None, prob.mu,
                                         prob.rho, prob.dt, True, 1e-10, False)
    return h(u1.data)
report('velocity correction (CG)', correction)


def magnitude():
    m = fem.project_magnitude(prob.u0, tol=1e-9)
    return h(m.vector().data if hasattr(m.vector(), 'data') else m.data)
report('project |u| (CG)', magnitude)

# building blocks
prm = _hip.NsParams(prob.dt, prob.rho, prob.mu, 1.0, 0.0)
bfm = device.to_device(mesh.cell_bfacet_mask())
f0s, keep = ops.coef_struct(fem.as_cell_coefficient(fem.Constant((0.0, 0.0)), mesh, 2), mesh, lay.degree)
buf = ops.scratch(mesh, max(2 * lay.nloc, 4 * lay.nloc**2) * nc)


def residual():
    F = device.zeros(n2)
    _hip.check(lib.flow_assemble_momentum(
        ctypes.byref(ops.mesh_struct(mesh)), ctypes.byref(ops.space_struct(lay)),
        ctypes.byref(ops.space_struct(P.layout)), _hip.i32(bfm), _hip.f64(prob.u0.data),
        _hip.f64(prob.u0.data), _hip.f64(prob.p0.data), ctypes.byref(f0s), ctypes.byref(f0s),
        ctypes.byref(prm), _hip.f64(buf), _hip.f64(F), None, 0, _hip.stream()))
    return h(F)
report('momentum residual', residual)
g = torch.Generator().manual_seed(1)
v = (torch.rand(n2, generator=g, dtype=torch.float64) - 0.5).to(device.get())
none = device.to_device(numpy.zeros(0, dtype=numpy.int32))
Jop = ops.MomentumJacobian(W, bfm, prob.u0.data, prm, none)


def jvp():
    out = device.zeros(n2); Jop.apply(v, out); return h(out)
report('matrix-free J v', jvp)


def bicg_jacobi_matfree():
    x = device.zeros(n2)
    s = ops.krylov_solve('bicgstab', Jop, v, x, rtol=1e-8, maxit=500, dinv=None, check_every=2)
    return (h(x), s.iterations)
report('BiCGStab (no prec) on J', bicg_jacobi_matfree)
M = ops.assemble_mass(W)
A = ops.Matrix(lay, 1)
for p in (0, 1):
    ops.copy(A.plane(p), M.vals[:lay.nnz])


def bicg_mass():
    x = device.zeros(n2)
    s = ops.krylov_solve('bicgstab', A, v, x, rtol=1e-10, maxit=500, check_every=2)
    return (h(x), s.iterations)
report('BiCGStab + Jacobi on mass matrix', bicg_mass)


def cg_mass():
    x = device.zeros(n2)
    s = ops.krylov_solve('cg', A, v, x, rtol=1e-10, maxit=500, check_every=2)
    return (h(x), s.iterations)
report('CG + Jacobi on mass matrix', cg_mass)


def dots():
    return (ops.dot(v, v), ops.vector_norm(v), ops.vector_norm(v, 'linf'))
report('dot / norms', dots)
