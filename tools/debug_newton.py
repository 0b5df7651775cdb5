# Development helper: true vs reported BiCGStab residuals in the Newton solve.
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import karman, device
from flow_amd.fem import ops
import flow_amd.navier_stokes.pressure_correction as pc
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 600
orig = ops.krylov_solve
def patched(method, A, b, x, *a, **k):
    info = orig(method, A, b, x, *a, **k)
    if method == 'bicgstab':
        tmp = device.empty(A.size)
        A.apply(x, tmp)
        true = ops.vector_norm(b - tmp)
        print('   bicgstab its %d reported %.2e true %.2e  |b| %.2e' % (info.iterations, info.residual, true, ops.vector_norm(b)))
    return info
ops.krylov_solve = patched
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
for k in range(14):
    info = prob.step()
    print(k, 'dt %.2e' % info['dt'], ['%.1e' % r for r in info['newton_residuals']], info['newton_linear_iterations'])
