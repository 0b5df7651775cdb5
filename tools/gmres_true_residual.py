# Development: is the residual GMRES reports (least-squares estimate) the true
# one?  Checks |b - A x| after every Newton linear solve of a short run.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flow_amd import karman
from flow_amd.fem import ops
import flow_amd.navier_stokes as navsto

real = ops.krylov_solve


def patched(method, A, b, x, rtol, atol=0.0, **kw):
    sol = real(method, A, b, x, rtol, atol, **kw)
    if method in ('gmres', 'bicgstab'):
        t = torch.empty_like(b)
        A.apply(x, t)
        true = float((b - t).norm())
        print('   %s: %d its, reported %.3e, true %.3e, |b| %.3e, target %.3e' % (
            method, sol.iterations, sol.residual, true, float(b.norm()),
            max(rtol * float(b.norm()), atol)), flush=True)
    return sol


ops.krylov_solve = patched
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
for solver in ('gmres', 'bicgstab'):
    navsto.solver_parameters['newton']['linear_solver'] = solver
    prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
    prob.set_initial_profile()
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
        info = prob.step()
        print(solver, 'step', k, ['%.2e' % r for r in info['newton_residuals']], flush=True)
