# Development: would GMRES need fewer operator/preconditioner applications than
# BiCGStab on the Newton systems of the headline workload?  Counts them on the
# very systems the time loop solves (torch-level GMRES, counting only).
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flow_amd import karman
from flow_amd.fem import ops

real = ops.krylov_solve
log = []


def gmres_count(A, ilu, b, target, mmax=30):
    beta = float(b.norm())
    V = [b / beta]
    H = [[0.0] * mmax for _ in range(mmax + 1)]
    cs, sn, g = [], [], [beta]
    z = torch.empty_like(b)
    w = torch.empty_like(b)
    for j in range(mmax):
        ilu.solve(V[j], z)
        A.apply(z, w)
        for i in range(j + 1):
            H[i][j] = float(torch.dot(w, V[i]))
            w = w - H[i][j] * V[i]
        H[j + 1][j] = float(w.norm())
        V.append(w / H[j + 1][j])
        w = torch.empty_like(b)
        for i in range(j):
            t = cs[i] * H[i][j] + sn[i] * H[i + 1][j]
            H[i + 1][j] = -sn[i] * H[i][j] + cs[i] * H[i + 1][j]
            H[i][j] = t
        d = math.hypot(H[j][j], H[j + 1][j])
        cs.append(H[j][j] / d)
        sn.append(H[j + 1][j] / d)
        g.append(-sn[j] * g[j])
        g[j] = cs[j] * g[j]
        if abs(g[j + 1]) <= target:
            return j + 1, abs(g[j + 1])
    return mmax, abs(g[-1])


def patched(method, A, b, x, rtol, atol=0.0, **kw):
    if method == 'bicgstab' and kw.get('ilu') is not None:
        target = max(rtol * float(b.norm()), atol)
        ng, rg = gmres_count(A, kw['ilu'], b, target)
        sol = real(method, A, b, x, rtol, atol, **kw)
        log.append((ng, 2 * sol.iterations, float(b.norm()), target, rg,
                    sol.residual))
        return sol
    return real(method, A, b, x, rtol, atol, **kw)


ops.krylov_solve = patched
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
for k in range(nsteps):
    n0 = len(log)
    info = prob.step()
    for e in log[n0:]:
        print('step %2d dt %.2e  applications: gmres %2d  bicgstab %2d   |b| %.2e '
              'target %.2e  gmres res %.2e bicg res %.2e' % ((k, info['dt']) + e),
              flush=True)
print('total gmres %d bicgstab %d' % (sum(e[0] for e in log), sum(e[1] for e in log)))
