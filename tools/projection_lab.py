# -*- coding: utf-8 -*-
'''How much better than the time-extrapolated start would a projection of the
right-hand side onto the previous solutions be (Fischer 1998: successive
right-hand sides of ONE matrix)?  At every pressure solve of a Karman run the
residual of the start the solver is given is compared with the residual of
  x_p = X c,  (X^T K X) c = X^T b,  X = the last m solutions
(host-side, torch: a lab, not a product path).  SPIN=n steps first.
  python tools/projection_lab.py [steps] [m]'''
from __future__ import print_function
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                            # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    import torch
    from flow_amd import karman, device
    from flow_amd.navier_stokes import pressure_correction as pc
    nx = int(os.environ.get('NX', '2182'))
    prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
    prob.set_initial_stokes()
    prob.dt = 1e-5
    spin = int(os.environ.get('SPIN', '40'))
    for _ in range(spin):
        prob.step()
    hist = []
    rows = []
    orig = pc._pressure_cg

    def spy(A, dinv, coarse, b, x, tol, par):
        n = b.numel()
        tmp = device.empty(n)

        def res(v):
            A.apply(v, tmp)
            return float((b - tmp).norm() / b.norm())
        r_e = res(x)
        r_p = float('nan')
        if len(hist) >= 2:
            X = torch.stack(hist[-m:], dim=1)              # n x k
            KX = torch.empty_like(X)
            for j in range(X.shape[1]):
                A.apply(X[:, j].contiguous(), tmp)
                KX[:, j] = tmp
            G = X.T @ KX
            c = torch.linalg.solve(G, X.T @ b)
            xp = (X @ c).contiguous()
            r_p = res(xp)
        out = orig(A, dinv, coarse, b, x, tol, par)
        hist.append(x.clone())
        del hist[:-m]
        rows.append((r_e, r_p, out.iterations))
        return out
    pc._pressure_cg = spy
    for _ in range(steps):
        prob.step()
    rows = rows[m:]
    re = numpy.array([r[0] for r in rows])
    rp = numpy.array([r[1] for r in rows])
    print('after %d steps, %d solves, m = %d: relative residual of the start '
          '-- extrapolated: median %.2e (%.2e .. %.2e); projected: median '
          '%.2e (%.2e .. %.2e); CG iterations %.1f'
          % (spin, len(rows), m, numpy.median(re), re.min(), re.max(),
             numpy.median(rp), rp.min(), rp.max(),
             numpy.mean([r[2] for r in rows])))


if __name__ == '__main__':
    main()
