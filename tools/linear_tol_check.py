# -*- coding: utf-8 -*-
'''What does the tolerance of the Newton systems' linear solves buy?  From the
settled state: N steps with linear_atol_factor 1e-6 (the default of mode
'parity'), then the same N steps with looser factors -- rel-L2 distance of u
and p to the first run, GMRES applications per step, ms per step.
  python tools/linear_tol_check.py [nx] [steps] [mu]'''
from __future__ import print_function
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 772
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.00565
    import torch
    from flow_amd import karman
    import flow_amd.navier_stokes as navsto
    ny = max(2, int(round(nx * 509.0 / 2182.0)))
    prob = karman.KarmanProblem(nx, ny, mu=mu)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    prob.settle()
    for _ in range(8):           # (fill the start vectors' history)
        prob.step()
    snap = prob.snapshot()
    from flow_amd.navier_stokes import start_vectors as sv
    lay = prob.W.layout
    hist = sv.snapshot_state(lay)
    umag = list(prob._umag_hist)

    def run(factor):
        prob.restore(snap)
        # (with the histories of the snapshot: restore_state installs a copy)
        sv.restore_state(lay, hist)
        prob._umag_hist = list(umag)
        navsto.solver_parameters['newton']['linear_atol_factor'] = \
            factor if factor else 1.0e-6
        navsto.solver_parameters['newton']['linear_remainder_fraction'] = \
            0.0 if factor else 1.0e-3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infos = [prob.step() for _ in range(steps)]
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        apps = sum(sum(i['newton_linear_applications']) for i in infos) / float(steps)
        res = max(i['newton_residuals'][-1] for i in infos)
        return prob.u0.data.clone(), prob.p0.data.clone(), apps, ms, res

    u0, p0, apps, ms, res = run(1.0e-6)
    print('factor 1e-06: %.2f applications/step, %.3f ms/step, max |F| %.2e'
          % (apps, ms, res))
    # (0: 1e-3 of the predicted Newton remainder, 'linear_remainder_fraction')
    for f in (0, 1.0e-5, 1.0e-4, 1.0e-3, 1.0e-2):
        u, p, apps, ms, res = run(f)
        du = float((u - u0).norm() / u0.norm())
        dp = float((p - p0).norm() / p0.norm())
        print('factor %.0e: %.2f applications/step, %.3f ms/step, max |F| %.2e, '
              'u %.2e  p %.2e from the 1e-6 run after %d steps'
              % (f, apps, ms, res, du, dp, steps))


if __name__ == '__main__':
    main()
