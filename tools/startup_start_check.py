# -*- coding: utf-8 -*-
'''Start-up transient (impulsive start, dt doubling from 1e-5): do the start
vectors extrapolated in time change the fields?  Single GPU, zero-start vs
default, relative l2 distances per step.
  python tools/startup_start_check.py [nx] [steps]'''
from __future__ import print_function
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                            # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1091
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from flow_amd import karman, device
    import flow_amd.navier_stokes as navsto
    ny = max(2, int(round(nx * 509.0 / 2182.0)))
    runs = {}
    for mode in ('zero', 'extrapolated'):
        navsto.set_mode('parity')
        navsto.solver_parameters['newton']['linear_start'] = mode
        navsto.solver_parameters['pressure']['start'] = mode
        navsto.solver_parameters['correction']['increment_start'] = mode
        prob = karman.KarmanProblem(nx, ny)
        prob.set_initial_profile()
        rows = []
        for _ in range(steps):
            info = prob.step()
            rows.append((device.to_host(prob.u0.data).numpy().copy(),
                         device.to_host(prob.p0.data).numpy().copy(),
                         info['dt'], sum(info['newton_linear_applications']),
                         info['pressure'].iterations, info['pressure'].residual,
                         info['correction'].iterations))
        runs[mode] = rows
    for k in range(steps):
        a, b = runs['zero'][k], runs['extrapolated'][k]
        du = numpy.linalg.norm(a[0] - b[0]) / numpy.linalg.norm(a[0])
        dp = numpy.linalg.norm(a[1] - b[1]) / numpy.linalg.norm(a[1])
        print('step %d dt %.2e: du %.2e dp %.2e  (gmres, p its, |Br|, corr) %r -> %r'
              % (k, a[2], du, dp, a[3:], b[3:]))


if __name__ == '__main__':
    main()
