# -*- coding: utf-8 -*-
'''Does the start vector of the Newton linear solve move the trajectory?
N plateau steps of the Karman channel from the same settled state, once with
'linear_start': 'zero' and once with 'extrapolated' (flow_amd/navier_stokes):
relative l2 distance of u and p after every step, GMRES applications per step.
  python tools/linear_start_check.py [nx] [steps] [mu]
'''
from __future__ import print_function
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                            # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 772
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.00565
    from flow_amd import karman, device
    import flow_amd.navier_stokes as navsto
    ny = max(2, int(round(nx * 509.0 / 2182.0)))
    prob = karman.KarmanProblem(nx, ny, mu=mu)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    navsto.solver_parameters['newton']['linear_start'] = 'zero'
    prob.settle()
    snap = prob.snapshot()
    runs = {}
    for mode in ('zero', 'extrapolated'):
        navsto.solver_parameters['newton']['linear_start'] = mode
        navsto.solver_parameters['newton']['linear_start_points'] = \
            int(os.environ.get('POINTS', '5'))
        navsto.solver_parameters['newton']['linear_start_degree'] = \
            int(os.environ.get('DEGREE', '3'))
        navsto.solver_parameters['correction']['increment_start'] = mode
        for grp in ('pressure', 'correction'):
            navsto.solver_parameters[grp]['start_points'] = \
                int(os.environ.get('POINTS', '5'))
            navsto.solver_parameters[grp]['start_degree'] = \
                int(os.environ.get('DEGREE', '3'))
        navsto.solver_parameters['pressure']['start'] = mode
        prob.restore(snap)
        fields, apps = [], []
        for _ in range(steps):
            info = prob.step()
            apps.append((sum(info['newton_linear_applications']),
                         info['pressure'].iterations,
                         info['correction'].iterations))
            fields.append((device.to_host(prob.u0.data).numpy().copy(),
                           device.to_host(prob.p0.data).numpy().copy(),
                           info['dt']))
        runs[mode] = (fields, apps)
    (fa, aa), (fb, ab) = runs['zero'], runs['extrapolated']
    for k in range(steps):
        du = numpy.linalg.norm(fa[k][0] - fb[k][0]) / numpy.linalg.norm(fa[k][0])
        dp = numpy.linalg.norm(fa[k][1] - fb[k][1]) / numpy.linalg.norm(fa[k][1])
        print('step %2d: du %.2e dp %.2e ddt %.1e  GMRES applications, pressure its, corrections '
              '%r -> %r' % (k, du, dp, abs(fa[k][2] - fb[k][2]) / fa[k][2],
                            aa[k], ab[k]))


if __name__ == '__main__':
    main()
