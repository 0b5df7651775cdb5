# -*- coding: utf-8 -*-
'''The developed vortex street: run the headline workload until the wake sheds
(default 2400 steps from the Stokes start, t ~ 70), then time a window of steps
there -- ms per step, Newton iterations, GMRES applications per Newton
iteration, pressure iterations, corrections.
  python tools/developed_flow.py [spin_up_steps] [window] [group.key=value ...]
(NX=... for other resolutions, MODE=fast for the other solver mode)'''
from __future__ import print_function
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = [a for a in sys.argv[1:] if '=' not in a]
    spin = int(args[0]) if len(args) > 0 else 2400
    window = int(args[1]) if len(args) > 1 else 200
    from flow_amd import karman, device
    import flow_amd.navier_stokes as navsto
    for kv in [a for a in sys.argv[1:] if '=' in a]:
        key, val = kv.split('=')
        grp, name = key.split('.')
        old = navsto.solver_parameters[grp][name]
        navsto.solver_parameters[grp][name] = \
            val if isinstance(old, str) else type(old)(float(val))
        print('set', grp, name, navsto.solver_parameters[grp][name])
    if os.environ.get('MODE'):
        navsto.set_mode(os.environ['MODE'])
    nx = int(os.environ.get('NX', '2182'))
    prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
    prob.set_initial_stokes()
    prob.dt = 1e-5
    t0 = time.time()
    for k in range(spin):
        prob.step()
    device.synchronize()
    rebuilds0 = navsto.last_step_info.get('newton_preconditioner_rebuilds', 0)
    print('spin-up: %d steps to t = %.2f in %.1f s, %d rebuilds of the Newton '
          'preconditioner' % (spin, prob.t, time.time() - t0, rebuilds0))
    device.synchronize()
    t1 = time.time()
    infos = [prob.step() for _ in range(window)]
    device.synchronize()
    ms = 1e3 * (time.time() - t1) / window
    newton = [len(i['newton_linear_iterations']) for i in infos]
    apps = [i['newton_linear_applications'] for i in infos]
    by_it = {}
    for a in apps:
        for j, v in enumerate(a):
            by_it.setdefault(j, []).append(v)
    print('window of %d steps at t = %.2f: %.2f ms/step (%.1f steps/s)'
          % (window, prob.t, ms, 1e3 / ms))
    print('  Newton iterations/step %.2f; GMRES applications/step %.1f; per '
          'Newton iteration: %s'
          % (sum(newton) / float(window), sum(sum(a) for a in apps) / float(window),
             ', '.join('#%d: %.1f (%d solves)' % (j, sum(v) / float(len(v)), len(v))
                       for j, v in sorted(by_it.items()))))
    print('  pressure iterations/step %.1f; corrections %.2f; projection %.2f'
          % (sum(i['pressure'].iterations for i in infos) / float(window),
             sum(i['correction'].iterations for i in infos) / float(window),
             sum(i.get('projection_iterations', 0) for i in infos) / float(window)))
    print('  rebuilds of the Newton preconditioner in the window: %d'
          % (navsto.last_step_info.get('newton_preconditioner_rebuilds', 0)
             - rebuilds0))
    cmp_path = os.environ.get('COMPARE')
    if cmp_path:
        import torch
        if os.path.exists(cmp_path):
            ref = torch.load(cmp_path)
            print('  against %s: u %.3e  p %.3e (rel-L2)' % (
                cmp_path,
                float((prob.u0.data - ref['u']).norm() / ref['u'].norm()),
                float((prob.p0.data - ref['p']).norm() / ref['p'].norm())))
        else:
            torch.save({'u': prob.u0.data.clone(), 'p': prob.p0.data.clone()},
                       cmp_path)
    print('  dt %.3e .. %.3e; |u|inf %.4f .. %.4f; |F0| %.2e .. %.2e; |F1| %.2e .. %.2e'
          % (min(i['dt'] for i in infos), max(i['dt'] for i in infos),
             min(i['unorm'] for i in infos), max(i['unorm'] for i in infos),
             min(i['newton_residuals'][0] for i in infos),
             max(i['newton_residuals'][0] for i in infos),
             min(i['newton_residuals'][min(1, len(i['newton_residuals']) - 1)]
                 for i in infos),
             max(i['newton_residuals'][min(1, len(i['newton_residuals']) - 1)]
                 for i in infos)))


if __name__ == '__main__':
    main()
