# -*- coding: utf-8 -*-
'''The developed vortex street as a laboratory: ONE spin-up (default 2400 steps
of the headline workload, t ~ 74), a snapshot, and then any number of windows
from that snapshot, each under its own overrides of
navier_stokes.solver_parameters -- ms per step, GMRES applications per Newton
iteration, pressure iterations, corrections, sub-step times.

  python tools/developed_lab.py [SPIN] [WINDOW] -- name:group.key=value,group.key=value name2:...

e.g.  python tools/developed_lab.py 2400 24 -- base: pmg22:newton.pmg.pre=2 r6:newton.gmres_restart=6
Every window starts from the same fields and step size with empty start-vector
histories (8 warm-up steps refill them) and a freshly built preconditioner.
NX=... for other resolutions; MU=... viscosity.
'''
from __future__ import print_function
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def set_param(navsto, key, val):
    path = key.split('.')
    d = navsto.solver_parameters
    for part in path[:-1]:
        d = d[part]
    old = d.get(path[-1])
    if isinstance(old, str) or old is None:
        d[path[-1]] = val
    elif isinstance(old, bool):
        d[path[-1]] = val in ('1', 'True', 'true')
    else:
        d[path[-1]] = type(old)(float(val))


def main():
    argv = sys.argv[1:]
    cut = argv.index('--') if '--' in argv else len(argv)
    pos, exps = argv[:cut], argv[cut + 1:]
    spin = int(pos[0]) if len(pos) > 0 else 2400
    window = int(pos[1]) if len(pos) > 1 else 24
    warm = int(os.environ.get('WARM', '8'))
    import copy
    from flow_amd import karman, device
    import flow_amd.navier_stokes as navsto
    nx = int(os.environ.get('NX', '2182'))
    mu = float(os.environ.get('MU', '0.002'))
    prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)), mu=mu)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    prob.settle()
    t0 = time.time()
    for _ in range(spin):
        prob.step()
    device.synchronize()
    print('spin-up: %d steps to t = %.2f, dt = %.4f in %.1f s'
          % (spin, prob.t, prob.dt, time.time() - t0), flush=True)
    snap = prob.snapshot()
    base = copy.deepcopy(navsto.solver_parameters)
    if not exps:
        exps = ['base:']
    for exp in exps:
        name, _, rest = exp.partition(':')
        for g in base:
            if isinstance(base[g], dict):
                navsto.solver_parameters[g].clear()
                navsto.solver_parameters[g].update(copy.deepcopy(base[g]))
        for kv in [x for x in rest.split(',') if x]:
            key, val = kv.split('=')
            set_param(navsto, key, val)
        prob.restore(snap)
        for slot in ('jacobian_ilu', 'jacobian_pmg'):
            pre = prob.W.layout._dev.get(slot)
            if pre is not None:
                pre.stale = True
        # (a changed cycle shape needs a new Pmg object)
        if 'pmg' in rest:
            prob.W.layout._dev.pop('jacobian_pmg', None)
        for _ in range(warm):
            prob.step()
        device.synchronize()
        t1 = time.time()
        infos = [prob.step() for _ in range(window)]
        device.synchronize()
        ms = 1e3 * (time.time() - t1) / window
        apps = [i['newton_linear_applications'] for i in infos]
        by_it = {}
        for a in apps:
            for j, v in enumerate(a):
                by_it.setdefault(j, []).append(v)
        tim = {k: 1e3 * sum(i['timings'][k] for i in infos) / window
               for k in ('tentative_s', 'pressure_s', 'correction_s')}
        print('%-14s %6.2f ms/step | newton %.2f its, GMRES %5.1f (%s) | '
              'pressure %4.1f | corr %.2f | proj %.2f | tent %.2f pres %.2f '
              'corr %.2f ms | F1 %.1e | dropped %d'
              % (name, ms,
                 sum(len(a) for a in apps) / float(window),
                 sum(sum(a) for a in apps) / float(window),
                 ' '.join('#%d %.1f' % (j, sum(v) / float(len(v)))
                          for j, v in sorted(by_it.items())),
                 sum(i['pressure'].iterations for i in infos) / float(window),
                 sum(i['correction'].iterations for i in infos) / float(window),
                 sum(i.get('projection_iterations', 0) for i in infos) / float(window),
                 tim['tentative_s'], tim['pressure_s'], tim['correction_s'],
                 sum(i['newton_residuals'][1] for i in infos
                     if len(i['newton_residuals']) > 1) / float(window),
                 sum(i.get('pressure_starts_dropped', 0) for i in infos)),
              flush=True)


if __name__ == '__main__':
    main()
