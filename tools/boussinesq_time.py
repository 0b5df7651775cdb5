# -*- coding: utf-8 -*-
'''BASELINE config 4 (Boussinesq box, coupled N-S + heat) in steady stepping:
wall time per coupled time step after setup, Banach sweeps per step, where a
sweep's time goes (heat assemble + solve / N-S step).
  python tools/boussinesq_time.py [nx] [steps] [group.key=value ...]'''
from __future__ import print_function
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    from flow_amd import fem, boussinesq, device, heat
    import flow_amd.navier_stokes as navsto
    for kv in sys.argv[3:]:            # e.g. newton.pmg.coarse_max=24
        key, val = kv.split('=')
        path = key.split('.')
        where = heat.solver_parameters if path[0] == 'heat' \
            else navsto.solver_parameters[path[0]]
        for part in path[1:-1]:
            where = where[part]
        old = where[path[-1]]
        if isinstance(old, bool):
            where[path[-1]] = val in ('1', 'True', 'true')
        else:
            where[path[-1]] = val if isinstance(old, str) \
                else type(old)(float(val))
    mesh = fem.heater_box(nx, fitted=nx >= 12)
    t0 = time.time()
    stepper = boussinesq.FixedPointStepper(boussinesq.HeaterBox(mesh), 1.0e-2)
    device.synchronize()
    print('setup %.1f s' % (time.time() - t0))
    tim = {'heat_init': 0.0, 'heat_solve': 0.0, 'ns': 0.0}
    orig_init, orig_solve = heat.Heat.__init__, heat.Heat.solve_alpha_M_beta_F

    def timed(name, fn):
        def wrapper(*a, **kw):
            device.synchronize()
            t = time.time()
            out = fn(*a, **kw)
            device.synchronize()
            tim[name] += time.time() - t
            return out
        return wrapper
    heat.Heat.__init__ = timed('heat_init', orig_init)
    heat.Heat.solve_alpha_M_beta_F = timed('heat_solve', orig_solve)
    flows = []
    orig_flow = boussinesq.CoupledStep.flow

    def flow(self):
        device.synchronize()
        t = time.time()
        out = orig_flow(self)
        device.synchronize()
        i = navsto.last_step_info
        flows.append((1e3 * (time.time() - t), len(i['newton_residuals']) - 1,
                      sum(i['newton_linear_applications']),
                      i['pressure'].iterations, i['correction'].iterations,
                      '%s/%s' % (i.get('newton_preconditioner'),
                                 i.get('pmg_coarse_steps'))))
        return out
    boussinesq.CoupledStep.flow = flow
    for k in range(steps):
        for key in tim:
            tim[key] = 0.0
        del flows[:]
        device.synchronize()
        t = time.time()
        stepper.advance()
        device.synchronize()
        wall = time.time() - t
        row = stepper.log[-1]
        umax = stepper.u.vector().norm('linf')
        h = 0.1 / nx
        print('         |u|max %.2e: CFL %.1f, thermal cell Peclet %.1f, viscous '
              'diffusion number %.1f' % (
                  umax, umax * row['dt'] / h, umax * h / (2.0 * 1.43e-7),
                  1.0e-6 * row['dt'] / h**2))
        print('step %2d  dt %.3e  sweeps %s  wall %.1f ms  (heat assembly %.1f, '
              'heat solve %.1f [%s])'
              % (k + 1, row['dt'], row.get('banach_steps'), 1e3 * wall,
                 1e3 * tim['heat_init'], 1e3 * tim['heat_solve'],
                 '%s, contraction %s, coarse steps %s' % (
                     heat.last_solve_info.get('heat'),
                     heat.last_solve_info.get('heat_pmg_contraction'),
                     heat.last_solve_info.get('heat_pmg_coarse_steps'))))
        print('         N-S steps (ms, Newton its, GMRES applications, pressure, '
              'corrections, preconditioner): %s'
              % ', '.join('(%.1f, %d, %d, %d, %d, %s)' % f for f in flows))


if __name__ == '__main__':
    main()
