# -*- coding: utf-8 -*-
'''Two runs of the start-up ramp from the same snapshot, launched kernel by
kernel: what is the first number that differs?'''
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'tests'))
import numpy                                   # noqa: E402


def main():
    from flow_amd import _hip, device
    import flow_amd.navier_stokes as navsto
    import test_graph_replay as T
    prob = T._problem()
    snap = prob.snapshot()
    _hip.graph_mode(0)

    def run():
        prob.restore(snap)
        for slot in ('jacobian_ilu', 'jacobian_pmg'):
            pre = prob.W.layout._dev.get(slot)
            if pre is not None:
                pre.stale = True
        out = []
        for _ in range(9):
            info = prob.step()
            out.append(dict(
                u=device.to_host(prob.u0.data).numpy().copy(),
                p=device.to_host(prob.p0.data).numpy().copy(),
                dt=info['dt'], unorm=info.get('unorm'),
                proj=info.get('projection_iterations'),
                newton=tuple(info['newton_residuals']),
                lin=tuple(info['newton_linear_residuals']),
                apps=tuple(info['newton_linear_applications']),
                pres=(info['pressure'].iterations, info['pressure'].residual),
                corr=(info['correction'].iterations, info['correction'].residual),
                pre=navsto.last_step_info.get('newton_preconditioner'),
                dropped=info.get('pressure_starts_dropped')))
        return out
    run()
    a, b = run(), run()
    for k, (x, y) in enumerate(zip(a, b)):
        print('step %d: du %.1e dp %.1e' % (
            k, numpy.abs(x['u'] - y['u']).max(), numpy.abs(x['p'] - y['p']).max()))
        for key in ('dt', 'unorm', 'proj', 'newton', 'lin', 'apps', 'pres', 'corr',
                    'pre', 'dropped'):
            print('   %-7s %s%s' % (key, x[key],
                                    '' if x[key] == y[key] else '   !=   %s' % (y[key],)))


if __name__ == '__main__':
    main()
