# -*- coding: utf-8 -*-
'''Which replayed loop, if any, moves the numbers?  The 14 steps of
tests/test_graph_replay.py launched kernel by kernel (twice: is the run itself
repeatable?) and with the graphs of one loop at a time.'''
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), 'tests'))
import numpy                                   # noqa: E402


def main():
    from flow_amd import _hip, device
    import test_graph_replay as T
    prob = T._problem()
    snap = prob.snapshot()

    def run(mode, sites):
        _hip.graph_mode(mode, sites=sites)
        prob.restore(snap)
        # (the Newton preconditioner is lagged and its eigenvalue estimates are
        # warm-started from the previous build: every run builds a new one)
        for slot in ('jacobian_ilu', 'jacobian_pmg'):
            prob.W.layout._dev.pop(slot, None)
        out = []
        for _ in range(14):
            info = prob.step()
            out.append((device.to_host(prob.u0.data).numpy().copy(),
                        device.to_host(prob.p0.data).numpy().copy(),
                        info['pressure'].iterations,
                        tuple(info['newton_linear_applications']),
                        info['correction'].iterations))
        return out
    run(0, 7)          # (the first run after the set-up is not the reference)
    ref = run(0, 7)
    for name, mode, sites in (('off again', 0, 7), ('cg', 1, 1), ('gmres', 1, 2),
                              ('mass', 1, 4), ('all', 1, 7)):
        got = run(mode, sites)
        first = None
        for k, (a, b) in enumerate(zip(ref, got)):
            if not (numpy.array_equal(a[0], b[0]) and numpy.array_equal(a[1], b[1])):
                first = k
                break
        print('%-10s first differing step: %s   %s' % (
            name, first, '' if first is None else
            'du %.2e dp %.2e, iterations %s | %s' % (
                numpy.abs(ref[first][0] - got[first][0]).max(),
                numpy.abs(ref[first][1] - got[first][1]).max(),
                ref[first][2:], got[first][2:])), flush=True)
    print(_hip.graph_stats())


if __name__ == '__main__':
    main()
