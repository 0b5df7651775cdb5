# -*- coding: utf-8 -*-
'''Which extrapolation predicts the next increment best?  A Karman run (the
bench's protocol at proxy size) in which, before every step, each candidate
scheme (points m, polynomial degree q; q = m - 1: interpolation) predicts the
Newton increment, the pressure increment and the velocity-correction
increment from the stored histories, and after the step the prediction is
compared with what the solves found: median relative error per block of steps.
  python tools/extrapolation_lab.py [nx] [steps] [mu]
SPIN=n: n steps before the measurement starts (the developed vortex street of
the headline mesh: nx 2182, mu 0.002, SPIN=2400).
'''
from __future__ import print_function
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                            # noqa: E402

CANDIDATES = [(2, 1), (3, 2), (4, 3), (5, 4), (6, 5),
              (4, 2), (5, 2), (6, 2), (5, 3), (6, 3), (6, 4), (4, 1), (6, 1)]


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 772
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.00565
    import ctypes
    from flow_amd import karman, device, _hip
    from flow_amd.fem import ops
    import flow_amd.navier_stokes as navsto
    from flow_amd.navier_stokes import pressure_correction as pc
    ny = max(2, int(round(nx * 509.0 / 2182.0)))
    prob = karman.KarmanProblem(nx, ny, mu=mu)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    prob.settle()
    for _ in range(int(os.environ.get('SPIN', '0'))):
        prob.step()
    fields = (('newton', prob.W.layout, 'newton_increments', 1),
              ('newton#2', prob.W.layout, ('newton_increments', 1), 1),
              ('pressure', prob.W.layout, 'pressure_increments', 1),
              ('correction', prob.W.layout, 'correction_increments', 2))

    def history(lay, key):
        # (round 5: the histories belong to the trajectory of the run,
        # flow_amd/navier_stokes/start_vectors.py)
        st = lay._dev.get('start_vector_state')
        if st is None or not st.trajectories:
            return []
        return max(st.trajectories, key=lambda t: t.used).hist.get(key, [])
    errs = {f[0]: {c: [] for c in CANDIDATES} for f in fields}
    counts = []
    for k in range(steps):
        preds = {}
        for name, lay, key, power in fields:
            hist = history(lay, key)
            for (m, q) in CANDIDATES:
                if len(hist) < m:
                    continue
                h = hist[:m]
                w = pc.extrapolation_weights([x[1] for x in h], prob.dt, power, q)
                n = h[0][0].numel()
                out = device.empty(n)
                coef = (ctypes.c_double * m)(*w)
                ptrs = (ctypes.c_void_p * m)(*[_hip.f64(x[0], n).value for x in h])
                _hip.check(_hip.lib().flow_lincomb(n, m, coef, ptrs,
                                                   _hip.f64(out, n), _hip.stream()))
                preds[(name, m, q)] = out
        info = prob.step()
        counts.append((sum(info['newton_linear_applications']),
                       info['pressure'].iterations,
                       info['correction'].iterations))
        for name, lay, key, power in fields:
            if not history(lay, key):
                continue
            actual = history(lay, key)[0][0]
            na = ops.vector_norm(actual)
            for (m, q) in CANDIDATES:
                p = preds.get((name, m, q))
                if p is None:
                    continue
                ops.axpby(-1.0, actual, 1.0, p)
                errs[name][(m, q)].append(ops.vector_norm(p) / na)
    block = 50
    for name in errs:
        print('== %s increment: median relative prediction error per %d steps'
              % (name, block))
        print('%-8s' % '(m, q)' + ''.join('%10d' % (b * block)
                                            for b in range(steps // block)))
        for c in CANDIDATES:
            e = errs[name][c]
            off = steps - len(e)
            row = []
            for b in range(steps // block):
                seg = [e[i - off] for i in range(b * block, (b + 1) * block)
                       if i - off >= 0]
                row.append(numpy.median(seg) if seg else float('nan'))
            print('%-8s' % (c,) + ''.join('%10.1e' % v for v in row))
    print('iterations (GMRES applications, pressure, corrections), every 25th '
          'step: %r' % (counts[::25],))


if __name__ == '__main__':
    main()
