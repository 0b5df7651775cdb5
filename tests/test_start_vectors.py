# -*- coding: utf-8 -*-
'''
Start vectors extrapolated in time (flow_amd/navier_stokes: `linear_start`,
pressure `start`, correction `increment_start`).  A time loop's solves start
from the previous steps' increments, fitted in time; every solve still
converges to the reference's stopping test and the Newton iteration still
starts from u0 (reference :204-220) -- so the trajectory must not move beyond
solver tolerance, and the iteration counts must drop.  GPU: -m gpu.
CPU: the extrapolation weights themselves.
'''
import numpy
import pytest

from flow_amd.navier_stokes.start_vectors import extrapolation_weights


def test_extrapolation_weights_reproduce_polynomials():
    '''Rates that are polynomials of degree <= q in time are extrapolated
    exactly (interpolation and least-squares fit, uneven step sizes, both
    scalings of the increment with the step size); more points than q + 1
    lower the noise amplification sum(w^2).'''
    rng = numpy.random.RandomState(3)
    for power in (1, 2):
        for m, q in ((2, None), (3, None), (5, None), (5, 3), (6, 2), (4, 1)):
            dts = list(0.03 * (1.0 + 0.2 * rng.uniform(-1, 1, size=m)))
            dt = 0.031
            deg = m - 1 if q is None else q
            coef = rng.standard_normal(deg + 1)
            # mid points of the past steps (time 0 = end of the newest) ...
            mids, t = [], 0.0
            for dtk in dts:
                mids.append(t - 0.5 * dtk)
                t -= dtk
            rate = lambda x: sum(c * x**k for k, c in enumerate(coef))  # noqa: E731
            incs = [rate(mk) * dtk**power for mk, dtk in zip(mids, dts)]
            w = extrapolation_weights(dts, dt, power, q)
            pred = sum(wi * di for wi, di in zip(w, incs))
            exact = rate(0.5 * dt) * dt**power
            assert abs(pred - exact) <= 1e-9 * max(abs(exact), 1.0), (m, q, power)
    even = [0.03] * 6
    w_interp = extrapolation_weights(even[:4], 0.03, 1, None)      # cubic, 4 pts
    w_fit = extrapolation_weights(even[:5], 0.03, 1, 3)            # cubic, 5 pts
    assert sum(x * x for x in w_fit) < 0.6 * sum(x * x for x in w_interp)
    assert numpy.allclose(extrapolation_weights(even[:3], 0.03), [3.0, -3.0, 1.0])


@pytest.mark.gpu
def test_start_vectors_do_not_move_the_trajectory(hip):
    '''The bench's protocol at a sixteenth of its size (viscosity scaled with
    the mesh width: the headline's cell Peclet number; Stokes start; settled
    step size): 16 steps with every Krylov solve started as a single call
    would start it, and 16 with the extrapolated start vectors, from the same
    state.  Fields agree far inside the north-star tolerance (1e-6) after every
    step, step sizes too, and the extrapolated run needs fewer GMRES
    applications, pressure CG iterations and defect corrections.'''
    from flow_amd import karman, device
    import flow_amd.navier_stokes as navsto
    saved = {g: dict(navsto.solver_parameters[g])
             for g in ('newton', 'pressure', 'correction')}
    try:
        navsto.solver_parameters['pressure']['mg_coarsest'] = 200
        prob = karman.KarmanProblem(386, 90, mu=0.0113)
        prob.prepare()
        prob.reset(1.0e-5)
        prob.set_initial_stokes()
        navsto.set_mode('parity')
        prob.settle()
        snap = prob.snapshot()
        runs = {}
        for mode in ('zero', 'extrapolated'):
            navsto.solver_parameters['newton']['linear_start'] = mode
            navsto.solver_parameters['pressure']['start'] = mode
            navsto.solver_parameters['correction']['increment_start'] = mode
            prob.restore(snap)
            rows = []
            for _ in range(16):
                info = prob.step()
                rows.append(dict(
                    u=device.to_host(prob.u0.data).numpy().copy(),
                    p=device.to_host(prob.p0.data).numpy().copy(),
                    dt=info['dt'], newton=len(info['newton_residuals']) - 1,
                    gmres=sum(info['newton_linear_applications']),
                    pressure=info['pressure'].iterations,
                    correction=info['correction'].iterations))
            runs[mode] = rows
    finally:
        for g, vals in saved.items():
            navsto.solver_parameters[g].clear()
            navsto.solver_parameters[g].update(vals)
    a, b = runs['zero'], runs['extrapolated']
    for k in range(16):
        du = numpy.linalg.norm(a[k]['u'] - b[k]['u']) / numpy.linalg.norm(a[k]['u'])
        dp = numpy.linalg.norm(a[k]['p'] - b[k]['p']) / numpy.linalg.norm(a[k]['p'])
        assert du < 1e-8 and dp < 1e-7, (k, du, dp)
        assert abs(a[k]['dt'] - b[k]['dt']) <= 1e-8 * a[k]['dt']
        assert a[k]['newton'] == b[k]['newton']         # the same Newton path
    # once the histories are full (5 steps) the counts drop
    for key in ('gmres', 'pressure', 'correction'):
        za = sum(r[key] for r in a[6:])
        zb = sum(r[key] for r in b[6:])
        assert zb < 0.75 * za, (key, za, zb)
    print('16 steps, zero start -> extrapolated start: GMRES applications '
          '%d -> %d, pressure iterations %d -> %d, corrections %d -> %d; '
          'du %.1e dp %.1e at the last step' % (
              sum(r['gmres'] for r in a), sum(r['gmres'] for r in b),
              sum(r['pressure'] for r in a), sum(r['pressure'] for r in b),
              sum(r['correction'] for r in a), sum(r['correction'] for r in b),
              du, dp))


@pytest.mark.gpu
def test_stamped_histories_only_count_without_gaps(hip):
    '''The increment histories of the second / third Newton iteration are only
    written by the calls that get that far: `extrapolated_increment` with a
    stamp uses the entries of the calls just before this one and stops at the
    first gap (an older entry would be extrapolated over a time it does not
    belong to).'''
    from types import SimpleNamespace
    from flow_amd import device
    from flow_amd.navier_stokes import start_vectors as sv
    lay = SimpleNamespace(_dev={})
    n = 1000
    dt = 0.03
    key = ('newton_increments', 1)
    # calls 1..4 wrote t, 2t, 3t, 4t (a linear rate: exactly extrapolated)
    for call in (1, 2, 3, 4):
        sv.remember_increment(lay, dt, device.to_device(
            numpy.full(n, float(call))), key=key, stamp=call)
    out = device.zeros(n)
    assert sv.extrapolated_increment(lay, dt, out, 5, key=key, degree=3,
                                      stamp=5)
    assert numpy.allclose(device.to_host(out).numpy(), 5.0)
    # call 5 did not get to this Newton iteration: at call 6 the history has a
    # gap right at its head -> nothing to start from
    out = device.zeros(n)
    assert not sv.extrapolated_increment(lay, dt, out, 5, key=key, degree=3,
                                          stamp=6)
    assert (device.to_host(out).numpy() == 0.0).all()
    # calls 6 and 7 write again: at call 8 only those two count (a straight
    # line through them: 6, 7 -> 8), the older four lie behind the gap
    for call in (6, 7):
        sv.remember_increment(lay, dt, device.to_device(
            numpy.full(n, float(call))), key=key, stamp=call)
    out = device.zeros(n)
    assert sv.extrapolated_increment(lay, dt, out, 5, key=key, degree=3,
                                      stamp=8)
    assert numpy.allclose(device.to_host(out).numpy(), 8.0)
    # without a stamp (the first iteration's history, written on every call)
    # all entries within the step-size rule count, as before
    out = device.zeros(n)
    assert sv.extrapolated_increment(lay, dt, out, 2, key=key)
    assert numpy.allclose(device.to_host(out).numpy(), 8.0)
