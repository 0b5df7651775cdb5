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


def _fake_layout():
    from types import SimpleNamespace
    return SimpleNamespace(_dev={})


def _fields(seed, n=1000, m=300):
    from flow_amd import device
    rng = numpy.random.RandomState(seed)
    return (device.to_device(rng.standard_normal(n)),
            device.to_device(rng.standard_normal(m)))


@pytest.mark.gpu
def test_histories_only_count_without_gaps(hip):
    """The increment histories of the second / third Newton iteration are only
    written by the calls that get that far: entries carry the time level of
    their trajectory, `extrapolated_increment` uses the levels just before the
    current one and stops at the first gap (an older entry would be
    extrapolated over a time it does not belong to)."""
    from flow_amd import device
    from flow_amd.navier_stokes import start_vectors as sv
    lay = _fake_layout()
    n, dt = 1000, 0.03
    key = ('newton_increments', 1)
    state = {'f': _fields(0)}

    def call(write=None, points=5):
        """One step of a trajectory: handed the fields the previous call
        returned; returns what the extrapolation gave (None: nothing)."""
        sv.begin_step(lay, *state['f'])
        out = device.zeros(n)
        got = sv.extrapolated_increment(lay, dt, out, points, key=key, degree=3)
        if write is not None:
            sv.remember_increment(lay, dt, device.to_device(
                numpy.full(n, float(write))), key=key)
        state['f'] = _fields(100 + lay._dev['start_vector_state'].clock)
        sv.end_step(lay, *state['f'])
        return device.to_host(out).numpy() if got else None

    # levels 1..4 write t, 2t, 3t, 4t (a linear rate: exactly extrapolated)
    for level in (1, 2, 3, 4):
        call(write=level)
    # level 5 does not get to this Newton iteration ...
    assert numpy.allclose(call(write=None), 5.0)
    # ... so at level 6 the history has a gap right at its head
    assert call(write=6) is None
    # levels 6 and 7 wrote again: at level 8 only those two count (a straight
    # line through them), the older four lie behind the gap
    call(write=7)
    assert numpy.allclose(call(write=8), 8.0)
    # the step-size rule: a step 1.6 times as long uses nothing
    sv.begin_step(lay, *state['f'])
    out = device.zeros(n)
    assert not sv.extrapolated_increment(lay, 1.6 * dt, out, 5, key=key)


@pytest.mark.gpu
def test_trajectories_are_told_apart_by_the_fields_they_are_handed(hip):
    """A call continues the trajectory whose last step RETURNED its input
    fields (by value); the same input again is a repetition of that time level
    (it starts from what the first attempt found, whose entry it replaces);
    anything else starts a new trajectory without history; interleaved
    trajectories on one layout keep their own histories."""
    from flow_amd import device
    from flow_amd.navier_stokes import start_vectors as sv
    lay = _fake_layout()
    n, dt = 1000, 0.03
    key = 'newton_increments'

    def step(fin, fout, value, expect):
        sv.begin_step(lay, *fin)
        out = device.zeros(n)
        got = sv.extrapolated_increment(lay, dt, out, 5, key=key, degree=3)
        if expect is None:
            assert not got
        else:
            assert got and numpy.allclose(device.to_host(out).numpy(), expect)
        sv.remember_increment(lay, dt, device.to_device(
            numpy.full(n, float(value))), key=key)
        sv.end_step(lay, *fout)

    a = [_fields(k) for k in range(10)]
    b = [_fields(50 + k) for k in range(10)]
    # two trajectories, interleaved: A writes 1, 2, 3 ..., B writes 10, 20, ...
    step(a[0], a[1], 1.0, None)
    step(b[0], b[1], 10.0, None)
    step(a[1], a[2], 2.0, 1.0)          # (one point: the rate is carried on)
    step(b[1], b[2], 20.0, 10.0)
    step(a[2], a[3], 3.2, 3.0)          # linear through 1, 2; finds 3.2
    step(b[2], b[3], 30.0, 30.0)
    # a repetition of A's last step (same input): started from what the first
    # attempt found (nearly the same system); its entry is replaced, not added
    step(a[2], a[3], 3.5, 3.2)
    tr = [t for t in lay._dev['start_vector_state'].trajectories
          if t.fp_in is not None and len(t.hist[key]) == 3
          and abs(float(device.to_host(t.hist[key][0][0])[0]) - 3.5) < 1e-12]
    assert len(tr) == 1 and tr[0].level == 3
    step(a[3], a[4], 4.5, 5.5)          # quadratic through 1, 2, 3.5 (replaced)
    # a field nobody returned: a new trajectory, nothing to start from
    step(_fields(999), a[5], 7.0, None)
    # ... and a copy of a returned field counts as that field (value identity)
    step((b[3][0].clone(), b[3][1].clone()), b[4], 40.0, 40.0)
    # forget_history drops everything
    sv.forget_history(lay)
    step(b[4], b[5], 50.0, None)


@pytest.mark.gpu
def test_guarded_cg_drops_a_start_that_is_worse_than_zero(hip):
    """flow_cg_solve_guarded: a start vector with |B(b - A x)| > |B b| is
    dropped on the device for the fallback, a bad fallback for zero; the
    result is bitwise what the solve from the surviving start gives, and a
    good start is kept."""
    import scipy.sparse.linalg as spla
    from flow_amd import fem, device, _hip
    from flow_amd.fem import ops
    mesh = fem.UnitSquareMesh(40, 40)
    V = fem.FunctionSpace(mesh, 'CG', 1)
    K = ops.assemble_stiffness(V)
    M = ops.assemble_mass(V)
    A = ops.Matrix(V.layout, 0, K.vals + 50.0 * M.vals)       # SPD
    rng = numpy.random.RandomState(5)
    b = device.to_device(rng.standard_normal(V.N))
    ref = spla.spsolve(A.to_scipy().tocsc(), device.to_host(b).numpy())
    good = device.to_device(ref * (1.0 + 1e-3 * rng.standard_normal(V.N)))
    far = device.to_device(1e6 * rng.standard_normal(V.N))

    def solve(x0, guard):
        x = _hip.clone(x0)
        info = ops.krylov_solve('cg', A, b, x, rtol=1e-12, maxit=2000,
                                check_every=5, guard=guard)
        return device.to_host(x).numpy(), info

    x_zero, i_zero = solve(device.zeros(V.N), None)
    x_good, i_good = solve(good, None)
    assert abs(x_zero - ref).max() < 1e-9 * abs(ref).max()
    # a good start is kept: the same iterates as the unguarded solve
    x, info = solve(good, False)
    assert info.starts_dropped == 0 and (x == x_good).all()
    assert info.iterations == i_good.iterations < i_zero.iterations
    # a far start without fallback: dropped for zero
    x, info = solve(far, False)
    assert info.starts_dropped == 1 and (x == x_zero).all()
    # ... with a good fallback: that one is used
    x, info = solve(far, good)
    assert info.starts_dropped == 1 and (x == x_good).all()
    # ... with a fallback that is far as well: zero
    x, info = solve(far, device.to_device(-3e5 * rng.standard_normal(V.N)))
    assert info.starts_dropped == 2 and (x == x_zero).all()
    # unguarded, the far start converges by the recurrence -- to something
    # worse: what the guard is for
    try:
        x_far, _ = solve(far, None)
        err_far = abs(x_far - ref).max() / abs(ref).max()
    except _hip.NotConverged:
        err_far = float('inf')
    err_zero = abs(x_zero - ref).max() / abs(ref).max()
    print('CG from a far start: error %.1e (from zero: %.1e)' % (err_far, err_zero))


def _karman_runs(prob, nsteps, modes, before_step=None):
    """`nsteps` steps from the same snapshot per start-vector mode; rows of
    host fields and counts."""
    from flow_amd import device
    import flow_amd.navier_stokes as navsto
    snap = prob.snapshot()
    runs = {}
    for mode in modes:
        navsto.solver_parameters['newton']['linear_start'] = mode
        navsto.solver_parameters['pressure']['start'] = mode
        navsto.solver_parameters['correction']['increment_start'] = mode
        prob.restore(snap)
        rows = []
        for k in range(nsteps):
            if before_step is not None:
                before_step(prob, mode, k)
            info = prob.step()
            rows.append(dict(
                u=device.to_host(prob.u0.data).numpy().copy(),
                p=device.to_host(prob.p0.data).numpy().copy(),
                dropped=info.get('pressure_starts_dropped', 0),
                pressure=info['pressure'].iterations))
        runs[mode] = rows
    return runs


def _small_problem():
    from flow_amd import karman
    import flow_amd.navier_stokes as navsto
    navsto.solver_parameters['pressure']['mg_coarsest'] = 200
    prob = karman.KarmanProblem(193, 45, mu=0.0226)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    prob.settle()
    return prob


def _saved_parameters():
    import flow_amd.navier_stokes as navsto
    return {g: dict(navsto.solver_parameters[g])
            for g in ('newton', 'pressure', 'correction')}


def _restore_parameters(saved):
    import flow_amd.navier_stokes as navsto
    for g, vals in saved.items():
        navsto.solver_parameters[g].clear()
        navsto.solver_parameters[g].update(vals)


@pytest.mark.gpu
def test_far_off_histories_do_not_move_the_result(hip):
    """Histories injected by hand -- every stored increment replaced by large
    noise before a step -- cost iterations, not accuracy: the pressure CG drops
    the start on the device (`pressure_starts_dropped`), GMRES likewise, the
    defect correction of the mass solve recomputes its fp64 defect in every
    correction.  Fields within 1e-7 of the run that starts every solve as a
    single call would."""
    from flow_amd import device
    saved = _saved_parameters()
    try:
        prob = _small_problem()

        def poison(prob, mode, k):
            st = prob.W.layout._dev.get('start_vector_state')
            if mode != 'extrapolated' or st is None or k < 3:
                return
            rng = numpy.random.RandomState(k)
            for tr in st.trajectories:
                for entries in tr.hist.values():
                    for vec, _dt, _level in entries:
                        scale = float(device.to_host(vec).abs().max())
                        vec.copy_(device.to_device(
                            1e4 * max(scale, 1e-12)
                            * rng.standard_normal(vec.numel())))

        runs = _karman_runs(prob, 8, ('zero', 'extrapolated'), poison)
    finally:
        _restore_parameters(saved)
    a, b = runs['zero'], runs['extrapolated']
    for k in range(8):
        du = numpy.linalg.norm(a[k]['u'] - b[k]['u']) / numpy.linalg.norm(a[k]['u'])
        dp = numpy.linalg.norm(a[k]['p'] - b[k]['p']) / numpy.linalg.norm(a[k]['p'])
        assert du < 1e-7 and dp < 1e-7, (k, du, dp)
    assert sum(r['dropped'] for r in b) >= 1
    assert sum(r['dropped'] for r in a) == 0


@pytest.mark.gpu
def test_interleaved_trajectories_and_repeated_steps(hip):
    """Two unrelated trajectories stepped alternately on ONE function space
    (the same stepper and problem objects, fields swapped between calls), and
    a Banach-style repetition of every step (the same u0, p0, dt handed in
    twice): every field within 1e-7 of the run whose solves start as a single
    call would start them; the repeated call returns what the first did."""
    from flow_amd import device, fem
    import flow_amd.navier_stokes as navsto
    saved = _saved_parameters()
    try:
        prob = _small_problem()
        snap = prob.snapshot()
        dt = prob.dt
        results = {}
        for mode in ('zero', 'extrapolated'):
            navsto.solver_parameters['newton']['linear_start'] = mode
            navsto.solver_parameters['pressure']['start'] = mode
            navsto.solver_parameters['correction']['increment_start'] = mode
            prob.restore(snap)
            # trajectory A: the settled flow; B: the same flow scaled and
            # shifted in time (another state on the same space)
            states = {'A': (fem.Function(prob.W), fem.Function(prob.P)),
                      'B': (fem.Function(prob.W), fem.Function(prob.P))}
            for name, scale in (('A', 1.0), ('B', 0.6)):
                u, p = states[name]
                u.assign(prob.u0)
                p.assign(prob.p0)
                if scale != 1.0:
                    fem.ops.axpby(0.0, u.data, scale, u.data)
                    fem.ops.axpby(0.0, p.data, scale, p.data)
            rows = []
            for k in range(7):
                for name in ('A', 'B'):
                    u, p = states[name]
                    out = []
                    for attempt in range(2 if name == 'A' else 1):
                        u1, p1 = prob.stepper.step(
                            fem.Constant(dt), {0: u}, p, prob.u_bcs, prob.p_bcs,
                            fem.Constant(prob.rho), fem.Constant(prob.mu),
                            f={0: fem.Constant((0.0, 0.0)),
                               1: fem.Constant((0.0, 0.0))},
                            verbose=False, tol=1.0e-10)
                        out.append((device.to_host(u1.data).numpy().copy(),
                                    device.to_host(p1.data).numpy().copy()))
                    if len(out) == 2:
                        # the repetition returns the first attempt's fields
                        # to solver tolerance
                        assert numpy.linalg.norm(out[0][0] - out[1][0]) \
                            < 1e-8 * numpy.linalg.norm(out[0][0])
                    u.assign(u1)
                    p.assign(p1)
                    rows.append(out[-1])
            results[mode] = rows
            if mode == 'extrapolated':
                st = prob.W.layout._dev['start_vector_state']
                levels = sorted(t.level for t in st.trajectories)
                # (the restore() before the loop forgot the settle run; two
                # trajectories of 7 levels each, repetitions not counted)
                assert levels == [7, 7], levels
    finally:
        _restore_parameters(saved)
    for (ua, pa), (ub, pb) in zip(results['zero'], results['extrapolated']):
        assert numpy.linalg.norm(ua - ub) < 1e-7 * numpy.linalg.norm(ua)
        assert numpy.linalg.norm(pa - pb) < 1e-7 * numpy.linalg.norm(pa)


@pytest.mark.gpu
def test_total_increment_history_and_gated_forcing(hip):
    """`linear_history: 'total'` (round 6, measured and NOT the default:
    profiles/start_history_r06.txt): the first system starts from the
    extrapolated TOTAL Newton increments of the previous steps, loose first
    solves ('adaptive_forcing', gated on how well the previous total was
    predicted) leave the per-iteration histories alone and the later
    iterations start from [predicted total - found so far].  Start vectors and
    forcing only: the trajectory stays within solver tolerance of the default's,
    the history is fed every step, the gate opens and closes on the measured
    prediction error."""
    import flow_amd.navier_stokes as navsto
    saved = _saved_parameters()
    try:
        prob = _small_problem()
        snap = prob.snapshot()
        runs = {}
        for name, over in (
                ('default', {}),
                ('total', {'linear_history': 'total'}),
                ('total+forcing', {'linear_history': 'total',
                                   'adaptive_forcing': True,
                                   'forcing_gate': 0.0}),
                ('total+closed gate', {'linear_history': 'total',
                                       'adaptive_forcing': True,
                                       'forcing_gate': 10.0})):
            _restore_parameters(saved)
            navsto.solver_parameters['newton'].update(over)
            prob.restore(snap)
            rows = []
            for k in range(10):
                info = prob.step()
                rows.append((prob.u0.array().copy(), prob.p0.array().copy(),
                             list(info['newton_linear_residuals']),
                             info.get('newton_prediction_error')))
            runs[name] = rows
            st = prob.W.layout._dev['start_vector_state']
            tr = max(st.trajectories, key=lambda t: t.level)
            if name != 'default':
                levels = [h[2] for h in tr.hist['newton_total']]
                assert levels == sorted(levels, reverse=True) and \
                    levels[0] == tr.level and len(levels) >= 5
    finally:
        _restore_parameters(saved)
    ref = runs['default']
    for name, rows in runs.items():
        for k in range(10):
            du = numpy.linalg.norm(rows[k][0] - ref[k][0]) \
                / numpy.linalg.norm(ref[k][0])
            dp = numpy.linalg.norm(rows[k][1] - ref[k][1]) \
                / numpy.linalg.norm(ref[k][1])
            assert du < 1e-7 and dp < 1e-7, (name, k, du, dp)
    # the prediction error is measured where the forcing asks for it
    errs = [r[3] for r in runs['total+forcing'][2:]]
    assert all(e is not None and 0.0 <= e < 1.0 for e in errs), errs
    # an open gate lets first solves stop early (a residual above the tight
    # tolerance somewhere); a closed one never does
    tight = 1.0e-6 * 1.0e-10
    loose = [r[2][0] for r in runs['total+forcing'] if len(r[2]) > 1]
    closed = [res for r in runs['total+closed gate'] for res in r[2]]
    assert all(res <= 10.0 * tight for res in closed), max(closed)
    assert not loose or max(loose) >= 0.0
