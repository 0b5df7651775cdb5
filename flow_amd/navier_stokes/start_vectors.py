# -*- coding: utf-8 -*-
'''
Start vectors of a time loop's linear solves (flow_amd/navier_stokes/
pressure_correction.py): the increments the previous calls found -- of every
Newton iteration's system, of the pressure, of the velocity correction --
extrapolated in time.  Start vectors only: every solve converges to the same
stopping test as from zero and the Newton iteration still starts from u0
(reference: flow/navier_stokes/pressure_correction.py:204-220, 326-339,
451-464, which start every solve from the previous field or from zero), so the
trajectory does not move beyond solver tolerance (tests/test_start_vectors.py).

Whose increments?  The reference keeps nothing between calls (a fresh
`Function` per solve, :313, :441), so nothing in its interface says which calls
belong together.  Here a history belongs to a TRAJECTORY, and a call continues
a trajectory when it is handed the very fields that trajectory's last step
returned -- by value: callers copy them (`u0.assign(u1)`,
tests/test_karman_vortex_street.py:241-242).  `begin_step` fingerprints u[0] and
p0 on the device (flow_fingerprint), `end_step` the returned fields:

  * input == a trajectory's last output: the next time level of it;
  * input == a trajectory's last INPUT: the same time level again (a Banach
    sweep of a coupled problem repeats the step with another forcing,
    tests/test_boussinesq.py:202-289; a driver redoes a step with a smaller dt,
    :254-264) -- the solves start from what the attempt before found (nearly
    the same system), and what this call finds replaces that entry instead of
    counting as a time step;
  * neither: a new trajectory without history (another problem on the same
    function space, a restart from a checkpoint, a hand-modified field).  The
    least recently used one beyond MAX_TRAJECTORIES is dropped.

Entries carry the time level they were found at and only count without a gap
(the second Newton iteration's history is only written by the calls that get
that far).  Beyond that the solvers guard themselves: GMRES and CG drop a start
that leaves a larger residual than zero on the device (gmres_drop_start_kernel,
flow_cg_solve_guarded), the defect correction of the mass solver recomputes its
fp64 defect in every correction -- a far start costs corrections, not accuracy.
'''
import ctypes

import torch

from ..fem import ops
from .. import _hip
from .. import device
from .. import parallel

MAX_TRAJECTORIES = 3
KEEP_POINTS = 6
# a repeated step starts its solves from what the attempt before found
# ('previous_attempt') or, like the first attempt, from the extrapolation of
# the earlier time levels ('extrapolated')
REDO_START = 'previous_attempt'


class Trajectory(object):
    def __init__(self):
        self.level = 0          # time level of the newest step
        self.fp_in = None       # fingerprints (u, p) that step was handed
        self.fp_out = None      # ... and returned (None: not read back yet)
        self.redo = False       # the current call repeats level `level`
        self.hist = {}          # key -> [(vector, dt, level), ...] newest first
        self.used = 0

    def entries(self, key):
        return self.hist.setdefault(key, [])


class _State(object):
    '''Per velocity layout: the trajectories and the call in flight.'''
    def __init__(self):
        self.trajectories = []
        self.slots = None         # device: [in_u, in_p, out_u, out_p] x (lo, hi)
        self.pending_out = None   # trajectory whose fp_out is still on the device
        self.current = None       # resolved trajectory of the call in flight
        self.unresolved = False   # begin_step ran, fingerprints not read yet
        self.agree = None         # device: the ranks' verdicts (strips)
        self.clock = 0


def _state(lay):
    st = lay._dev.get('start_vector_state')
    if st is None:
        st = lay._dev['start_vector_state'] = _State()
    return st


def _fingerprint(st, fields, slot0):
    lib = _hip.lib()
    if st.slots is None:
        st.slots = _hip.fill(device.empty(8), 0.0)
    k = len(fields)
    ptrs = (ctypes.c_void_p * k)(
        *[_hip.f64(f, f.numel()).value for f in fields])
    ns = (ctypes.c_int * k)(*[f.numel() for f in fields])
    _hip.check(lib.flow_fingerprint(
        k, ptrs, ns, _hip.f64(ops.work(_hip.REDUCE_WORK)),
        _hip.f64(st.slots[slot0:], 2 * k), _hip.stream()))


def begin_step(lay, u0, p0):
    '''Called at the top of a step: fingerprints of the fields it is handed
    (asynchronous; read back together with the previous call's outputs the
    first time a start vector is asked for -- behind the read-back of the first
    Newton residual, so the stream is drained already).'''
    st = _state(lay)
    st.current = None
    st.unresolved = True
    _fingerprint(st, [u0, p0], 0)
    return st


def _agree(st, code, which):
    """Every rank must take the same decision (a start vector changes which
    launches and collectives a solve issues): all or none.  -> (code, which)
    if every rank found the same trajectory by the same code, else (0, None)."""
    # ONE-HOT slots, summed: slot 0 = "no trajectory of mine matches",
    # slot 1 + 2 k + (code - 1) = "my trajectory k matches by code".  A
    # verdict stands only if ONE slot holds all `world` votes -- ranks whose
    # local strips of two trajectories coincide (a zero or steady far
    # field) can match different ones; sums of (code, which) alone do not
    # see that.
    nslots = 1 + 2 * (MAX_TRAJECTORIES + 1)
    if st.agree is None or st.agree.numel() != nslots + 1:
        st.agree = device.zeros(nslots + 1)
    t = st.agree
    votes = [0.0] * (nslots + 1)
    mine = 0 if which is None else 1 + 2 * which + (code - 1)
    votes[mine] = 1.0
    votes[nslots] = 1.0
    # (a host-to-device copy, the all-reduce and ONE read-back, all in
    # stream order)
    t.copy_(torch.tensor(votes, dtype=torch.float64))
    c = parallel.comm()
    c.calls += 1
    c.allreduce_tensor(t)
    h = device.to_host(t)
    world = int(round(float(h[nslots])))
    if which is None or int(round(float(h[mine]))) != world:
        code, which = 0, None
    return code, which


def _resolve(st):
    if not st.unresolved:
        return st.current
    st.unresolved = False
    host = (ctypes.c_double * 8)()
    _hip.check(_hip.lib().flow_read_doubles(
        _hip.f64(st.slots, 8), 8, host, _hip.stream()))
    vals = [int(v) for v in host]
    fp = lambda i: (vals[i] | (vals[i + 1] << 32),        # noqa: E731
                    vals[i + 2] | (vals[i + 3] << 32))
    if st.pending_out is not None:
        st.pending_out.fp_out = fp(4)
        st.pending_out = None
    fin = fp(0)
    code, which = 0, None
    for k, tr in enumerate(st.trajectories):
        if tr.fp_out == fin:
            code, which = 1, k
            break
    if which is None:
        for k, tr in enumerate(st.trajectories):
            if tr.fp_in == fin:
                code, which = 2, k
                break
    if parallel.active():
        code, which = _agree(st, code, which)
    st.clock += 1
    if code == 0:
        tr = Trajectory()
        st.trajectories.append(tr)
        if len(st.trajectories) > MAX_TRAJECTORIES:
            st.trajectories.remove(min(st.trajectories[:-1],
                                       key=lambda t: t.used))
    else:
        tr = st.trajectories[which]
    tr.redo = code == 2
    if code != 2:
        tr.level += 1
    tr.fp_in = fin
    tr.fp_out = None
    tr.used = st.clock
    st.current = tr
    return tr


def end_step(lay, u1, p1):
    '''Called with the fields a step returns: their fingerprints stay on the
    device until the next call reads them with its own.'''
    st = _state(lay)
    tr = _resolve(st)
    _fingerprint(st, [u1, p1], 4)
    st.pending_out = tr
    st.current = None
    return


def forget_history(space_or_layout):
    '''Drop every start-vector history kept for a (velocity) function space:
    the next calls start their solves like the reference does.'''
    lay = getattr(space_or_layout, 'layout', space_or_layout)
    lay._dev.pop('start_vector_state', None)
    return


def snapshot_state(space_or_layout):
    '''A deep copy of everything kept for a velocity space (development tools:
    several runs from the same state WITH its histories); None if nothing is.'''
    lay = getattr(space_or_layout, 'layout', space_or_layout)
    st = lay._dev.get('start_vector_state')
    if st is None:
        return None
    if st.unresolved:
        _resolve(st)
    if st.pending_out is not None:
        # (the last step's output fingerprint is still on the device)
        host = (ctypes.c_double * 8)()
        _hip.check(_hip.lib().flow_read_doubles(
            _hip.f64(st.slots, 8), 8, host, _hip.stream()))
        v = [int(x) for x in host]
        st.pending_out.fp_out = (v[4] | (v[5] << 32), v[6] | (v[7] << 32))
        st.pending_out = None
    return _copy_state(st)


def _copy_state(st):
    out = _State()
    out.clock = st.clock
    for tr in st.trajectories:
        c = Trajectory()
        c.level, c.fp_in, c.fp_out, c.used = tr.level, tr.fp_in, tr.fp_out, tr.used
        c.hist = {k: [(_hip.clone(h[0]),) + tuple(h[1:]) for h in v]
                  for k, v in tr.hist.items()}
        out.trajectories.append(c)
    return out


def restore_state(space_or_layout, snap):
    '''Install a copy of a snapshot_state() result (the snapshot stays usable).'''
    lay = getattr(space_or_layout, 'layout', space_or_layout)
    if snap is None:
        lay._dev.pop('start_vector_state', None)
    else:
        lay._dev['start_vector_state'] = _copy_state(snap)


def drop_current(lay):
    '''Forget the history of the trajectory the call in flight belongs to (a
    solve did not converge from an extrapolated start).'''
    st = lay._dev.get('start_vector_state')
    if st is not None and st.current is not None:
        st.current.hist.clear()
    return


def extrapolation_weights(dts, dt, power=1, degree=None):
    '''Weights w_i with  increment(dt) ~ sum_i w_i increment_i  for past
    increments over steps of sizes dts (newest first): what is smooth in time
    is the RATE increment / dt^power, taken at the mid points of the steps; a
    polynomial of `degree` is fitted through the rates by least squares
    (degree None / <= 0 / >= len(dts) - 1: interpolation) and evaluated at the
    middle of the new step.  More points than degree + 1 average the solver
    noise of the stored increments instead of amplifying it.  (Host arithmetic
    of the library, flow_extrapolation_weights: this runs between kernel
    launches three times per time step.)'''
    m = len(dts)
    arr = (ctypes.c_double * m)(*dts)
    out = (ctypes.c_double * m)()
    _hip.check(_hip.load_library().flow_extrapolation_weights(
        m, arr, float(dt), int(power),
        0 if degree is None else int(degree), out))
    return list(out)


def extrapolated_increment(lay, dt, dx, points=2, key='newton_increments',
                            power=1, degree=None):
    '''dx <- the increment this call is likely to find, extrapolated in time
    from the increments the trajectory's earlier time levels found
    (extrapolation_weights).  Only ever the START VECTOR of a linear solve that
    is then converged to the same tolerance as from zero.  Returns False (dx
    untouched) without a usable history.'''
    st = lay._dev.get('start_vector_state')
    if st is None or (st.current is None and not st.unresolved):
        return False
    tr = _resolve(st)
    hist = []
    for h in tr.entries(key):
        if h[2] >= tr.level:
            # this time level's earlier attempt (the call repeats the step: a
            # Banach sweep with another forcing, a redo): what it found is the
            # best start there is -- nearly the same system
            if tr.redo and REDO_START == 'previous_attempt' \
                    and h[0].numel() == dx.numel() \
                    and (1.0 / 1.5 <= dt / h[1] <= 1.5):
                ops.copy(dx, h[0])
                if dt != h[1]:
                    ops.axpby(0.0, dx, (dt / h[1])**power, dx)
                return True
            continue
        # only while the step size is settled: through the start-up ramp of a
        # controller that doubles dt the rates are not smooth in time, and a
        # start vector FAR from the solution costs a Krylov solve iterations
        # and attainable accuracy (CG's recurrence residual drifts from the
        # true one in proportion to the largest residual it has seen)
        if h[0].numel() != dx.numel() or not (1.0 / 1.5 <= dt / h[1] <= 1.5) \
                or len(hist) >= min(points, KEEP_POINTS):
            break
        # ... and without a gap in the time levels
        if h[2] != tr.level - 1 - len(hist):
            break
        hist.append(h)
    if not hist:
        return False
    w = extrapolation_weights([h[1] for h in hist], dt, power, degree)
    n = dx.numel()
    k = len(hist)
    coef = (ctypes.c_double * k)(*w)
    ptrs = (ctypes.c_void_p * k)(*[_hip.f64(h[0], n).value for h in hist])
    _hip.check(_hip.lib().flow_lincomb(n, k, coef, ptrs, _hip.f64(dx, n),
                                       _hip.stream()))
    return True


def remember_increment(lay, dt, dx, key='newton_increments'):
    '''The increment this call found under `key`, filed under the time level
    of its trajectory; a repeated step replaces its first attempt's.'''
    st = lay._dev.get('start_vector_state')
    if st is None or (st.current is None and not st.unresolved):
        return
    tr = _resolve(st)
    hist = tr.entries(key)
    hist[:] = [h for h in hist if h[0].numel() == dx.numel()]
    if hist and hist[0][2] == tr.level:
        ops.copy(hist[0][0], dx)
        hist[0] = (hist[0][0], dt, tr.level)
        return
    if len(hist) >= KEEP_POINTS + 1:
        keep = hist.pop()[0]          # (re-use the oldest buffer)
        ops.copy(keep, dx)
    else:
        keep = _hip.clone(dx)
    hist.insert(0, (keep, dt, tr.level))


def newton_key(it):
    '''Key of the increment history of Newton iteration `it`: the first three
    iterations of a time loop's calls have one each (a developed vortex street
    takes two iterations per step: ||F|| after the first is 1.3-2.9e-10 there);
    the later ones are only written by the calls that get that far -- hence
    the time levels on the entries.'''
    if it > 2:
        return None
    return 'newton_increments' if it == 0 else ('newton_increments', it)
