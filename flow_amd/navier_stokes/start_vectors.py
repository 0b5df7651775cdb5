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
'''
import ctypes

from ..fem import ops
from .. import _hip


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
                            power=1, degree=None, stamp=None):
    '''dx <- the increment this call is likely to find, extrapolated in time
    from the increments of the previous calls (extrapolation_weights).  Only
    ever the START VECTOR of a linear solve that is then converged to the same
    tolerance as from zero.  Returns False (dx untouched) without a
    history.  stamp: the number of this call -- only entries remembered under
    the stamps just before it, without a gap, count (a history that is not
    written on every call: the second Newton iteration's).'''
    hist = []
    for h in lay._dev.get(key, []):
        # only while the step size is settled: through the start-up ramp of a
        # controller that doubles dt the rates are not smooth in time, and a
        # start vector FAR from the solution costs a Krylov solve iterations
        # and attainable accuracy (CG's recurrence residual drifts from the
        # true one in proportion to the largest residual it has seen)
        if h[0].numel() != dx.numel() or not (1.0 / 1.5 <= dt / h[1] <= 1.5) \
                or len(hist) >= min(points, 6):
            break
        if stamp is not None and (len(h) < 3 or h[2] != stamp - 1 - len(hist)):
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


def remember_increment(lay, dt, dx, keep_points=6, key='newton_increments',
                        stamp=None):
    hist = lay._dev.setdefault(key, [])
    hist[:] = [h for h in hist if h[0].numel() == dx.numel()]
    if len(hist) >= keep_points:
        keep = hist.pop()[0]          # (re-use the oldest buffer)
        ops.copy(keep, dx)
    else:
        keep = _hip.clone(dx)
    hist.insert(0, (keep, dt, stamp))


def newton_history(lay, it):
    '''(key, stamp) of the increment history of Newton iteration `it` of this
    call: the first iteration of a time loop's calls has its own since round
    4; so have the second and the third (a developed vortex street takes two
    iterations per step: ||F|| after the first is 1.3-2.9e-10 there) -- those
    are only written on the calls that get that far, hence the stamps.'''
    if it > 2:
        return None, None
    key = 'newton_increments' if it == 0 else ('newton_increments', it)
    return key, lay._dev.get('newton_call', 0)
