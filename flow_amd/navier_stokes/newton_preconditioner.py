# -*- coding: utf-8 -*-
'''
Life cycle of the lagged preconditioners of the Newton systems of the tentative
velocity (flow_amd/navier_stokes/pressure_correction.py; the reference solves
these systems with a sparse LU, flow/navier_stokes/pressure_correction.py:
224-254): when a lagged preconditioner is rebuilt (`age`), the acceptance test
of the p-multigrid cycle on one GPU and on the strips (`contraction`,
`contraction_on_strips`), and the assembly of its P1 coarse level
(`coarse_jacobian`).
'''
import ctypes

import numpy

from ..fem import ops
from .. import _hip
from .. import device
from .. import parallel


def age(pre, kind, rebuilt, its, applications, npar, it=0):
    """When is a lagged preconditioner rebuilt?  The ILU(0): when a solve needs
    more than twice the (BiCGStab-equivalent) iterations the fresh factors
    needed.  The p-multigrid cycle ages gently (13 -> 16 applications over 200
    plateau steps of the 10 M-DoF run) and a rebuild costs most of a time step
    (Jacobian assembly, packing, 2 x 10 power-method products from the
    previous rebuild's iterate: ~5 ms): it is
    rebuilt when the solves need 2 applications (or 15 %) more than the best
    one since the rebuild -- a smoothed count, and not within `pmg_min_solves`
    of the rebuild: the counts also move with the quality of the start vectors
    and with the Newton iteration the solve belongs to (each has its own
    yardstick) -- and in any case after `pmg_refresh` solves where the solves are long
    (`pmg_refresh_min` applications on average since the rebuild: a rebuild
    costs as much as ten applications, and where a solve takes four there is
    little a fresh cycle could save): the linearisation state the cycle was
    built on drifts away without any single solve showing it.  Developed
    vortex street (tools/developed_flow.py): 17.8 applications per step with a
    cycle built on the symmetric start-up flow 2000 steps earlier, 15.4 when
    it is rebuilt every 50 solves; 23.5 -> 21.5 ms per step.  (Tried and not
    kept as the trigger: the relative change of dt u since the build -- it
    grows fastest on the early plateau, where the counts do not move at
    all.)"""
    pre.uses = 0 if rebuilt else getattr(pre, 'uses', 0) + 1
    if rebuilt:
        pre.base_its = max(its, npar['check_every'])
        pre.base_applications = {}
    if kind == 'pmg':
        # (the yardstick of Newton iteration `it`: its first solve with the
        # fresh cycle -- the second iteration's systems are shorter solves)
        base = getattr(pre, 'base_applications', None)
        if not isinstance(base, dict):
            base = pre.base_applications = {}
        # [fewest applications since the rebuild, smoothed count]: one long
        # solve (a poor start vector, a step-size jump) is not ageing
        best, smooth = base.get(it, (applications, float(applications)))
        best = min(best, applications)
        smooth = 0.6 * smooth + 0.4 * applications
        base[it] = (best, smooth)
        pre.work = (0 if rebuilt else getattr(pre, 'work', 0)) + applications
        refresh = int(npar.get('pmg_refresh', 50))
        if refresh and pre.uses >= refresh and \
                pre.work >= float(npar.get('pmg_refresh_min', 6.0)) * (
                    pre.uses + 1):
            pre.stale = True
        elif pre.uses >= int(npar.get('pmg_min_solves', 3)) and \
                smooth >= best + max(2, int(round(0.15 * best))) - 0.25:
            pre.stale = True
    elif not rebuilt and its > 2 * pre.base_its:
        pre.stale = True


def contraction(pre, operator, lay, bc_dofs_host, sweeps=1):
    """|v - M^-1 J v| / |v| for the preconditioner `pre` of the operator J on a
    fixed full-spectrum vector v that vanishes on the Dirichlet dofs (as the
    Krylov vectors of the Newton systems do) -- and, for sweeps > 1, the same
    ratio for the next applications of the error propagation (power_probe):
    < 1 where one application is a convergent iteration, NaN / > 1 where it
    amplifies."""
    n2 = 2 * lay.N
    hold = lay._dev.setdefault('pmg_probe', {})
    bkey = hash(bc_dofs_host.tobytes())
    if hold.get('key') != bkey:
        v = numpy.random.RandomState(7).standard_normal(n2)
        v[bc_dofs_host] = 0.0
        hold.update(key=bkey, v=device.to_device(v), w=device.empty(n2),
                    z=device.empty(n2))
    v, w, z = hold['v'], hold['w'], hold['z']
    return power_probe(operator.apply, pre.apply, v, w, z, sweeps)


def choose_cycle(cycle, bare, operator, lay, bc_dofs_host, npar):
    """The two-level ILU cycle (flow_amd/fem/tlilu.py) or its fine smoother
    alone, for the Newton systems: `rate_verdict` on the fixed probe vector of
    `contraction`.  -> (use the cycle?, (contraction per application of the
    cycle, of the smoother, ms per application of the cycle, of the
    smoother))."""
    select = npar.get('tl_select', 'rate')
    if select != 'rate':
        return select == 'cycle', (float('nan'), float('nan'), 0.0, 0.0)
    contraction(bare, operator, lay, bc_dofs_host)      # (builds the vectors)
    hold = lay._dev['pmg_probe']
    return rate_verdict(operator.apply, cycle.apply, bare.apply, hold['v'],
                        hold['w'], hold['z'],
                        smooth=int(npar.get('tl_probe_smooth', 6)),
                        sweeps=int(npar.get('tl_probe_sweeps', 4)),
                        accept=float(npar.get('tl_accept', 0.9)))


def rate_verdict(apply_operator, apply_cycle, apply_bare, v, w, z, smooth=6,
                 sweeps=4, accept=0.9):
    """Which of two preconditioners converges faster PER UNIT OF TIME on the
    part of the spectrum that decides a solve?  Both are run as stationary
    iterations e <- (I - M^-1 A) e for `sweeps` applications, timed, from the
    SAME start: the probe vector v after `smooth` applications of the bare
    smoother's error propagation -- what is left then is what the smoother is
    slow on, which is what the Krylov iteration spends its time on (from the
    raw random vector the two read 0.54 / 0.76 after six sweeps in a regime
    where GMRES needs 100 applications with the cycle and 400 without: the
    high frequencies, which both treat alike, still dominate).  The
    contraction per application is the geometric mean over the sweeps; the
    better -log(contraction) per millisecond wins.  Time, not launches or
    bytes: which of them bounds an application changes with the size of the
    problem.  (The verdict can flip from run to run where the two are within a
    few per cent of each other, where it does not matter; a converged solve
    does not depend on it.)  The cycle is only ever taken where it CONTRACTS
    (per application < `accept`): on an under-resolved mesh -- cell Peclet
    ~12 on the P2 level, twice that on the rediscretised P1 level -- neither
    iteration contracts (0.99 / 1.9 per application, then 2.8 / 8.0), the
    coarse correction buys nothing (66 + 58 + 35 GMRES applications with it,
    64 + 56 + 37 without) and at the next step GMRES stalls on the cycle where
    the bare sweeps, outliers and all, converge
    (tests/test_large_parity.py::test_twelve_steps...).
    -> (cycle wins?, (c_cycle, c_bare, ms_cycle, ms_bare) per application)."""
    import time
    sweeps = max(1, int(sweeps))
    power_probe(apply_operator, apply_bare, v, w, z, sweeps=max(1, int(smooth)))
    start = device.empty(v.numel())
    ops.copy(start, z)
    den = ops.vector_norm(start)
    if not (den > 0.0 and numpy.isfinite(den)):
        # (the smoother alone has blown up or annihilated the probe)
        ops.copy(start, v)
        den = ops.vector_norm(start)
    out = []
    for apply_pre in (apply_cycle, apply_bare):
        power_probe(apply_operator, apply_pre, start, w, z, sweeps=1)  # (untimed)
        device.synchronize()
        t0 = time.time()
        power_probe(apply_operator, apply_pre, start, w, z, sweeps=sweeps)
        num = ops.vector_norm(z)
        device.synchronize()
        ms = 1.0e3 * (time.time() - t0) / sweeps
        c = (num / den)**(1.0 / sweeps) if numpy.isfinite(num) else float('inf')
        out.append((c, ms))
    (c_tl, t_tl), (c_b, t_b) = out

    def rate(c, t):
        return -numpy.log(c) / t if 0.0 < c < 1.0 else -float(c)
    if c_tl == 0.0:
        return True, (c_tl, c_b, t_tl, t_b)
    use = c_tl < accept and rate(c_tl, t_tl) > rate(c_b, t_b)
    return bool(use), (c_tl, c_b, t_tl, t_b)


def power_probe(apply_operator, apply_preconditioner, v, w, z, sweeps=3):
    """max_k |E^k v| / |E^(k-1) v|, k = 1 .. sweeps, E = I - M^-1 A: the
    first ratio is the contraction of one application on the full-spectrum
    vector v; the later ones home in on the dominant eigenvalue of E -- a cycle
    that amplifies only LOCALLY (a plume, a shear layer at CFL >> 1: a handful
    of modes a random vector carries little of) shows up there, where the
    first ratio still reads 0.15.  (Not the acceptance test by default: a
    rediscretised coarse level leaves E a few eigenvalues of modulus 2-3 even
    where the cycle is an excellent GMRES preconditioner -- outliers of M^-1 A
    far from 0 cost an iteration each, no more; measured: 2.8 on the Newton
    system of tests/test_pmg.py, which GMRES solves in 14 applications.  The
    solvers cap the iterations of a solve with the cycle instead and hand a
    stalled one to the ILU(0).)  z holds the last E^k v on return."""
    worst = 0.0
    cur = v
    den = ops.vector_norm(v)
    for _ in range(max(1, int(sweeps))):
        apply_operator(cur, w)
        apply_preconditioner(w, z)
        ops.axpby(1.0, cur, -1.0, z)          # z = cur - M^-1 A cur
        num = ops.vector_norm(z)
        ratio = num / den if den > 0.0 else float('inf')
        if not numpy.isfinite(ratio):
            return float('inf')
        worst = max(worst, ratio)
        if cur is v:
            cur = device.empty(v.numel())
        ops.copy(cur, z)
        den = num
    return worst


def contraction_on_strips(pre, Jop, lay, bc_dofs_host):
    """`contraction` for the block preconditioner of the calling rank: the same
    probe vector on every rank (seeded: its ghost rows need no exchange), the
    Jacobian action on the rank's rows, the rank's cycle on them, and
    |v - M^-1 J v| / |v| summed over the ranks -- every rank gets the same
    number and takes the same decision."""
    import torch
    n = lay.N
    v2 = parallel.view(lay)
    r0, r1 = v2.r0, v2.r1
    hold = lay._dev.setdefault('pmg_probe_strip', {})
    bkey = (hash(bc_dofs_host.tobytes()), r0, r1)
    if hold.get('key') != bkey:
        v = numpy.random.RandomState(7).standard_normal(2 * n)
        v[bc_dofs_host] = 0.0
        own = numpy.concatenate([v[r0:r1], v[n + r0:n + r1]])
        hold.update(key=bkey, v=device.to_device(v), w=device.empty(2 * n),
                    vo=device.to_device(own), zo=device.empty(2 * (r1 - r0)))
    v, w, vo, zo = hold['v'], hold['w'], hold['vo'], hold['zo']
    Jop.apply(v, w)
    wo = torch.cat([w[r0:r1], w[n + r0:n + r1]]).contiguous()
    pre.apply(wo, zo)
    ops.axpby(1.0, vo, -1.0, zo)
    sums = torch.stack([(zo * zo).sum(), (vo * vo).sum()])
    parallel.comm().allreduce_tensor(sums)
    num, den = [float(x) for x in device.to_host(sums)]
    ratio = numpy.sqrt(num / den) if den > 0.0 else float('inf')
    return ratio if numpy.isfinite(ratio) else float('inf')


def mass_share(pre, J1):
    """A lower bound for the share of the mass term in the diagonal of the P1
    Jacobian `J1` (kind 2: the diagonal blocks are planes 0 and 3) over its
    free rows, min_i M_ii / (J1)_ii: what sizes the Chebyshev treatment of the
    coarse level (Pmg.refactor, `coarse_auto`).  At CFL-sized steps the level
    is mass-dominated (share ~ 0.5, a handful of steps); at dt >> h^2 / nu --
    the Boussinesq driver at dt = 1 -- it is a diffusion problem (share ~
    1e-2) and wants ~ sqrt(1 / share) of them."""
    lay1 = pre.lay1
    hold = lay1._dev.setdefault('pmg_mass_diag', {})
    if 'm' not in hold:
        M1 = ops.assemble_scalar_matrix(lay1, ops.MASS)
        hold['m'] = M1.vals[lay1.dev('diag_idx').long()].clone()
    m = hold['m']
    di = lay1.dev('diag_idx').long()
    share = None
    for plane in (0, 3):
        d = J1.plane(plane)[di]
        s = m / d
        # (Dirichlet rows are identity rows: d = 1, tiny share -- not counted)
        s = s[d != 1.0]
        if s.numel():
            v = float(device.to_host(s.min()))
            share = v if share is None else min(share, v)
    return share if share is not None and share > 0.0 else None


def coarse_jacobian(pre, W, P, ui, p0, f0, f1, prm, bfmask, bc_dofs_host,
                     mesh_s=None, space1_s=None, pspace_s=None):
    """The Jacobian of the P1 discretisation of the same Newton system at the
    vertex values of `ui` (the coarse level of flow_amd/fem/pmg.py): assembled
    by the P1 instance of the momentum kernel (the P1-P1 element pair of
    BASELINE config 2 runs through it), Dirichlet rows -> identity rows.
    mesh_s / space1_s / pspace_s: a rank's views of the mesh, the P1 velocity
    space and the pressure space (the strips of flow_amd.parallel: its cells,
    its rows)."""
    lib = _hip.lib()
    mesh = W.mesh()
    lay, lay1 = W.layout, pre.lay1
    n, n1 = lay.N, lay1.N
    nc = mesh.num_cells()
    st = _hip.stream()
    hold = lay._dev.setdefault('pmg_coarse', {})
    if 'J1' not in hold:
        hold['J1'] = ops.Matrix(lay1, 2)
        hold['ui1'] = device.empty(2 * n1)
        hold['vd'] = device.to_device(lay.vertex_dofs.astype(numpy.int32))
    J1, ui1 = hold['J1'], hold['ui1']
    _hip.check(lib.flow_gather_rows(
        2, _hip.i32(hold['vd'], n1), n1, _hip.f64(ui.data, 2 * n), n,
        _hip.f64(ui1, 2 * n1), n1, st))
    f0s, keep0 = ops.coef_struct(f0, mesh, 1)
    f1s, keep1 = ops.coef_struct(f1, mesh, 1)
    s1 = space1_s if space1_s is not None else ops.space_struct(lay1)
    buf = ops.scratch(mesh, 4 * lay1.nloc**2 * nc)
    _hip.check(lib.flow_assemble_momentum(
        ctypes.byref(mesh_s if mesh_s is not None else ops.mesh_struct(mesh)),
        ctypes.byref(s1),
        ctypes.byref(pspace_s if pspace_s is not None
                     else ops.space_struct(P.layout)),
        _hip.i32(bfmask, nc, 'bfmask'), _hip.f64(ui1, 2 * n1),
        _hip.f64(ui1, 2 * n1), _hip.f64(p0.data, P.size()),
        ctypes.byref(f0s), ctypes.byref(f1s), ctypes.byref(prm),
        _hip.f64(buf), None, _hip.f64(J1.vals, 4 * J1.stride), J1.stride, st
        ))
    del keep0, keep1
    bc1_host, bc1 = pre.set_bcs(bc_dofs_host)
    if len(bc1_host):
        _hip.check(lib.flow_bc_identity_rows(
            ctypes.byref(J1.operator()), _hip.f64(J1.vals),
            _hip.i32(lay1.dev('diag_idx')), len(bc1_host), _hip.i32(bc1), st
            ))
    return J1
