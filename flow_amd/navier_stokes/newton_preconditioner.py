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


def contraction(pre, operator, lay, bc_dofs_host):
    """|v - M^-1 J v| / |v| for the preconditioner `pre` of the operator J on a
    fixed full-spectrum vector v that vanishes on the Dirichlet dofs (as the
    Krylov vectors of the Newton systems do): < 1 where one application is a
    convergent iteration, NaN / > 1 where it amplifies."""
    n2 = 2 * lay.N
    hold = lay._dev.setdefault('pmg_probe', {})
    bkey = hash(bc_dofs_host.tobytes())
    if hold.get('key') != bkey:
        v = numpy.random.RandomState(7).standard_normal(n2)
        v[bc_dofs_host] = 0.0
        hold.update(key=bkey, v=device.to_device(v), w=device.empty(n2),
                    z=device.empty(n2))
    v, w, z = hold['v'], hold['w'], hold['z']
    operator.apply(v, w)
    pre.apply(w, z)
    ops.axpby(1.0, v, -1.0, z)
    num = ops.vector_norm(z)
    den = ops.vector_norm(v)
    ratio = num / den
    return ratio if numpy.isfinite(ratio) else float('inf')


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
