# -*- coding: utf-8 -*-
'''
Host side of the mass-matrix solver (`flow_mass` in include/flow_hip.h, kernels
in flow_amd/csrc/mass_kernels.hip): mixed-precision defect correction with a
fixed Chebyshev polynomial of the Jacobi-scaled mass matrix as the approximate
inverse.  Stands in for `solve(a3 == L3, u1, bcs)` with CG + hypre_amg of the
reference's velocity correction (flow/navier_stokes/pressure_correction.py:
451-464) and for the `project(...)` of its drivers' step-size control
(tests/test_karman_vortex_street.py:262-267).

Nothing is estimated at run time: the spectrum of D^-1 M is bounded a priori
by the extreme eigenvalues of the diagonally scaled ELEMENT mass matrix
(Wathen, IMA J. Numer. Anal. 7 (1987)): triangles, P1: [1/2, 2]; P2:
[0.3924, 2.0598] -- on any mesh, with or without identity rows (eigenvalue 1).
'''
import ctypes

import torch

from . import ops
from .space import csr_stream_rowblocks
from .. import _hip
from .. import device

# (lam_min, lam_max) of D^-1 M per Lagrange degree on triangles
WATHEN = {1: (0.5, 2.0), 2: (0.39237, 2.05982)}
# the interval is padded for the rounding of the entries to fp16
_PAD = (0.98, 1.01)
# |p(A~)| |A~ - A| for entries rounded to fp16 (relative 2^-11, row sums of
# |D^-1 M| <= 3.5, |p| <= 1.02 / lam_min): the part of |I - B M| the polynomial
# cannot see
_FP16_TERM = 4.5e-3


def chebyshev_contraction(lo, hi, steps):
    '''max |1 - lam p(lam)| on [lo, hi] after `steps` Chebyshev steps.'''
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    s = theta / delta
    sigma = 1.0 / (s + (s * s - 1.0)**0.5)
    return 2.0 * sigma**steps / (1.0 + sigma**(2 * steps))


class MassSolver(object):
    '''Defect-correction solver for a mass Matrix `A` (ops.Matrix of kind 0, or
    kind 4: one plane for both components with identity rows by mask).
    `dinv`: 1 / diagonal (1 on masked rows), A.size doubles.'''

    def __init__(self, A, dinv, steps=6, safety=1.5, packed=True):
        assert A.kind in (0, 4)
        lay = A.layout
        self.A = A
        self.dinv = dinv
        self.ncomp = 2 if A.kind == 4 else 1
        n, nnz = lay.N, lay.nnz
        if lay.degree not in WATHEN:
            raise ValueError(
                'MassSolver: no spectral bounds for Lagrange degree %r on '
                'triangles (known: %s); use the Jacobi-CG (ops.krylov_solve)'
                % (lay.degree, sorted(WATHEN)))
        lo, hi = WATHEN[lay.degree]
        lo, hi = _PAD[0] * lo, _PAD[1] * hi
        # (readable three halves past nnz: quads of nonzeros per lane)
        self.vals16 = torch.zeros(nnz + 8, dtype=torch.float16,
                                  device=device.get())
        assert self.vals16.data_ptr() % 16 == 0
        _hip.check(_hip.lib().flow_mass_pack(
            n, _hip.i32(lay.dev('rowptr'), n + 1, 'rowptr'),
            _hip.i32(lay.dev('diag_idx'), n, 'diag_idx'),
            _hip.f64(A.vals, nnz, 'vals'), _hip.f16(self.vals16, nnz),
            _hip.stream()))
        if 'pmg_rowblocks' not in lay._dev:
            lay._dev['pmg_rowblocks'] = device.to_device(csr_stream_rowblocks(
                lay.pattern('rowptr'), nnz_per_block=_hip.PMG_NNZ_PER_BLOCK))
        self.rowblocks16 = lay._dev['pmg_rowblocks']
        self.work16 = torch.zeros(5 * self.ncomp * n + 4, dtype=torch.float32,
                                  device=device.get())
        self.contraction = min(1.0, safety * (
            chebyshev_contraction(lo, hi, steps) + _FP16_TERM))
        s = _hip.MassS()
        s.A = ctypes.pointer(A.operator())
        s.dinv = _hip.f64(dinv, A.size, 'dinv').value
        s.nblocks16 = self.rowblocks16.numel() - 1
        s.rowblocks16 = _hip.i32(self.rowblocks16).value
        s.vals16 = _hip.f16(self.vals16, nnz).value
        s.lam_min, s.lam_max = lo, hi
        s.steps = int(steps)
        s.contraction = self.contraction
        s.work16 = _hip.f32(self.work16, 5 * self.ncomp * n).value
        # (cols is read in quads too)
        assert lay.dev('cols').numel() >= nnz + 4
        # the packed stream (4 B per nonzero): wherever every tile's columns
        # span < 65536 -- any mesh numbered along one axis
        self.packed16 = None
        if packed:
            nb = s.nblocks16
            pk = torch.zeros(nnz + 8, dtype=torch.int32, device=device.get())
            cb = torch.zeros(nb, dtype=torch.int32, device=device.get())
            flag = torch.zeros(1, dtype=torch.int32, device=device.get())
            assert pk.data_ptr() % 16 == 0
            _hip.check(_hip.lib().flow_mass_pack16(
                n, nb, _hip.i32(self.rowblocks16),
                _hip.i32(lay.dev('rowptr'), n + 1),
                _hip.i32(lay.dev('cols'), nnz), _hip.i32(lay.dev('diag_idx'), n),
                _hip.f64(A.vals, nnz), _hip.i32(cb, nb), _hip.i32(pk, nnz),
                _hip.i32(flag, 1), _hip.stream()))
            if int(device.to_host(flag).item()) == 0:
                self.packed16 = (pk, cb)
                s.packed16 = _hip.i32(pk, nnz).value
                s.cbase16 = _hip.i32(cb, nb).value
        self.struct = s
        self.history = {}

    @classmethod
    def cached(cls, A, dinv, **kw):
        # (the fp16 copy is made from A.vals as they are now: a matrix whose
        # values are replaced or rewritten in place gets a new solver)
        key = ('mass_solver', tuple(sorted(kw.items())), A.vals.data_ptr(),
               getattr(A.vals, '_version', 0))
        held = A.__dict__.setdefault('_mass_solvers', {})
        for old in [k for k in held if k[:2] == key[:2] and k != key]:
            del held[old]
        if key not in held:
            held[key] = cls(A, dinv, **kw)
        return held[key]

    # The stopping test rests on an a-priori contraction; the device watches
    # the observed one (mass_scalar_kernel) and reports a violation -- or a
    # NaN, or maxit -- as non-convergence.  guard: fall back to the Jacobi-CG
    # (flow_cg_solve: PETSc's test on the preconditioned residual) with a
    # message instead of failing the caller's step.
    # (Running into the caller's `maxit` stays an error like the Krylov
    # solvers', and dolfin's 'error_on_nonconvergence'.)
    guard = True

    def _bound_failed(self, err):
        msg = str(err)
        return self.guard and ('contracted by' in msg or 'broke down' in msg)

    def _fallback(self, err, b, x, rtol, atol):
        from ..message import info
        info('mass solve: %s -- falling back to Jacobi-CG' % err)
        self.fallbacks = getattr(self, 'fallbacks', 0) + 1
        if self.ncomp == 2:
            # (the identity-row form needs the boundary values in x on the
            # masked rows: they are what b carries there)
            mask = self.A.rowmask == 0
            x[mask] = b[mask]
        sol = ops.krylov_solve('cg', self.A, b, x, rtol=rtol, atol=atol,
                               maxit=10000, dinv=self.dinv, check_every=5)
        sol.method = 'defect correction -> ' + sol.method
        return sol

    def solve_increment(self, g, xbase, x, rtol, atol=0.0, maxit=50, tag=None,
                        first_check=0, delta0=None):
        '''x = xbase + increment, M increment = g (g = b - M xbase handed in by
        the caller: no product with M to form it, and fp64 only has to resolve
        the increment); x may be xbase.'''
        n = self.A.size
        head = _hip.REDUCE_WORK + 2 * self.struct.nblocks16
        wk = ops.work(head + 2 + n)
        if first_check == 0 and tag is not None and tag in self.history:
            first_check = self.history[tag]
        its = ctypes.c_int(0)
        res = ctypes.c_double(0.0)
        base = _hip.clone(xbase) if x.data_ptr() == xbase.data_ptr() \
            and self.guard else xbase
        try:
            _hip.check(_hip.lib().flow_mass_solve_increment(
                ctypes.byref(self.struct), _hip.f64(g, n, 'g'),
                _hip.f64(xbase, n, 'xbase'),
                _hip.f64(delta0, n, 'delta0') if delta0 is not None else None,
                _hip.f64(x, n, 'x'),
                float(rtol), float(atol), int(maxit), int(first_check),
                _hip.f64(wk), wk.numel(), ctypes.byref(its), ctypes.byref(res),
                _hip.stream()))
        except _hip.NotConverged as err:
            if not self._bound_failed(err):
                raise
            # M increment = g by Jacobi-CG from zero, x = xbase + increment
            delta = _hip.fill(device.empty(n), 0.0)
            sol = self._fallback(err, g, delta, rtol, atol)
            ops.copy(x, base)
            ops.axpby(1.0, delta, 1.0, x)
            return sol
        if tag is not None:
            self.history[tag] = its.value
        return ops.SolveInfo(its.value, res.value,
                             'defect correction (increment) + chebyshev%d/fp16'
                             % self.struct.steps)

    def solve(self, b, x, rtol, atol=0.0, maxit=50, tag=None, first_check=0):
        '''x holds the initial guess; raises _hip.NotConverged like the Krylov
        solvers.  tag: the solve recurs in a time loop under that name -- as
        many corrections as the previous call needed are enqueued before the
        first read-back.'''
        n = self.A.size
        wk = ops.work(_hip.REDUCE_WORK + 2 * self.struct.nblocks16)
        if first_check == 0 and tag is not None and tag in self.history:
            first_check = self.history[tag]
        its = ctypes.c_int(0)
        res = ctypes.c_double(0.0)
        try:
            _hip.check(_hip.lib().flow_mass_solve(
                ctypes.byref(self.struct), _hip.f64(b, n, 'b'),
                _hip.f64(x, n, 'x'),
                float(rtol), float(atol), int(maxit), int(first_check),
                _hip.f64(wk), wk.numel(), ctypes.byref(its), ctypes.byref(res),
                _hip.stream()))
        except _hip.NotConverged as err:
            if not self._bound_failed(err):
                raise
            # (x holds the last iterate: a start vector like any other -- the
            # CG below only ever moves it on the free rows)
            if not bool(torch.isfinite(x).all()):
                _hip.fill(x, 0.0)
            return self._fallback(err, b, x, rtol, atol)
        if tag is not None:
            self.history[tag] = its.value
        return ops.SolveInfo(its.value, res.value,
                             'defect correction + chebyshev%d/fp16'
                             % self.struct.steps)
