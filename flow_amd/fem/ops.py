# -*- coding: utf-8 -*-
'''
Host-side operations that run on the HIP path: matrices over a space's CSR
pattern, Krylov solves, norms, and the callers' utilities `project`,
`interpolate`, `errornorm`, `norm` (tests/test_navier_stokes.py:296-308, 333;
tests/test_karman_vortex_street.py:262-268; tests/test_boussinesq.py:85).

Everything numerical is a call into libflow_hip.so (flow_amd/_hip.py); torch
only owns the HBM buffers.
'''
import ctypes

import numpy
import torch

from . import reference
from .function import (
    Function, Expression, Constant, as_cell_coefficient, cell_lattice_points,
    )
from .space import mesh_geometry_dev
from .. import _hip
from .. import device


# -- C structs ----------------------------------------------------------------
def mesh_struct(mesh):
    if 'mesh_struct' not in mesh._cache:
        xy = mesh_geometry_dev(mesh)
        s = _hip.MeshS(mesh.num_cells(), _hip.f64(xy, 6 * mesh.num_cells(), 'xy'))
        mesh._cache['mesh_struct'] = (s, xy)
    return mesh._cache['mesh_struct'][0]


def space_struct(layout):
    if 'struct' not in layout._dev:
        nc = layout.mesh.num_cells()
        cd = layout.dev('cell_dofs')
        s = _hip.SpaceS(
            layout.degree, layout.N, layout.nnz,
            _hip.i32(cd, layout.nloc * nc, 'cell_dofs'),
            _hip.i32(layout.dev('cptr'), layout.nnz + 1, 'cptr'),
            _hip.i32(layout.dev('csrc'), layout.nloc**2 * nc, 'csrc'),
            _hip.i32(layout.dev('vptr'), layout.N + 1, 'vptr'),
            _hip.i32(layout.dev('vsrc'), layout.nloc * nc, 'vsrc'),
            )
        layout._dev['struct'] = s
    return layout._dev['struct']


_G_CACHE = {}


def coef_struct(coef, mesh, test_degree):
    '''flow_coef of a CellCoefficient for test functions of `test_degree`.
    Returns (struct, keepalive).'''
    key = (coef.k, test_degree, str(device.get()))
    if key not in _G_CACHE:
        _G_CACHE[key] = device.to_device(
            reference.source_matrix(coef.k, test_degree)
            )
    G = _G_CACHE[key]
    nce = mesh.num_cells() if coef.cell_stride else 1
    s = _hip.CoefS(
        coef.nl, coef.cell_stride,
        _hip.f64(coef.values, coef.dim * coef.nl * nce, 'coefficient values'),
        _hip.f64(G, coef.nl * reference.nloc(test_degree), 'G'),
        )
    return s, (coef.values, G)


def scratch(mesh, ndoubles):
    '''Per-mesh scratch buffer for the cell kernels (grown on demand).'''
    buf = mesh._cache.get('scratch')
    if buf is None or buf.numel() < ndoubles:
        mesh._cache['scratch'] = None
        buf = device.empty(ndoubles)
        mesh._cache['scratch'] = buf
    if device._POISON:      # debugging aid: stale scratch reads become NaNs
        _hip.fill(buf, float('nan'))
    return buf


_WORK = {}


def work(ndoubles):
    key = str(device.get())
    buf = _WORK.get(key)
    if buf is None or buf.numel() < ndoubles:
        _WORK[key] = None
        buf = device.empty(ndoubles)
        _WORK[key] = buf
    return buf


# -- matrices -----------------------------------------------------------------
def plane_stride(layout):
    '''Distance between value planes: nnz rounded up to even, so that every
    plane is 16-byte aligned and readable one entry past nnz (the SpMV loads
    value pairs; include/flow_hip.h, flow_operator).'''
    return layout.nnz + (layout.nnz & 1)


def value_plane(layout):
    return device.zeros(plane_stride(layout))


class Matrix(object):
    '''Value planes over the CSR pattern of a scalar layout.
    kind 0: scalar (1 plane), 1: block-diagonal (2 planes), 2: 2x2 (4 planes),
    4: one plane for both components of a vector field with identity rows
    where `rowmask` (uint8, 2N) is 0 (include/flow_hip.h, flow_operator);
    `vals` is one tensor of nplanes*stride doubles, stride = plane_stride().'''
    NPLANES = {0: 1, 1: 2, 2: 4, 4: 1}

    def __init__(self, layout, kind, vals=None, rowmask=None):
        self.layout = layout
        self.kind = kind
        assert (rowmask is not None) == (kind == 4)
        if rowmask is not None:
            assert rowmask.dtype == torch.uint8 and \
                rowmask.numel() == 2 * layout.N
        self.rowmask = rowmask
        self.nplanes = self.NPLANES[kind]
        nnz = layout.nnz
        self.stride = plane_stride(layout)
        if vals is None:
            vals = device.zeros(self.nplanes * self.stride)
        elif vals.numel() == self.nplanes * nnz and self.stride != nnz:
            # compact planes -> aligned planes
            packed = device.zeros(self.nplanes * self.stride)
            for p in range(self.nplanes):
                packed[p * self.stride:p * self.stride + nnz] = \
                    vals[p * nnz:(p + 1) * nnz]
            vals = packed
        assert vals.numel() == self.nplanes * self.stride
        assert vals.data_ptr() % 16 == 0
        self.vals = vals
        self._op = None
        return

    @property
    def size(self):
        return self.layout.N * (1 if self.kind == 0 else 2)

    def plane(self, p):
        return self.vals[p * self.stride:p * self.stride + self.layout.nnz]

    def operator(self):
        if self._op is None:
            lay = self.layout
            op = _hip.Operator()
            op.kind = self.kind
            op.n = lay.N
            op.nnz = lay.nnz
            rb = lay.dev('rowblocks2' if self.kind in (2, 4) else 'rowblocks')
            op.nblocks = rb.numel() - 1
            op.rowptr = _hip.i32(lay.dev('rowptr'), lay.N + 1, 'rowptr')
            op.cols = _hip.i32(lay.dev('cols'), lay.nnz, 'cols')
            op.rowblocks = _hip.i32(rb, None, 'rowblocks')
            base = _hip.f64(self.vals, self.nplanes * self.stride, 'vals').value
            for p in range(self.nplanes):
                op.vals[p] = base + 8 * p * self.stride
            if self.rowmask is not None:
                op.rowmask = _hip.u8(self.rowmask, 2 * lay.N, 'rowmask').value
            self._op = op
        return self._op

    def apply(self, x, y):
        lib = _hip.lib()
        assert x.data_ptr() != y.data_ptr()
        _hip.check(lib.flow_operator_apply(
            ctypes.byref(self.operator()), _hip.f64(x, self.size, 'x'),
            _hip.f64(y, self.size, 'y'), _hip.stream()
            ))
        return y

    def diag_inv(self):
        lib = _hip.lib()
        out = device.empty(self.size)
        _hip.check(lib.flow_operator_diag_inv(
            ctypes.byref(self.operator()),
            _hip.i32(self.layout.dev('diag_idx'), self.layout.N, 'diag_idx'),
            _hip.f64(out), _hip.stream()
            ))
        return out

    def to_scipy(self):
        '''Host copy (tests / debugging only).'''
        import scipy.sparse as sp
        lay = self.layout
        rp = lay.pattern('rowptr')
        cols = lay.pattern('cols')
        v = device.to_host(self.vals).numpy()
        nnz = lay.nnz
        st = self.stride
        P = [sp.csr_matrix((v[p * st:p * st + nnz], cols, rp),
                           shape=(lay.N, lay.N)) for p in range(self.nplanes)]
        if self.kind == 0:
            return P[0]
        if self.kind == 4:
            m = device.to_host(self.rowmask).numpy().astype(float)
            blocks = [
                sp.diags(m[c * lay.N:(c + 1) * lay.N]).dot(P[0])
                + sp.diags(1.0 - m[c * lay.N:(c + 1) * lay.N])
                for c in range(2)
                ]
            return sp.block_diag(blocks, format='csr')
        if self.kind == 1:
            return sp.block_diag(P, format='csr')
        return sp.bmat([[P[0], P[1]], [P[2], P[3]]], format='csr')


class MomentumJacobian(object):
    '''Matrix-free Jacobian of the momentum residual at `ui` (flow_operator
    kind 3, include/flow_hip.h `flow_momentum_jvp`): `derivative(F1, ui)` of the
    reference (pressure_correction.py:202) applied cell by cell instead of
    assembled.  Quacks like a Matrix for krylov_solve ('bicgstab').'''
    kind = 3

    def __init__(self, W, bfmask, ui, prm, bc_dofs, mesh_s=None, space_s=None):
        '''mesh_s / space_s: a rank's views of the mesh and space structs
        (cell and row ranges: flow_amd/parallel.py); default: everything.'''
        mesh = W.mesh()
        self.layout = W.layout
        self.size = W.size()
        n2 = self.size
        nc = mesh.num_cells()
        if mesh_s is None:
            mesh_s = mesh_struct(mesh)
        if space_s is None:
            space_s = space_struct(W.layout)
        self._keep = (bfmask, ui, bc_dofs,
                      device.empty(2 * W.layout.nloc * nc), mesh_s, space_s)
        self.struct = _hip.MomentumJvp(
            ctypes.pointer(mesh_s),
            ctypes.pointer(space_s),
            _hip.i32(bfmask, nc, 'bfmask'), _hip.f64(ui, n2, 'ui'), prm,
            _hip.f64(self._keep[3]), int(bc_dofs.numel()),
            _hip.i32(bc_dofs) if bc_dofs.numel() else None,
            self._bc_mask(bc_dofs),
            )
        op = _hip.Operator()
        op.kind = 3
        op.n = W.layout.N
        op.matfree = ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)
        self._op = op

    def operator(self):
        return self._op

    def _bc_mask(self, bc_dofs):
        '''Byte mask of the Dirichlet dofs (2n, component-blocked): with it the
        gather of an application writes the identity rows itself.  Built once
        per dof set (the caller hands in the same tensor while the conditions
        are unchanged).'''
        if not bc_dofs.numel():
            return None
        held = getattr(self, '_mask', None)
        # (the same tensor, not rewritten in place since: identity alone would
        # keep a stale mask after an in-place update of the dof list)
        stamp = (bc_dofs.data_ptr(), bc_dofs.numel(), bc_dofs._version)
        if held is None or held[0] is not bc_dofs or held[2] != stamp:
            mask = torch.zeros(self.size, dtype=torch.uint8,
                               device=bc_dofs.device)
            mask[bc_dofs.long()] = 1
            device.synchronize()
            self._mask = held = (bc_dofs, mask, stamp)
        return _hip.u8(held[1], self.size, 'bc_mask')

    def rebind(self, bfmask, ui, prm, bc_dofs):
        '''Point the operator at another linearisation state (a time loop
        builds one per Newton iteration: the 0.2 GB scratch buffer and the
        structs stay).'''
        nc = self.layout.mesh.num_cells()
        self._keep = (bfmask, ui, bc_dofs) + self._keep[3:]
        s = self.struct
        s.bfmask = _hip.i32(bfmask, nc, 'bfmask')
        s.ui = _hip.f64(ui, self.size, 'ui')
        s.prm = prm
        s.nbc = int(bc_dofs.numel())
        s.bc_dofs = _hip.i32(bc_dofs) if bc_dofs.numel() else None
        s.bc_mask = self._bc_mask(bc_dofs)
        return self

    @classmethod
    def cached(cls, W, bfmask, ui, prm, bc_dofs, mesh_s=None, space_s=None):
        '''The operator of W's layout, created on first use and re-pointed
        afterwards.'''
        key = ('jvp_operator', id(mesh_s), id(space_s))
        held = W.layout._dev.get(key)
        if held is None:
            held = cls(W, bfmask, ui, prm, bc_dofs, mesh_s, space_s)
            W.layout._dev[key] = held
            return held
        return held.rebind(bfmask, ui, prm, bc_dofs)

    def apply(self, x, y):
        lib = _hip.lib()
        assert x.data_ptr() != y.data_ptr()
        _hip.check(lib.flow_momentum_jvp_apply(
            ctypes.byref(self.struct), _hip.f64(x, self.size, 'x'),
            _hip.f64(y, self.size, 'y'), _hip.stream()
            ))
        return y


STIFFNESS, MASS, LUMPED_MASS = 0, 1, 2


def assemble_scalar_matrix(layout, kind):
    '''Step-invariant scalar matrices, cached per layout (K1, K3).'''
    key = ('matrix', kind)
    if key not in layout._dev:
        lib = _hip.lib()
        mesh = layout.mesh
        A = Matrix(layout, 0)
        buf = scratch(mesh, layout.nloc**2 * mesh.num_cells())
        _hip.check(lib.flow_assemble_scalar_matrix(
            kind, ctypes.byref(mesh_struct(mesh)),
            ctypes.byref(space_struct(layout)), _hip.f64(buf),
            _hip.f64(A.vals), _hip.stream()
            ))
        layout._dev[key] = A
    return layout._dev[key]


def assemble_mass(V):
    return assemble_scalar_matrix(V.layout, MASS)


def assemble_stiffness(V):
    return assemble_scalar_matrix(V.layout, STIFFNESS)


def symmetric_bc_matrix(A, isbc):
    '''assemble_system-style symmetric elimination of one scalar plane.'''
    lib = _hip.lib()
    lay = A.layout
    out = Matrix(lay, 0)
    _hip.check(lib.flow_bc_symmetric_matrix(
        lay.N, _hip.i32(lay.dev('rowptr')), _hip.i32(lay.dev('cols')),
        _hip.f64(A.vals, lay.nnz), _hip.u8(isbc, lay.N, 'isbc'),
        _hip.f64(out.vals, lay.nnz), _hip.stream()
        ))
    return out


class CoarseSpace(object):
    '''Piecewise-constant aggregate coarse space of a scalar SPD operator for
    the two-level additive preconditioner of flow_cg_solve (include/flow_hip.h,
    `flow_coarse`): dofs are binned into square patches of `s` mesh widths,
    Dirichlet dofs are left out, Ac = P^T A P is inverted densely on the host
    once per (operator, BC set).  `singular`: pure Neumann operator (constants
    in the kernel) -> pseudo-inverse via the rank-one shift.  Setup only; the
    per-iteration work runs in the HIP kernels.'''

    def __init__(self, A, isbc=None, singular=False, target_nc=4096):
        import scipy.sparse as sp
        lay = A.layout
        n = lay.N
        assert A.kind == 0
        isbc = numpy.zeros(n, dtype=bool) if isbc is None else \
            numpy.asarray(isbc, dtype=bool)
        x = lay.dof_coords
        h = numpy.sqrt(2.0 * lay.mesh.cell_areas().mean())
        nfree = int(n - isbc.sum())
        s = max(2.0, numpy.sqrt(max(nfree, 1) / float(target_nc)))
        while True:
            ix = numpy.floor((x[:, 0] - x[:, 0].min()) / (s * h) + 1e-9)
            iy = numpy.floor((x[:, 1] - x[:, 1].min()) / (s * h) + 1e-9)
            key = (ix * 2000003 + iy).astype(numpy.int64)
            key[isbc] = -1
            ukey, agg = numpy.unique(key, return_inverse=True)
            if len(ukey) and ukey[0] == -1:
                agg = agg - 1          # excluded dofs -> -1
                nc = len(ukey) - 1
            else:
                nc = len(ukey)
            if nc <= 1.25 * target_nc or s > 1.0e6:
                break
            s *= 1.1
        assert nc >= 1
        self.s = s
        self.nc = nc
        self.n = n
        free = numpy.nonzero(agg >= 0)[0]
        Pm = sp.csr_matrix(
            (numpy.ones(len(free)), (free, agg[free])), shape=(n, nc)
            )
        Ah = A.to_scipy()
        Ac = (Pm.T.dot(Ah).dot(Pm)).toarray()
        if singular:
            e = numpy.ones(nc)
            beta = numpy.trace(Ac) / nc
            Ainv = numpy.linalg.inv(Ac + beta * numpy.outer(e, e) / nc) \
                - numpy.outer(e, e) / (beta * nc)
        else:
            Ainv = numpy.linalg.inv(Ac)
        Ainv = 0.5 * (Ainv + Ainv.T)
        order = numpy.argsort(agg[free], kind='stable')
        agg_dofs = free[order].astype(numpy.int32)
        agg_ptr = numpy.zeros(nc + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(agg[free], minlength=nc), out=agg_ptr[1:])
        self.agg_of_host = agg.astype(numpy.int32)
        # fp32 storage, rows padded to a multiple of 4 floats (float4 loads)
        lda = (nc + 3) // 4 * 4
        A32 = numpy.zeros((nc, lda), dtype=numpy.float32)
        A32[:, :nc] = Ainv
        self._keep = (
            device.to_device(agg_ptr.astype(numpy.int32)),
            device.to_device(agg_dofs),
            device.to_device(self.agg_of_host),
            device.to_device(A32),
            )
        k = self._keep
        self.struct = _hip.CoarseS(
            n, nc, _hip.i32(k[0], nc + 1), _hip.i32(k[1], len(agg_dofs)),
            _hip.i32(k[2], n), lda, _hip.f32(k[3], nc * lda)
            )


# -- BLAS-1 / norms -----------------------------------------------------------
def vector_norm(x, kind='l2'):
    lib = _hip.lib()
    code = {'l2': 0, 'linf': 1}[kind]
    res = ctypes.c_double(0.0)
    _hip.check(lib.flow_norm_host(
        x.numel(), _hip.f64(x), code, _hip.f64(work(_hip.REDUCE_WORK)),
        ctypes.byref(res), _hip.stream()
        ))
    return res.value


def dot(x, y):
    lib = _hip.lib()
    assert x.numel() == y.numel()
    res = ctypes.c_double(0.0)
    _hip.check(lib.flow_dot_host(
        x.numel(), _hip.f64(x), _hip.f64(y), _hip.f64(work(_hip.REDUCE_WORK)),
        ctypes.byref(res), _hip.stream()
        ))
    return res.value


def axpby(a, x, b, y):
    '''y = a x + b y'''
    lib = _hip.lib()
    assert x.numel() == y.numel()
    _hip.check(lib.flow_axpby(
        x.numel(), float(a), _hip.f64(x), float(b), _hip.f64(y), _hip.stream()
        ))
    return y


copy = _hip.copy
fill = _hip.fill


def vmul(x, y, out=None, a=1.0):
    '''out = a * x .* y (out may alias x or y)'''
    lib = _hip.lib()
    assert x.numel() == y.numel()
    if out is None:
        out = device.empty(x.numel())
    _hip.check(lib.flow_vmul(
        x.numel(), float(a), _hip.f64(x), _hip.f64(y), _hip.f64(out),
        _hip.stream()
        ))
    return out


# -- Krylov -------------------------------------------------------------------
class StartChooser(object):
    '''Which extrapolation order (1 linear, 2 quadratic) gives a recurring
    solve the better start vector?  Quadratic wins while the fields evolve,
    linear once they only differ by solver noise (which the quadratic weights
    amplify more).  Decided by the iteration counts themselves: the mode in
    use is kept, the other one is tried every `period`-th call.'''

    def __init__(self, period=8):
        self.period = period
        self.calls = 0
        self.mode = 2
        self.its = {}

    def pick(self):
        self.calls += 1
        if self.calls % self.period == 0:
            return 3 - self.mode          # trial of the other mode
        return self.mode

    def report(self, mode, iterations):
        self.its[mode] = iterations
        other = 3 - self.mode
        if other in self.its and self.mode in self.its:
            if self.its[other] < self.its[self.mode] or (
                    self.its[other] == self.its[self.mode] and other == 1):
                self.mode = other


class SolveInfo(object):
    def __init__(self, iterations, residual, method):
        self.iterations = iterations
        self.residual = residual
        self.method = method

    def __repr__(self):
        return '%s: %d iterations, |r| = %.3e' % (
            self.method, self.iterations, self.residual
            )


def krylov_solve(method, A, b, x, rtol, atol=0.0, maxit=1000, dinv='jacobi',
                 check_every=None, coarse=None, ilu=None, mg=None,
                 first_check=0, tag=None, restart=20, x_is_zero=False,
                 pmg=None, verify=True, guard=None):
    '''Solve A x = b on the device; x holds the initial guess.  Raises
    _hip.NotConverged (a RuntimeError) like dolfin's
    'error_on_nonconvergence'.  first_check > 0: iterations before the first
    residual read-back (then every check_every); the device itself freezes
    the solution at the first iterate that passes the stopping test, so a
    late read-back costs idle launches, never accuracy.  tag (CG): the solve
    recurs in a time loop under that name -- the first read-back comes one
    iteration after the count the previous call (kept on A) needed.
    method 'gmres': GMRES(restart); x_is_zero promises x = 0 on entry;
    `iterations` then counts operator applications (a BiCGStab iteration is
    two); first_check = the applications the caller expects the solve to need
    (that many Arnoldi steps are enqueued before the first read-back);
    verify: check the accepted iterate with the true residual (one more
    operator application; off for the Newton systems, whose outer iteration
    recomputes the nonlinear residual anyway).
    guard (CG): x is a start vector the caller does not vouch for (one
    extrapolated from earlier calls): False / a device vector = its fallback
    (none / that vector).  A start that leaves a larger preconditioned residual
    than zero is dropped on the device for the fallback, and that for zero
    (flow_cg_solve_guarded); `starts_dropped` of the returned info counts.'''
    lib = _hip.lib()
    n = A.size
    if isinstance(dinv, str):
        assert dinv == 'jacobi'
        dinv = A.diag_inv() if ilu is None and pmg is None else None
    assert method in ('cg', 'bicgstab', 'gmres'), method
    nvec = {'cg': 5, 'bicgstab': 7, 'gmres': 2 * restart + 2}[method]
    nparts = A.operator().nblocks * (2 if A.kind == 1 else 1) + 2
    wk = work(_hip.REDUCE_WORK + nvec * n
              + (nparts if method == 'cg' else 0)
              + (2 * coarse.struct.lda if coarse else 0)
              + (2 * max(mg.struct.Ps[0].nblocks, mg.struct.up_nblocks[0])
                 if mg is not None else 0)
              + (n if ilu is not None and method == 'bicgstab' else 0)
              + (_hip.GMRES_PARTIALS + _hip.GMRES_STATE
                 if method == 'gmres' else 0))
    if device._POISON:      # debugging aid: stale workspace reads become NaNs
        _hip.fill(wk, float('nan'))
    if check_every is None:
        check_every = 10 if method == 'bicgstab' else 50
    its = ctypes.c_int(0)
    res = ctypes.c_double(0.0)
    history = None
    if tag is not None and method == 'cg':
        history = A.__dict__.setdefault('_solve_history', {})
        if first_check == 0 and tag in history:
            first_check = history[tag] + 1
    dropped = ctypes.c_int(0)
    if method == 'cg' and guard is not None:
        fallback = None if guard is False else _hip.f64(guard, n, 'x_fallback')
        rc = lib.flow_cg_solve_guarded(
            ctypes.byref(A.operator()), _hip.f64(dinv, n, 'dinv'),
            ctypes.byref(coarse.struct) if coarse is not None else None,
            ctypes.byref(mg.struct) if mg is not None else None,
            _hip.f64(b, n, 'b'), _hip.f64(x, n, 'x'), fallback, float(rtol),
            float(atol), int(maxit), int(check_every), int(first_check),
            _hip.f64(wk), wk.numel(), ctypes.byref(its), ctypes.byref(res),
            ctypes.byref(dropped), _hip.stream()
            )
    elif method == 'cg':
        rc = lib.flow_cg_solve(
            ctypes.byref(A.operator()), _hip.f64(dinv, n, 'dinv'),
            ctypes.byref(coarse.struct) if coarse is not None else None,
            ctypes.byref(mg.struct) if mg is not None else None,
            _hip.f64(b, n, 'b'), _hip.f64(x, n, 'x'), float(rtol), float(atol),
            int(maxit), int(check_every), int(first_check), _hip.f64(wk),
            wk.numel(),
            ctypes.byref(its), ctypes.byref(res), _hip.stream()
            )
    elif method == 'gmres':
        assert coarse is None and mg is None
        rc = lib.flow_gmres_solve(
            ctypes.byref(A.operator()), _hip.f64(dinv, n, 'dinv'),
            ctypes.byref(ilu.struct) if ilu is not None else None,
            ctypes.byref(pmg.struct) if pmg is not None else None,
            _hip.f64(b, n, 'b'), _hip.f64(x, n, 'x'), float(rtol), float(atol),
            int(maxit), int(restart), int(bool(x_is_zero)), int(first_check),
            int(bool(verify)),
            _hip.f64(wk), wk.numel(), ctypes.byref(its), ctypes.byref(res),
            _hip.stream()
            )
    else:
        assert coarse is None and mg is None
        rc = lib.flow_bicgstab_solve(
            ctypes.byref(A.operator()), _hip.f64(dinv, n, 'dinv'),
            ctypes.byref(ilu.struct) if ilu is not None else None,
            _hip.f64(b, n, 'b'), _hip.f64(x, n, 'x'), float(rtol), float(atol),
            int(maxit), int(check_every), int(first_check), _hip.f64(wk),
            wk.numel(),
            ctypes.byref(its), ctypes.byref(res), _hip.stream()
            )
    _hip.check(rc)
    if history is not None:
        history[tag] = its.value
    out = SolveInfo(its.value, res.value,
                    method + ('+2level' if coarse is not None else '')
                    + ('+mg%d' % mg.nlevels if mg is not None else '')
                    + (('+tlilu' if hasattr(ilu, 'cycle') else '+ilu0')
                       if ilu is not None else '')
                    + ('+pmg' if pmg is not None else ''))
    out.starts_dropped = dropped.value
    return out


# -- load vectors, projection, norms -----------------------------------------
def assemble_source(V, f):
    '''(f, v) for v in V (scalar or vector space).'''
    lib = _hip.lib()
    mesh = V.mesh()
    lay = V.layout
    coef = as_cell_coefficient(f, mesh, V.dim)
    cs, keep = coef_struct(coef, mesh, lay.degree)
    b = device.empty(V.size())
    buf = scratch(mesh, V.dim * lay.nloc * mesh.num_cells())
    _hip.check(lib.flow_assemble_source(
        ctypes.byref(mesh_struct(mesh)), ctypes.byref(space_struct(lay)), V.dim,
        ctypes.byref(cs), _hip.f64(buf), _hip.f64(b), _hip.stream()
        ))
    del keep
    return b


def project(f, V, tol=1.0e-14):
    '''L2 projection (dolfin.project): mass solve per component, CG + Jacobi on
    the device.'''
    b = assemble_source(V, f)
    M = assemble_mass(V)
    dinv = M.diag_inv()
    out = Function(V)
    n = V.N
    for c in range(V.dim):
        x = out.data[c * n:(c + 1) * n]
        krylov_solve('cg', M, b[c * n:(c + 1) * n].contiguous(), x, tol,
                     maxit=1000, dinv=dinv, check_every=10)
    return out


# how the mass systems of the projections below are solved: 'chebyshev' (defect
# correction with a fixed Chebyshev polynomial, flow_amd/fem/mass.py) or 'cg'
MASS_SOLVER = {'method': 'chebyshev', 'steps': 6}


def project_magnitude(u, mode=0, tol=1.0e-12, initial_guess=None):
    '''`project(sqrt(ux**2 + uy**2), FunctionSpace(mesh, 'Lagrange', k))`
    (mode 0; tests/test_karman_vortex_street.py:262-267) or
    `project(abs(ux) + abs(uy), Q)` (mode 1; tests/test_boussinesq.py:268-272)
    for a vector field u of degree k.  `initial_guess`: a previous projection
    to start the mass solve from (step-size controllers call this every step).'''
    from .. import parallel
    lib = _hip.lib()
    W = u.function_space()
    assert W.dim == 2
    mesh = W.mesh()
    lay = W.layout
    S = W.collapse()
    strips = parallel.active()
    # (single GPU: right-hand side and iterate of the mass solve in buffers of
    # the layout -- a loop whose arguments repeat from call to call is replayed
    # as a HIP graph, csrc/graph_replay.hip; the result is copied out)
    # -- only where a replay is possible at all (the option is off by default:
    # the two copies then are not made)
    fixed = not strips and _hip.graphs_possible()
    held = lay._dev.get(('persistent', 'magnitude')) if fixed else None
    if fixed and held is None:
        held = lay._dev[('persistent', 'magnitude')] = (
            device.empty(lay.N), device.empty(lay.N))
    b = _hip.fill(held[0], 0.0) if fixed else device.zeros(lay.N)
    buf = scratch(mesh, lay.nloc * mesh.num_cells())
    _hip.check(lib.flow_assemble_magnitude(
        ctypes.byref(parallel.mesh_view(mesh) if strips else mesh_struct(mesh)),
        ctypes.byref(parallel.view(lay).space if strips else space_struct(lay)),
        int(mode), _hip.f64(u.data, 2 * lay.N), _hip.f64(buf), _hip.f64(b),
        _hip.stream()
        ))
    M = assemble_mass(S)
    key = ('M_dinv',)
    if key not in lay._dev:
        lay._dev[key] = M.diag_inv()
    out = Function(S)
    if initial_guess is not None:
        out.assign(initial_guess)
    tag = 'project_magnitude' if initial_guess is not None else None
    if strips and MASS_SOLVER['method'] == 'chebyshev':
        from .mass import MassSolver
        out.solve_info = parallel.mass_solve(
            MassSolver.cached(M, lay._dev[key], steps=MASS_SOLVER['steps']),
            b, out.data, tol, maxit=100, tag=tag)
    elif strips:
        out.solve_info = parallel.cg(M, lay._dev[key], b, out.data, tol,
                                     maxit=1000, check_every=2, tag=tag)
    elif MASS_SOLVER['method'] == 'chebyshev':
        from .mass import MassSolver
        x = copy(held[1], out.data) if fixed else out.data
        out.solve_info = MassSolver.cached(
            M, lay._dev[key], steps=MASS_SOLVER['steps']).solve(
                b, x, tol, maxit=100, tag=tag)
        if fixed:
            copy(out.data, x)
    else:
        out.solve_info = krylov_solve(
            'cg', M, b, out.data, tol, maxit=1000, dinv=lay._dev[key],
            check_every=2, tag=tag)
    return out


def interpolate(f, V):
    out = Function(V)
    if isinstance(f, Constant):
        out.assign(f)
        return out
    assert isinstance(f, Expression)
    vals = f.eval(V.layout.dof_coords.T)
    assert vals.shape[0] == V.dim
    out.set_array(vals.reshape(-1))
    return out


def norm(u, kind='L2'):
    '''norm(Function, 'L2') = sqrt(u^T M u) (tests/test_boussinesq.py:85);
    norm(vector, 'linf'|'l2').'''
    if isinstance(u, Function):
        assert kind == 'L2'
        V = u.function_space()
        M = assemble_mass(V)
        n = V.N
        tmp = device.empty(n)
        total = 0.0
        for c in range(V.dim):
            x = u.data[c * n:(c + 1) * n]
            M.apply(x, tmp)
            total += dot(x, tmp)
        return numpy.sqrt(total)
    data = u.data if hasattr(u, 'data') else u
    return vector_norm(data, kind.lower())


def integral(u):
    '''assemble(u*dx) for a scalar Function: 1^T M u.'''
    V = u.function_space()
    assert V.dim == 1
    M = assemble_mass(V)
    tmp = device.empty(V.N)
    M.apply(u.data, tmp)
    one = torch.ones(V.N, dtype=torch.float64, device=device.get())
    return dot(one, tmp)


def errornorm(exact, uh, degree_rise=3):
    '''L2 error between an Expression and a discrete Function, both
    interpolated per cell into P_(k+degree_rise) as dolfin.errornorm does
    (tests/test_navier_stokes.py:333, 360).  A measurement for the harness:
    evaluated on the host with numpy.'''
    V = uh.function_space()
    mesh = V.mesh()
    k = min(V.degree + degree_rise, 5)
    lat = reference.lattice(k)
    X = cell_lattice_points(mesh, k)                        # (Nc, nl, 2)
    nc, nl = X.shape[:2]
    ue = exact.eval(X.reshape(-1, 2).T).reshape(V.dim, nc, nl)
    tab = reference.tabulate(V.degree, lat)                 # (nl, nloc)
    U = uh.array().reshape(V.dim, V.N)
    uhl = numpy.einsum('acj,lj->acl', U[:, V.layout.cell_dofs], tab)
    e = ue - uhl
    Mk = reference.mass_matrix(k)
    detj = 2.0 * mesh.cell_areas()
    return float(numpy.sqrt(numpy.einsum('acl,lm,acm,c->', e, Mk, e, detj)))
