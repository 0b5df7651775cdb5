# -*- coding: utf-8 -*-
'''
Host-side setup of the smoothed-aggregation multigrid preconditioner of the
pressure-Poisson CG (`flow_mg` in include/flow_hip.h; stands in for the
reference's 'hypre_amg', flow/navier_stokes/pressure_correction.py:331,
414-418).

Setup only (numpy/scipy, once per operator and Dirichlet set): every
application of the V-cycle runs in the HIP kernels behind flow_cg_solve.

  aggregates   square patches of `s` mesh widths (spatial binning of the dof
               coordinates, as the two-level CoarseSpace does; coarser levels
               bin the aggregate centroids); Dirichlet dofs stay out
  P            (I - 4/(3 rho) D^-1 A) P0, rho = spectral radius of D^-1 A
               (power iteration), P0 the piecewise-constant prolongation
  A_{l+1}      P^T A_l P
  coarsest     dense (pseudo-)inverse in fp32, as CoarseSpace
  per level    the device cycle works with Ah = A w D^-1 (column-scaled) and
               Ps = (I - w D^-1 A) P: with them the textbook V(1,1) cycle from
               a zero start, x = w D^-1 r; r' = P^T (r - A x); ...;
               x += P x'; x += w D^-1 (r - A x), becomes t = r - Ah r;
               r' = P^T t; ...; x = Ps x' + w D^-1 (r + t) -- one product with
               A and three launches per level (include/flow_hip.h, flow_mg)
Measured on the 1.1 M-row pressure system of the headline workload: 21 CG
iterations instead of 180-210 with the two-level preconditioner.
'''
import ctypes

import numpy

from .space import csr_stream_rowblocks
from .. import _hip
from .. import device


class CsrOperator(object):
    '''A (possibly rectangular) CSR matrix on the device as a kind-0
    flow_operator: padded, aligned arrays + CSR-stream row blocks.'''

    def __init__(self, M):
        M = M.tocsr()
        M.sort_indices()
        self.shape = M.shape
        rowptr = M.indptr.astype(numpy.int64)
        nnz = int(rowptr[-1])
        assert nnz > 0
        pad = 2 + (nnz & 1)
        self._rowptr = device.to_device(rowptr.astype(numpy.int32))
        self._cols = device.to_device(numpy.concatenate(
            [M.indices.astype(numpy.int32), numpy.zeros(pad, numpy.int32)]))
        self._vals = device.to_device(numpy.concatenate(
            [M.data.astype(numpy.float64), numpy.zeros(pad)]))
        rb = csr_stream_rowblocks(rowptr)
        self._rb = device.to_device(rb.astype(numpy.int32))
        op = _hip.Operator()
        op.kind = 0
        op.n = M.shape[0]
        op.nnz = nnz
        op.nblocks = len(rb) - 1
        op.rowptr = _hip.i32(self._rowptr, M.shape[0] + 1)
        op.cols = _hip.i32(self._cols, nnz)
        op.rowblocks = _hip.i32(self._rb)
        op.vals[0] = _hip.f64(self._vals, nnz).value
        self.op = op


def _scale_rows(M, d):
    '''diag(d) M for a CSR matrix M.'''
    out = M.copy()
    out.data = out.data * numpy.repeat(d, numpy.diff(out.indptr))
    return out


def _bin_aggregates(x, free, width):
    ix = numpy.floor((x[:, 0] - x[:, 0].min()) / width + 1e-9).astype(numpy.int64)
    iy = numpy.floor((x[:, 1] - x[:, 1].min()) / width + 1e-9).astype(numpy.int64)
    key = ix * 2000003 + iy
    key[~free] = -1
    ukey, agg = numpy.unique(key, return_inverse=True)
    if len(ukey) and ukey[0] == -1:
        return agg - 1, len(ukey) - 1
    return agg, len(ukey)


def _kd_aggregates(x, free, target):
    '''Aggregates of about `target` free points each by recursive median
    splits along the longer axis of a group's bounding box (a k-d tree's
    leaves): spatially compact and of equal COUNT whatever the local mesh
    size -- for graded meshes, where bins of one width hold hundreds of points
    at the fine end and none at the coarse one.'''
    idx = numpy.nonzero(free)[0]
    agg = numpy.full(len(x), -1, dtype=numpy.int64)
    nagg = 0
    stack = [idx]
    while stack:
        g = stack.pop()
        if len(g) <= 1.5 * target:
            agg[g] = nagg
            nagg += 1
            continue
        p = x[g]
        ext = p.max(axis=0) - p.min(axis=0)
        axis = int(numpy.argmax(ext))
        # (split into parts that are multiples of the target, as even as that
        # allows: leaves between target and 1.5 target points)
        k = int(round(len(g) / float(target)))
        half = (k // 2) * len(g) // k
        order = numpy.argpartition(p[:, axis], half)
        stack.append(g[order[:half]])
        stack.append(g[order[half:]])
    return agg, nagg


def _algebraic_aggregates(Ah, free, theta):
    '''Aggregates from the MATRIX alone (flow_aggregate_host, a host routine
    of the library: strength of connection + the three passes of Vanek, Mandel
    and Brezina) -- what the reference's BoomerAMG works from
    (flow/navier_stokes/pressure_correction.py:331, 414): no coordinates.'''
    Ah = Ah.tocsr()
    n = Ah.shape[0]
    agg = numpy.empty(n, dtype=numpy.int32)
    na = ctypes.c_int(0)
    rowptr = numpy.ascontiguousarray(Ah.indptr, dtype=numpy.int32)
    cols = numpy.ascontiguousarray(Ah.indices, dtype=numpy.int32)
    vals = numpy.ascontiguousarray(Ah.data, dtype=numpy.float64)
    fr = numpy.ascontiguousarray(free, dtype=numpy.uint8)
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _hip.check(_hip.load_library().flow_aggregate_host(
        n, as_p(rowptr), as_p(cols), as_p(vals), float(theta), as_p(fr),
        as_p(agg), ctypes.byref(na)))
    return agg.astype(numpy.int64), na.value


class Multigrid(object):
    '''Hierarchy for a scalar SPD Matrix `A` (kind 0).  isbc: Dirichlet dofs
    (identity rows of A); singular: pure Neumann operator -> pseudo-inverse on
    the coarsest level.'''

    def __init__(self, A, isbc=None, singular=False, s=3.0, coarsest=4200,
                 omega=0.8, keep_host=False, two_launch=True,
                 aggregation='geometric', theta=0.08):
        '''two_launch: also form C = R (I - Ah) and the row blocks common to
        Ps and Ah per level, so that the device cycle runs its two-launch
        form (include/flow_hip.h, flow_mg).
        aggregation: 'geometric' (default: patches of the dof coordinates --
        bins on a uniform mesh, k-d-tree leaves on a graded one) or
        'algebraic' (from the matrix alone: `_algebraic_aggregates`, strength
        threshold `theta`, halved per level; for operators that come without
        coordinates).'''
        import scipy.sparse as sp
        assert A.kind == 0
        lay = A.layout
        n = lay.N
        isbc = numpy.zeros(n, dtype=bool) if isbc is None else \
            numpy.asarray(isbc, dtype=bool)
        assert aggregation in ('geometric', 'algebraic'), aggregation
        if aggregation == 'algebraic':
            x, width = None, 0.0
            self.aggregation = 'algebraic'
        else:
            x = lay.dof_coords.copy()
            areas = lay.mesh.cell_areas()
            width = s * numpy.sqrt(2.0 * areas.mean())
            # a graded mesh (cell sizes a factor 3 apart and more): aggregates
            # of equal count (s^2 points: what a bin of s widths holds on a
            # uniform mesh) instead of bins of equal width
            self.aggregation = 'kd-tree' if numpy.percentile(areas, 99) > \
                9.0 * numpy.percentile(areas, 1) else 'bins'
        free = ~isbc
        Ah = A.to_scipy().tocsr()
        self.levels = []          # device operators Ah, Ps, R + dinv, t per level
        self.host_levels = []     # keep_host: scipy (A, D, P) for the tests
        self.sizes = [n]
        self.R0_host = None
        self.C0_host = None
        rng = numpy.random.RandomState(1)
        self.fine = A
        while Ah.shape[0] > coarsest and len(self.levels) < _hip.MG_MAX_LEVELS - 1:
            m = Ah.shape[0]
            D = Ah.diagonal()
            if self.aggregation == 'algebraic':
                agg, nc = _algebraic_aggregates(
                    Ah, free, theta * 0.5**len(self.levels))
            elif self.aggregation == 'kd-tree':
                agg, nc = _kd_aggregates(x, free, s * s)
            else:
                agg, nc = _bin_aggregates(x, free, width)
            if nc < 2 or nc >= m:
                break
            idx = numpy.nonzero(agg >= 0)[0]
            P0 = sp.csr_matrix((numpy.ones(len(idx)), (idx, agg[idx])),
                               shape=(m, nc))
            v = rng.standard_normal(m)
            lam = 1.0
            for _ in range(15):
                v = Ah.dot(v) / D
                lam = numpy.linalg.norm(v)
                v /= lam
            # (diagonal scalings act on the value arrays: a product with a
            # sparse diagonal matrix costs scipy a full sparse matmat)
            P = (P0 - _scale_rows(Ah.dot(P0).tocsr(), (4.0 / (3.0 * lam)) / D)
                 ).tocsr()
            AP = Ah.dot(P).tocsr()
            R = P.T.tocsr()
            Ac = (R.dot(AP)).tocsr()
            Ahat = Ah.copy()
            Ahat.data = Ahat.data * (omega / D)[Ahat.indices]
            Ps = (P - _scale_rows(AP, omega / D)).tocsr()
            self.levels.append(dict(
                Ah=CsrOperator(Ahat),
                dinv=device.to_device(1.0 / D),
                Ps=CsrOperator(Ps), R=CsrOperator(R),
                t=device.zeros(m),
                ))
            if two_launch:
                # C = R (I - Ah): restriction of the residual behind the
                # pre-smoothing step in one product; and row blocks that hold
                # a tile of Ps AND of Ah (the up-sweep does both products)
                Cm = (R - R.dot(Ahat)).tocsr()
                Cm.eliminate_zeros()
                L = self.levels[-1]
                L['C'] = CsrOperator(Cm)
                Ps.sort_indices()
                L['up_rb'] = device.to_device(csr_stream_rowblocks(
                    [Ps.indptr, Ahat.indptr]))
                if not self.levels[1:]:
                    self.C0_host = Cm
            if keep_host:
                self.host_levels.append((Ah, D, P))
            if not self.levels[1:]:
                # the finest restriction, for the strip-sharded cycle
                # (flow_amd/parallel.py cuts it by columns)
                self.R0_host = R
            if x is not None:
                cnt = numpy.bincount(agg[idx], minlength=nc)
                x = numpy.stack([
                    numpy.bincount(agg[idx], weights=x[idx, d], minlength=nc)
                    / cnt for d in (0, 1)], axis=1)
            free = numpy.ones(nc, dtype=bool)
            width *= s
            Ah = Ac
            self.sizes.append(nc)
        nc = Ah.shape[0]
        Ad = Ah.toarray()
        if singular:
            e = numpy.ones(nc)
            # constants are (to rounding) in the kernel of every Galerkin level
            beta = numpy.trace(Ad) / nc
            Ainv = numpy.linalg.inv(Ad + beta * numpy.outer(e, e) / nc) \
                - numpy.outer(e, e) / (beta * nc)
        else:
            Ainv = numpy.linalg.inv(Ad)
        Ainv = 0.5 * (Ainv + Ainv.T)
        lda = (nc + 3) // 4 * 4
        A32 = numpy.zeros((nc, lda), dtype=numpy.float32)
        A32[:, :nc] = Ainv
        self._Ainv = device.to_device(A32)
        self.nlevels = len(self.levels) + 1
        self._r = [None] + [device.zeros(m) for m in self.sizes[1:]]
        self._x = [None] + [device.zeros(m) for m in self.sizes[1:]]
        M = _hip.MgS()
        M.nlevels = self.nlevels
        for l, L in enumerate(self.levels):
            ctypes.memmove(ctypes.byref(M.Ah[l]), ctypes.byref(L['Ah'].op),
                           ctypes.sizeof(_hip.Operator))
            ctypes.memmove(ctypes.byref(M.Ps[l]), ctypes.byref(L['Ps'].op),
                           ctypes.sizeof(_hip.Operator))
            ctypes.memmove(ctypes.byref(M.R[l]), ctypes.byref(L['R'].op),
                           ctypes.sizeof(_hip.Operator))
            M.dinv[l] = _hip.f64(L['dinv'], self.sizes[l]).value
            M.t[l] = _hip.f64(L['t'], self.sizes[l]).value
            if 'C' in L:
                ctypes.memmove(ctypes.byref(M.C[l]), ctypes.byref(L['C'].op),
                               ctypes.sizeof(_hip.Operator))
                M.up_rowblocks[l] = _hip.i32(L['up_rb']).value
                M.up_nblocks[l] = L['up_rb'].numel() - 1
        for l in range(1, self.nlevels):
            M.r[l] = _hip.f64(self._r[l], self.sizes[l]).value
            M.x[l] = _hip.f64(self._x[l], self.sizes[l]).value
        M.nc, M.lda = nc, lda
        M.Ainv = _hip.f32(self._Ainv, nc * lda).value
        M.omega = float(omega)
        self.struct = M
        self.omega = float(omega)
        self.Ainv_host = Ainv if keep_host else None

    def apply(self, r, z):
        '''z = V-cycle(r): one application of the preconditioner.'''
        n = self.sizes[0]
        _hip.check(_hip.lib().flow_mg_apply(
            ctypes.byref(self.struct), n, _hip.f64(r, n, 'r'),
            _hip.f64(z, n, 'z'), _hip.stream()
            ))
        return z
