# -*- coding: utf-8 -*-
'''
Host side of the p-multigrid / Chebyshev preconditioner of the Newton systems
(`flow_pmg` in include/flow_hip.h, kernels in flow_amd/csrc/pmg_kernels.hip):
a second stand-in -- next to the multicolour ILU(0) of flow_amd/fem/ilu.py --
for the sparse LU behind the reference's Newton solve
(flow/navier_stokes/pressure_correction.py:224-254), as the right
preconditioner of the flexible GMRES.

Two levels on the same mesh: the diagonal blocks of the assembled P2 Jacobian,
smoothed with a few Chebyshev steps, and the P1 discretisation of the same
linearised operator as the coarse problem (P1 is a subspace of P2: vertex dofs
copy their vertex, edge dofs average their two end points), treated with
Chebyshev steps as well.  Both matrices are kept row-scaled (D^-1 A) in fp16,
the two velocity blocks interleaved: 8 bytes per nonzero with the index.  Everything an application does is a CSR-stream
product over a pattern that is already resident -- no dependent sweeps, no
colouring.  Measured on the oracle's Jacobian in the regime of the 10 M-DoF
workload (tools/precond_lab.py: CFL 1.85, diffusion number 0.85): 14-15
GMRES iterations against 34 with the multicolour ILU(0) (natural-order ILU(0),
which has no parallel form: 24).

Setup here is index tables only (numpy, once per mesh); `refactor` packs the
matrices and estimates the spectral radii with kernels of the library.
'''
import ctypes
import os

import numpy
import torch

from . import ops
from .space import scalar_layout, csr_stream_rowblocks
from .. import _hip
from .. import device


# 16-bit column offsets in the packed levels (include/flow_hip.h,
# flow_pmg_level.cols16); False: plain int32 columns
COLS16 = True
# ONE plane for both velocity components (flow_pmg_level.packed: the mean of
# the two diagonal blocks = the Oseen operator, 4 B per nonzero with the 16-bit
# column offset; needs COLS16's tile bases) instead of the two blocks as half2.
# Measured on the 10 M-DoF workload (tools/ab_env_long.sh, 60 steps): the
# products are 10 % shorter (the kernel is bound by its chain of dependent
# loads more than by bytes), an application 45 us of 590 -- and the GMRES needs
# 5.4 applications per step instead of 4.9: 9.66 against 9.45 ms per step.
# Off by default.
ONE_PLANE = os.environ.get('FLOW_AMD_PMG_ONE_PLANE', '0') == '1'


def transfer_tables(lay2):
    '''(ends, rptr, rsrc) between the P2 layout and the P1 layout of its mesh:
    ends[i] = the two P1 rows (= vertices) P2 dof i interpolates from (a vertex
    dof names its vertex twice); rptr/rsrc = the transposed lists, per vertex
    its own P2 dof FIRST, then the dofs of the edges that end there.'''
    mesh = lay2.mesh
    assert lay2.degree == 2
    nv = mesh.num_vertices()
    edges = mesh.edges.astype(numpy.int64)
    ends = numpy.empty((lay2.N, 2), dtype=numpy.int32)
    vd = lay2.vertex_dofs.astype(numpy.int64)
    ed = lay2.edge_dofs.astype(numpy.int64)
    ends[vd, 0] = numpy.arange(nv)
    ends[vd, 1] = numpy.arange(nv)
    ends[ed, 0] = edges[:, 0]
    ends[ed, 1] = edges[:, 1]
    # per vertex: [own dof | edge dofs]
    cnt = 1 + numpy.bincount(edges.ravel(), minlength=nv)
    rptr = numpy.zeros(nv + 1, dtype=numpy.int64)
    numpy.cumsum(cnt, out=rptr[1:])
    rsrc = numpy.empty(int(rptr[-1]), dtype=numpy.int32)
    rsrc[rptr[:-1]] = vd
    owner = numpy.concatenate([edges[:, 0], edges[:, 1]])
    dofs = numpy.concatenate([ed, ed])
    order = numpy.argsort(owner, kind='stable')
    owner, dofs = owner[order], dofs[order]
    # position inside the vertex's list: 1 + running index
    start = numpy.zeros(nv + 1, dtype=numpy.int64)
    numpy.cumsum(numpy.bincount(owner, minlength=nv), out=start[1:])
    within = numpy.arange(len(owner)) - start[owner]
    rsrc[rptr[owner] + 1 + within] = dofs
    return ends, rptr.astype(numpy.int32), rsrc


def local_transfer_tables(lay2, rows, vrows):
    '''The same tables for the diagonal block of a strip (flow_amd/parallel.py):
    P2 rows [r0, r1) and P1 rows (vertices) [v0, v1) in LOCAL numbering.  An
    end point outside the block becomes the dummy coarse row v1 - v0 (the
    kernels read a zero there); the restriction lists hold the block's own
    dofs only.'''
    r0, r1 = rows
    v0, v1 = vrows
    ends, rptr, rsrc = transfer_tables(lay2)
    n1 = v1 - v0
    e = ends[r0:r1].astype(numpy.int64)
    inside = (e >= v0) & (e < v1)
    e_loc = numpy.where(inside, e - v0, n1).astype(numpy.int32)
    # a vertex dof of the block names its own vertex: always inside
    starts = rptr[v0:v1].astype(numpy.int64)
    stops = rptr[v0 + 1:v1 + 1].astype(numpy.int64)
    seg = rsrc[starts[0]:stops[-1]].astype(numpy.int64) if n1 else \
        numpy.zeros(0, dtype=numpy.int64)
    owner = numpy.repeat(numpy.arange(n1), stops - starts)
    keep = (seg >= r0) & (seg < r1)
    first = numpy.zeros(len(seg), dtype=bool)
    first[starts - starts[0]] = True
    assert keep[first].all(), 'a vertex dof outside its vertex\'s block'
    cnt = numpy.bincount(owner[keep], minlength=n1)
    rptr_loc = numpy.zeros(n1 + 1, dtype=numpy.int64)
    numpy.cumsum(cnt, out=rptr_loc[1:])
    rsrc_loc = (seg[keep] - r0).astype(numpy.int32)
    return e_loc, rptr_loc.astype(numpy.int32), rsrc_loc


class _Level(object):
    def __init__(self, lay, rows=None):
        '''rows = (r0, r1): the diagonal block of those rows in local numbering
        (block Jacobi on a strip).'''
        self.lay = lay
        self.rows = rows
        if rows is not None:
            self._init_block(lay, rows)
            return
        self.offset, self.keep = 0, None
        n, nnz = lay.N, lay.nnz
        self.n, self.nnz = n, nnz
        self.diag_idx = lay.dev('diag_idx')
        self.rowptr = lay.dev('rowptr')
        # (half2 per nonzero; readable three entries past nnz: the kernels load
        # QUADS of nonzeros from a base aligned down to a multiple of four)
        self.vals = torch.zeros(2 * (nnz + 4), dtype=torch.float16,
                                device=device.get())
        self.diag = torch.zeros(2 * n, dtype=torch.float32, device=device.get())
        self.dinv = torch.zeros(2 * n, dtype=torch.float32, device=device.get())
        assert self.vals.data_ptr() % 16 == 0
        # row blocks of their own: at most 2044 nonzeros (the tile of 2048
        # minus the alignment slack of a quad)
        if 'pmg_rowblocks' not in lay._dev:
            lay._dev['pmg_rowblocks'] = device.to_device(csr_stream_rowblocks(
                lay.pattern('rowptr'), nnz_per_block=_hip.PMG_NNZ_PER_BLOCK))
        rb = lay._dev['pmg_rowblocks']
        self._fill_struct(lay.dev('rowptr'), lay.dev('cols'), rb)

    def _fill_struct(self, rowptr, cols, rb):
        n, nnz = self.n, self.nnz
        s = _hip.PmgLevelS()
        s.n, s.nnz, s.nblocks = n, nnz, rb.numel() - 1
        s.rowptr = _hip.i32(rowptr, n + 1, 'rowptr')
        s.cols = _hip.i32(cols, nnz + 4, 'cols')
        s.rowblocks = _hip.i32(rb, None, 'rowblocks')
        s.vals = _hip.f16(self.vals, 2 * nnz, 'vals')
        s.diag = _hip.f32(self.diag, 2 * n, 'diag')
        s.dinv = _hip.f32(self.dinv, 2 * n, 'dinv')
        s.lam_min, s.lam_max = 0.25, 2.0
        # 16-bit column offsets from each row block's lowest column (6 B per
        # nonzero with the values instead of 8): any banded numbering allows it
        self._cols16 = None
        if COLS16:
            nb = rb.numel() - 1
            c16 = torch.zeros(nnz + 8, dtype=torch.int16, device=device.get())
            cb = torch.zeros(nb, dtype=torch.int32, device=device.get())
            flag = torch.zeros(1, dtype=torch.int32, device=device.get())
            assert c16.data_ptr() % 16 == 0
            _hip.check(_hip.lib().flow_pmg_cols16(
                nb, _hip.i32(rb), _hip.i32(rowptr, n + 1), _hip.i32(cols, nnz),
                _hip.i32(cb, nb), ctypes.c_void_p(c16.data_ptr()),
                _hip.i32(flag, 1), _hip.stream()))
            if int(device.to_host(flag).item()) == 0:
                self._cols16 = (c16, cb)
                s.cols16 = c16.data_ptr()
                s.cbase = _hip.i32(cb, nb).value
        self._packed = None
        self._cols, self._rb_used = cols, rb
        if ONE_PLANE and self._cols16 is not None:
            pk = torch.zeros(nnz + 8, dtype=torch.int32, device=device.get())
            idr = torch.zeros(2 * n, dtype=torch.uint8, device=device.get())
            assert pk.data_ptr() % 16 == 0
            self._packed = (pk, idr)
            s.packed = _hip.i32(pk, nnz).value
            s.idrows = _hip.u8(idr, 2 * n).value
        self.struct = s

    def _init_block(self, lay, rows):
        r0, r1 = rows
        rp = lay.pattern('rowptr').astype(numpy.int64)
        k0, k1 = int(rp[r0]), int(rp[r1])
        n, nnz = r1 - r0, k1 - k0
        self.n, self.nnz, self.offset = n, nnz, k0
        cols = lay.pattern('cols')[k0:k1].astype(numpy.int64)
        row_of = numpy.repeat(numpy.arange(r0, r1), numpy.diff(rp[r0:r1 + 1]))
        inside = (cols >= r0) & (cols < r1)
        # couplings that leave the block: value 0 (keep mask), column = own row
        cols_loc = numpy.where(inside, cols - r0, row_of - r0)
        self._host = dict(
            rowptr=(rp[r0:r1 + 1] - k0).astype(numpy.int32),
            cols=numpy.concatenate([cols_loc, numpy.zeros(4)]).astype(numpy.int32),
            diag_idx=(lay.pattern('diag_idx')[r0:r1].astype(numpy.int64)
                      - k0).astype(numpy.int32),
            keep=inside.astype(numpy.uint8))
        self.rowptr = device.to_device(self._host['rowptr'])
        self._cols = device.to_device(self._host['cols'])
        self.diag_idx = device.to_device(self._host['diag_idx'])
        self.keep = device.to_device(self._host['keep'])
        self.vals = torch.zeros(2 * (nnz + 4), dtype=torch.float16,
                                device=device.get())
        self.diag = torch.zeros(2 * n, dtype=torch.float32, device=device.get())
        self.dinv = torch.zeros(2 * n, dtype=torch.float32, device=device.get())
        self._rb = device.to_device(csr_stream_rowblocks(
            self._host['rowptr'], nnz_per_block=_hip.PMG_NNZ_PER_BLOCK))
        self._fill_struct(self.rowptr, self._cols, self._rb)

    def pack(self, J):
        '''Diagonal blocks (planes 0 and 3) of a kind-2 Matrix on this layout;
        a scalar Matrix (kind 0) fills both planes (flow_pmg.scalar).'''
        lay = self.lay
        assert J.layout is lay and J.kind in (0, 2)
        n, nnz, k0 = self.n, self.nnz, self.offset
        if J.kind == 0:
            assert self._packed is None
            a = J.vals[k0:k0 + nnz]
            _hip.check(_hip.lib().flow_pmg_pack(
                n, nnz, _hip.i32(self.rowptr, n + 1, 'rowptr'),
                _hip.i32(self.diag_idx, n, 'diag_idx'),
                _hip.f64(a, nnz), _hip.f64(a, nnz),
                _hip.u8(self.keep, nnz) if self.keep is not None else None,
                _hip.f16(self.vals, 2 * nnz), _hip.f32(self.diag, 2 * n),
                _hip.f32(self.dinv, 2 * n), _hip.stream()))
            return
        if self._packed is not None:
            pk, idr = self._packed
            rb, cb = self._rb_used, self._cols16[1]
            _hip.check(_hip.lib().flow_pmg_pack1(
                n, nnz, rb.numel() - 1, _hip.i32(rb),
                _hip.i32(self.rowptr, n + 1, 'rowptr'),
                _hip.i32(self._cols, nnz, 'cols'),
                _hip.i32(self.diag_idx, n, 'diag_idx'),
                _hip.f64(J.plane(0)[k0:k0 + nnz], nnz),
                _hip.f64(J.plane(3)[k0:k0 + nnz], nnz),
                _hip.u8(self.keep, nnz) if self.keep is not None else None,
                _hip.i32(cb), _hip.u8(idr, 2 * n), _hip.i32(pk, nnz),
                _hip.f32(self.diag, 2 * n), _hip.f32(self.dinv, 2 * n),
                _hip.stream()))
            return
        _hip.check(_hip.lib().flow_pmg_pack(
            n, nnz, _hip.i32(self.rowptr, n + 1, 'rowptr'),
            _hip.i32(self.diag_idx, n, 'diag_idx'),
            _hip.f64(J.plane(0)[k0:k0 + nnz], nnz),
            _hip.f64(J.plane(3)[k0:k0 + nnz], nnz),
            _hip.u8(self.keep, nnz) if self.keep is not None else None,
            _hip.f16(self.vals, 2 * nnz), _hip.f32(self.diag, 2 * n),
            _hip.f32(self.dinv, 2 * n), _hip.stream()))

    def lambda_max(self, work, iterations=32, warm_iterations=10):
        '''Spectral radius of D^-1 A by the power method: `iterations` steps
        from a fixed vector the first time, `warm_iterations` from the iterate
        of the previous call afterwards (a rebuild of the same level some time
        steps later: the dominant eigenvector has hardly moved).'''
        first = getattr(self, '_eigvec', None) is None or \
            os.environ.get('FLOW_AMD_PMG_COLD_POWER') == '1'
        if first:
            self._eigvec = torch.zeros(2 * self.n, dtype=torch.float32,
                                       device=device.get())
        res = ctypes.c_double(0.0)
        _hip.check(_hip.lib().flow_pmg_lambda_max(
            ctypes.byref(self.struct),
            int(iterations if first else warm_iterations),
            _hip.f32(work, 6 * self.n),
            _hip.f64(ops.work(_hip.REDUCE_WORK)),
            _hip.f32(self._eigvec, 2 * self.n), ctypes.byref(res),
            _hip.stream()))
        return res.value


class Pmg(object):
    '''The preconditioner for the velocity space W (degree 2).  `refactor`
    takes the assembled Jacobian (kind 2, Dirichlet rows already identity
    rows) on the P2 pattern and the one of the P1 discretisation.

    pre / post / coarse_steps: Chebyshev steps; ratio_fine / ratio_coarse: the
    intervals are [lam_max / ratio, safety * lam_max] with lam_max from the
    power method (tools/precond_lab.py: 2 / 2 / 4 steps and ratios 8 / 8 are a
    flat optimum).'''

    def __init__(self, W, pre=1, post=2, coarse_steps=6, ratio_fine=5.0,
                 ratio_coarse=12.0, safety=1.1, rows=None, vrows=None,
                 scalar=False, coarse_auto=False, coarse_max=48):
        '''rows / vrows = (r0, r1) / (v0, v1): block Jacobi on a strip -- the
        cycle on the diagonal block of those P2 / P1 rows in local numbering,
        couplings that leave the block dropped (flow_amd/parallel.py).
        scalar: W is a scalar P2 space, the operator a scalar one (the heat
        system): `refactor` takes kind-0 matrices, `apply` vectors of W.N
        entries, `set_bcs` scalar dof numbers.
        coarse_auto: the Chebyshev treatment of the P1 level is sized from an
        estimate of its condition number at every `refactor` (`mass_share`):
        the hand-tuned steps at CFL-sized time steps, more of them -- with the
        square root of the estimate, at most coarse_max -- where the level is
        a diffusion problem (dt >> h^2 / nu).'''
        lay = W.layout
        assert lay.degree == 2, 'the p-multigrid needs a P2 space'
        self.scalar = bool(scalar)
        self.coarse_auto = bool(coarse_auto)
        self.coarse_max = int(coarse_max)
        self.coarse_steps0, self.ratio_coarse0 = int(coarse_steps), ratio_coarse
        self.lay = lay
        self.lay1 = scalar_layout(lay.mesh, 1)
        self.rows, self.vrows = rows, vrows
        self.fine = _Level(lay, rows)
        self.coarse = _Level(self.lay1, vrows)
        self.ratio_fine, self.ratio_coarse = ratio_fine, ratio_coarse
        self.safety = safety
        if rows is None:
            ends, rptr, rsrc = transfer_tables(lay)
        else:
            ends, rptr, rsrc = local_transfer_tables(lay, rows, vrows)
        n, n1 = self.fine.n, self.coarse.n
        self._keep = dict(
            ends=device.to_device(ends.reshape(-1)),
            rptr=device.to_device(rptr), rsrc=device.to_device(rsrc),
            bc_fine=torch.zeros(2 * n, dtype=torch.uint8, device=device.get()),
            bc_coarse=torch.zeros(2 * n1, dtype=torch.uint8,
                                  device=device.get()),
            # (+ one zero float2 behind the coarse vectors: the dummy coarse
            # row of dofs whose partner vertex lies outside a block)
            work=torch.zeros(12 * n + 8 * n1 + 2, dtype=torch.float32,
                             device=device.get()),
            )
        k = self._keep
        s = _hip.PmgS()
        s.fine, s.coarse = self.fine.struct, self.coarse.struct
        s.pre, s.post, s.coarse_steps = int(pre), int(post), int(coarse_steps)
        s.ends = _hip.i32(k['ends'], 2 * n, 'ends')
        s.rptr = _hip.i32(k['rptr'], n1 + 1, 'rptr')
        s.rsrc = _hip.i32(k['rsrc'], len(rsrc), 'rsrc')
        s.bc_fine = _hip.u8(k['bc_fine'], 2 * n).value
        s.bc_coarse = _hip.u8(k['bc_coarse'], 2 * n1).value
        s.work = _hip.f32(k['work'], 12 * n + 8 * n1 + 2).value
        s.scalar = int(self.scalar)
        assert k['work'].data_ptr() % 16 == 0
        self.struct = s
        self._bc_key = None
        self.lam = (None, None)

    # -- Dirichlet rows ----------------------------------------------------------
    def coarse_bc_dofs(self, bc_dofs_host):
        '''The Dirichlet dofs of the P1 level (component-blocked numbering of
        the P1 vector space): the vertex dofs among the P2 ones.'''
        lay, n, n1 = self.lay, self.lay.N, self.lay1.N
        vertex_of = numpy.full(n, -1, dtype=numpy.int64)
        vertex_of[lay.vertex_dofs] = numpy.arange(n1)
        d = numpy.asarray(bc_dofs_host, dtype=numpy.int64)
        comp, row = d // n, d % n
        v = vertex_of[row]
        sel = v >= 0
        return numpy.sort(comp[sel] * n1 + v[sel]).astype(numpy.int32)

    def set_bcs(self, bc_dofs_host):
        '''Upload the Dirichlet masks of both levels (once per set).'''
        key = hash(numpy.asarray(bc_dofs_host).tobytes())
        if key == self._bc_key:
            return self._bc1
        n, n1 = self.lay.N, self.lay1.N
        if self.scalar:
            # (both lanes carry the one component: the masks are duplicated;
            # the coarse dof list that is returned is the scalar one)
            d = numpy.asarray(bc_dofs_host, dtype=numpy.int64)
            bc_dofs_host = numpy.concatenate([d, n + d])
        m0 = numpy.zeros(2 * n, dtype=numpy.uint8)
        m0[bc_dofs_host] = 1
        bc1 = self.coarse_bc_dofs(bc_dofs_host)
        m1 = numpy.zeros(2 * n1, dtype=numpy.uint8)
        m1[bc1] = 1
        if self.rows is not None:
            (r0, r1), (v0, v1) = self.rows, self.vrows
            m0 = numpy.concatenate([m0[r0:r1], m0[n + r0:n + r1]])
            m1 = numpy.concatenate([m1[v0:v1], m1[n1 + v0:n1 + v1]])
        self._keep['bc_fine'].copy_(torch.from_numpy(m0))
        self._keep['bc_coarse'].copy_(torch.from_numpy(m1))
        device.synchronize()
        self._bc_key = key
        if self.scalar:
            bc1 = bc1[bc1 < n1]
        self._bc1 = (bc1, device.to_device(bc1) if len(bc1) else None)
        return self._bc1

    # -- numbers -------------------------------------------------------------------
    def refactor(self, J, J1, mass_share=None):
        '''mass_share (coarse_auto): a lower bound for the share of the mass
        term in the diagonal of the P1 operator, min_i (alpha M)_ii / (J1)_ii
        over its free rows.  The operator is the mass term plus a (nearly)
        positive semi-definite rest, so lam_min(D^-1 J1) >= lam_min(D_M^-1 M)
        * mass_share ~ 0.4 * mass_share: the condition number the Chebyshev
        steps on that level face.'''
        self.fine.pack(J)
        self.coarse.pack(J1)
        work = self._keep['work']
        lam0 = self.fine.lambda_max(work)
        lam1 = self.coarse.lambda_max(work)
        self.lam = (lam0, lam1)
        if self.coarse_auto and mass_share is not None:
            # The hand-tuned pair (coarse_steps, ratio_coarse) belongs to
            # CFL-sized steps, where this estimate of the condition number
            # reads ~70 (tools/debug_pmg_auto.py: 69 on the 160 x 37 channel at
            # diffusion number 0.5 -- the bound is pessimistic, the Krylov
            # method deals with the low end).  Up to three times that nothing
            # changes (a longer polynomial over a wider interval is WORSE
            # there: contraction 0.38 instead of 0.24, and it amplifies the
            # complex eigenvalues of the convection term).  Beyond -- dt >>
            # h^2 / nu -- the interval grows with the estimate and the step
            # count with its square root, which keeps the quality of the
            # tuned point; with the count capped the interval is cut to what
            # those steps cover.
            est = self.safety * lam1 / max(0.4 * mass_share, 1e-6)
            scale = est / 70.0
            steps, ratio = self.coarse_steps0, self.ratio_coarse0
            if scale > 3.0:
                steps = int(numpy.ceil(self.coarse_steps0 * numpy.sqrt(scale)))
                steps = min(steps, self.coarse_max)
                ratio = self.ratio_coarse0 * (
                    steps / float(self.coarse_steps0))**2
            self.ratio_coarse = ratio
            self.struct.coarse_steps = steps
        for lvl, lam, ratio in ((self.fine, lam0, self.ratio_fine),
                                (self.coarse, lam1, self.ratio_coarse)):
            lvl.struct.lam_max = self.safety * lam
            lvl.struct.lam_min = lam / ratio
        # (the level structs are embedded BY VALUE in flow_pmg)
        self.struct.fine = self.fine.struct
        self.struct.coarse = self.coarse.struct
        return self

    def apply(self, r, z):
        '''z = M^-1 r (tests / direct use); on a block: vectors of the block's
        rows, component stride = its size.'''
        n2 = (1 if self.scalar else 2) * self.fine.n
        _hip.check(_hip.lib().flow_pmg_apply(
            ctypes.byref(self.struct), _hip.f64(r, n2, 'r'), _hip.f64(z, n2, 'z'),
            _hip.stream()))
        return z
