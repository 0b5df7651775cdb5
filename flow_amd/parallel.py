# -*- coding: utf-8 -*-
'''
Row-block sharding of the pressure-Poisson solve over the GPUs of one node
(nothing in the reference: DOLFIN/PETSc would do this
implicitly under mpirun).

One process per GPU, torch.distributed over RCCL ('nccl' backend) on xGMI.
Rank g owns the contiguous rows [r_g, r_{g+1}) of the P1 stiffness matrix,
balanced by nonzeros.  With the x-major vertex numbering of the channel mesh
the matrix is banded (bandwidth ~ ny), so the columns a rank's rows reference
(its ghost rows) belong to its left and right neighbour only.

Chronopoulos-Gear single-reduction CG with exactly ONE collective per
iteration.  Every rank keeps x, r, p, s, z current on its ghost rows as well:
their updates are pointwise, so all they need there is w = A z, which the
owners publish.  The all-reduced buffer

    [ r.z, z.w, r.r, 0 | omega = P^T w | boundary entries of w, rank by rank ]

therefore carries the dot products, the coarse restriction of the two-level
preconditioner (P^T r is advanced by the recurrence that mirrors r -= alpha s)
AND the halo: each rank fills its own partial sums and its own boundary slots,
zeros elsewhere, and the sum is the concatenation.  No point-to-point traffic,
no second synchronisation point; everything between two all-reduces is one
library call (flow_cg_shard_step, include/flow_hip.h).  The message is a few
tens of KB: latency bound.

Every rank keeps full-length vectors (the pressure space is small: 9 MB at
10 M DoF); the start (r = b - A x, z = M^-1 r) is computed redundantly on all
rows, and the solution is all-gathered at the end because the other sub-steps
are replicated.
'''
import ctypes
import os

import numpy
import torch
import torch.distributed as dist

from . import _hip
from . import device
from .fem.space import csr_stream_rowblocks

_STATE = {'group': None}


def enable(group, force=False):
    '''Shard subsequent pressure solves over `group`.  Collective: every rank
    of the group must call it (the first NCCL operation on a group has to
    involve all of its ranks).
    force: take the sharded loop even on a 1-rank group or for systems below
    min_rows() (development / tests).'''
    _STATE['group'] = group
    _STATE['force'] = bool(force)
    if dist.get_world_size(group) > 1:
        t = torch.zeros(1, dtype=torch.float64, device=device.get()
                        if dist.get_backend(group) != 'gloo' else 'cpu')
        dist.all_reduce(t, group=group)


def disable():
    _STATE['group'] = None


# Rows of the pressure system from which sharding pays.  A sharded CG iteration
# is latency bound: one all-reduce plus ~10 stream-ordered launches whose cost
# does not shrink with the local row count (~5 us each), and the dense coarse
# solve is replicated; a complete single-GPU iteration on the 1.1 M-row system
# of the headline workload takes 72 us (~65 us per million rows).  Below a few
# million rows one GPU is as fast as eight.
DEFAULT_MIN_ROWS = 4000000


def min_rows():
    return int(os.environ.get('FLOW_AMD_SHARD_MIN_ROWS', DEFAULT_MIN_ROWS))


def active(nrows=None):
    '''Is the pressure solve sharded?  nrows: size of the system about to be
    solved -- the `auto` policy shards only from min_rows() on.'''
    if _STATE['group'] is None:
        return False
    if _STATE.get('force', False):
        return True
    if dist.get_world_size(_STATE['group']) <= 1:
        return False
    return nrows is None or nrows >= min_rows()


def describe(world, nrows):
    '''One line for bench.py's `config.parallelism`.'''
    if active(nrows):
        return 'pressure-poisson row-block x%d' % world
    if world == 1:
        return 'single GPU'
    return ('replicated x%d (pressure system of %d rows is below the sharding '
            'threshold of %d rows: DESIGN.md section 6; --shard always forces '
            'it)' % (world, nrows, min_rows()))


# -- partition (pure host logic, CPU-testable) --------------------------------
class HaloLayout(object):
    '''What rank g puts into / takes out of the halo section of the all-reduce
    buffer.  Index 0 = left neighbour, 1 = right neighbour; len 0 = none.'''

    def __init__(self):
        self.e0 = self.e1 = 0
        self.nhalo = 0
        self.send_row = [0, 0]
        self.send_len = [0, 0]
        self.send_slot = [0, 0]
        self.recv_row = [0, 0]
        self.recv_len = [0, 0]
        self.recv_slot = [0, 0]


class Partition(object):
    '''Row-block partition of a CSR pattern, balanced by nonzeros.'''

    def __init__(self, rowptr, cols, world):
        rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
        n = len(rowptr) - 1
        nnz = int(rowptr[-1])
        assert 1 <= world <= n
        targets = (numpy.arange(1, world) * nnz) // world
        cuts = numpy.searchsorted(rowptr, targets, side='left')
        bounds = numpy.concatenate([[0], cuts, [n]]).astype(numpy.int64)
        # strictly increasing (tiny problems)
        for g in range(1, world + 1):
            bounds[g] = max(bounds[g], bounds[g - 1] + 1)
        bounds[world] = n
        assert (numpy.diff(bounds) > 0).all(), 'more ranks than rows'
        self.n = n
        self.world = world
        self.bounds = bounds
        # columns referenced by each rank's rows
        self.lo = numpy.empty(world, dtype=numpy.int64)
        self.hi = numpy.empty(world, dtype=numpy.int64)
        cols = numpy.asarray(cols)
        for g in range(world):
            seg = cols[rowptr[bounds[g]]:rowptr[bounds[g + 1]]]
            self.lo[g] = min(seg.min(), bounds[g])
            self.hi[g] = max(seg.max() + 1, bounds[g + 1])
        for g in range(world):
            # ghost rows must belong to the immediate neighbours only
            left = bounds[g - 1] if g > 0 else 0
            right = bounds[g + 2] if g + 2 <= world else n
            assert self.lo[g] >= left and self.hi[g] <= right, \
                'matrix bandwidth exceeds the neighbour row blocks'

    def rows(self, g):
        return int(self.bounds[g]), int(self.bounds[g + 1])

    def sends(self, g):
        '''[(row, len)] for the left and the right neighbour: the owned rows
        of rank g that are ghost rows there.'''
        r0, r1 = self.rows(g)
        left = (r0, 0)
        right = (r1, 0)
        if g > 0:
            end = int(min(self.hi[g - 1], r1))
            left = (r0, max(end - r0, 0))
        if g + 1 < self.world:
            start = int(max(self.lo[g + 1], r0))
            right = (start, max(r1 - start, 0))
        return [left, right]

    def halo_layout(self, g):
        '''Slots of the halo section: rank by rank, [to-left | to-right].'''
        slot = {}
        off = 0
        for q in range(self.world):
            for side, (row, ln) in enumerate(self.sends(q)):
                slot[(q, side)] = (off, row, ln)
                off += ln
        lay = HaloLayout()
        lay.nhalo = off
        lay.e0, lay.e1 = int(self.lo[g]), int(self.hi[g])
        for side in (0, 1):
            o, row, ln = slot[(g, side)]
            lay.send_row[side], lay.send_len[side], lay.send_slot[side] = \
                row, ln, o
        r0, r1 = self.rows(g)
        if g > 0:
            # my left ghost rows = what the left neighbour sends to ITS right
            o, row, ln = slot[(g - 1, 1)]
            assert row == lay.e0 and row + ln == r0
            lay.recv_row[0], lay.recv_len[0], lay.recv_slot[0] = row, ln, o
        else:
            assert lay.e0 == r0
        if g + 1 < self.world:
            o, row, ln = slot[(g + 1, 0)]
            assert row == r1 and row + ln == lay.e1
            lay.recv_row[1], lay.recv_len[1], lay.recv_slot[1] = row, ln, o
        else:
            assert lay.e1 == r1
        return lay


class Comm(object):
    '''The collectives of the solver on a torch.distributed group.  The gloo
    backend cannot reduce device tensors, so with gloo (CPU tests, single-GPU
    rehearsals) buffers are staged through the host.'''

    def __init__(self, group):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) == 'gloo'

    def global_rank(self, r):
        return dist.get_global_rank(self.group, r) \
            if self.group is not dist.group.WORLD else r

    def allreduce_sum(self, t):
        if self.staged and t.is_cuda:
            # (to_host synchronises first: see device.to_host)
            h = device.to_host(t)
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            # RCCL: enqueued behind the kernels of the current stream by
            # torch's event hand-over, and the stream waits for its result
            dist.all_reduce(t, group=self.group)
        return t

    def allgather_rows(self, vec, bounds):
        '''Make the full vector current on every rank (owned slices -> all).'''
        for g in range(self.world):
            r0, r1 = int(bounds[g]), int(bounds[g + 1])
            seg = vec[r0:r1]
            if self.staged and vec.is_cuda:
                h = device.to_host(seg)
                dist.broadcast(h, self.global_rank(g), group=self.group)
                seg.copy_(h)
            else:
                dist.broadcast(seg, self.global_rank(g), group=self.group)


# -- local kernels ------------------------------------------------------------
class HipLocal(object):
    '''The local side of the sharded CG on the HIP path: owns the vectors and
    the flow_cg_shard context; `step` is one library call.'''

    def __init__(self, A, dinv, coarse, part, rank):
        self.lib = _hip.lib()
        lay = A.layout
        n = lay.N
        self.A = A
        self.dinv = dinv
        self.coarse = coarse
        self.n = n
        r0, r1 = part.rows(rank)
        self.r0, self.r1 = r0, r1
        hl = part.halo_layout(rank)
        rowptr = lay.pattern('rowptr').astype(numpy.int64)
        rb = csr_stream_rowblocks(rowptr[r0:r1 + 1] - rowptr[r0]) + r0
        self.rowblocks = device.to_device(rb.astype(numpy.int32))
        base = A.operator()
        op = _hip.Operator()
        ctypes.memmove(ctypes.byref(op), ctypes.byref(base),
                       ctypes.sizeof(_hip.Operator))
        op.rowblocks = _hip.i32(self.rowblocks)
        op.nblocks = len(rb) - 1
        self.op = op
        nc = coarse.nc if coarse is not None else 0
        lda = coarse.struct.lda if coarse is not None else 0
        z = device.zeros
        self.r, self.z, self.w, self.p, self.s = z(n), z(n), z(n), z(n), z(n)
        self.S = z(16)
        self.buf = z(4 + nc + hl.nhalo)
        self.work = device.empty(_hip.REDUCE_WORK)
        self.rc, self.zc, self.sigma = z(max(lda, 1)), z(max(lda, 1)), \
            z(max(lda, 1))
        self.x = None
        c = _hip.CgShard()
        c.A = ctypes.pointer(self.op)
        c.dinv = _hip.f64(dinv, n, 'dinv')
        if coarse is not None:
            c.coarse = ctypes.pointer(coarse.struct)
        c.n, c.r0, c.r1, c.e0, c.e1 = n, r0, r1, hl.e0, hl.e1
        c.nhalo = hl.nhalo
        for side in (0, 1):
            c.send_row[side] = hl.send_row[side]
            c.send_len[side] = hl.send_len[side]
            c.send_slot[side] = hl.send_slot[side]
            c.recv_row[side] = hl.recv_row[side]
            c.recv_len[side] = hl.recv_len[side]
            c.recv_slot[side] = hl.recv_slot[side]
        for name in ('r', 'z', 'w', 'p', 's', 'rc', 'zc', 'sigma', 'S', 'buf',
                     'work'):
            setattr(c, name, _hip.f64(getattr(self, name)))
        self.ctx = c

    def _start(self, b, q):
        '''r = b - q, z = B r on ALL rows (replicated).'''
        lib, n, st = self.lib, self.n, _hip.stream()
        _hip.check(lib.flow_residual_dev(
            n, _hip.f64(b, n), _hip.f64(q), _hip.f64(self.dinv),
            _hip.f64(self.r), _hip.f64(self.z), st
            ))
        if self.coarse is not None:
            cs = ctypes.byref(self.coarse.struct)
            _hip.check(lib.flow_coarse_restrict_dev(
                cs, _hip.f64(self.r), 0, n, _hip.f64(self.rc), st
                ))
            _hip.check(lib.flow_coarse_solve_dev(
                cs, _hip.f64(self.rc), _hip.f64(self.zc), st
                ))
            _hip.check(lib.flow_coarse_prolong_dev(
                cs, _hip.f64(self.dinv), _hip.f64(self.r), _hip.f64(self.zc),
                _hip.f64(self.z), 0, n, st
                ))

    def begin(self, b, x):
        '''Replicated start on ALL rows: r = b - A x, z = B r, p = s = 0.
        Returns |B b|^2 (the stopping test is in the preconditioned norm, as
        in flow_cg_solve).'''
        from .fem import ops
        n = self.n
        self.x = x
        self.ctx.x = _hip.f64(x, n, 'x')
        for v in (self.p, self.s, self.S, self.sigma, self.buf, self.w):
            _hip.fill(v, 0.0)
        self._start(b, self.w)
        bb2 = ops.dot(self.z, self.z)
        self.A.apply(x, self.w)
        self._start(b, self.w)
        return bb2

    def step(self, phase):
        _hip.check(self.lib.flow_cg_shard_step(
            ctypes.byref(self.ctx), int(phase), _hip.stream()
            ))

    def res2(self):
        return float(device.to_host(self.buf[2:3])[0])


def sharded_cg(local, comm, part, b, x, rtol, atol, maxit, check_every):
    '''Chronopoulos-Gear CG on the row partition `part`.  `local` provides the
    kernels (HipLocal in the product; the CPU tests inject a numpy stand-in to
    exercise the partition + communication logic under gloo).  Returns
    (iterations, residual norm); raises _hip.NotConverged.'''
    # The replicated start needs bitwise identical b and x on every rank.  The
    # ranks compute them redundantly (deterministic kernels, so they agree),
    # but a sharded solve must not depend on that: take every row from its
    # owner.
    comm.allgather_rows(b, part.bounds)
    comm.allgather_rows(x, part.bounds)
    b2 = local.begin(b, x)
    local.step(0)
    comm.allreduce_sum(local.buf)
    res2 = local.res2()
    target = max(rtol * numpy.sqrt(b2), atol)
    it = 0
    while True:
        if res2 != res2:
            raise _hip.NotConverged('sharded CG broke down (NaN residual)')
        if numpy.sqrt(res2) <= target:
            break
        if it >= maxit:
            raise _hip.NotConverged(
                'sharded CG did not converge in %d iterations: |r| = %.3e > %.3e'
                % (it, numpy.sqrt(res2), target)
                )
        todo = min(check_every, maxit - it)
        for k in range(todo):
            local.step(1 if it + k == 0 else 2)
            comm.allreduce_sum(local.buf)
        it += todo
        res2 = local.res2()
    comm.allgather_rows(x, part.bounds)
    return it, float(numpy.sqrt(res2))


_PART_CACHE = {}


def pressure_cg(A, dinv, coarse, b, x, rtol, atol, maxit, check_every):
    '''Sharded replacement of ops.krylov_solve('cg', ...) for the pressure
    system (called from navier_stokes._compute_pressure when enabled).'''
    from .fem.ops import SolveInfo
    comm = Comm(_STATE['group'])
    lay = A.layout
    key = (id(lay), comm.world)
    if key not in _PART_CACHE:
        _PART_CACHE[key] = Partition(
            lay.pattern('rowptr'), lay.pattern('cols'), comm.world
            )
    part = _PART_CACHE[key]
    lkey = (id(A), id(dinv), id(coarse), comm.world, comm.rank)
    if lkey not in _PART_CACHE:
        # keep the keyed objects alive: ids must not be recycled
        _PART_CACHE[lkey] = (HipLocal(A, dinv, coarse, part, comm.rank),
                             A, dinv, coarse)
    local = _PART_CACHE[lkey][0]
    its, res = sharded_cg(local, comm, part, b, x, rtol, atol, maxit,
                          check_every)
    return SolveInfo(its, res, 'cg%s[row-sharded x%d]' % (
        '+2level' if coarse is not None else '', comm.world))
