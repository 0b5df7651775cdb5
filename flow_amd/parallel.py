# -*- coding: utf-8 -*-
'''
Row-block sharding of the pressure-Poisson solve over the GPUs of one node
(nothing in the reference: DOLFIN/PETSc would do this
implicitly under mpirun).

One process per GPU, torch.distributed over RCCL ('nccl' backend) on xGMI.
Rank g owns the contiguous rows [r_g, r_{g+1}) of the P1 stiffness matrix,
balanced by nonzeros.  With the x-major vertex numbering of the channel mesh
the matrix is banded (bandwidth ~ ny), so a rank only needs `halo` entries from
its left and right neighbour.  Per CG iteration (Chronopoulos-Gear single
reduction form) there is ONE neighbour halo exchange of z and ONE all-reduce
carrying the three scalars (r.z, z.w, r.r) and, with the two-level
preconditioner, the partial coarse restriction omega = P^T w of the same
iteration (P^T r is then advanced by the recurrence that mirrors r -= alpha s);
both are latency bound (a few KB / <= 32 KB).
The local pieces are the same HIP kernels as the single-GPU solver, launched on
the owned row range through the C ABI (flow_cg_update_dev, flow_operator_apply
with the rank's row blocks, flow_dot3_dev, flow_cg_scalars_dev).

Every rank keeps full-length vectors (the pressure space is small: 9 MB at
10 M DoF); only the owned slice and the halos are kept current during the
iteration, and the solution is all-gathered at the end because the other
sub-steps are replicated.
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
    involve all of its ranks before point-to-point traffic may start).
    force: take the sharded loop even on a 1-rank group (development: measures
    the loop's host overhead).'''
    _STATE['group'] = group
    _STATE['force'] = bool(force)
    if dist.get_world_size(group) > 1:
        t = torch.zeros(1, dtype=torch.float64, device=device.get()
                        if dist.get_backend(group) != 'gloo' else 'cpu')
        dist.all_reduce(t, group=group)


def disable():
    _STATE['group'] = None


# Rows of the pressure system from which sharding pays.  A sharded CG iteration
# costs two collectives plus ~10 stream-ordered launches whose cost does not
# shrink with the local row count (measured on MI355X: >= 110 us even on a
# 1-rank group, against 65-75 us for a complete single-GPU iteration on 1.1 M
# rows, i.e. ~65 us per million rows), so below a few million rows one GPU is
# faster than eight and the ranks solve the pressure system redundantly.
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


# -- partition (pure host logic, CPU-testable) --------------------------------
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
            self.lo[g] = seg.min()
            self.hi[g] = seg.max() + 1
        for g in range(world):
            # halos must come from the immediate neighbours only
            left = bounds[g - 1] if g > 0 else 0
            right = bounds[g + 2] if g + 2 <= world else n
            assert self.lo[g] >= left and self.hi[g] <= right, \
                'matrix bandwidth exceeds the neighbour row blocks'

    def rows(self, g):
        return int(self.bounds[g]), int(self.bounds[g + 1])

    def exchanges(self, g):
        '''[(peer, send_slice, recv_slice)] of rank g: what it must send to and
        receive from each neighbour before a local SpMV.'''
        r0, r1 = self.rows(g)
        out = []
        if g > 0:
            # left neighbour needs my first rows up to its hi; I need [lo, r0)
            send = (r0, int(max(r0, min(self.hi[g - 1], r1))))
            recv = (int(min(self.lo[g], r0)), r0)
            out.append((g - 1, send, recv))
        if g + 1 < self.world:
            send = (int(min(r1, max(self.lo[g + 1], r0))), r1)
            recv = (r1, int(max(self.hi[g], r1)))
            out.append((g + 1, send, recv))
        return out


class Comm(object):
    '''The two collectives of the solver on a torch.distributed group.
    The gloo backend cannot move device tensors point-to-point, so with gloo
    (CPU tests, single-GPU rehearsals) buffers are staged through the host.'''

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
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def halo_exchange(self, vec, plan):
        '''plan: [(peer, (s0, s1), (r0, r1))] slices of the full-length vec.'''
        ops_ = []
        staged = []
        for peer, (s0, s1), (r0, r1) in plan:
            gp = self.global_rank(peer)
            if s1 > s0:
                buf = vec[s0:s1]
                if self.staged and vec.is_cuda:
                    buf = buf.cpu()
                ops_.append(dist.P2POp(dist.isend, buf, gp, self.group))
            if r1 > r0:
                buf = vec[r0:r1]
                if self.staged and vec.is_cuda:
                    buf = torch.empty(r1 - r0, dtype=vec.dtype)
                    staged.append((buf, r0, r1))
                ops_.append(dist.P2POp(dist.irecv, buf, gp, self.group))
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        for buf, r0, r1 in staged:
            vec[r0:r1].copy_(buf)

    def allgather_rows(self, vec, bounds):
        '''Make the full vector current on every rank (owned slices -> all).'''
        for g in range(self.world):
            r0, r1 = int(bounds[g]), int(bounds[g + 1])
            seg = vec[r0:r1]
            if self.staged and vec.is_cuda:
                h = seg.cpu()
                dist.broadcast(h, self.global_rank(g), group=self.group)
                seg.copy_(h)
            else:
                dist.broadcast(seg, self.global_rank(g), group=self.group)


# -- local kernels ------------------------------------------------------------
class HipLocal(object):
    '''Local pieces of the sharded CG on the HIP path.'''

    def __init__(self, A, r0, r1):
        self.lib = _hip.lib()
        lay = A.layout
        self.A = A
        self.r0, self.r1 = r0, r1
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
        self.work = device.empty(_hip.REDUCE_WORK)

    def zeros(self, n):
        return device.zeros(n)

    def spmv_rows(self, x, y):
        _hip.check(self.lib.flow_operator_apply(
            ctypes.byref(self.op), _hip.f64(x, self.A.size),
            _hip.f64(y, self.A.size), _hip.stream()
            ))

    def residual(self, b, q, dinv, r, z):
        s = slice(self.r0, self.r1)
        _hip.check(self.lib.flow_residual_dev(
            self.r1 - self.r0, _hip.f64(b[s]), _hip.f64(q[s]), _hip.f64(dinv[s]),
            _hip.f64(r[s]), _hip.f64(z[s]), _hip.stream()
            ))

    def dots(self, r, z, w, b, out):
        '''out[0:3] = local (r.z, z.w, r.r); out[3] = b.b if b is given.'''
        s = slice(self.r0, self.r1)
        n = self.r1 - self.r0
        _hip.check(self.lib.flow_dot3_dev(
            n, 3, _hip.f64(r[s]), _hip.f64(z[s]), _hip.f64(z[s]), _hip.f64(w[s]),
            _hip.f64(r[s]), _hip.f64(r[s]), _hip.f64(self.work),
            _hip.f64(out[0:3]), _hip.stream()
            ))
        if b is not None:
            _hip.check(self.lib.flow_dot3_dev(
                n, 1, _hip.f64(b[s]), _hip.f64(b[s]), None, None, None, None,
                _hip.f64(self.work), _hip.f64(out[3:4]), _hip.stream()
                ))

    def scalars(self, first, sums, S):
        _hip.check(self.lib.flow_cg_scalars_dev(
            int(first), _hip.f64(sums), _hip.f64(S), _hip.stream()
            ))

    def update(self, S, dinv, w, z, p, s_, x, r, want_z=True):
        s = slice(self.r0, self.r1)
        _hip.check(self.lib.flow_cg_update_dev(
            self.r1 - self.r0, _hip.f64(S), _hip.f64(dinv[s]), _hip.f64(w[s]),
            _hip.f64(z[s]), _hip.f64(p[s]), _hip.f64(s_[s]), _hip.f64(x[s]),
            _hip.f64(r[s]), int(want_z), _hip.stream()
            ))

    # two-level preconditioner: partial restriction / coarse solve / prolongation
    def coarse_restrict(self, coarse, r, rc):
        _hip.check(self.lib.flow_coarse_restrict_dev(
            ctypes.byref(coarse.struct), _hip.f64(r, coarse.n), self.r0, self.r1,
            _hip.f64(rc, coarse.nc), _hip.stream()
            ))

    def coarse_solve(self, coarse, rc, zc):
        _hip.check(self.lib.flow_coarse_solve_dev(
            ctypes.byref(coarse.struct), _hip.f64(rc, coarse.nc),
            _hip.f64(zc, coarse.nc), _hip.stream()
            ))

    def coarse_recur(self, coarse, S, omega, sigma, rc):
        _hip.check(self.lib.flow_coarse_recur_dev(
            coarse.nc, _hip.f64(S), _hip.f64(omega, coarse.nc),
            _hip.f64(sigma, coarse.nc), _hip.f64(rc, coarse.nc), _hip.stream()
            ))

    def coarse_prolong(self, coarse, dinv, r, zc, z):
        _hip.check(self.lib.flow_coarse_prolong_dev(
            ctypes.byref(coarse.struct), _hip.f64(dinv, coarse.n),
            _hip.f64(r, coarse.n), _hip.f64(zc, coarse.nc), _hip.f64(z, coarse.n),
            self.r0, self.r1, _hip.stream()
            ))


def sharded_cg(local, comm, part, b, x, dinv, rtol, atol, maxit, check_every,
               coarse=None):
    '''Chronopoulos-Gear CG on the row partition `part`.  `local` provides the
    kernels (HipLocal in the product; the CPU tests inject a numpy stand-in to
    exercise the partition + communication logic under gloo).  Returns
    (iterations, residual norm); raises _hip.NotConverged.'''
    g = comm.rank
    plan = part.exchanges(g)
    n = part.n
    r = local.zeros(n)
    z = local.zeros(n)
    w = local.zeros(n)
    p = local.zeros(n)
    s = local.zeros(n)
    S = local.zeros(16)
    # ONE all-reduce per iteration: [r.z, z.w, r.r, b.b | omega = P^T w]
    nc = coarse.nc if coarse is not None else 0
    buf = local.zeros(4 + nc)
    sums = buf[0:4]
    if coarse is not None:
        omega = buf[4:4 + nc]
        rc = local.zeros(nc)
        zc = local.zeros(nc)
        sigma = local.zeros(nc)

    def precondition():
        # z = D^-1 r + P Ac^-1 rc with rc = P^T r kept current by recurrence
        # (rc -= alpha (omega + beta sigma), mirroring r -= alpha s); the dense
        # coarse solve is replicated on every rank
        local.coarse_solve(coarse, rc, zc)
        local.coarse_prolong(coarse, dinv, r, zc, z)

    def reduce_and_scalars(first, with_b):
        local.dots(r, z, w, b if with_b else None, sums)
        if coarse is not None:
            local.coarse_restrict(coarse, w, omega)
        comm.allreduce_sum(buf)
        local.scalars(first, sums, S)
        if coarse is not None:
            local.coarse_recur(coarse, S, omega, sigma, rc)

    comm.halo_exchange(x, plan)
    local.spmv_rows(x, w)
    local.residual(b, w, dinv, r, z)
    if coarse is not None:
        # the only separate coarse all-reduce: rc_0 = P^T r_0
        local.coarse_restrict(coarse, r, rc)
        comm.allreduce_sum(rc)
        precondition()
    comm.halo_exchange(z, plan)
    local.spmv_rows(z, w)
    reduce_and_scalars(True, True)
    host = sums.cpu()
    b2 = float(host[3])
    res2 = float(host[2])
    sums[3:4].zero_()       # the slot rides along in every later all-reduce
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
        for _ in range(todo):
            local.update(S, dinv, w, z, p, s, x, r, coarse is None)
            if coarse is not None:
                precondition()
            comm.halo_exchange(z, plan)
            local.spmv_rows(z, w)
            reduce_and_scalars(False, False)
        it += todo
        res2 = float(sums[2].item())
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
    r0, r1 = part.rows(comm.rank)
    lkey = (id(A), comm.world, comm.rank)
    if lkey not in _PART_CACHE:
        _PART_CACHE[lkey] = HipLocal(A, r0, r1)
    local = _PART_CACHE[lkey]
    its, res = sharded_cg(local, comm, part, b, x, dinv, rtol, atol, maxit,
                          check_every, coarse)
    return SolveInfo(its, res, 'cg%s[row-sharded x%d]' % (
        '+2level' if coarse is not None else '', comm.world))
