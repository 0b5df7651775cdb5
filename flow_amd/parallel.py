# -*- coding: utf-8 -*-
'''
Domain decomposition of the WHOLE pressure-correction step over the GPUs of
one node (nothing in the reference: DOLFIN/PETSc would do
this implicitly under mpirun).

One process per GPU, torch.distributed over RCCL ('nccl' backend) on xGMI.  The
channel is cut into strips along x.  With the x-major vertex numbering of the
mesh every scalar space splits into contiguous row blocks: rank g OWNS the
vertices [v_g, v_{g+1}), the P1 rows of those vertices, the P2 rows whose lowest
vertex is one of them, and works on the cells that touch an owned vertex.  What
its rows are coupled to beyond that -- matrix columns, dofs of those cells --
are GHOST rows, a vertex column wide, owned by the left and right neighbour.

Every sub-step runs on the strips (flow_amd/navier_stokes):
  tentative velocity   residual and matrix-free Jacobian action over the rank's
                       cells, rows gathered for the owned dofs; GMRES with the
                       rank's own ILU(0) (block Jacobi: the factor of the
                       diagonal block of the owned rows, no communication)
  pressure             CG with the SAME smoothed-aggregation V-cycle as on one
                       GPU: finest level row-sharded, coarse residual summed
                       over the ranks, the small levels replicated
  velocity correction, step-size projection
                       Jacobi-CG on the P2 mass matrix, one collective per
                       iteration

ONE communication primitive carries everything (flow_comm in
include/flow_hip.h): an all-reduce (sum) of the head of a device buffer.  Dot
products, the partial coarse residuals of the V-cycle and the halos travel in
it -- for a halo every rank writes its boundary rows into its own slots and
zeros into everybody else's, so the sum is the concatenation and the ghost
values are bitwise the owners'.  The Krylov loops live in the library
(flow_shard_*_solve) and call back into `Comm._allreduce` below, which hands
the buffer to torch.distributed.all_reduce: RCCL on the stream the kernels run
on, or -- gloo backend, for CPU tests and several-ranks-on-one-GPU rehearsals --
staged through the host.

Fields stay global-length on every rank (memory is not the scarce resource:
288 GB), valid on the owned + ghost rows; `gather_field` makes one whole (tests,
output).
'''
import atexit
import ctypes
import os
import traceback

import numpy
import torch
import torch.distributed as dist

from . import _hip
from . import device
from .fem.space import csr_stream_rowblocks

_STATE = {'group': None, 'force': False, 'comm': None}


def enable(group, force=False):
    '''Run subsequent steps on the strips of `group`.  Collective: every rank
    of the group must call it (the first NCCL operation on a group has to
    involve all of its ranks).  force: also on a 1-rank group (development /
    tests: measures the host overhead of the sharded loops).'''
    _STATE['group'] = group
    _STATE['force'] = bool(force)
    _STATE['comm'] = Comm(group)
    # the first collective brings the communicator up: here, not inside a
    # solver loop
    t = torch.zeros(1, dtype=torch.float64, device=device.get()
                    if dist.get_backend(group) != 'gloo' else 'cpu')
    dist.all_reduce(t, group=group)
    device.synchronize()
    # halos from neighbour to neighbour instead of through the all-reduce
    # (Comm.enable_peer): opt-in -- the path has only ever run between
    # processes that share one device (the development box has one GPU)
    if os.environ.get('FLOW_AMD_PEER_HALO', '0') == '1':
        _STATE['comm'].enable_peer()


def disable():
    c = _STATE['comm']
    if c is not None:
        c.close()
    _STATE['group'] = None
    _STATE['comm'] = None


@atexit.register
def _close_at_exit():
    c = _STATE.get('comm')
    if c is not None:
        c.close()


def active(nrows=None):
    '''Does the step run on strips?  (nrows: kept for callers that ask about
    one system; the policy no longer depends on it.)'''
    if _STATE['group'] is None:
        return False
    return _STATE['force'] or dist.get_world_size(_STATE['group']) > 1


def comm():
    return _STATE['comm']


def describe(world, nrows=None):
    '''One line for bench.py's `config.parallelism`: names the solvers the
    strips are actually bound to (the Newton preconditioner is whatever the
    last step ran with: navier_stokes.last_step_info).'''
    if active():
        from .navier_stokes import pressure_correction as pc
        pre = pc.last_step_info.get('newton_preconditioner')
        if pre is None:
            pre = '%s (block Jacobi)' % pc.solver_parameters['newton'].get(
                'preconditioner', 'ilu0')
        return ('x-strips x%d: every sub-step sharded (GMRES + %s, row-sharded '
                'V-cycle CG, %s mass solves); collectives: %s' % (
                    world, pre,
                    'Jacobi-CG (strips thinner than the mass solver\'s halo)'
                    if _MASS_FALLBACK[0] else MASS_SOLVER_ON_STRIPS,
                    'ncclAllReduce issued by the library on its stream'
                    if comm().direct is not None else
                    'torch.distributed.all_reduce (%s)' % (
                        'gloo, host-staged' if comm().staged else 'RCCL')))
    if world == 1:
        return 'single GPU'
    return 'replicated x%d' % world


# collectives per iteration of the sharded V-cycle CG: 1 (default: the halo of
# w two layers deep, z formed on the first ghost layer by the rank itself), 2
# (the halo of z exchanged), 3 (round 3: the coarse residual too)
MGCG_COLLECTIVES = int(os.environ.get('FLOW_AMD_MGCG_COLLECTIVES', '1'))

# what parallel.cg-based mass solves are (bench.py's parallelism line)
MASS_SOLVER_ON_STRIPS = 'defect-correction (deep halo)'
_MASS_FALLBACK = [False]      # mass_solve fell back to Jacobi-CG (thin strips)


# -- communicator ---------------------------------------------------------------
class Comm(object):
    '''The exchange buffer and the all-reduce callback of flow_comm.'''

    def __init__(self, group, capacity=1 << 16):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) == 'gloo'
        self.calls = 0
        self._cb = _hip.ALLREDUCE_FN(self._allreduce)
        self.buf = None
        self.struct = None
        self.direct = None          # RcclBinding when the library calls RCCL
        self._rccl_comm = None
        self.failed = False         # a collective failed: no further ones
        self.peer = None            # _hip.PeerS: halos from neighbour to neighbour
        self._peer_maps = []
        self._peer_base = None
        self.ensure(capacity)
        # The all-reduce issued by the library itself (no Python, no event
        # hand-over per collective) is OPT-IN: it has only ever run on 1-rank
        # groups (the development box has one GPU), torch.distributed's
        # all_reduce is the path every multi-rank test goes through.
        if not self.staged and device.on_gpu() and \
                os.environ.get('FLOW_AMD_RCCL_DIRECT', '0') == '1':
            self._bind_rccl()

    # -- halos from neighbour to neighbour (flow_peer, csrc/la_kernels.hip) -----
    def enable_peer(self, land_cap=1 << 20, spin_limit=2000000, selftest=24):
        '''Map the neighbours' landing buffers (hipIpc: xGMI peers on a node,
        or processes sharing one device in a rehearsal) and take the halos off
        the all-reduce: a pure halo then costs no collective, a [sums | halo]
        exchange one of <= 8 doubles.  Every rank tries; a self-test of
        `selftest` exchanges with known data (both directions, both buffer
        parities, a deliberately late rank) must pass on EVERY rank --
        agreed on through an all-reduce -- or all of them stay on the
        all-reduce path.  Returns whether the peer path is on.'''
        if self.world == 1 or not device.on_gpu():
            return False
        lib = _hip.lib()

        def all_ok(ok):
            flag = torch.tensor([float(ok)], dtype=torch.float64)
            if not self.staged:
                flag = flag.to(device.get())
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return float(flag.item()) >= 1.0
        base = ctypes.c_void_p(None)
        handle = ctypes.create_string_buffer(64)
        ok = 1
        try:
            _hip.check(lib.flow_peer_alloc(int(land_cap), ctypes.byref(base),
                                           handle))
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
            ok = 0
        if not all_ok(ok):
            if ok:
                lib.flow_peer_free(base)
            return False
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw), group=self.group)
        maps = [None, None]
        try:
            for sd, nb in ((0, self.rank - 1), (1, self.rank + 1)):
                if 0 <= nb < self.world:
                    m = ctypes.c_void_p(None)
                    _hip.check(lib.flow_peer_open(handles[nb], ctypes.byref(m)))
                    maps[sd] = m
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
            ok = 0
        if not all_ok(ok):
            for m in maps:
                if m is not None:
                    lib.flow_peer_close(m)
            lib.flow_peer_free(base)
            return False
        fbytes = 8 * _hip.PEER_FLAGS
        peer = _hip.PeerS()
        peer.flags = base.value
        peer.land = base.value + fbytes
        peer.land_cap, peer.spin_limit = int(land_cap), int(spin_limit)
        for sd in (0, 1):
            if maps[sd] is not None:
                peer.nb_flags[sd] = maps[sd].value
                peer.nb_land[sd] = maps[sd].value + fbytes
        self._peer_seq = (ctypes.c_ulonglong * 5)()
        peer.seq_host = ctypes.pointer(self._peer_seq)
        self._peer_base, self._peer_maps = base, [m for m in maps if m is not None]
        self.peer = peer
        self.struct.peer = ctypes.pointer(peer)
        good = 1
        try:
            good = int(self._peer_selftest(selftest))
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
            good = 0
        if not all_ok(good):
            self.disable_peer()
            return False
        return True

    def _peer_selftest(self, rounds):
        '''Pure halos of a tiny 1-D decomposition (4 owned rows per rank, 2
        ghost rows per side) with data that names rank, round and row; odd
        ranks sleep before every third round so that their neighbours run
        ahead into the flags.'''
        import time
        w = self.world
        n = 4 * w
        bounds = numpy.arange(w + 1) * 4
        lo = numpy.maximum(bounds[:-1] - 2, 0)
        hi = numpy.minimum(bounds[1:] + 2, n)
        rows = RowBlocks(n, bounds, lo, hi).struct(self.rank)
        self.ensure(2 * rows.nhalo + 8)
        x = device.zeros(2 * n)
        idx = torch.arange(n, dtype=torch.float64, device=x.device)
        own = slice(rows.r0, rows.r1)
        for k in range(rounds):
            if k % 3 == 2 and self.rank % 2 == 1:
                device.synchronize()
                time.sleep(0.02)
            x.fill_(-1.0)
            for a in (0, 1):
                x[a * n:(a + 1) * n][own] = \
                    (1000.0 * k + 100.0 * a + idx)[own]
            _hip.check(_hip.lib().flow_shard_halo(
                ctypes.byref(self.struct), ctypes.byref(rows), 2,
                _hip.f64(x, 2 * n), n, _hip.stream()))
            got = device.to_host(x).numpy()
            for a in (0, 1):
                want = 1000.0 * k + 100.0 * a + numpy.arange(n)
                seg = got[a * n:(a + 1) * n]
                if not numpy.array_equal(seg[rows.e0:rows.e1],
                                         want[rows.e0:rows.e1]):
                    return False
        return self.peer_error() == 0

    def peer_error(self):
        '''The error word of this rank's peer block (0: every wait was served;
        else (sequence number << 8) | what timed out).'''
        if self.peer is None:
            return 0
        err = ctypes.c_ulonglong(0)
        _hip.check(_hip.lib().flow_peer_status(
            ctypes.byref(self.peer), ctypes.byref(err), _hip.stream()))
        return int(err.value)

    def disable_peer(self):
        '''Back to halos in the all-reduce; unmap and free (collective in
        spirit: call it on every rank, neighbours first unmap, then owners
        free).'''
        if self.peer is None and self._peer_base is None:
            return
        lib = _hip.load_library()
        self.peer = None
        if self.struct is not None:
            self.struct.peer = None
        try:
            device.synchronize()
            for m in self._peer_maps:
                lib.flow_peer_close(m)
            self._peer_maps = []
            if dist.is_initialized():
                dist.barrier(group=self.group)
            if self._peer_base is not None:
                lib.flow_peer_free(self._peer_base)
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
        self._peer_base = None

    def close(self):
        '''Destroy the library's own RCCL communicator (if one was made).'''
        self.disable_peer()
        comm, self._rccl_comm = self._rccl_comm, None
        if comm is not None:
            self.direct = None
            try:
                device.synchronize()
                _hip.load_library().flow_rccl_comm_destroy(comm)
            except Exception:                                  # noqa: BLE001
                traceback.print_exc()

    def ensure(self, capacity):
        '''Grow the exchange buffer to at least `capacity` doubles.'''
        if self.buf is None or self.buf.numel() < capacity:
            self.buf = device.zeros(int(capacity) + (int(capacity) & 1))
            assert self.buf.data_ptr() % 16 == 0
            # (on the CPU -- host-logic tests -- the buffer is a host tensor and
            # the struct is only ever handed to the numpy stand-ins)
            self.struct = _hip.CommS(
                self.rank, self.world, ctypes.c_void_p(self.buf.data_ptr()),
                self.buf.numel(), self._cb, None,
                ctypes.pointer(self.peer) if getattr(self, 'peer', None)
                is not None else None)
            if self.direct is not None:
                self._point_struct_at_rccl()
        if self.direct is not None:
            # the solvers enqueue on torch's CURRENT stream, whatever it was
            # when the binding was made: ncclAllReduce must run on the same one
            self.direct.stream = ctypes.c_void_p(device.stream_handle())
        if self.failed:
            raise _hip.HipError(
                'a collective of this communicator failed earlier: the ranks '
                'are no longer in step, no further collectives are issued')
        return self.struct

    # -- the all-reduce issued by the library itself (csrc/rccl_direct.hip) -----
    def _bind_rccl(self):
        '''Create a communicator of the library's own on the RCCL instance
        torch has loaded and bind flow_comm's callback to ncclAllReduce on the
        solvers' stream: no Python, no event hand-over per collective.  Every
        rank tries; the binding is only used if EVERY rank succeeded (agreed
        on through torch's all-reduce), otherwise all stay on the torch path.'''
        lib = _hip.lib()

        def all_ok(ok):
            flag = torch.tensor([float(ok)], dtype=torch.float64,
                                device=device.get())
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return float(flag.item()) >= 1.0

        # 1. everybody loads the library, rank 0 draws the unique id; nobody
        #    goes on unless everybody got this far (a rank that dropped out
        #    here would leave the others waiting in the steps below)
        ok = 1
        raw = ctypes.create_string_buffer(128)
        try:
            path = os.path.join(os.path.dirname(torch.__file__), 'lib',
                                'librccl.so')
            _hip.check(lib.flow_rccl_load(path.encode()))
            if self.rank == 0:
                _hip.check(lib.flow_rccl_unique_id(
                    ctypes.cast(raw, ctypes.c_void_p)))
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
            ok = 0
        if not all_ok(ok):
            return
        # 2. the id travels through torch.distributed
        dev_id = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8) \
            .to(device.get())
        src = dist.get_global_rank(self.group, 0) \
            if self.group is not dist.group.WORLD else 0
        dist.broadcast(dev_id, src, group=self.group)
        raw = ctypes.create_string_buffer(
            bytes(dev_id.cpu().numpy().tobytes()), 128)
        # 3. the communicator (collective)
        comm = ctypes.c_void_p(None)
        try:
            _hip.check(lib.flow_rccl_comm_create(
                ctypes.cast(raw, ctypes.c_void_p), self.rank, self.world,
                ctypes.byref(comm)))
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
            ok = 0
        if not all_ok(ok):
            if ok:
                lib.flow_rccl_comm_destroy(comm)
            return
        self._rccl_comm = comm
        binding = _hip.RcclBinding(
            comm, ctypes.c_void_p(self.buf.data_ptr()),
            ctypes.c_void_p(device.stream_handle()))
        # 4. a reduction whose result is known, through the binding itself:
        #    every rank contributes (rank + 1, 1); used only if every rank got
        #    (world (world + 1) / 2, world)
        self.buf[:2] = torch.tensor([self.rank + 1.0, 1.0], dtype=torch.float64,
                                    device=self.buf.device)
        device.synchronize()
        rc = lib.flow_rccl_allreduce(
            ctypes.cast(ctypes.pointer(binding), ctypes.c_void_p), 2)
        device.synchronize()
        got = device.to_host(self.buf[:2]).numpy()
        ok = int(rc == 0 and got[0] == 0.5 * self.world * (self.world + 1)
                 and got[1] == float(self.world))
        self.buf[:2] = 0.0
        if not all_ok(ok):
            lib.flow_rccl_comm_destroy(comm)
            self._rccl_comm = None
            return
        self.direct = binding
        self._point_struct_at_rccl()

    def benchmark(self, calls=50, count=8):
        '''Microseconds per all-reduce of `count` doubles through each binding
        this communicator can use, `calls` back to back after a warm-up (the
        slowest rank's time, agreed on by all): {'torch': us[, 'library':
        us]} -- 'library' only on the nccl backend, and only if every rank
        could bring the library's own communicator up.'''
        import time
        out = {}
        lib = _hip.load_library() if device.on_gpu() else None

        def timed(call):
            for _ in range(5):
                call()
            device.synchronize()
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(calls):
                call()
            device.synchronize()
            us = 1.0e6 * (time.perf_counter() - t0) / calls
            t = torch.tensor([us], dtype=torch.float64,
                             device=self.buf.device if not self.staged
                             else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return float(t.item())
        self.buf[:count] = 0.0
        out['torch'] = timed(lambda: self.allreduce_tensor(self.buf[:count]))
        if not self.staged and lib is not None:
            had = self.direct
            if had is None:
                self._bind_rccl()
            if self.direct is not None:
                user = ctypes.cast(ctypes.pointer(self.direct), ctypes.c_void_p)
                self.direct.stream = ctypes.c_void_p(device.stream_handle())
                out['library'] = timed(
                    lambda: lib.flow_rccl_allreduce(user, count))
        self.buf[:count] = 0.0
        return out

    def use_fastest(self, calls=50):
        '''Benchmark both bindings and keep the faster one (every rank takes
        the same decision: the timings are maxima over the ranks).  -> the
        timings, with 'used'.'''
        us = self.benchmark(calls)
        if 'library' in us and us['library'] < us['torch']:
            self._point_struct_at_rccl()
            us['used'] = 'library'
        else:
            if self.direct is not None:
                # back to the torch callback (the communicator stays: unused)
                self.struct.allreduce = self._cb
                self.struct.user = None
                self.direct_unused, self.direct = self.direct, None
            us['used'] = 'torch'
        return us

    def _point_struct_at_rccl(self):
        lib = _hip.load_library()
        self.direct.buf = ctypes.c_void_p(self.buf.data_ptr())
        self.struct.allreduce = ctypes.cast(lib.flow_rccl_allreduce,
                                            _hip.ALLREDUCE_FN)
        self.struct.user = ctypes.cast(ctypes.pointer(self.direct),
                                       ctypes.c_void_p)

    def _allreduce(self, _user, count):
        # called from inside the library's solver loops (ctypes re-acquires
        # the GIL); exceptions must not propagate through the C frames
        try:
            self.calls += 1
            self.allreduce_tensor(self.buf[:count])
            return 0
        except Exception:                                  # noqa: BLE001
            traceback.print_exc()
            self.failed = True
            return 1

    def allreduce_tensor(self, t):
        if self.world == 1 and self.staged:
            return t
        if self.staged and t.is_cuda:
            # gloo cannot reduce device tensors (to_host synchronises first)
            h = device.to_host(t)
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            # RCCL: enqueued behind the kernels of the current stream by
            # torch's event hand-over; the stream waits for its result
            dist.all_reduce(t, group=self.group)
        return t


# -- partition (pure host logic, CPU-testable) ------------------------------------
class StripsTooThin(ValueError):
    '''A ghost range would reach past the immediate neighbour: the caller
    picks the algorithm with the shallower halo (raised explicitly -- not an
    assert: the choice must not depend on `python -O`, and a genuine invariant
    failure must not be mistaken for it).'''


class RowBlocks(object):
    '''Contiguous row blocks of ONE scalar space over the ranks, with the ghost
    ranges [lo_g, r0_g) and [r1_g, hi_g) each rank needs and the layout of the
    halo slots in the exchange buffer: rank by rank, [to-left | to-right].'''

    def __init__(self, n, bounds, lo, hi):
        self.n = int(n)
        self.bounds = numpy.asarray(bounds, dtype=numpy.int64)
        self.world = len(self.bounds) - 1
        self.lo = numpy.asarray(lo, dtype=numpy.int64)
        self.hi = numpy.asarray(hi, dtype=numpy.int64)
        assert self.bounds[0] == 0 and self.bounds[-1] == n
        assert (numpy.diff(self.bounds) > 0).all(), 'more ranks than rows'
        for g in range(self.world):
            r0, r1 = self.rows(g)
            assert self.lo[g] <= r0 and self.hi[g] >= r1
        if not self.fits(self.bounds, self.lo, self.hi):
            # (decided over ALL ranks before anything is raised: every rank
            # sees the same verdict)
            raise StripsTooThin(
                'strips too thin: a ghost range reaches past the neighbour')
        slot = {}
        off = 0
        for q in range(self.world):
            for side, (row, ln) in enumerate(self.sends(q)):
                slot[(q, side)] = (off, row, ln)
                off += ln
        self.nhalo = off
        self._slot = slot

    @staticmethod
    def fits(bounds, lo, hi):
        '''Do the ghost ranges [lo_g, hi_g) stay within the immediate
        neighbours' rows, for every rank?'''
        world = len(bounds) - 1
        n = int(bounds[-1])
        for g in range(world):
            left = bounds[g - 1] if g > 0 else 0
            right = bounds[g + 2] if g + 2 <= world else n
            if lo[g] < left or hi[g] > right:
                return False
        return True

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

    def struct(self, g):
        '''flow_rows of rank g.'''
        s = _hip.RowsS()
        r0, r1 = self.rows(g)
        s.n, s.r0, s.r1 = self.n, r0, r1
        s.e0, s.e1 = int(self.lo[g]), int(self.hi[g])
        s.nhalo = self.nhalo
        for side in (0, 1):
            o, row, ln = self._slot[(g, side)]
            s.send_row[side], s.send_len[side], s.send_slot[side] = row, ln, o
        if g > 0:
            # my left ghost rows = what the left neighbour sends to ITS right
            o, row, ln = self._slot[(g - 1, 1)]
            assert row == s.e0 and row + ln == r0
            s.recv_row[0], s.recv_len[0], s.recv_slot[0] = row, ln, o
        else:
            assert s.e0 == r0
        if g + 1 < self.world:
            o, row, ln = self._slot[(g + 1, 0)]
            assert row == r1 and row + ln == s.e1
            s.recv_row[1], s.recv_len[1], s.recv_slot[1] = row, ln, o
        else:
            assert s.e1 == r1
        return s


class Strips(object):
    '''The strip decomposition of a mesh for `world` ranks: vertex bounds,
    per-rank cell ranges, and RowBlocks of every scalar space on the mesh.'''

    def __init__(self, mesh, world):
        from .fem.space import scalar_layout
        self.mesh = mesh
        self.world = world
        p1 = scalar_layout(mesh, 1)
        nv = p1.N
        assert 1 <= world <= nv
        # balance the work: cells per vertex ~ nonzeros of the P1 pattern
        rowptr = p1.pattern('rowptr').astype(numpy.int64)
        targets = (numpy.arange(1, world) * int(rowptr[-1])) // world
        cuts = numpy.searchsorted(rowptr, targets, side='left')
        vb = numpy.concatenate([[0], cuts, [nv]]).astype(numpy.int64)
        for g in range(1, world + 1):
            vb[g] = max(vb[g], vb[g - 1] + 1)
        vb[world] = nv
        self.vbounds = vb
        # cells of a rank: the index range spanned by the cells that touch one
        # of its vertices (x-major cell numbering: hardly any cell in that
        # range touches none)
        cv = mesh.cell_vertices
        owner = numpy.searchsorted(vb, cv, side='right') - 1      # (nc, 3)
        self.cells = []
        for g in range(world):
            idx = numpy.nonzero((owner == g).any(axis=1))[0]
            self.cells.append((int(idx.min()), int(idx.max()) + 1))
        self._blocks = {}

    def deep_ranges(self, layout, depth):
        '''Per rank the nested row ranges [lo_d, hi_d), d = 0 .. depth: its own
        rows, then what the rows of the range before are coupled to (d = 1:
        the ghost range of `blocks`, which also covers the dofs of the rank's
        cells).'''
        base = self.blocks(layout)
        rowptr = layout.pattern('rowptr').astype(numpy.int64)
        cols = layout.pattern('cols')
        out = []
        for g in range(self.world):
            r0, r1 = base.rows(g)
            rng = [(r0, r1), (int(base.lo[g]), int(base.hi[g]))]
            while len(rng) <= depth:
                lo, hi = rng[-1]
                seg = cols[rowptr[lo]:rowptr[hi]]
                rng.append((min(int(seg.min()), lo), max(int(seg.max()) + 1, hi)))
            out.append(rng[:depth + 1])
        return out

    def deep_blocks(self, layout, depth):
        '''RowBlocks whose ghost ranges are `depth` layers deep (the halo of
        the sharded mass solver); strips thinner than that are refused.'''
        key = (layout.degree, depth)
        if key not in self._blocks:
            base = self.blocks(layout)
            rng = self.deep_ranges(layout, depth)
            lo = numpy.array([r[depth][0] for r in rng], dtype=numpy.int64)
            hi = numpy.array([r[depth][1] for r in rng], dtype=numpy.int64)
            self._blocks[key] = RowBlocks(layout.N, base.bounds, lo, hi)
        return self._blocks[key]

    def blocks(self, layout):
        '''RowBlocks of a scalar layout (P1 or P2) on this decomposition.'''
        key = layout.degree
        if key not in self._blocks:
            vb = self.vbounds
            n = layout.N
            if layout.degree == 1:
                bounds = vb.copy()
            else:
                # P2 dofs are numbered by (lowest vertex, kind): the vertex dof
                # of v comes first among those with lowest vertex v
                vd = layout.vertex_dofs.astype(numpy.int64)
                bounds = numpy.concatenate([vd[vb[:-1]], [n]])
                bounds[0] = 0
            rowptr = layout.pattern('rowptr').astype(numpy.int64)
            cols = layout.pattern('cols')
            cd = layout.cell_dofs
            lo = numpy.empty(self.world, dtype=numpy.int64)
            hi = numpy.empty(self.world, dtype=numpy.int64)
            for g in range(self.world):
                r0, r1 = int(bounds[g]), int(bounds[g + 1])
                seg = cols[rowptr[r0]:rowptr[r1]]
                c0, c1 = self.cells[g]
                # everything the rank's rows are coupled to AND every dof of
                # every cell its cell kernels visit (they index compact
                # vectors of exactly this extent)
                lo[g] = min(int(seg.min()), int(cd[c0:c1].min()), r0)
                hi[g] = max(int(seg.max()), int(cd[c0:c1].max())) + 1
                hi[g] = max(hi[g], r1)
            self._blocks[key] = RowBlocks(n, bounds, lo, hi)
        return self._blocks[key]


_STRIPS = {}


def strips(mesh):
    c = comm()
    key = (id(mesh), c.world)
    if key not in _STRIPS:
        # (the mesh is kept alive with its decomposition: ids are not recycled)
        _STRIPS[key] = (Strips(mesh, c.world), mesh)
    return _STRIPS[key][0]


# -- per-rank views of the library's structs ---------------------------------------
class View(object):
    '''Everything rank-specific about one scalar layout: flow_rows, the
    space struct with the owned row / nonzero range, owned row blocks.'''

    def __init__(self, layout, st, rank):
        from .fem import ops
        self.layout = layout
        self.blocks = st.blocks(layout)
        self.rows = self.blocks.struct(rank)
        self.r0, self.r1 = self.rows.r0, self.rows.r1
        self.e0, self.e1 = self.rows.e0, self.rows.e1
        base = ops.space_struct(layout)
        s = _hip.SpaceS()
        ctypes.memmove(ctypes.byref(s), ctypes.byref(base),
                       ctypes.sizeof(_hip.SpaceS))
        rowptr = layout.pattern('rowptr')
        s.r0, s.r1 = self.r0, self.r1
        s.nnz0, s.nnz1 = int(rowptr[self.r0]), int(rowptr[self.r1])
        self.space = s
        rp = rowptr.astype(numpy.int64)
        local = rp[self.r0:self.r1 + 1] - rp[self.r0]
        rb = csr_stream_rowblocks(local) + self.r0
        self.rowblocks = device.to_device(rb.astype(numpy.int32))
        # (operator kinds 2 and 4 park two products per nonzero: smaller tiles)
        rb2 = csr_stream_rowblocks(
            local, nnz_per_block=_hip.SPMV_NNZ_PER_BLOCK) + self.r0
        self.rowblocks2 = device.to_device(rb2.astype(numpy.int32))

    def operator(self, A):
        '''A copy of A's flow_operator that covers the owned rows only.'''
        return owned_operator(
            A.operator(),
            self.rowblocks2 if A.kind in (2, 4) else self.rowblocks)


def owned_operator(base, rowblocks):
    op = _hip.Operator()
    ctypes.memmove(ctypes.byref(op), ctypes.byref(base),
                   ctypes.sizeof(_hip.Operator))
    op.rowblocks = _hip.i32(rowblocks)
    op.nblocks = rowblocks.numel() - 1
    return op


def view(layout):
    '''The calling rank's View of a scalar layout (cached on the layout).'''
    c = comm()
    key = ('strip_view', c.world, c.rank)
    if key not in layout._dev:
        layout._dev[key] = View(layout, strips(layout.mesh), c.rank)
    return layout._dev[key]


def mesh_view(mesh):
    '''flow_mesh restricted to the calling rank's cells.'''
    from .fem import ops
    c = comm()
    key = ('strip_mesh', c.world, c.rank)
    if key not in mesh._cache:
        base = ops.mesh_struct(mesh)
        s = _hip.MeshS(base.nc, base.xy, 0, 0)
        s.c0, s.c1 = strips(mesh).cells[c.rank]
        mesh._cache[key] = s
    return mesh._cache[key]


# -- collectives on fields ------------------------------------------------------------
def halo(vec, layout, ncomp=1):
    '''Make the ghost rows of a global-length field current.'''
    c = comm()
    v = view(layout)
    c.ensure(ncomp * v.rows.nhalo)
    _hip.check(_hip.lib().flow_shard_halo(
        ctypes.byref(c.struct), ctypes.byref(v.rows), ncomp,
        _hip.f64(vec, ncomp * layout.N), layout.N, _hip.stream()))
    return vec


def _reduce(x, y, layout, ncomp, kind):
    from .fem import ops
    c = comm()
    v = view(layout)
    c.ensure(16)
    res = ctypes.c_double(0.0)
    _hip.check(_hip.lib().flow_shard_reduce_host(
        ctypes.byref(c.struct), ctypes.byref(v.rows), ncomp,
        _hip.f64(x, ncomp * layout.N),
        _hip.f64(y, ncomp * layout.N) if y is not None else None, layout.N,
        kind, _hip.f64(ops.work(_hip.REDUCE_WORK)), ctypes.byref(res),
        _hip.stream()))
    return res.value


def dot(x, y, layout, ncomp=1):
    '''x . y over the owned rows of all ranks.'''
    return _reduce(x, y, layout, ncomp, 0)


def norm_linf(x, layout, ncomp=1):
    return _reduce(x, None, layout, ncomp, 1)


def gather_field(vec, layout, ncomp=1):
    '''Make a field whole on every rank (tests, output): every rank keeps its
    owned rows, zeros elsewhere, and the sum is the field.'''
    c = comm()
    v = view(layout)
    n = layout.N
    out = torch.zeros_like(vec)
    for a in range(ncomp):
        out[a * n + v.r0:a * n + v.r1] = vec[a * n + v.r0:a * n + v.r1]
    device.synchronize()
    c.allreduce_tensor(out)
    device.synchronize()
    vec.copy_(out)
    return vec


# -- solvers ------------------------------------------------------------------------
def _solve_info(its, res, name):
    from .fem.ops import SolveInfo
    return SolveInfo(its, res, '%s[x-strips x%d]' % (name, comm().world))


def _guarded(solve, x, guard):
    '''The fallback chain of ops.krylov_solve(guard=...) around a sharded CG:
    solve(rejected) runs the library's loop from x (rejected: a c_int to
    receive the verdict on the start, or None: unguarded).  Returns the number
    of starts dropped.'''
    from .fem import ops
    if guard is None:
        solve(None)
        return 0
    rejected = ctypes.c_int(0)
    solve(rejected)
    if not rejected.value:
        return 0
    dropped = 1
    if guard is not False:
        ops.copy(x, guard)
        solve(rejected)
        if not rejected.value:
            return dropped
        dropped = 2
    ops.fill(x, 0.0)
    solve(None)
    return dropped


def cg(A, dinv, b, x, rtol, atol=0.0, maxit=1000, check_every=2, tag=None,
       guard=None):
    '''Jacobi-CG on the strips (operator kind 0 or 4); b valid on the owned
    rows, x on the owned rows (start) -> owned + ghost rows (solution).
    guard: as ops.krylov_solve.'''
    from .fem import ops
    c = comm()
    lay = A.layout
    v = view(lay)
    ncomp = 2 if A.kind == 4 else 1
    c.ensure(4 + ncomp * v.rows.nhalo)
    op = v.operator(A)
    n = A.size
    wlen = _hip.REDUCE_WORK + 10 * ncomp * (v.e1 - v.e0) + op.nblocks + 2
    wk = ops.work(wlen)
    history = A.__dict__.setdefault('_solve_history', {}) if tag else None
    first = history[tag] + 1 if history is not None and tag in history else 0
    its = ctypes.c_int(0)
    res = ctypes.c_double(0.0)

    def solve(rejected):
        _hip.check(_hip.lib().flow_shard_cg_solve(
            ctypes.byref(c.struct), ctypes.byref(v.rows), ctypes.byref(op),
            _hip.f64(dinv, n, 'dinv'), _hip.f64(b, n, 'b'), _hip.f64(x, n, 'x'),
            float(rtol), float(atol), int(maxit), int(check_every), int(first),
            _hip.f64(wk), wk.numel(), ctypes.byref(its), ctypes.byref(res),
            ctypes.byref(rejected) if rejected is not None else None,
            _hip.stream()))
    dropped = _guarded(solve, x, guard)
    if history is not None:
        history[tag] = its.value
    out = _solve_info(its.value, res.value, 'cg')
    out.starts_dropped = dropped
    return out


class MgShard(object):
    '''The finest level of a Multigrid hierarchy cut to the calling rank's
    strip (flow_mg_shard): Ah0 / Ps0 with the row blocks of the owned rows, the
    restriction restricted to the owned columns.'''

    def __init__(self, mg, v, rows2=None, zrange=None):
        '''rows2 / zrange: the one-collective form -- the flow_rows whose ghost
        ranges reach TWO coupling layers out, and the rows [z_lo, z_hi) (owned +
        one layer) the up-sweep then covers.'''
        from .fem.multigrid import CsrOperator
        assert mg.nlevels >= 2 and mg.R0_host is not None
        self.mg = mg
        self._keep = []
        self.rows2 = rows2
        lvl = mg.levels[0]
        Rg = mg.R0_host[:, v.r0:v.r1].tocsr()
        assert Rg.nnz > 0
        self.Rg = CsrOperator(Rg)
        s = _hip.MgShardS()
        s.mg = ctypes.pointer(mg.struct)
        s.Ah0 = owned_operator(lvl['Ah'].op, self._blocks(lvl['Ah'], v))
        s.Ps0 = owned_operator(lvl['Ps'].op, self._blocks(lvl['Ps'], v))
        s.Rg = self.Rg.op
        self.Cg = None
        if mg.C0_host is not None and MGCG_COLLECTIVES != 3:
            # the two-collective form: C = R (I - Ah) cut to the owned COLUMNS
            # (no ghost rows needed for the rank's share of C r), and row
            # blocks of the owned rows that hold a tile of Ps AND of Ah
            Cg = mg.C0_host[:, v.r0:v.r1].tocsr()
            assert Cg.nnz > 0
            self.Cg = CsrOperator(Cg)
            s.Cg = self.Cg.op
            rps = [device.to_host(lvl[k]._rowptr).numpy().astype(numpy.int64)
                   for k in ('Ps', 'Ah')]
            z0, z1 = (v.r0, v.r1) if zrange is None else zrange
            rb = csr_stream_rowblocks(
                [rp[z0:z1 + 1] - rp[z0] for rp in rps]) + z0
            t = device.to_device(rb.astype(numpy.int32))
            self._keep.append(t)
            s.up_rowblocks0 = _hip.i32(t).value
            s.up_nblocks0 = t.numel() - 1
            if zrange is not None:
                assert rows2 is not None
                assert rows2.e0 <= z0 <= v.r0 and v.r1 <= z1 <= rows2.e1
                s.z_lo, s.z_hi = int(z0), int(z1)
        self.struct = s

    def _blocks(self, op, v):
        rp = device.to_host(op._rowptr).numpy().astype(numpy.int64)
        rb = csr_stream_rowblocks(rp[v.r0:v.r1 + 1] - rp[v.r0]) + v.r0
        t = device.to_device(rb.astype(numpy.int32))
        self._keep.append(t)
        return t


def mgcg(A, dinv, mg, b, x, rtol, atol=0.0, maxit=1000, check_every=2,
         tag=None, guard=None):
    '''CG + the strip-sharded V-cycle (the pressure solve).  guard: as
    ops.krylov_solve.'''
    from .fem import ops
    c = comm()
    lay = A.layout
    v = view(lay)
    key = ('mg_shard', c.world, c.rank)
    if key not in mg.__dict__:
        rows2 = zrange = None
        if mg.C0_host is not None and MGCG_COLLECTIVES == 1:
            # one collective per iteration: ghost ranges two layers deep, z
            # formed on the first layer by the rank itself
            st = strips(lay.mesh)
            try:
                rows2 = st.deep_blocks(lay, 2).struct(c.rank)
                zrange = st.deep_ranges(lay, 2)[c.rank][1]
            except StripsTooThin:
                rows2 = zrange = None       # (two collectives per iteration)
        mg.__dict__[key] = MgShard(mg, v, rows2, zrange)
    ms = mg.__dict__[key]
    n1 = ms.struct.Rg.n
    two = ms.Cg is not None
    rows = ms.rows2 if ms.rows2 is not None else v.rows
    c.ensure(max(4 + rows.nhalo + (n1 if two else 0), 2 * n1 if two else n1))
    op = v.operator(A)
    n = A.size
    wlen = _hip.REDUCE_WORK + 11 * (rows.e1 - rows.e0) + op.nblocks \
        + 2 * max(ms.struct.Ps0.nblocks, ms.struct.up_nblocks0) + 2 \
        + (3 * n1 + 2 if two else 0)
    wk = ops.work(wlen)
    history = A.__dict__.setdefault('_solve_history', {}) if tag else None
    first = history[tag] + 1 if history is not None and tag in history else 0
    its = ctypes.c_int(0)
    res = ctypes.c_double(0.0)

    def solve(rejected):
        _hip.check(_hip.lib().flow_shard_mgcg_solve(
            ctypes.byref(c.struct), ctypes.byref(rows), ctypes.byref(op),
            _hip.f64(dinv, n, 'dinv'), ctypes.byref(ms.struct),
            _hip.f64(b, n, 'b'), _hip.f64(x, n, 'x'), float(rtol), float(atol),
            int(maxit), int(check_every), int(first), _hip.f64(wk), wk.numel(),
            ctypes.byref(its), ctypes.byref(res),
            ctypes.byref(rejected) if rejected is not None else None,
            _hip.stream()))
    dropped = _guarded(solve, x, guard)
    if history is not None:
        history[tag] = its.value
    out = _solve_info(its.value, res.value, 'cg+mg%d' % mg.nlevels)
    out.starts_dropped = dropped
    return out


def gmres(Jop, pre, b, x, rtol, atol=0.0, maxit=1000, restart=20,
          x_is_zero=True, expected=0, verify=False):
    '''GMRES + a block-Jacobi preconditioner on the strips -- `pre`: the rank's
    own ILU(0) (local_ilu) or two-level cycle (local_pmg) of its diagonal
    block.  Jop: a MomentumJacobian built on the rank's views (kind 3; b, x:
    global-length velocity fields), or a scalar ops.Matrix (kind 0: applied on
    the rank's rows; b, x: global-length scalar fields).  b valid on the owned
    rows; x (start: owned rows) -> the owned rows of the solution.'''
    from .fem.pmg import Pmg
    is_pmg = isinstance(pre, Pmg)
    from .fem import ops
    c = comm()
    lay = Jop.layout
    v = view(lay)
    ncomp = 1 if Jop.kind == 0 else 2
    op = v.operator(Jop) if Jop.kind == 0 else Jop.operator()
    c.ensure(max(ncomp * v.rows.nhalo, _hip.GMRES_MAX_RESTART + 2))
    mo, me = v.r1 - v.r0, v.e1 - v.e0
    wlen = _hip.REDUCE_WORK + (2 * restart + 4) * ncomp * mo + 2 * me \
        + _hip.GMRES_PARTIALS + _hip.GMRES_STATE
    wk = ops.work(wlen)
    its = ctypes.c_int(0)
    res = ctypes.c_double(0.0)
    nn = ncomp * lay.N
    _hip.check(_hip.lib().flow_shard_gmres_solve(
        ctypes.byref(c.struct), ctypes.byref(v.rows), ctypes.byref(op),
        None if is_pmg else ctypes.byref(pre.struct),
        ctypes.byref(pre.struct) if is_pmg else None,
        _hip.f64(b, nn, 'b'), _hip.f64(x, nn, 'x'), float(rtol), float(atol),
        int(maxit), int(restart), int(bool(x_is_zero)), int(expected),
        int(bool(verify)),
        _hip.f64(wk), wk.numel(), ctypes.byref(its), ctypes.byref(res),
        _hip.stream()))
    return _solve_info(its.value, res.value,
                       'gmres+pmg(block)' if is_pmg else 'gmres+ilu0(block)')


class MassStrips(object):
    '''What the sharded mass solver needs beside the fem.mass.MassSolver of the
    whole matrix: the rank's deep ghost range and halo slots, the row blocks of
    its shrinking products, fp32 work vectors over its window.'''

    def __init__(self, solver, st, rank):
        lay = solver.A.layout
        steps = solver.struct.steps
        products = steps - 1
        rng = st.deep_ranges(lay, steps)[rank]
        try:
            self.rows = st.deep_blocks(lay, steps).struct(rank)
        except StripsTooThin as e:
            raise StripsTooThin(
                'the mass solver\'s halo is %d coupling layers deep, a strip '
                'of this decomposition is thinner (%s): fewer ranks, or '
                "solver_parameters['correction']['method'] = 'cg' and "
                "fem.ops.MASS_SOLVER['method'] = 'cg'" % (steps, e))
        rp = lay.pattern('rowptr').astype(numpy.int64)
        L = _hip.MassStripsS()
        L.nlevels = products
        self._keep = []
        for j in range(products):
            lo, hi = rng[products - j]          # product j: depth products - j
            rb = csr_stream_rowblocks(
                rp[lo:hi + 1] - rp[lo],
                nnz_per_block=_hip.PMG_NNZ_PER_BLOCK) + lo
            t = device.to_device(rb.astype(numpy.int32))
            self._keep.append(t)
            L.rowblocks16[j] = _hip.i32(t).value
            L.nblocks16[j] = t.numel() - 1
        L.row_lo_last, L.row_hi_last = rng[1]
        self.levels = L
        # (the View owns the device row blocks the operator points to)
        self.view = View(lay, st, rank)
        self.op = self.view.operator(solver.A)
        me = self.rows.e1 - self.rows.e0
        self.work16 = torch.zeros(5 * solver.ncomp * me + 4, dtype=torch.float32,
                                  device=device.get())
        M = _hip.MassS()
        ctypes.memmove(ctypes.byref(M), ctypes.byref(solver.struct),
                       ctypes.sizeof(_hip.MassS))
        M.A = ctypes.pointer(self.op)
        M.work16 = _hip.f32(self.work16, 5 * solver.ncomp * me).value
        M.work16_rows = me
        M.packed16 = None              # (the packed stream is tiled once)
        M.cbase16 = None
        self.struct = M
        self.nlast = L.nblocks16[products - 1]


def mass_solve(solver, b, x, rtol, atol=0.0, maxit=50, tag=None, xbase=None,
               delta0=None):
    '''fem.mass.MassSolver on the strips (flow_shard_mass_solve): one
    collective per defect correction (+ one for the verdict on the last).
    xbase None: x holds the start (valid on own + ghost rows); else the
    increment form, b the defect of xbase, delta0 an optional start of the
    increment; x = xbase + increment on the own + ghost rows.'''
    from .fem import ops
    from .message import info
    c = comm()
    key = ('mass_strips', c.world, c.rank)
    if key not in solver.__dict__:
        try:
            solver.__dict__[key] = MassStrips(
                solver, strips(solver.A.layout.mesh), c.rank)
        except StripsTooThin as e:
            # (a decomposition is thin for every rank or for none: RowBlocks
            # looks at all ranks before it raises)
            info('mass solves on these strips by Jacobi-CG: %s' % e)
            solver.__dict__[key] = None
            _MASS_FALLBACK[0] = True
    ms = solver.__dict__[key]
    n = solver.A.size
    if ms is None:
        # strips thinner than the halo: Jacobi-CG, one collective per iteration
        if xbase is None:
            return cg(solver.A, solver.dinv, b, x, rtol, atol, maxit=1000,
                      check_every=2, tag=tag)
        # M delta = g from delta0 (or zero); (not ops.work: cg's own buffer)
        delta = _hip.clone(delta0) if delta0 is not None else device.zeros(n)
        sol = cg(solver.A, solver.dinv, b, delta, rtol, atol, maxit=1000,
                 check_every=2, tag=tag)
        if x.data_ptr() != xbase.data_ptr():
            ops.copy(x, xbase)
        ops.axpby(1.0, delta, 1.0, x)
        return sol
    c.ensure(4 + solver.ncomp * ms.rows.nhalo)
    head = _hip.REDUCE_WORK + 2 * ms.nlast
    wk = ops.work(head + 2 + (n if xbase is not None else 0))
    first = solver.history.get(('strips', tag), 0) if tag is not None else 0
    its = ctypes.c_int(0)
    res = ctypes.c_double(0.0)
    base = _hip.clone(xbase) if xbase is not None and solver.guard \
        and x.data_ptr() == xbase.data_ptr() else xbase
    try:
        _hip.check(_hip.lib().flow_shard_mass_solve(
            ctypes.byref(c.struct), ctypes.byref(ms.rows),
            ctypes.byref(ms.struct), ctypes.byref(ms.levels),
            _hip.f64(b, n, 'b'),
            _hip.f64(xbase, n, 'xbase') if xbase is not None else None,
            _hip.f64(delta0, n, 'delta0') if delta0 is not None else None,
            _hip.f64(x, n, 'x'), float(rtol), float(atol), int(maxit),
            int(first), _hip.f64(wk), wk.numel(), ctypes.byref(its),
            ctypes.byref(res), _hip.stream()))
    except _hip.NotConverged as err:
        # The verdict (an iteration that does not contract, a NaN) comes out
        # of sums that are the same on every rank: all ranks land here
        # together and all take the Jacobi-CG of the strips, like
        # MassSolver._fallback on one GPU.
        if not solver._bound_failed(err):
            raise
        info('sharded mass solve: %s -- falling back to Jacobi-CG' % err)
        solver.fallbacks = getattr(solver, 'fallbacks', 0) + 1
        if xbase is None:
            if not bool(torch.isfinite(x).all()):
                _hip.fill(x, 0.0)
            sol = cg(solver.A, solver.dinv, b, x, rtol, atol, maxit=1000,
                     check_every=2, tag=tag)
        else:
            delta = device.zeros(n)
            sol = cg(solver.A, solver.dinv, b, delta, rtol, atol, maxit=1000,
                     check_every=2, tag=tag)
            ops.copy(x, base)
            ops.axpby(1.0, delta, 1.0, x)
        sol.method = 'defect correction -> ' + sol.method
        return sol
    if tag is not None:
        solver.history[('strips', tag)] = its.value
    return _solve_info(its.value, res.value,
                       'defect correction + chebyshev%d/fp16'
                       % solver.struct.steps)


def local_pmg(W, **kw):
    '''The two-level cycle of flow_amd/fem/pmg.py on the calling rank's
    diagonal block: P2 rows / P1 rows (vertices) of its strip in local
    numbering, couplings that leave the block dropped -- block Jacobi, no
    communication inside the preconditioner.'''
    from .fem.pmg import Pmg
    from .fem.space import scalar_layout
    lay = W.layout
    v2 = view(lay)
    v1 = view(scalar_layout(lay.mesh, 1))
    return Pmg(W, rows=(v2.r0, v2.r1), vrows=(v1.r0, v1.r1), **kw)


def local_ilu(J, packed=False, single_vector=False):
    '''ILU(0) of the calling rank's diagonal block of the (block-diagonal part
    of the) Jacobian J: plan over the owned rows in local numbering.'''
    from .fem import ilu
    c = comm()
    lay = J.layout
    key = ('ilu_plan_strip', c.world, c.rank)
    if key not in lay._dev:
        v = view(lay)
        lay._dev[key] = ilu.IluPlan(lay, rows=(v.r0, v.r1))
    return ilu.Ilu0(J, plan=lay._dev[key], packed=packed,
                    single_vector=single_vector)
