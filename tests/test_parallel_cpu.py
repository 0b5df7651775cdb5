# -*- coding: utf-8 -*-
'''
Multi-process CPU tests (gloo, world_size 2 and 3) of the row-sharded pressure
solve: partition, halo layout, the one-collective communication pattern and the
single-reduction CG recurrence of flow_amd/parallel.py.  The local side is a
numpy stand-in that lives HERE (test infrastructure); the product's local
side is parallel.HipLocal (flow_cg_shard_step), covered by the `-m gpu` tests.
'''
import os
import socket

import numpy
import pytest
import scipy.sparse.linalg as spla
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flow_amd import fem, parallel
from oracle import fem_oracle as orc

import oracle_harness as H


class NumpyLocal(object):
    '''numpy restatement of parallel.HipLocal (tests only): the vectors, the
    replicated start and everything flow_cg_shard_step does between two
    all-reduces (include/flow_hip.h), driven by the same HaloLayout.'''

    def __init__(self, A, dinv, coarse, part, rank):
        self.A = A.tocsr()
        self.dinv = dinv.numpy()
        self.coarse = coarse
        self.n = n = A.shape[0]
        self.r0, self.r1 = part.rows(rank)
        self.hl = part.halo_layout(rank)
        self.rows = self.A[self.r0:self.r1]
        nc = coarse.nc if coarse is not None else 0
        self.nc = nc
        z = lambda m: numpy.zeros(m)
        self.r, self.z, self.w, self.p, self.s = z(n), z(n), z(n), z(n), z(n)
        self.rc, self.zc, self.sigma = z(nc), z(nc), z(nc)
        self.alpha = self.beta = self.gamma = 0.0
        self.buf = torch.zeros(4 + nc + self.hl.nhalo, dtype=torch.float64)

    def _precondition(self, rows):
        self.z[rows] = self.dinv[rows] * self.r[rows]
        if self.coarse is not None:
            self.zc[:] = self.coarse.Ainv.dot(self.rc)
            self.z[rows] += self.coarse.P.dot(self.zc)[rows]

    def begin(self, b, x):
        self.x = x.numpy()
        bn = b.numpy()
        self.p[:] = 0.0
        self.s[:] = 0.0
        self.sigma[:] = 0.0
        self.alpha = self.beta = self.gamma = 0.0
        self.buf.zero_()
        # |B b|^2: the stopping test is in the preconditioned norm
        self.r[:] = bn
        if self.coarse is not None:
            self.rc[:] = self.coarse.P.T.dot(self.r)
        self._precondition(slice(0, self.n))
        bb2 = float(self.z.dot(self.z))
        self.r[:] = bn - self.A.dot(self.x)
        if self.coarse is not None:
            self.rc[:] = self.coarse.P.T.dot(self.r)
        self._precondition(slice(0, self.n))
        return bb2

    def step(self, phase):
        hl, buf = self.hl, self.buf.numpy()
        halo = buf[4 + self.nc:]
        ext = slice(hl.e0, hl.e1)
        own = slice(self.r0, self.r1)
        if phase > 0:
            for side in (0, 1):
                row, ln, slot = hl.recv_row[side], hl.recv_len[side], \
                    hl.recv_slot[side]
                self.w[row:row + ln] = halo[slot:slot + ln]
            g, d = buf[0], buf[1]
            if phase == 1:
                beta = 0.0
                alpha = g / d if d != 0.0 else 0.0
            else:
                beta = g / self.gamma if self.gamma != 0.0 else 0.0
                den = d - beta * g / self.alpha if self.alpha != 0.0 else 0.0
                alpha = g / den if den != 0.0 else 0.0
            self.gamma, self.alpha, self.beta = g, alpha, beta
            if self.coarse is not None:
                self.sigma[:] = buf[4:4 + self.nc] + beta * self.sigma
                self.rc -= alpha * self.sigma
            self.p[ext] = self.z[ext] + beta * self.p[ext]
            self.s[ext] = self.w[ext] + beta * self.s[ext]
            self.x[ext] += alpha * self.p[ext]
            self.r[ext] -= alpha * self.s[ext]
            self._precondition(ext)
        self.w[own] = self.rows.dot(self.z)
        buf[0] = self.r[own].dot(self.z[own])
        buf[1] = self.z[own].dot(self.w[own])
        buf[2] = self.z[own].dot(self.z[own])
        buf[3] = 0.0
        if self.coarse is not None:
            v = numpy.zeros(self.n)
            v[own] = self.w[own]
            buf[4:4 + self.nc] = self.coarse.P.T.dot(v)
        halo[:] = 0.0
        for side in (0, 1):
            row, ln, slot = hl.send_row[side], hl.send_len[side], \
                hl.send_slot[side]
            halo[slot:slot + ln] = self.w[row:row + ln]

    def res2(self):
        return float(self.buf[2])


class NumpyCoarse(object):
    '''Aggregates of 6 x 6 vertices, dense inverse of P^T A P (tests only).'''

    def __init__(self, A, points, isbc):
        import scipy.sparse as sp
        n = A.shape[0]
        h = 0.6 / 36
        ix = numpy.floor(points[:, 0] / (6 * h) + 1e-9).astype(int)
        iy = numpy.floor((points[:, 1] + 0.07) / (6 * h) + 1e-9).astype(int)
        _, agg = numpy.unique(ix * 1000 + iy, return_inverse=True)
        free = numpy.nonzero(~isbc)[0]
        P = sp.csr_matrix((numpy.ones(len(free)), (free, agg[free])),
                          shape=(n, agg.max() + 1))
        keep = numpy.nonzero(numpy.asarray(P.sum(axis=0)).ravel() > 0)[0]
        self.P = P[:, keep].tocsr()
        self.Ainv = numpy.linalg.inv(self.P.T.dot(A).dot(self.P).toarray())
        self.n = n
        self.nc = self.P.shape[1]


def _system():
    mesh = fem.karman_channel(36, 9)
    P = H.oracle_space(mesh, 1)
    A = orc.stiffness_matrix(P)
    # Dirichlet at the outlet, as in the Karman pressure system
    bc = numpy.nonzero(mesh.points[:, 0] > 0.6 - 1e-12)[0]
    rng = numpy.random.RandomState(0)
    b = rng.standard_normal(P.N)
    A, b = orc.symmetric_bc(A, b, bc, numpy.zeros(len(bc)))
    A.sort_indices()
    isbc = numpy.zeros(P.N, dtype=bool)
    isbc[bc] = True
    _system.extra = (mesh.points, isbc)
    return A, b


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, two_level, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        A, b = _system()
        coarse = NumpyCoarse(A, *_system.extra) if two_level else None
        part = parallel.Partition(A.indptr, A.indices, world)
        comm = parallel.Comm(dist.group.WORLD)
        dinv = torch.from_numpy(1.0 / A.diagonal())
        local = NumpyLocal(A, dinv, coarse, part, rank)
        x = torch.zeros(A.shape[0], dtype=torch.float64)
        its, res = parallel.sharded_cg(
            local, comm, part, torch.from_numpy(b), x, 1e-12, 0.0, 5000, 7
            )
        out[rank] = (its, res, x.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_partition_and_halo_layout():
    A, _ = _system()
    n = A.shape[0]
    for world in (1, 2, 3, 5):
        part = parallel.Partition(A.indptr, A.indices, world)
        assert part.bounds[0] == 0 and part.bounds[-1] == n
        nnz = numpy.diff(A.indptr[part.bounds])
        assert nnz.max() - nnz.min() <= 2 * 9, 'balanced by nonzeros'
        layouts = [part.halo_layout(g) for g in range(world)]
        nhalo = layouts[0].nhalo
        owner = numpy.full(nhalo, -1)
        for g, hl in enumerate(layouts):
            assert hl.nhalo == nhalo
            r0, r1 = part.rows(g)
            assert hl.e0 <= r0 and r1 <= hl.e1
            cols = A[r0:r1].indices
            assert cols.min() >= hl.e0 and cols.max() < hl.e1, \
                'ghost rows cover every referenced column'
            for side in (0, 1):
                row, ln, slot = hl.send_row[side], hl.send_len[side], \
                    hl.send_slot[side]
                assert r0 <= row and row + ln <= r1
                assert (owner[slot:slot + ln] == -1).all(), 'slots are disjoint'
                owner[slot:slot + ln] = g
        assert (owner >= 0).all()
        # what a rank receives is exactly what the neighbour sends, row by row
        for g, hl in enumerate(layouts):
            r0, r1 = part.rows(g)
            got = []
            for side, peer in ((0, g - 1), (1, g + 1)):
                ln = hl.recv_len[side]
                if ln == 0:
                    continue
                ps = layouts[peer]
                assert ps.send_slot[1 - side] == hl.recv_slot[side]
                assert ps.send_len[1 - side] == ln
                assert ps.send_row[1 - side] == hl.recv_row[side]
                got.append((hl.recv_row[side], ln))
            ghosts = sum(ln for _, ln in got)
            assert ghosts == (r0 - hl.e0) + (hl.e1 - r1)
    with pytest.raises(AssertionError):
        parallel.Partition(A.indptr, A.indices, 200)    # blocks thinner than band


@pytest.mark.parametrize('world,two_level', [(2, False), (3, False), (2, True)])
def test_sharded_cg_gloo(world, two_level):
    A, b = _system()
    ref = spla.splu(A.tocsc()).solve(b)
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), two_level, out), nprocs=world,
             join=True)
    its = {out[r][0] for r in range(world)}
    assert len(its) == 1
    for r in range(world):
        x = out[r][2]
        assert numpy.linalg.norm(x - ref) < 1e-9 * numpy.linalg.norm(ref)
        assert numpy.array_equal(x, out[0][2]), 'all ranks hold the solution'
    # same recurrence on one rank (no communication): same iteration count
    part = parallel.Partition(A.indptr, A.indices, 1)

    class Solo(object):
        rank, world = 0, 1

        def allreduce_sum(self, t):
            return t

        def allgather_rows(self, vec, bounds):
            pass
    x = torch.zeros(A.shape[0], dtype=torch.float64)
    coarse = NumpyCoarse(A, *_system.extra) if two_level else None
    dinv = torch.from_numpy(1.0 / A.diagonal())
    it1, _ = parallel.sharded_cg(
        NumpyLocal(A, dinv, coarse, part, 0), Solo(), part, torch.from_numpy(b),
        x, 1e-12, 0.0, 5000, 7
        )
    n_its = its.pop()
    assert abs(it1 - n_its) <= 7
    if two_level:
        # the coarse space pays off also through the recurrence form
        x0 = torch.zeros(A.shape[0], dtype=torch.float64)
        it0, _ = parallel.sharded_cg(
            NumpyLocal(A, dinv, None, part, 0), Solo(), part,
            torch.from_numpy(b), x0, 1e-12, 0.0, 5000, 7
            )
        assert n_its < 0.75 * it0, (n_its, it0)
