# -*- coding: utf-8 -*-
'''
Multi-process CPU tests (gloo, world_size 2 and 3) of the row-sharded pressure
solve: partition, halo plans, the communication pattern and the single-reduction
CG recurrence of flow_amd/parallel.py.  The local kernels are replaced by a
numpy stand-in that lives HERE (test infrastructure); the product's local
kernels are the HIP ones (parallel.HipLocal), covered by the `-m gpu` tests.
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
    '''numpy restatement of parallel.HipLocal's interface (tests only).'''

    def __init__(self, A, r0, r1):
        self.A = A.tocsr()
        self.rows = self.A[r0:r1]
        self.r0, self.r1 = r0, r1

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def spmv_rows(self, x, y):
        y.numpy()[self.r0:self.r1] = self.rows.dot(x.numpy())

    def residual(self, b, q, dinv, r, z):
        s = slice(self.r0, self.r1)
        r.numpy()[s] = b.numpy()[s] - q.numpy()[s]
        z.numpy()[s] = dinv.numpy()[s] * r.numpy()[s]

    def dots(self, r, z, w, b, out):
        s = slice(self.r0, self.r1)
        rn, zn, wn = r.numpy()[s], z.numpy()[s], w.numpy()[s]
        out[0] = rn.dot(zn)
        out[1] = zn.dot(wn)
        out[2] = rn.dot(rn)
        if b is not None:
            out[3] = b.numpy()[s].dot(b.numpy()[s])

    def scalars(self, first, sums, S):
        g, d, rr = float(sums[0]), float(sums[1]), float(sums[2])
        if first:
            beta = 0.0
            alpha = g / d if d != 0.0 else 0.0
        else:
            beta = g / float(S[0]) if float(S[0]) != 0.0 else 0.0
            den = d - beta * g / float(S[1]) if float(S[1]) != 0.0 else 0.0
            alpha = g / den if den != 0.0 else 0.0
        S[0], S[1], S[2], S[3] = g, alpha, beta, rr

    def update(self, S, dinv, w, z, p, s_, x, r, want_z=True):
        s = slice(self.r0, self.r1)
        alpha, beta = float(S[1]), float(S[2])
        pn, sn = p.numpy(), s_.numpy()
        pn[s] = z.numpy()[s] + beta * pn[s]
        sn[s] = w.numpy()[s] + beta * sn[s]
        x.numpy()[s] += alpha * pn[s]
        r.numpy()[s] -= alpha * sn[s]
        z.numpy()[s] = dinv.numpy()[s] * r.numpy()[s]

    # two-level preconditioner pieces (numpy coarse object: see NumpyCoarse)
    def coarse_restrict(self, coarse, vec, out):
        v = numpy.zeros(coarse.n)
        v[self.r0:self.r1] = vec.numpy()[self.r0:self.r1]
        out.numpy()[:] = coarse.P.T.dot(v)

    def coarse_solve(self, coarse, rc, zc):
        zc.numpy()[:] = coarse.Ainv.dot(rc.numpy())

    def coarse_prolong(self, coarse, dinv, r, zc, z):
        s = slice(self.r0, self.r1)
        z.numpy()[s] = dinv.numpy()[s] * r.numpy()[s] \
            + coarse.P.dot(zc.numpy())[s]

    def coarse_recur(self, coarse, S, omega, sigma, rc):
        alpha, beta = float(S[1]), float(S[2])
        sg = sigma.numpy()
        sg[:] = omega.numpy() + beta * sg
        rc.numpy()[:] -= alpha * sg


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
        r0, r1 = part.rows(rank)
        local = NumpyLocal(A, r0, r1)
        x = torch.zeros(A.shape[0], dtype=torch.float64)
        dinv = torch.from_numpy(1.0 / A.diagonal())
        its, res = parallel.sharded_cg(
            local, comm, part, torch.from_numpy(b), x, dinv, 1e-12, 0.0, 5000, 7,
            coarse
            )
        out[rank] = (its, res, x.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_partition_and_halo_plans():
    A, _ = _system()
    n = A.shape[0]
    for world in (1, 2, 3, 5):
        part = parallel.Partition(A.indptr, A.indices, world)
        assert part.bounds[0] == 0 and part.bounds[-1] == n
        nnz = numpy.diff(A.indptr[part.bounds])
        assert nnz.max() - nnz.min() <= 2 * 9, 'balanced by nonzeros'
        for g in range(world):
            r0, r1 = part.rows(g)
            cols = A[r0:r1].indices
            have = numpy.zeros(n, dtype=bool)
            have[r0:r1] = True
            for peer, (s0, s1), (q0, q1) in part.exchanges(g):
                have[q0:q1] = True
                # the peer's matching send is exactly my receive
                back = [e for e in part.exchanges(peer) if e[0] == g][0]
                assert back[1] == (q0, q1)
                assert s0 >= r0 and s1 <= r1
            assert have[cols].all(), 'halo covers every referenced column'
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

        def halo_exchange(self, vec, plan):
            assert plan == []

        def allreduce_sum(self, t):
            return t

        def allgather_rows(self, vec, bounds):
            pass
    x = torch.zeros(A.shape[0], dtype=torch.float64)
    coarse = NumpyCoarse(A, *_system.extra) if two_level else None
    it1, _ = parallel.sharded_cg(
        NumpyLocal(A, 0, A.shape[0]), Solo(), part, torch.from_numpy(b), x,
        torch.from_numpy(1.0 / A.diagonal()), 1e-12, 0.0, 5000, 7, coarse
        )
    n_its = its.pop()
    assert abs(it1 - n_its) <= 7
    if two_level:
        # the coarse space pays off also through the recurrence form
        x0 = torch.zeros(A.shape[0], dtype=torch.float64)
        it0, _ = parallel.sharded_cg(
            NumpyLocal(A, 0, A.shape[0]), Solo(), part, torch.from_numpy(b), x0,
            torch.from_numpy(1.0 / A.diagonal()), 1e-12, 0.0, 5000, 7, None
            )
        assert n_its < 0.75 * it0, (n_its, it0)
