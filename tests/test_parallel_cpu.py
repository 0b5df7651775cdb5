# -*- coding: utf-8 -*-
'''
The strip decomposition of flow_amd.parallel on the CPU (gloo, world_size 2
and 3): partition + halo-slot logic, and the sharded Krylov loop's
communication pattern -- ONE primitive, an all-reduce of the head of a buffer
that carries dot products and halos (own slots filled, zeros elsewhere) --
driven through the real `parallel.Comm` callback.  The kernels' side is
restated in numpy here (NumpyShardCg mirrors what flow_shard_cg_solve does
between two collectives); the HIP side is covered by the `-m gpu` tests.
'''
import os
import socket

import numpy
import pytest
import scipy.sparse.linalg as spla
import torch.distributed as dist
import torch.multiprocessing as mp

from flow_amd import fem, parallel
from flow_amd.fem.space import scalar_layout
from oracle import fem_oracle as orc

import oracle_harness as H


def _mesh(world=2):
    '''The channel, long enough for `world` strips (6 and 8 ranks -- a whole
    node -- get strips of 20 cell columns: room for the deep halos).'''
    return fem.karman_channel(36 if world <= 3 else 20 * world, 9)


def _long_mesh(world=2):
    return fem.karman_channel(72 if world <= 3 else 24 * world, 9)


def _system(degree=1, world=2):
    '''SPD system on the channel: stiffness + mass (P1) or mass (P2), as the
    pressure / correction systems; pattern = the layout's pattern.'''
    mesh = _mesh(world)
    S = H.oracle_space(mesh, degree)
    A = orc.mass_matrix(S)
    if degree == 1:
        A = A + orc.stiffness_matrix(S)
    A = A.tocsr()
    A.sort_indices()
    rng = numpy.random.RandomState(degree)
    return mesh, A, rng.standard_normal(S.N)


@pytest.mark.parametrize('world', [1, 2, 3, 5, 6, 8])
def test_strips_cover_the_mesh_and_halo_slots_are_consistent(world):
    mesh = _mesh(world if world > 5 else 2)
    st = parallel.Strips(mesh, world)
    nv = mesh.num_vertices()
    assert st.vbounds[0] == 0 and st.vbounds[-1] == nv
    # every cell that touches a rank's vertices is in its cell range
    owner = numpy.searchsorted(st.vbounds, mesh.cell_vertices, side='right') - 1
    for g in range(world):
        c0, c1 = st.cells[g]
        idx = numpy.nonzero((owner == g).any(axis=1))[0]
        assert c0 <= idx.min() and idx.max() < c1
    for degree in (1, 2):
        lay = scalar_layout(mesh, degree)
        rb = st.blocks(lay)
        n = lay.N
        rowptr, cols = lay.pattern('rowptr'), lay.pattern('cols')
        structs = [rb.struct(g) for g in range(world)]
        owner_of_slot = numpy.full(rb.nhalo, -1)
        for g, s in enumerate(structs):
            assert s.n == n and s.nhalo == rb.nhalo
            assert s.e0 <= s.r0 < s.r1 <= s.e1
            seg = cols[rowptr[s.r0]:rowptr[s.r1]]
            assert seg.min() >= s.e0 and seg.max() < s.e1, \
                'ghost rows cover every referenced column'
            c0, c1 = st.cells[g]
            cd = lay.cell_dofs[c0:c1]
            assert cd.min() >= s.e0 and cd.max() < s.e1, \
                'ghost rows cover every dof of the cells the rank visits'
            for side in (0, 1):
                row, ln, slot = s.send_row[side], s.send_len[side], \
                    s.send_slot[side]
                assert ln == 0 or (s.r0 <= row and row + ln <= s.r1)
                assert (owner_of_slot[slot:slot + ln] == -1).all()
                owner_of_slot[slot:slot + ln] = g
        assert (owner_of_slot >= 0).all()
        assert structs[0].r0 == 0 and structs[-1].r1 == n
        if degree == 2:
            # a P2 dof belongs to the rank of its lowest vertex
            vd = lay.vertex_dofs
            for g, s in enumerate(structs):
                v0, v1 = st.vbounds[g], st.vbounds[g + 1]
                assert s.r0 <= vd[v0] and vd[v1 - 1] < s.r1
        # the exchange: pack own slots, zeros elsewhere, SUM, unpack
        x = numpy.random.RandomState(7).standard_normal(n)
        buf = numpy.zeros(rb.nhalo)
        for s in structs:
            for side in (0, 1):
                row, ln, slot = s.send_row[side], s.send_len[side], \
                    s.send_slot[side]
                buf[slot:slot + ln] += x[row:row + ln]
        for s in structs:
            got = numpy.full(n, numpy.nan)
            got[s.r0:s.r1] = x[s.r0:s.r1]
            for side in (0, 1):
                row, ln, slot = s.recv_row[side], s.recv_len[side], \
                    s.recv_slot[side]
                got[row:row + ln] = buf[slot:slot + ln]
            assert numpy.array_equal(got[s.e0:s.e1], x[s.e0:s.e1]), \
                'ghost rows = the owners values, bit for bit'


def test_strips_too_thin_are_refused():
    mesh = _mesh()
    with pytest.raises(parallel.StripsTooThin):
        parallel.Strips(mesh, 150).blocks(scalar_layout(mesh, 2))
    # (a ValueError of its own, so that callers can choose the algorithm with
    # the shallower halo without catching genuine invariant failures)
    assert issubclass(parallel.StripsTooThin, ValueError)


@pytest.mark.parametrize('world', [2, 3, 6, 8])
@pytest.mark.parametrize('degree,depth', [(1, 2), (2, 6)])
def test_deep_ghost_ranges(world, degree, depth):
    """The deep halos (round 4): ghost ranges `depth` coupling layers out -- two
    for the pressure solve with one collective per iteration, six for the mass
    solver's five products per collective.  Layer d + 1 holds every column of
    the rows of layer d; the halo slots of the deep blocks exchange exactly
    the ghost rows; k products on shrinking ranges reproduce the whole
    matrix' k-th power on the owned rows."""
    mesh = _long_mesh(world)
    lay = scalar_layout(mesh, degree)
    st = parallel.Strips(mesh, world)
    rowptr = lay.pattern('rowptr').astype(numpy.int64)
    cols = lay.pattern('cols').astype(numpy.int64)
    ranges = st.deep_ranges(lay, depth)
    deep = st.deep_blocks(lay, depth)
    n = lay.N
    import scipy.sparse as sp
    A = sp.csr_matrix((numpy.ones(len(cols)), cols, rowptr), shape=(n, n))
    A = sp.diags(1.0 / numpy.asarray(A.sum(axis=1)).ravel()).dot(A).tocsr()
    x = numpy.random.RandomState(0).standard_normal(n)
    # the exchange buffer as the all-reduce leaves it: every rank's send slots
    structs = [deep.struct(g) for g in range(world)]
    buf = numpy.zeros(deep.nhalo)
    for s in structs:
        for side in (0, 1):
            row, ln, slot = s.send_row[side], s.send_len[side], s.send_slot[side]
            buf[slot:slot + ln] += x[row:row + ln]
    want = x.copy()
    for _ in range(depth - 1):
        want = A.dot(want)
    for g, s in enumerate(structs):
        rng = ranges[g]
        assert rng[0] == (s.r0, s.r1) and rng[depth] == (s.e0, s.e1)
        for d in range(depth):
            (lo, hi), (lo1, hi1) = rng[d], rng[d + 1]
            assert lo1 <= lo and hi <= hi1
            seg = cols[rowptr[lo]:rowptr[hi]]
            assert lo1 <= seg.min() and seg.max() < hi1
        # the rank's window: own rows + what the neighbours sent
        win = numpy.full(n, numpy.nan)
        win[s.r0:s.r1] = x[s.r0:s.r1]
        for side in (0, 1):
            row, ln, slot = s.recv_row[side], s.recv_len[side], s.recv_slot[side]
            win[row:row + ln] = buf[slot:slot + ln]
        assert numpy.array_equal(win[s.e0:s.e1], x[s.e0:s.e1])
        # depth - 1 products, product j on the rows of layer depth - 1 - j
        v = win
        for j in range(depth - 1):
            lo, hi = rng[depth - 1 - j]
            nxt = numpy.full(n, numpy.nan)
            nxt[lo:hi] = A[lo:hi].dot(numpy.nan_to_num(v, nan=1e300))
            assert numpy.isfinite(nxt[lo:hi]).all() and \
                abs(nxt[lo:hi]).max() < 1e100, 'a product read outside its layer'
            v = nxt
        assert numpy.allclose(v[s.r0:s.r1], want[s.r0:s.r1], rtol=1e-12,
                              atol=1e-14)


# -- the sharded CG loop under gloo ------------------------------------------------
class NumpyShardCg(object):
    '''What flow_shard_cg_solve does on one rank, in numpy: ext-compact vectors,
    Chronopoulos-Gear recurrences on owned + ghost rows, masked dots, and ONE
    all-reduce per iteration of [r.z, z.w, z.z, |Bb|^2 | halo of w].'''

    def __init__(self, A, s, comm):
        self.A, self.s, self.comm = A, s, comm
        self.me = s.e1 - s.e0
        self.own = numpy.zeros(self.me, dtype=bool)
        self.own[s.r0 - s.e0:s.r1 - s.e0] = True
        self.rows = A[s.r0:s.r1][:, s.e0:s.e1].tocsr()

    def exchange(self, head, w):
        s, buf = self.s, self.comm.buf
        count = 4 + s.nhalo
        b = buf.numpy()
        b[:4] = head
        b[4:count] = 0.0
        for side in (0, 1):
            row, ln, slot = s.send_row[side], s.send_len[side], \
                s.send_slot[side]
            b[4 + slot:4 + slot + ln] = w[row - s.e0:row - s.e0 + ln]
        assert self.comm._allreduce(None, count) == 0
        for side in (0, 1):
            row, ln, slot = s.recv_row[side], s.recv_len[side], \
                s.recv_slot[side]
            w[row - s.e0:row - s.e0 + ln] = b[4 + slot:4 + slot + ln]
        return b[:4].copy()

    def halo(self, v):
        self.exchange(numpy.zeros(4), v)

    def solve(self, b, rtol, maxit):
        s, own = self.s, self.own
        dinv = 1.0 / self.A.diagonal()[s.e0:s.e1]
        bc = b[s.e0:s.e1].copy()
        x = numpy.zeros(self.me)
        r = numpy.zeros(self.me)
        r[own] = bc[own] - self.rows.dot(x)
        self.halo(r)
        z = dinv * r
        p = numpy.zeros(self.me)
        sv = numpy.zeros(self.me)
        w = numpy.zeros(self.me)
        w[own] = self.rows.dot(z)
        zb = dinv * bc
        head = self.exchange(
            [r[own].dot(z[own]), z[own].dot(w[own]), z[own].dot(z[own]),
             zb[own].dot(zb[own])], w)
        target2 = rtol**2 * head[3]
        gamma, alpha = head[0], head[0] / head[1]
        beta = 0.0
        it = 0
        while head[2] > target2 and it < maxit:
            p = z + beta * p
            sv = w + beta * sv
            x += alpha * p
            r -= alpha * sv
            z = dinv * r
            w[own] = self.rows.dot(z)
            head = self.exchange(
                [r[own].dot(z[own]), z[own].dot(w[own]), z[own].dot(z[own]),
                 0.0], w)
            beta = head[0] / gamma
            alpha = head[0] / (head[1] - beta * head[0] / alpha)
            gamma = head[0]
            it += 1
        return x, it


def _free_port():
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def _worker(rank, world, port, degree, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        parallel.enable(dist.group.WORLD, force=True)
        comm = parallel.comm()
        assert comm.rank == rank and comm.world == world and comm.staged
        mesh, A, b = _system(degree, world)
        lay = scalar_layout(mesh, degree)
        s = parallel.strips(mesh).blocks(lay).struct(rank)
        comm.ensure(4 + s.nhalo)
        x, its = NumpyShardCg(A, s, comm).solve(b, 1e-12, 2000)
        out[rank] = (its, s.e0, s.e1, s.r0, s.r1, x, comm.calls)
    finally:
        parallel.disable()
        dist.destroy_process_group()


@pytest.mark.parametrize('world,degree', [(2, 1), (3, 1), (2, 2), (3, 2),
                                          (6, 2), (8, 1)])
def test_sharded_cg_pattern_under_gloo(world, degree):
    _mesh_, A, b = _system(degree, world)
    ref = spla.splu(A.tocsc()).solve(b)
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), degree, out), nprocs=world,
             join=True)
    its = {out[r][0] for r in range(world)}
    assert len(its) == 1, 'every rank runs the same number of iterations'
    n_its = its.pop()
    full = numpy.full(A.shape[0], numpy.nan)
    for r in range(world):
        _, e0, e1, r0, r1, x, calls = out[r]
        full[r0:r1] = x[r0 - e0:r1 - e0]
        # one collective per iteration + the two of the start
        assert calls == n_its + 2, (calls, n_its)
    assert numpy.linalg.norm(full - ref) < 1e-9 * numpy.linalg.norm(ref)
    for r in range(world):
        _, e0, e1, r0, r1, x, _ = out[r]
        # the recurrences on the ghost rows reproduce the owners' values
        assert numpy.array_equal(x, full[e0:e1]), 'ghost rows = owners, bitwise'


# -- the sharded defect-correction mass solver under gloo ---------------------------
class NumpyShardMass(object):
    '''What flow_shard_mass_solve does on one rank, in numpy (fp64 throughout:
    the communication pattern is the subject, not the mixed precision): per
    correction the scaled defect on the OWN rows, ONE all-reduce of [the norms
    of the correction before | the deep halo of the defect], then the Chebyshev
    polynomial's products on shrinking ghost ranges; the verdict on correction
    k arrives with the collective of pass k + 1.'''

    def __init__(self, M, ranges, s, comm, steps, lo, hi):
        self.M, self.rng, self.s, self.comm = M, ranges, s, comm
        self.steps, self.lo, self.hi = steps, lo, hi
        d = M.diagonal()
        import scipy.sparse as sp
        self.A = sp.diags(1.0 / d).dot(M).tocsr()
        self.dinv = 1.0 / d

    def exchange(self, head, rho):
        s, b = self.s, self.comm.buf.numpy()
        count = 4 + s.nhalo
        b[:4] = head
        b[4:count] = 0.0
        for side in (0, 1):
            row, ln, slot = s.send_row[side], s.send_len[side], s.send_slot[side]
            b[4 + slot:4 + slot + ln] = rho[row:row + ln]
        assert self.comm._allreduce(None, count) == 0
        for side in (0, 1):
            row, ln, slot = s.recv_row[side], s.recv_len[side], s.recv_slot[side]
            rho[row:row + ln] = b[4 + slot:4 + slot + ln]
        return b[:4].copy()

    def polynomial(self, rho0):
        '''z = p(D^-1 M) rho0 on the own rows; rho0 valid on the deepest range,
        product j on the rows of layer steps - 1 - j.'''
        theta, delta = 0.5 * (self.hi + self.lo), 0.5 * (self.hi - self.lo)
        sigma = theta / delta
        rk = 1.0 / sigma
        n = len(rho0)
        rho = rho0.copy()
        d = rho / theta
        acc = d.copy()
        for j in range(self.steps - 1):
            lo, hi = self.rng[self.steps - 1 - j]
            nxt = numpy.full(n, numpy.nan)
            nxt[lo:hi] = rho[lo:hi] - self.A[lo:hi].dot(numpy.nan_to_num(
                d, nan=1e300))
            assert abs(nxt[lo:hi]).max() < 1e100, 'read outside the layer'
            rn = 1.0 / (2.0 * sigma - rk)
            dn = numpy.full(n, numpy.nan)
            dn[lo:hi] = rn * rk * d[lo:hi] + 2.0 * rn / delta * nxt[lo:hi]
            rk = rn
            acc[lo:hi] = acc[lo:hi] + dn[lo:hi]
            rho, d = nxt, dn
        return acc

    def solve(self, b, rtol, contraction, maxit=30):
        s = self.s
        n = self.M.shape[0]
        x = numpy.zeros(n)
        zz = xx = 0.0
        for k in range(maxit + 1):
            rho = numpy.full(n, numpy.nan)
            rho[s.r0:s.r1] = self.dinv[s.r0:s.r1] * (
                b[s.r0:s.r1] - self.M[s.r0:s.r1].dot(numpy.nan_to_num(x)))
            head = self.exchange([zz, xx, 0.0, 0.0], rho)
            # the verdict on correction k - 1 (its norms summed over the ranks)
            if k > 0 and contraction * numpy.sqrt(head[0]) <= \
                    rtol * numpy.sqrt(head[1]):
                return x, k
            z = self.polynomial(rho)
            lo1, hi1 = self.rng[1]
            # (x is advanced on the own rows and the first ghost layer: the
            # next defect on the own rows reads it there)
            x[lo1:hi1] += z[lo1:hi1]
            zz = float(z[s.r0:s.r1].dot(z[s.r0:s.r1]))
            xx = float(x[s.r0:s.r1].dot(x[s.r0:s.r1]))
        raise AssertionError('no convergence')


def _mass_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        parallel.enable(dist.group.WORLD, force=True)
        comm = parallel.comm()
        mesh = _long_mesh(world)
        S = H.oracle_space(mesh, 2)
        M = orc.mass_matrix(S).tocsr()
        lay = scalar_layout(mesh, 2)
        steps = 6
        st = parallel.strips(mesh)
        rng = st.deep_ranges(lay, steps)[rank]
        s = st.deep_blocks(lay, steps).struct(rank)
        comm.ensure(4 + s.nhalo)
        b = M.dot(numpy.random.RandomState(4).standard_normal(M.shape[0]))
        from flow_amd.fem import mass as fmass
        lo, hi = fmass.WATHEN[2]
        bound = 1.5 * fmass.chebyshev_contraction(lo, hi, steps)
        x, k = NumpyShardMass(M, rng, s, comm, steps, lo, hi).solve(
            b, 1e-10, bound)
        out[rank] = (k, s.r0, s.r1, x[s.r0:s.r1], comm.calls)
    finally:
        parallel.disable()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_mass_solver_pattern_under_gloo(world):
    '''k corrections cost k + 1 collectives (the last one carries the verdict);
    every rank takes the same number; the own rows assemble the direct
    solution.'''
    mesh = _long_mesh(world)
    M = orc.mass_matrix(H.oracle_space(mesh, 2)).tocsr()
    xref = numpy.random.RandomState(4).standard_normal(M.shape[0])
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(_mass_worker, args=(world, _free_port(), out), nprocs=world,
             join=True)
    ks = {out[r][0] for r in range(world)}
    assert len(ks) == 1
    k = ks.pop()
    assert 3 <= k <= 8
    full = numpy.full(M.shape[0], numpy.nan)
    for r in range(world):
        _k, r0, r1, x, calls = out[r]
        full[r0:r1] = x
        assert calls == k + 1, (calls, k)
    assert numpy.linalg.norm(full - xref) < 1e-9 * numpy.linalg.norm(xref)


# -- the sharded two-level CG with ONE collective per iteration, under gloo --------
def _two_level(A, agg=6, omega=0.6):
    '''A two-level cycle in the product's algebra (flow_mg): Ah = A w D^-1,
    C = R (I - Ah), Ps = (I - w D^-1 A) P;  B r = Ps x' + w D^-1 (2 r - Ah r),
    x' = (P^T A P)^-1 C r.  P: piecewise constant over `agg` consecutive rows
    (any P exercises the pattern).'''
    import scipy.sparse as sp
    n = A.shape[0]
    nc = (n + agg - 1) // agg
    P = sp.csr_matrix((numpy.ones(n), (numpy.arange(n), numpy.arange(n) // agg)),
                      shape=(n, nc))
    wd = omega / A.diagonal()
    Ah = A.dot(sp.diags(wd)).tocsr()
    Ps = (P - sp.diags(wd).dot(A).dot(P)).tocsr()
    C = (P.T - P.T.dot(Ah)).tocsr()
    coarse = spla.splu(P.T.dot(A).dot(P).tocsc())
    return dict(wd=wd, Ah=Ah, Ps=Ps, C=C, coarse=coarse, nc=nc)


def _apply_two_level(T, r):
    x1 = T['coarse'].solve(T['C'].dot(r))
    return T['Ps'].dot(x1) + T['wd'] * (2.0 * r - T['Ah'].dot(r))


class NumpyShardMgCg(object):
    '''What flow_shard_mgcg_solve does on one rank in its one-collective form
    (flow_mg_shard.z_lo / z_hi), in numpy on GLOBAL-length vectors with NaN
    outside what the rank may know: ghost ranges two layers deep; r current
    two layers out through the halo of w that rides with the dots; z formed on
    the owned rows and the FIRST ghost layer; the coarse residual carried by
    CG's own recurrences, its collective part C[:, own] w_own summed in the
    same all-reduce.'''

    def __init__(self, A, T, rng, s, comm):
        self.A, self.T, self.rng, self.s, self.comm = A, T, rng, s, comm
        self.nc = T['nc']

    def exchange(self, head, w, coarse_part):
        s, b = self.s, self.comm.buf.numpy()
        off = 4 + s.nhalo
        count = off + self.nc
        b[:4] = head
        b[4:off] = 0.0
        for side in (0, 1):
            row, ln, slot = s.send_row[side], s.send_len[side], s.send_slot[side]
            b[4 + slot:4 + slot + ln] = w[row:row + ln]
        b[off:count] = coarse_part
        assert self.comm._allreduce(None, count) == 0
        for side in (0, 1):
            row, ln, slot = s.recv_row[side], s.recv_len[side], s.recv_slot[side]
            w[row:row + ln] = b[4 + slot:4 + slot + ln]
        return b[:4].copy(), b[off:count].copy()

    def solve(self, b, rtol, maxit):
        s, T, A = self.s, self.T, self.A
        n = A.shape[0]
        r0, r1 = s.r0, s.r1
        z0, z1 = self.rng[1]                    # owned + first ghost layer
        e0, e1 = s.e0, s.e1                     # two layers out
        own = slice(r0, r1)
        known = numpy.zeros(n, dtype=bool)
        known[e0:e1] = True

        def form_z(r, rc):
            # the up-sweep on [z0, z1): needs r two layers out
            xc = T['coarse'].solve(rc)
            rr = numpy.where(known, r, 1e300)
            z = numpy.zeros(n)
            z[z0:z1] = T['Ps'][z0:z1].dot(xc) + T['wd'][z0:z1] * (
                2.0 * r[z0:z1] - T['Ah'][z0:z1].dot(rr))
            assert abs(z[z0:z1]).max() < 1e100, 'z read r outside two layers'
            return z

        def A_own(z):
            zz = numpy.zeros(n) + 1e300
            zz[z0:z1] = z[z0:z1]
            w = numpy.zeros(n)
            w[own] = A[own].dot(zz)
            assert abs(w[own]).max() < 1e100, 'A z read z outside one layer'
            return w

        zero = numpy.zeros(self.nc)
        # start (x = 0): r = b on the own rows, its halo; the coarse residual
        # and |B b| need collectives of their own -- start-up, not the loop
        r = numpy.zeros(n)
        r[own] = b[own]
        _h, _c = self.exchange(numpy.zeros(4), r, zero)
        _h, rc_r = self.exchange(numpy.zeros(4), numpy.zeros(n),
                                 T['C'][:, own].dot(r[own]))
        z = form_z(r, rc_r)
        head, _c = self.exchange([0.0, 0.0, 0.0, z[own].dot(z[own])],
                                 numpy.zeros(n), zero)
        target2 = rtol**2 * head[3]
        x = numpy.zeros(n)
        p = numpy.zeros(n)
        sv = numpy.zeros(n)
        rc_s = numpy.zeros(self.nc)
        w = A_own(z)
        head, rc_w = self.exchange(
            [r[own].dot(z[own]), z[own].dot(w[own]), z[own].dot(z[own]), 0.0],
            w, T['C'][:, own].dot(w[own]))
        gamma, alpha = head[0], head[0] / head[1]
        beta = 0.0
        it = 0
        while head[2] > target2 and it < maxit:
            p = z + beta * p
            sv = w + beta * sv
            x += alpha * p
            r[e0:e1] -= alpha * sv[e0:e1]
            rc_s = rc_w + beta * rc_s
            rc_r = rc_r - alpha * rc_s
            z = form_z(r, rc_r)
            w = A_own(z)
            head, rc_w = self.exchange(
                [r[own].dot(z[own]), z[own].dot(w[own]), z[own].dot(z[own]),
                 0.0], w, T['C'][:, own].dot(w[own]))
            beta = head[0] / gamma
            alpha = head[0] / (head[1] - beta * head[0] / alpha)
            gamma = head[0]
            it += 1
        return x, it


def _reference_two_level_cg(A, T, b, rtol, maxit):
    '''The same recurrences on the whole system (one process).'''
    x = numpy.zeros_like(b)
    r = b.copy()
    z = _apply_two_level(T, r)
    zb = z.copy()
    target2 = rtol**2 * zb.dot(zb)
    w = A.dot(z)
    g, d = r.dot(z), z.dot(w)
    gamma, alpha, beta = g, g / d, 0.0
    p = numpy.zeros_like(b)
    sv = numpy.zeros_like(b)
    it = 0
    while z.dot(z) > target2 and it < maxit:
        p = z + beta * p
        sv = w + beta * sv
        x += alpha * p
        r -= alpha * sv
        z = _apply_two_level(T, r)
        w = A.dot(z)
        g, d = r.dot(z), z.dot(w)
        beta = g / gamma
        alpha = g / (d - beta * g / alpha)
        gamma = g
        it += 1
    return x, it


def _mgcg_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        parallel.enable(dist.group.WORLD, force=True)
        comm = parallel.comm()
        mesh, A, b = _system(1, world)
        lay = scalar_layout(mesh, 1)
        T = _two_level(A)
        st = parallel.strips(mesh)
        rng = st.deep_ranges(lay, 2)[rank]
        s = st.deep_blocks(lay, 2).struct(rank)
        comm.ensure(4 + s.nhalo + T['nc'])
        x, its = NumpyShardMgCg(A, T, rng, s, comm).solve(b, 1e-10, 500)
        out[rank] = (its, s.r0, s.r1, rng[1], x, comm.calls)
    finally:
        parallel.disable()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 6])
def test_one_collective_two_level_cg_pattern_under_gloo(world):
    '''The one-collective form of the sharded V-cycle CG as an algorithm: the
    same iterates as the whole-system recurrences (same count, same solution),
    ONE collective per iteration + the start-up ones, x valid on the owned rows
    and the first ghost layer.'''
    _mesh_, A, b = _system(1, world)
    T = _two_level(A)
    xref, its_ref = _reference_two_level_cg(A, T, b, 1e-10, 500)
    direct = spla.splu(A.tocsc()).solve(b)
    assert numpy.linalg.norm(xref - direct) < 1e-8 * numpy.linalg.norm(direct)
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(_mgcg_worker, args=(world, _free_port(), out), nprocs=world,
             join=True)
    for r in range(world):
        its, r0, r1, (z0, z1), x, calls = out[r]
        assert its == its_ref, (its, its_ref)
        assert calls == its + 4, (calls, its)
        # owned rows and the first ghost layer carry the solution
        assert numpy.allclose(x[z0:z1], xref[z0:z1], rtol=1e-9, atol=1e-12)


def _vote_worker(rank, world, port, cases_in, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from flow_amd import parallel
        from flow_amd.navier_stokes import start_vectors as sv
        parallel.enable(dist.group.WORLD)
        st = sv._State()
        out[rank] = [sv._agree(st, *case[rank]) for case in cases_in]
    finally:
        dist.destroy_process_group()


def test_trajectory_verdicts_are_unanimous_or_void():
    '''Which trajectory a call on the strips continues (start_vectors._agree):
    every rank votes for (code, trajectory) in a one-hot slot; the verdict
    stands only if one slot holds ALL the votes.  The split the sums of
    (code, which) could not see (local matches 0, 1, 1, 2 on four
    ranks, whose sum equals 4 x 1) is void on every rank.'''
    world = 4
    cases_in = [
        [(1, 1)] * 4,                                  # unanimous: stands
        [(1, 0), (1, 1), (1, 1), (1, 2)],              # the advisor's split
        [(2, 0)] * 3 + [(1, 0)],                       # same trajectory, other code
        [(0, None)] * 4,                               # nobody matches
        [(1, 2), (0, None), (1, 2), (1, 2)],           # one rank sees nothing
        [(2, 2)] * 4,
        ]
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(_vote_worker, args=(world, _free_port(), cases_in, out),
             nprocs=world, join=True)
    want = [(1, 1), (0, None), (0, None), (0, None), (0, None), (2, 2)]
    for r in range(world):
        assert list(out[r]) == want, (r, out[r])
