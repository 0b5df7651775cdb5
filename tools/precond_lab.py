# -*- coding: utf-8 -*-
'''
Development aid (CPU, scipy; not part of the product): which preconditioner for
the Newton systems of the tentative velocity?  Builds the oracle's Jacobian of a
Karman channel in the NON-DIMENSIONAL regime of the 10 M-DoF workload (same
CFL number u dt / h and diffusion number nu dt / h^2: the viscosity is scaled up
with the mesh width) and counts flexible-GMRES(10) iterations to a given
residual reduction for

  mc-ilu     multicolour ILU(0) of the two diagonal blocks (what the product runs)
  nat-ilu    natural-order ILU(0) (not parallel: yardstick)
  jacobi
  mg-*       V-cycles on the diagonal blocks (smoothed aggregation, p-multigrid)

    python tools/precond_lab.py --nx 300
'''
import argparse
import ctypes
import os
import sys
import time

import numpy
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))

from flow_amd import fem, karman                            # noqa: E402
from flow_amd.fem.bcs import collect                        # noqa: E402
from flow_amd.fem import reference                          # noqa: E402
from oracle import fem_oracle as orc                        # noqa: E402
import oracle_harness as H                                  # noqa: E402

_lib = ctypes.CDLL(os.path.join(HERE, 'lab', 'libilu0.so'))
_P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


# -- the system ------------------------------------------------------------------
def build_system(nx, nsteps=3, cache=True):
    path = '/tmp/precond_lab_%d.npz' % nx
    prob = karman.KarmanProblem(nx)
    mesh = prob.mesh
    W = H.oracle_space(mesh, 2)
    P = H.oracle_space(mesh, 1)
    h_ratio = 2182.0 / nx
    rho = prob.rho
    mu = 0.002 * h_ratio            # same cell Peclet number as at full size
    unorm = 0.0159
    dt = mesh.hmax() / unorm        # the controller's plateau step
    u_bc = collect(prob.u_bcs, prob.W.size())
    p_bc = collect(prob.p_bcs, prob.P.size())
    zero = (reference.lattice(0), numpy.zeros((mesh.num_cells(), 1, 2)))
    if cache and os.path.exists(path):
        d = numpy.load(path)
        u0, p0 = d['u0'], d['p0']
    else:
        prob.set_initial_profile()
        u0 = prob.u0.array().copy()
        p0 = numpy.zeros(P.N)
        for k in range(nsteps):
            t0 = time.time()
            u0, p0, _ = orc.step(W, P, u0, p0, zero, zero, u_bc, p_bc, rho, mu,
                                 dt, scheme='rotational')
            print('oracle step %d: %.1f s, |u|max %.4f' % (
                k, time.time() - t0, abs(u0).max()), flush=True)
        numpy.savez(path, u0=u0, p0=p0)
    # Newton system at ui = u0 (backward Euler): J dx = F
    M1 = orc.mass_matrix(W)
    M = sp.block_diag([M1] * 2, format='csr')
    Ri, dRi = orc.momentum_rhs(W, P, u0, p0, zero, rho, mu)
    F = -dt / rho * Ri
    J = (M - dt / rho * dRi).tocsr()
    bc = u_bc[0]
    F[bc] = u0[bc] - u_bc[1]
    keep = numpy.ones(J.shape[0])
    keep[bc] = 0.0
    J = (sp.diags(keep).dot(J) + sp.diags(1.0 - keep)).tocsr()
    J.sort_indices()
    K1 = orc.stiffness_matrix(W)
    info = dict(dt=dt, mu=mu, rho=rho, nu=mu / rho, h=0.6 / nx, N=W.N,
                cfl=unorm * dt / (0.6 / nx), diff=mu / rho * dt / (0.6 / nx)**2)
    lay = prob.W.layout
    return dict(J=J, F=F, M1=M1.tocsr(), K1=K1.tocsr(), bc=bc, info=info,
                lay=lay, mesh=mesh, W=W, P=P, play=prob.P.layout)


def diag_blocks(J, n):
    J = J.tocsr()
    return J[:n, :n].tocsr(), J[n:, n:].tocsr()


# -- ILU(0) ------------------------------------------------------------------------
class Ilu0(object):
    def __init__(self, A, order=None):
        A = A.tocsr()
        n = A.shape[0]
        self.order = order
        if order is not None:
            A = A[order][:, order].tocsr()
        A.sort_indices()
        self.n = n
        self.rowptr = A.indptr.astype(numpy.int32)
        self.cols = A.indices.astype(numpy.int32)
        rows = numpy.repeat(numpy.arange(n), numpy.diff(A.indptr))
        self.diag = numpy.nonzero(self.cols == rows)[0].astype(numpy.int32)
        assert len(self.diag) == n
        self.lu = A.data.astype(numpy.float64).copy()
        rc = _lib.ilu0_factor(n, _P(self.rowptr), _P(self.cols), _P(self.diag),
                              _P(self.lu))
        assert rc == 0, rc

    def solve(self, b):
        if self.order is not None:
            b = b[self.order]
        b = numpy.ascontiguousarray(b, dtype=numpy.float64)
        x = numpy.empty(self.n)
        _lib.ilu0_solve(self.n, _P(self.rowptr), _P(self.cols), _P(self.diag),
                        _P(self.lu), _P(b), _P(x))
        if self.order is not None:
            out = numpy.empty(self.n)
            out[self.order] = x
            return out
        return x


def greedy_colour_order(A):
    A = A.tocsr()
    n = A.shape[0]
    colour = numpy.full(n, -1, dtype=numpy.int64)
    indptr, indices = A.indptr, A.indices
    for i in range(n):
        used = set(colour[indices[indptr[i]:indptr[i + 1]]])
        c = 0
        while c in used:
            c += 1
        colour[i] = c
    return numpy.argsort(colour, kind='stable'), int(colour.max()) + 1


# -- multigrid -----------------------------------------------------------------------
def bin_aggregates(x, free, width):
    ix = numpy.floor((x[:, 0] - x[:, 0].min()) / width + 1e-9).astype(numpy.int64)
    iy = numpy.floor((x[:, 1] - x[:, 1].min()) / width + 1e-9).astype(numpy.int64)
    key = ix * 2000003 + iy
    key[~free] = -1
    ukey, agg = numpy.unique(key, return_inverse=True)
    if len(ukey) and ukey[0] == -1:
        return agg - 1, len(ukey) - 1
    return agg, len(ukey)


class SaMg(object):
    '''Smoothed-aggregation V(nu,nu) cycle with damped-Jacobi or Chebyshev
    smoothing on a (possibly nonsymmetric) matrix A; aggregates by spatial
    binning as flow_amd/fem/multigrid.py.'''

    def __init__(self, A, x, free, width, s=3.0, coarsest=500, omega=0.8, nu=1,
                 smoother='jacobi', cheb_deg=2, first_P=None, max_levels=8):
        self.levels = []
        self.nu = nu
        self.omega = omega
        self.smoother = smoother
        self.cheb_deg = cheb_deg
        A = A.tocsr()
        rng = numpy.random.RandomState(1)
        while A.shape[0] > coarsest and len(self.levels) < max_levels:
            m = A.shape[0]
            D = A.diagonal()
            v = rng.standard_normal(m)
            lam = 1.0
            for _ in range(20):
                v = A.dot(v) / D
                lam = numpy.linalg.norm(v)
                v /= lam
            if first_P is not None and not self.levels:
                P = first_P.tocsr()
                nc = P.shape[1]
                xc, freec = first_P_coords
            else:
                agg, nc = bin_aggregates(x, free, width)
                if nc < 2 or nc >= m:
                    break
                idx = numpy.nonzero(agg >= 0)[0]
                P0 = sp.csr_matrix((numpy.ones(len(idx)), (idx, agg[idx])),
                                   shape=(m, nc))
                As = 0.5 * (A + A.T)
                P = (P0 - sp.diags((4.0 / (3.0 * lam)) / D).dot(As.dot(P0))
                     ).tocsr()
                cnt = numpy.bincount(agg[idx], minlength=nc)
                xc = numpy.stack([
                    numpy.bincount(agg[idx], weights=x[idx, d], minlength=nc)
                    / cnt for d in (0, 1)], axis=1)
                freec = numpy.ones(nc, dtype=bool)
                width *= s
            Ac = (P.T.dot(A.dot(P))).tocsr()
            self.levels.append(dict(A=A, D=D, P=P, lam=lam))
            A, x, free = Ac, xc, freec
        self.Ac = spla.splu(A.tocsc())
        self.sizes = [l['A'].shape[0] for l in self.levels] + [A.shape[0]]
        self.nnz = [l['A'].nnz for l in self.levels] + [A.nnz]

    def smooth(self, l, x, r):
        L = self.levels[l]
        A, D = L['A'], L['D']
        if self.smoother == 'jacobi':
            for _ in range(self.nu):
                x = x + self.omega * (r - A.dot(x)) / D if x is not None \
                    else self.omega * r / D
            return x
        # Chebyshev on [lam/a, lam] of D^-1 A
        hi = 1.1 * L['lam']
        lo = hi / 8.0
        theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
        if x is None:
            x = numpy.zeros_like(r)
        res = r - A.dot(x) if x.any() else r.copy()
        sigma = theta / delta
        rho_k = 1.0 / sigma
        d = res / D / theta
        for k in range(self.cheb_deg):
            x = x + d
            if k + 1 < self.cheb_deg:
                res = res - A.dot(d)
                rho_n = 1.0 / (2.0 * sigma - rho_k)
                d = rho_n * rho_k * d + 2.0 * rho_n / delta * res / D
                rho_k = rho_n
        return x

    def cycle(self, l, r):
        if l == len(self.levels):
            return self.Ac.solve(r)
        L = self.levels[l]
        x = self.smooth(l, None, r)
        rc = L['P'].T.dot(r - L['A'].dot(x))
        x = x + L['P'].dot(self.cycle(l + 1, rc))
        return self.smooth(l, x, r)

    def solve(self, r):
        return self.cycle(0, r)


first_P_coords = None


def p2_to_p1_prolongation(lay2, lay1, mesh):
    '''P1 embedded in P2: vertex dofs copy, edge dofs average their ends.'''
    cd2 = lay2.cell_dofs            # (nc, 6): 3 vertex + 3 edge dofs
    cd1 = lay1.cell_dofs            # (nc, 3)
    n2, n1 = lay2.N, lay1.N
    rows, cols, vals = [], [], []
    # vertex dofs
    rows.append(cd2[:, :3].ravel())
    cols.append(cd1.ravel())
    vals.append(numpy.ones(cd1.size))
    # edge dofs: find which vertices each edge dof sits between by coordinates
    x2 = lay2.dof_coords
    x1 = lay1.dof_coords
    for e in range(3):
        ed = cd2[:, 3 + e]
        best = None
        for (a, b) in ((0, 1), (1, 2), (0, 2)):
            mid = 0.5 * (x1[cd1[:, a]] + x1[cd1[:, b]])
            ok = numpy.abs(mid - x2[ed]).max(axis=1) < 1e-12
            if best is None:
                best = numpy.full((len(ed), 2), -1, dtype=numpy.int64)
            best[ok, 0] = cd1[ok, a]
            best[ok, 1] = cd1[ok, b]
        assert (best >= 0).all()
        for k in (0, 1):
            rows.append(ed)
            cols.append(best[:, k])
            vals.append(numpy.full(len(ed), 0.5))
    Pm = sp.csr_matrix((numpy.concatenate(vals),
                        (numpy.concatenate(rows), numpy.concatenate(cols))),
                       shape=(n2, n1))
    # duplicates (shared vertices / edges) were summed: normalise rows
    Pm.data[:] = 1.0
    Pm = Pm.tocsr()
    cnt = numpy.diff(Pm.indptr)
    Pm.data = numpy.repeat(1.0 / cnt, cnt)
    return Pm


# -- flexible GMRES ------------------------------------------------------------------
def fgmres(A, b, M, rtol, restart=10, maxit=400):
    n = len(b)
    x = numpy.zeros(n)
    bn = numpy.linalg.norm(b)
    its = 0
    hist = []
    while its < maxit:
        r = b - A.dot(x)
        beta = numpy.linalg.norm(r)
        hist.append(beta / bn)
        if beta <= rtol * bn:
            break
        V = numpy.zeros((restart + 1, n))
        Z = numpy.zeros((restart, n))
        Hm = numpy.zeros((restart + 1, restart))
        V[0] = r / beta
        g = numpy.zeros(restart + 1)
        g[0] = beta
        k_used = 0
        for j in range(restart):
            Z[j] = M(V[j])
            w = A.dot(Z[j])
            for i in range(j + 1):
                Hm[i, j] = w.dot(V[i])
                w -= Hm[i, j] * V[i]
            Hm[j + 1, j] = numpy.linalg.norm(w)
            V[j + 1] = w / Hm[j + 1, j]
            its += 1
            k_used = j + 1
            y, res, _, _ = numpy.linalg.lstsq(Hm[:j + 2, :j + 1], g[:j + 2],
                                              rcond=None)
            rn = numpy.linalg.norm(g[:j + 2] - Hm[:j + 2, :j + 1].dot(y))
            hist.append(rn / bn)
            if rn <= rtol * bn or its >= maxit:
                break
        x = x + Z[:k_used].T.dot(y)
        if rn <= rtol * bn:
            break
    return x, its, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=300)
    ap.add_argument('--rtol', type=float, default=1e-8)
    ap.add_argument('--only', default='')
    ap.add_argument('--lab2', action='store_true')
    args = ap.parse_args()
    if args.lab2:
        return lab2(args)
    S = build_system(args.nx)
    J, F, info = S['J'], S['F'], S['info']
    n = info['N']
    print('system: N = %d per component, cfl %.2f, diffusion number %.2f, '
          'dt %.4g, |F| %.3e' % (n, info['cfl'], info['diff'], info['dt'],
                                 numpy.linalg.norm(F)), flush=True)
    B0, B1 = diag_blocks(J, n)
    lay = S['lay']
    x2 = lay.dof_coords
    isbc = numpy.zeros(2 * n, dtype=bool)
    isbc[S['bc']] = True
    rng = numpy.random.RandomState(0)
    rhs = [('F', F), ('random', rng.standard_normal(2 * n) * (~isbc))]
    only = set(args.only.split(',')) if args.only else None

    def blockwise(s0, s1):
        return lambda v: numpy.concatenate([s0(v[:n]), s1(v[n:])])

    precs = {}

    def want(name):
        return only is None or name in only

    if want('jacobi'):
        d = J.diagonal()
        precs['jacobi'] = lambda v: v / d
    if want('nat-ilu'):
        i0, i1 = Ilu0(B0), Ilu0(B1)
        precs['nat-ilu'] = blockwise(i0.solve, i1.solve)
    if want('mc-ilu'):
        t0 = time.time()
        order, nc = greedy_colour_order(B0)
        print('greedy colouring: %d colours (%.1f s)' % (nc, time.time() - t0))
        m0, m1 = Ilu0(B0, order), Ilu0(B1, order)
        precs['mc-ilu'] = blockwise(m0.solve, m1.solve)
    width = 3.0 * numpy.sqrt(2.0 * S['mesh'].cell_areas().mean())
    # symmetric part without convection: the same matrix for both components
    nu_dt = info['nu'] * info['dt']
    for name, kw in (
            ('mg-jac1', dict(nu=1)),
            ('mg-jac2', dict(nu=2)),
            ('mg-cheb2', dict(smoother='cheb', cheb_deg=2)),
            ('mg-cheb3', dict(smoother='cheb', cheb_deg=3)),
            ):
        if not want(name):
            continue
        t0 = time.time()
        g0 = SaMg(B0, x2, ~isbc[:n], width, **kw)
        g1 = SaMg(B1, x2, ~isbc[n:], width, **kw)
        print('%s: levels %r nnz %r (%.1f s)' % (name, g0.sizes, g0.nnz,
                                                 time.time() - t0))
        precs[name] = blockwise(g0.solve, g1.solve)
    if want('pmg-cheb2') or want('pmg-jac1') or want('pmg-cheb3'):
        global first_P_coords
        Pm = p2_to_p1_prolongation(lay, S['play'], S['mesh'])
        x1 = S['play'].dof_coords
        for name, kw in (('pmg-jac1', dict(nu=1)),
                         ('pmg-cheb2', dict(smoother='cheb', cheb_deg=2)),
                         ('pmg-cheb3', dict(smoother='cheb', cheb_deg=3))):
            if not want(name):
                continue
            gs = []
            for blk, bcmask in ((B0, isbc[:n]), (B1, isbc[n:])):
                # P1 dofs whose P2 vertex dof is a Dirichlet dof stay out
                vd = lay.vertex_dofs
                free1 = ~bcmask[vd]
                Pb = Pm.dot(sp.diags(free1.astype(float))).tocsr()
                Pb = sp.diags((~bcmask).astype(float)).dot(Pb).tocsr()
                # drop empty columns
                colsum = numpy.asarray(abs(Pb).sum(axis=0)).ravel()
                keepc = numpy.nonzero(colsum > 0)[0]
                Pb = Pb[:, keepc].tocsr()
                first_P_coords = (x1[keepc], numpy.ones(len(keepc), dtype=bool))
                gs.append(SaMg(blk, x2, ~bcmask, width, first_P=Pb, **kw))
            print('%s: levels %r nnz %r' % (name, gs[0].sizes, gs[0].nnz))
            precs[name] = blockwise(gs[0].solve, gs[1].solve)

    for rname, b in rhs:
        print('--- right-hand side: %s' % rname)
        for name, Mi in precs.items():
            t0 = time.time()
            x, its, hist = fgmres(J, b, Mi, args.rtol)
            true = numpy.linalg.norm(b - J.dot(x)) / numpy.linalg.norm(b)
            print('%-10s %4d iterations  (true rel. residual %.1e)  %.1f s' % (
                name, its, true, time.time() - t0), flush=True)


# -- round 2 of the lab: P2 -> P1 two-level cycle, Chebyshev on both levels -----------
class Cheb(object):
    '''k steps of the Chebyshev iteration for D^-1 A on [hi/ratio, hi].'''

    def __init__(self, A, k, ratio=8.0, lam=None, safety=1.1, fp32=False):
        self.A = A.tocsr()
        if fp32:
            self.A = self.A.copy()
            self.A.data = self.A.data.astype(numpy.float32).astype(numpy.float64)
        self.D = self.A.diagonal()
        self.k = k
        if lam is None:
            rng = numpy.random.RandomState(3)
            v = rng.standard_normal(A.shape[0])
            for _ in range(20):
                v = self.A.dot(v) / self.D
                lam = numpy.linalg.norm(v)
                v /= lam
        self.lam = lam
        self.hi = safety * lam
        self.lo = self.hi / ratio
        self.products = 0

    def run(self, r, x=None):
        '''x <- x + p_k(D^-1 A) D^-1 (r - A x)'''
        A, D = self.A, self.D
        theta, delta = 0.5 * (self.hi + self.lo), 0.5 * (self.hi - self.lo)
        sigma = theta / delta
        if x is None:
            x = numpy.zeros_like(r)
            res = r.copy()
        else:
            res = r - A.dot(x)
            self.products += 1
        rho_k = 1.0 / sigma
        d = res / D / theta
        for k in range(self.k):
            x = x + d
            if k + 1 < self.k:
                res = res - A.dot(d)
                self.products += 1
                rho_n = 1.0 / (2.0 * sigma - rho_k)
                d = rho_n * rho_k * d + 2.0 * rho_n / delta * res / D
                rho_k = rho_n
        return x


class TwoLevel(object):
    def __init__(self, A, P, Ac, pre, post, coarse, ratio_f=8.0, ratio_c=30.0,
                 fp32=False):
        self.A, self.P = A.tocsr(), P.tocsr()
        self.pre = Cheb(A, pre, ratio_f, fp32=fp32) if pre else None
        self.post = Cheb(A, post, ratio_f, fp32=fp32) if post else None
        if coarse == 'lu':
            lu = spla.splu(Ac.tocsc())
            self.coarse = lambda r: lu.solve(r)
        else:
            ch = Cheb(Ac, coarse, ratio_c, fp32=fp32)
            self.coarse = lambda r: ch.run(r)
        self.fine_products = 0

    def solve(self, r):
        x = self.pre.run(r) if self.pre else numpy.zeros_like(r)
        res = r - self.A.dot(x) if self.pre else r
        x = x + self.P.dot(self.coarse(self.P.T.dot(res)))
        if self.post:
            x = self.post.run(r, x)
        return x


def lab2(args):
    S = build_system(args.nx)
    J, F, info = S['J'], S['F'], S['info']
    n = info['N']
    B0, B1 = diag_blocks(J, n)
    lay, play = S['lay'], S['play']
    isbc = numpy.zeros(2 * n, dtype=bool)
    isbc[S['bc']] = True
    Pm = p2_to_p1_prolongation(lay, play, S['mesh'])
    vd = lay.vertex_dofs
    # rediscretised P1 operator (velocity interpolated to the vertices)
    W1 = H.oracle_space(S['mesh'], 1)
    u0 = numpy.load('/tmp/precond_lab_%d.npz' % args.nx)['u0']
    p0 = numpy.load('/tmp/precond_lab_%d.npz' % args.nx)['p0']
    u1 = numpy.concatenate([u0[:n][vd], u0[n:][vd]])
    zero = (reference.lattice(0), numpy.zeros((S['mesh'].num_cells(), 1, 2)))
    _, dR1 = orc.momentum_rhs(W1, S['P'], u1, p0, zero, info['rho'], info['mu'])
    M1 = orc.mass_matrix(W1)
    n1 = W1.N
    J1 = (sp.block_diag([M1] * 2) - info['dt'] / info['rho'] * dR1).tocsr()
    R0, R1 = diag_blocks(J1, n1)

    def blockwise(s0, s1):
        return lambda v: numpy.concatenate([s0(v[:n]), s1(v[n:])])

    def make(pre, post, coarse, galerkin=True, **kw):
        out = []
        for blk, red, bcmask in ((B0, R0, isbc[:n]), (B1, R1, isbc[n:])):
            free1 = ~bcmask[vd]
            Pb = sp.diags((~bcmask).astype(float)).dot(Pm).dot(
                sp.diags(free1.astype(float))).tocsr()
            if galerkin:
                Ac = Pb.T.dot(blk.dot(Pb)).tocsr()
            else:
                f = sp.diags(free1.astype(float))
                Ac = f.dot(red).dot(f).tocsr()
            Ac = (Ac + sp.diags((~free1).astype(float))).tocsr()
            out.append(TwoLevel(blk, Pb, Ac, pre, post, coarse, **kw))
        return blockwise(out[0].solve, out[1].solve)

    rng = numpy.random.RandomState(0)
    b = F
    cases = [('2/2 cheb4 rc8', dict(pre=2, post=2, coarse=4, ratio_c=8.0))]
    for name, kw in cases:
        Mi = make(**kw)
        x, its, hist = fgmres(J, b, Mi, args.rtol)
        print('%-20s %4d iterations' % (name, its), flush=True)

    # -- variants of the smoother MATRIX (the operator stays J) --------------------
    def rounded(A, dtype):
        '''D^-1 A rounded to `dtype`, scaled back.'''
        A = A.tocsr().copy()
        D = A.diagonal()
        S = sp.diags(1.0 / D).dot(A).tocsr()
        S.data = S.data.astype(dtype).astype(numpy.float64)
        return sp.diags(D).dot(S).tocsr()

    def make2(fine_of, coarse_of, pre=2, post=2, coarse=4):
        out = []
        for a, (blk, red, bcmask) in enumerate(((B0, R0, isbc[:n]),
                                                (B1, R1, isbc[n:]))):
            free1 = ~bcmask[vd]
            Pb = sp.diags((~bcmask).astype(float)).dot(Pm).dot(
                sp.diags(free1.astype(float))).tocsr()
            f = sp.diags(free1.astype(float))
            Ac = (f.dot(coarse_of(a)).dot(f)
                  + sp.diags((~free1).astype(float))).tocsr()
            out.append(TwoLevel(fine_of(a), Pb, Ac, pre, post, coarse,
                                ratio_c=8.0))
        return blockwise(out[0].solve, out[1].solve)

    Bs = [B0, B1]
    Rs = [R0, R1]
    half = numpy.float16
    variants = [
        ('own blocks fp64', lambda a: Bs[a], lambda a: Rs[a]),
        ('own blocks fp16', lambda a: rounded(Bs[a], half),
         lambda a: rounded(Rs[a], half)),
        ('own fine fp16, coarse fp32', lambda a: rounded(Bs[a], half),
         lambda a: rounded(Rs[a], numpy.float32)),
        ]
    # one averaged matrix for both components (Dirichlet rows differ per
    # component: keep each component's identity rows)
    def averaged(blocks, masks):
        avg = 0.5 * (blocks[0] + blocks[1])
        out = []
        for m in masks:
            keep = sp.diags((~m).astype(float))
            out.append((keep.dot(avg) + sp.diags(m.astype(float))).tocsr())
        return out
    Ba = averaged(Bs, [isbc[:n], isbc[n:]])
    Ra = averaged(Rs, [isbc[:n][vd], isbc[n:][vd]])
    variants += [
        ('averaged blocks', lambda a: Ba[a], lambda a: Ra[a]),
        ('averaged fine, own coarse', lambda a: Ba[a], lambda a: Rs[a]),
        ('averaged blocks fp16', lambda a: rounded(Ba[a], half),
         lambda a: rounded(Ra[a], half)),
        ]
    for name, fo, co in variants:
        Mi = make2(fo, co)
        x, its, hist = fgmres(J, b, Mi, args.rtol)
        print('%-32s %4d iterations' % (name, its), flush=True)


if __name__ == '__main__':
    main()
