# -*- coding: utf-8 -*-
'''
Development aid (CPU, scipy; not part of the product): which preconditioner for
the Newton systems of the tentative velocity where the Chebyshev-smoothed
p-multigrid cycle is REJECTED by its acceptance test -- cell Peclet numbers
4-11 at CFL-sized steps (the ~1 M-DoF channels at the driver's viscosity,
profiles/graded_mesh_r05.txt)?  The oracle's Jacobian of a Karman channel of
`--nx` columns in the non-dimensional regime of a channel of `--like` columns
(same CFL number and cell Peclet number: the viscosity is scaled with the mesh
width), flexible GMRES(10) to `--rtol`, preconditioned with

  mc-ilu          multicolour ILU(0) of the two diagonal blocks (the fallback
                  the product runs there today)
  nat-ilu         natural-order ILU(0) (not parallel: yardstick)
  chains-b        ILU(0) with chains of b consecutive dofs of the numbering
                  (x-major: ACROSS the flow) kept together, chains coloured
  pmg-cheb        the product's cycle (1 + 2 Chebyshev steps, 6 on P1)
  tl-ilu/<c>      two-level cycle, ONE multicolour-ILU(0) sweep before and one
                  after the coarse correction, the rediscretised P1 level
                  treated with <c>: lu (ideal), cheb6, ilu (one mc-ILU(0)
                  application), ilu2 (two: x += ILU^-1 (r - A x))
  tl-ilu-pre/<c>  the same without the post-smoothing sweep

    python tools/smoother_lab.py --nx 200 --like 680
'''
import argparse
import os
import sys
import time

import numpy
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
sys.path.insert(0, HERE)

from flow_amd import karman                                 # noqa: E402
from flow_amd.fem.bcs import collect                        # noqa: E402
from flow_amd.fem import reference                          # noqa: E402
from oracle import fem_oracle as orc                        # noqa: E402
import oracle_harness as H                                  # noqa: E402
import precond_lab as L                                     # noqa: E402
from heat_ilu_lab import block_colour_order                 # noqa: E402


def build_system(nx, like, nsteps, cfl_scale=1.0, graded=None):
    '''graded = (lcar, like_lcar): the unstructured graded Delaunay channel of
    fem.karman_channel_graded at `lcar` in the regime of the one at
    `like_lcar`.'''
    if graded is not None:
        from flow_amd import fem
        path = '/tmp/smoother_lab_g%g_%g_%g.npz' % (graded + (cfl_scale,))
        prob = karman.KarmanProblem(
            mesh=fem.karman_channel_graded(graded[0]).reordered())
        nx, like = 1.0 / graded[0], 1.0 / graded[1]
    else:
        path = '/tmp/smoother_lab_%d_%d_%g.npz' % (nx, like, cfl_scale)
        prob = karman.KarmanProblem(nx)
    mesh = prob.mesh
    W = H.oracle_space(mesh, 2)
    P = H.oracle_space(mesh, 1)
    rho = prob.rho
    mu = 0.002 * float(like) / nx
    unorm = 0.0159
    dt = cfl_scale * mesh.hmax() / unorm
    u_bc = collect(prob.u_bcs, prob.W.size())
    p_bc = collect(prob.p_bcs, prob.P.size())
    zero = (reference.lattice(0), numpy.zeros((mesh.num_cells(), 1, 2)))
    if os.path.exists(path):
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
    h = mesh.hmax() / 2**0.5 if graded is not None else 0.6 / nx
    nu = mu / rho
    info = dict(dt=dt, mu=mu, rho=rho, nu=nu, h=h, N=W.N,
                cfl=unorm * dt / h, peclet=unorm * h / (2.0 * nu))
    # rediscretised P1 level at the vertex values
    lay, play = prob.W.layout, prob.P.layout
    vd = lay.vertex_dofs
    n = W.N
    W1 = H.oracle_space(mesh, 1)
    u1 = numpy.concatenate([u0[:n][vd], u0[n:][vd]])
    _, dR1 = orc.momentum_rhs(W1, P, u1, p0, zero, rho, mu)
    Mc = orc.mass_matrix(W1)
    J1 = (sp.block_diag([Mc] * 2) - dt / rho * dR1).tocsr()
    return dict(J=J, F=F, bc=bc, info=info, lay=lay, play=play, mesh=mesh,
                J1=J1, n1=W1.N)


class TwoLevelIlu(object):
    def __init__(self, A, P, Ac, coarse, post=True, order=None, corder=None):
        self.A, self.P = A.tocsr(), P.tocsr()
        self.ilu = L.Ilu0(A, order)
        self.post = post
        Ac = Ac.tocsr()
        if coarse == 'lu':
            lu = spla.splu(Ac.tocsc())
            self.coarse = lu.solve
        elif coarse.startswith('cheb'):
            ch = L.Cheb(Ac, int(coarse[4:]), 12.0)
            self.coarse = lambda r: ch.run(r)
        elif coarse == 'ilu':
            ci = L.Ilu0(Ac, corder)
            self.coarse = ci.solve
        elif coarse == 'ilu2':
            ci = L.Ilu0(Ac, corder)

            def two(r):
                x = ci.solve(r)
                return x + ci.solve(r - Ac.dot(x))
            self.coarse = two
        elif coarse == 'none':
            self.coarse = None
        else:
            raise ValueError(coarse)

    def solve_additive(self, r):
        '''fine sweep + coarse correction of the SAME residual (the two could
        run side by side on two streams)'''
        return self.ilu.solve(r) + self.P.dot(self.coarse(self.P.T.dot(r)))

    def solve_coarse_first(self, r):
        x = self.P.dot(self.coarse(self.P.T.dot(r)))
        return x + self.ilu.solve(r - self.A.dot(x))

    def solve(self, r):
        x = self.ilu.solve(r)
        if self.coarse is not None:
            res = r - self.A.dot(x)
            x = x + self.P.dot(self.coarse(self.P.T.dot(res)))
        if self.post:
            x = x + self.ilu.solve(r - self.A.dot(x))
        return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=200)
    ap.add_argument('--like', type=int, default=680)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--cfl-scale', type=float, default=1.0)
    ap.add_argument('--rtol', type=float, default=1e-8)
    ap.add_argument('--only', default='')
    ap.add_argument('--graded', type=float, nargs=2, default=None,
                    metavar=('LCAR', 'LIKE_LCAR'))
    args = ap.parse_args()
    S = build_system(args.nx, args.like, args.steps, args.cfl_scale,
                     tuple(args.graded) if args.graded else None)
    J, F, info = S['J'], S['F'], S['info']
    n, n1 = info['N'], S['n1']
    print('system: N = %d per component, CFL %.2f, cell Peclet %.2f, dt %.4g, '
          '|F| %.3e' % (n, info['cfl'], info['peclet'], info['dt'],
                        numpy.linalg.norm(F)), flush=True)
    B = L.diag_blocks(J, n)
    R = L.diag_blocks(S['J1'], n1)
    lay = S['lay']
    isbc = numpy.zeros(2 * n, dtype=bool)
    isbc[S['bc']] = True
    vd = lay.vertex_dofs
    Pm = L.p2_to_p1_prolongation(lay, S['play'], S['mesh'])
    only = set(args.only.split(',')) if args.only else None

    def want(name):
        return only is None or name in only

    def blockwise(s0, s1):
        return lambda v: numpy.concatenate([s0(v[:n]), s1(v[n:])])

    t0 = time.time()
    order, ncol = L.greedy_colour_order(B[0])
    print('greedy colouring: %d colours (%.1f s)' % (ncol, time.time() - t0))
    precs = []
    if want('mc-ilu'):
        m = [L.Ilu0(b, order) for b in B]
        precs.append(('mc-ilu', blockwise(m[0].solve, m[1].solve)))
    if want('nat-ilu'):
        m = [L.Ilu0(b) for b in B]
        precs.append(('nat-ilu', blockwise(m[0].solve, m[1].solve)))
    for bsz in (4, 16, 64):
        name = 'chains-%d' % bsz
        if want(name):
            o, nc = block_colour_order(B[0], numpy.arange(n), bsz)
            m = [L.Ilu0(b, o) for b in B]
            precs.append(('%s (%d colours)' % (name, nc),
                          blockwise(m[0].solve, m[1].solve)))

    def levels():
        for a in (0, 1):
            bcmask = isbc[a * n:(a + 1) * n]
            free1 = ~bcmask[vd]
            Pb = sp.diags((~bcmask).astype(float)).dot(Pm).dot(
                sp.diags(free1.astype(float))).tocsr()
            f = sp.diags(free1.astype(float))
            Ac = (f.dot(R[a]).dot(f) + sp.diags((~free1).astype(float))).tocsr()
            yield B[a], Pb, Ac

    if want('pmg-cheb'):
        out = [L.TwoLevel(blk, Pb, Ac, 1, 2, 6, ratio_f=5.0, ratio_c=12.0)
               for blk, Pb, Ac in levels()]
        precs.append(('pmg-cheb', blockwise(out[0].solve, out[1].solve)))
    corder = None
    for post in (True, False):
        for coarse in ('lu', 'cheb6', 'ilu', 'ilu2', 'none'):
            name = 'tl-ilu%s/%s' % ('' if post else '-pre', coarse)
            if not want(name):
                continue
            if corder is None:
                corder, _ = L.greedy_colour_order(R[0])
            out = [TwoLevelIlu(blk, Pb, Ac, coarse, post, order, corder)
                   for blk, Pb, Ac in levels()]
            precs.append((name, blockwise(out[0].solve, out[1].solve)))

    for coarse in ('lu', 'ilu', 'ilu2'):
        for kind in ('add', 'cfirst'):
            name = 'tl-%s/%s' % (kind, coarse)
            if not want(name):
                continue
            if corder is None:
                corder, _ = L.greedy_colour_order(R[0])
            out = [TwoLevelIlu(blk, Pb, Ac, coarse, False, order, corder)
                   for blk, Pb, Ac in levels()]
            f = [o.solve_additive if kind == 'add' else o.solve_coarse_first
                 for o in out]
            precs.append((name, blockwise(f[0], f[1])))

    rng = numpy.random.RandomState(0)
    for rname, b in (('F', F), ('random', rng.standard_normal(2 * n) * (~isbc))):
        print('--- right-hand side: %s' % rname)
        for name, Mi in precs:
            t0 = time.time()
            try:
                x, its, hist = L.fgmres(J, b, Mi, args.rtol, maxit=300)
                true = numpy.linalg.norm(b - J.dot(x)) / numpy.linalg.norm(b)
            except Exception as e:          # (a cycle that blows up)
                its, true = -1, float('nan')
                print('  %s failed: %s' % (name, e))
            print('%-26s %4d applications (true rel. residual %.1e)  %.1f s' % (
                name, its, true, time.time() - t0), flush=True)




def heat_main(n=100, cfls=(6.0, 14.0, 40.0), diffusion=2.3):
    '''--heat: the same question for BASELINE config 4's heat system (the
    system and the plume of tools/heat_ilu_lab.py), P1 level = the heat
    operator rediscretised on P1 with the same P2 convection field.'''
    import heat_ilu_lab as Hl
    from flow_amd import fem, boussinesq
    from flow_amd.fem.space import scalar_layout
    mesh = fem.heater_box(n, fitted=True)
    pb = boussinesq.HeaterBox(mesh)
    Q, W = pb.Q, pb.W
    lay1 = scalar_layout(mesh, 1)
    Qo = orc.Space(mesh.points, mesh.cell_vertices, Q.layout.cell_dofs, 2, Q.N)
    Q1o = orc.Space(mesh.points, mesh.cell_vertices, lay1.cell_dofs, 1, lay1.N)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    h, dt = 0.1 / n, 1.0
    d_t, _ = collect(pb.temperature_bcs(12.0), Q.size())
    shape = Hl.plume(W.layout.dof_coords)
    kappa = diffusion * h**2 / dt * pb.rho_room * pb.cp
    vd = Q.layout.vertex_dofs
    Pm = L.p2_to_p1_prolongation(Q.layout, lay1, mesh)
    isbc = numpy.zeros(Q.N, dtype=bool)
    isbc[d_t] = True
    free1 = ~isbc[vd]
    Pb = sp.diags((~isbc).astype(float)).dot(Pm).dot(
        sp.diags(free1.astype(float))).tocsr()
    rng = numpy.random.RandomState(2)
    for cfl in cfls:
        conv = (cfl * h / dt * shape).reshape(-1)
        out = []
        for Qs in (Qo, Q1o):
            M, A, _b = orc.heat_operators(Qs, Wo, conv, kappa, pb.rho_room,
                                          pb.cp, 0.0, False)
            out.append((M - dt * A).tocsr())
        S, S1 = out
        keep = (~isbc).astype(float)
        S = (sp.diags(keep).dot(S) + sp.diags(1.0 - keep)).tocsr()
        # (row equilibration, as the product's solve)
        dscale = 1.0 / abs(S).sum(axis=1).A.ravel()
        S = sp.diags(dscale).dot(S).tocsr()
        S.sort_indices()
        f = sp.diags(free1.astype(float))
        Ac = (f.dot(S1).dot(f) + sp.diags((~free1).astype(float))).tocsr()
        # the cycle acts on the UNSCALED residual: M^-1 r~ = cycle(D r~)
        Su = sp.diags(1.0 / dscale).dot(S).tocsr()
        b = rng.standard_normal(Q.N) * keep
        order, _ = L.greedy_colour_order(S)
        corder, _ = L.greedy_colour_order(Ac)
        print('--- heat, cell CFL %.0f (cell Peclet number %.1f)' % (
            cfl, 0.5 * cfl / diffusion), flush=True)
        mc = L.Ilu0(S, order)
        rows = [('mc-ilu', mc.solve)]
        for name, coarse, post, how in (
                ('tl-ilu/ilu', 'ilu', True, 'solve'),
                ('tl-ilu/ilu2', 'ilu2', True, 'solve'),
                ('tl-ilu/lu', 'lu', True, 'solve'),
                ('tl-cfirst/ilu2', 'ilu2', False, 'solve_coarse_first'),
                ('tl-ilu-pre/ilu', 'ilu', False, 'solve')):
            T = TwoLevelIlu(Su, Pb, Ac, coarse, post, order, corder)
            fn = getattr(T, how)
            rows.append((name, lambda r, fn=fn: fn(r / dscale)))
        for name, Mi in rows:
            x, its, hist = L.fgmres(S, b, Mi, 1e-8, restart=30, maxit=400)
            true = numpy.linalg.norm(b - S.dot(x)) / numpy.linalg.norm(b)
            print('  %-16s %4d iterations (true residual %.1e)' % (
                name, its, true), flush=True)


if __name__ == '__main__' and '--heat' in sys.argv:
    heat_main()
elif __name__ == '__main__':
    main()
