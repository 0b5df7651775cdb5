# -*- coding: utf-8 -*-
'''
Development aid (CPU, scipy; not part of the product): BASELINE config 4's heat
system M + dt (convection + diffusion) of flow/heat.py at the cell CFL numbers
the plume reaches at dt = 1 -- where the product's p-multigrid cycle is
rejected and its multicolour ILU(0) needs 60-90 GMRES(30) iterations.  Which
ORDERING of an ILU(0) would do better?  The oracle's heat operators on the
body-fitted heater box (P2), convected by a synthetic plume (an upward jet
above the heater with a return flow along the walls), scaled to a given CFL;
right-preconditioned GMRES(30) to 1e-8 with ILU(0) in

  multicolour    greedy colouring, colour-major (what the product's sweeps are)
  natural        the generators' numbering (x-major)
  downwind       dofs sorted along the flow direction (y, the plume's axis)
  blocks-b       chains of b consecutive dofs of the downwind order (ACROSS
                 the flow) kept together, the chains coloured greedily
  vblocks-b      the same with chains of b consecutive dofs of the natural
                 numbering -- column by column: ALONG the plume -- (b rows per
                 lane, as many lanes as chains of one colour: what a GPU sweep
                 could run)

    python tools/heat_ilu_lab.py [--n 100] [--cfl 6 14 40]
'''
import argparse
import os
import sys
import time

import numpy
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
sys.path.insert(0, HERE)

from flow_amd import fem, boussinesq                        # noqa: E402
from flow_amd.fem.bcs import collect                        # noqa: E402
from oracle import fem_oracle as orc                        # noqa: E402
import precond_lab as L                                     # noqa: E402


def plume(x):
    '''(2, n): an upward jet above the heater, return flow near the side walls,
    zero on the walls; max |u| = 1.'''
    X, Y = x[:, 0], x[:, 1]
    wall = numpy.sin(numpy.pi * X / 0.1) * numpy.sin(numpy.pi * Y / 0.2)
    jet = numpy.exp(-((X - 0.05) / 0.012)**2) - 0.35
    uy = jet * wall * (Y > 0.07)
    ux = 0.3 * numpy.sin(2.0 * numpy.pi * X / 0.1) * numpy.cos(
        numpy.pi * Y / 0.2) * wall
    u = numpy.stack([ux, uy])
    return u / abs(u).max()


def block_colour_order(A, order, b):
    '''Chains of b consecutive entries of `order`; the chains coloured greedily
    by the graph of A contracted onto them; colour-major, chains intact.'''
    n = A.shape[0]
    chain_of = numpy.empty(n, dtype=numpy.int64)
    chain_of[order] = numpy.arange(n) // b
    nch = int(chain_of.max()) + 1
    A = A.tocoo()
    G = sp.csr_matrix((numpy.ones(A.nnz), (chain_of[A.row], chain_of[A.col])),
                      shape=(nch, nch))
    G.sum_duplicates()
    colour = numpy.full(nch, -1, dtype=numpy.int64)
    for c in range(nch):
        used = set(colour[G.indices[G.indptr[c]:G.indptr[c + 1]]])
        k = 0
        while k in used:
            k += 1
        colour[c] = k
    chain_rank = numpy.argsort(colour, kind='stable')
    pos = numpy.empty(nch, dtype=numpy.int64)
    pos[chain_rank] = numpy.arange(nch)
    key = pos[chain_of[order]] * b + (numpy.arange(n) % b)
    return order[numpy.argsort(key, kind='stable')], int(colour.max()) + 1


def gmres_right(A, b, Minv, rtol=1e-8, restart=30, maxit=400):
    '''Right-preconditioned GMRES(restart): precond_lab.fgmres.'''
    x, its, hist = L.fgmres(A, b, Minv, rtol, restart=restart, maxit=maxit)
    return its, numpy.linalg.norm(b - A.dot(x)) / numpy.linalg.norm(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=100)
    ap.add_argument('--cfl', type=float, nargs='*', default=[6.0, 14.0, 40.0])
    ap.add_argument('--diffusion', type=float, default=2.3,
                    help='thermal diffusion number alpha dt / h^2 of the 400-'
                         'cell box of config 4; the conductivity is scaled so '
                         'that the lab mesh has the same one')
    args = ap.parse_args()
    mesh = fem.heater_box(args.n, fitted=True)
    pb = boussinesq.HeaterBox(mesh)
    Q, W = pb.Q, pb.W
    Qo = orc.Space(mesh.points, mesh.cell_vertices, Q.layout.cell_dofs, 2, Q.N)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    h = 0.1 / args.n
    dt = 1.0
    d_t, v_t = collect(pb.temperature_bcs(12.0), Q.size())
    xq = Q.layout.dof_coords
    shape = plume(W.layout.dof_coords)
    rng = numpy.random.RandomState(2)
    kappa = args.diffusion * h**2 / dt * pb.rho_room * pb.cp
    print('heater box %d: %d dofs (scalar P2), h = %.2e, thermal diffusion '
          'number %.1f (conductivity x %.1f)' % (
              args.n, Q.N, h, args.diffusion, kappa / pb.kappa), flush=True)
    for cfl in args.cfl:
        umax = cfl * h / dt
        conv = (umax * shape).reshape(-1)
        M, A, _b = orc.heat_operators(Qo, Wo, conv, kappa, pb.rho_room, pb.cp,
                                      0.0, False)
        S = (M - dt * A).tocsr()
        keep = numpy.ones(Q.N)
        keep[d_t] = 0.0
        S = (sp.diags(keep).dot(S) + sp.diags(1.0 - keep)).tocsr()
        # row equilibration, as the product's solve
        S = sp.diags(1.0 / abs(S).sum(axis=1).A.ravel()).dot(S).tocsr()
        S.sort_indices()
        b = rng.standard_normal(Q.N) * keep
        down = numpy.argsort(xq[:, 1] + 1e-3 * xq[:, 0], kind='stable')
        orders = [('multicolour', L.greedy_colour_order(S)),
                  ('natural', (None, 0)),
                  ('downwind', (down, 0))]
        for bsz in (4, 16, 64):
            orders.append(('blocks-%d' % bsz, block_colour_order(S, down, bsz)))
        # ... and along the flow: the generators number the dofs column by
        # column (x fixed, y running), the plume's direction
        ident = numpy.arange(Q.N)
        for bsz in (4, 16, 64):
            orders.append(('vblocks-%d' % bsz, block_colour_order(S, ident, bsz)))
        print('--- cell CFL %.0f (max |u| = %.2e m/s; cell Peclet number %.1f)' % (
            cfl, umax, 0.5 * cfl / args.diffusion), flush=True)
        for name, (order, ncol) in orders:
            t0 = time.time()
            ilu = L.Ilu0(S, order)
            its, true = gmres_right(S, b, ilu.solve)
            print('  %-12s %4d iterations (true residual %.1e)%s  %.1f s' % (
                name, its, true,
                '  %d colours' % ncol if ncol else '', time.time() - t0),
                flush=True)


if __name__ == '__main__':
    main()
