# -*- coding: utf-8 -*-
'''
Development aid (CPU, scipy; not part of the product): what do the OFF-DIAGONAL
blocks of the Newton Jacobian cost the preconditioner in a developed vortex
street?  The two-level p-multigrid cycle of the product treats the two velocity
components separately (block Jacobi between them: the blocks
-/+ dt (du_a/dx_b) M are dropped).  On the early plateau they are small; in the
street dt |grad u| is 0.15-0.3.

The oracle's Jacobian on a Karman channel in the non-dimensional regime of the
headline workload (tools/precond_lab.py), linearised at a SYNTHETIC street --
channel profile plus a row of alternating Gaussian vortices behind the
cylinder, scaled so that dt |grad u| matches the street's --, and flexible
GMRES(10) iterations to 1e-8 with

  bj-exact     exact solves of the two diagonal blocks (block Jacobi)
  gs-exact     block Gauss-Seidel between the components, exact block solves
  bj-cycle     the product's cycle (1 + 2 Chebyshev steps on P2, 6 on P1)
  gs-cycle     the same cycle per component, Gauss-Seidel between them
  full-cycle   the cycle on the COUPLED operator (2 x 2 blocks on both levels)

    python tools/coupling_lab.py --nx 300 [--amp 0.008]
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


def street(x, amp, radius=0.012):
    '''(2, n): channel profile, zero near the cylinder, plus alternating
    Gaussian vortices (stream function amp * radius * exp(-r^2 / radius^2)).'''
    X, Y = x[:, 0], x[:, 1]
    prof = karman.ENTRANCE_VELOCITY * (karman.Y1 - Y) * (Y - karman.Y0) / 0.07**2
    r = numpy.sqrt((X - 0.1)**2 + (Y - 0.01)**2)
    mask = 1.0 - numpy.exp(-(numpy.maximum(r - 0.02, 0.0) / 0.012)**2)
    ux, uy = prof * mask, numpy.zeros_like(X)
    wall = (karman.Y1 - Y) * (Y - karman.Y0) / 0.07**2
    for k in range(9):
        cx, cy = 0.17 + 0.045 * k, 0.01 + (0.014 if k % 2 else -0.014)
        sgn = 1.0 if k % 2 else -1.0
        g = sgn * amp * radius * numpy.exp(
            -((X - cx)**2 + (Y - cy)**2) / radius**2) * numpy.e**0.5 / 2**0.5
        # u = (d psi / dy, -d psi / dx)
        ux += -2.0 * (Y - cy) / radius**2 * g * wall * mask
        uy += 2.0 * (X - cx) / radius**2 * g * wall * mask
    return numpy.concatenate([ux, uy])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=300)
    ap.add_argument('--amp', type=float, default=0.008)
    ap.add_argument('--rtol', type=float, default=1e-8)
    args = ap.parse_args()
    prob = karman.KarmanProblem.__new__(karman.KarmanProblem)
    from flow_amd import fem
    mesh = fem.karman_channel(args.nx, None, fitted=True)
    W = H.oracle_space(mesh, 2)
    P = H.oracle_space(mesh, 1)
    W1 = H.oracle_space(mesh, 1)
    Wv = fem.VectorFunctionSpace(mesh, 'Lagrange', 2)
    Pv = fem.FunctionSpace(mesh, 'Lagrange', 1)
    lay, play = Wv.layout, Pv.layout
    n = W.N
    h_ratio = 2182.0 / args.nx
    rho, mu = karman.RHO_WATER_293K, 0.002 * h_ratio
    dt = mesh.hmax() / 0.0159
    inflow = fem.Expression('%e * (%e - x[1]) * (x[1] - %e) / %e' % (
        karman.ENTRANCE_VELOCITY, karman.Y1, karman.Y0, 0.07**2), degree=2)
    u_bcs = [
        fem.DirichletBC(Wv, (0.0, 0.0), karman.UpperBoundary()),
        fem.DirichletBC(Wv, (0.0, 0.0), karman.LowerBoundary()),
        fem.DirichletBC(Wv, (0.0, 0.0), karman.ObstacleBoundary()),
        fem.DirichletBC(Wv.sub(0), inflow, karman.LeftBoundary()),
        fem.DirichletBC(Wv.sub(0), inflow, karman.RightBoundary())]
    bc, bcv = collect(u_bcs, Wv.size())
    u0 = street(lay.dof_coords, args.amp)
    u0[bc] = bcv
    p0 = numpy.zeros(P.N)
    zero = (reference.lattice(0), numpy.zeros((mesh.num_cells(), 1, 2)))
    t0 = time.time()
    M1 = orc.mass_matrix(W)
    M = sp.block_diag([M1] * 2, format='csr')
    Ri, dRi = orc.momentum_rhs(W, P, u0, p0, zero, rho, mu)
    F = -dt / rho * Ri
    J = (M - dt / rho * dRi).tocsr()
    F[bc] = 0.0
    keep = numpy.ones(2 * n)
    keep[bc] = 0.0
    J = (sp.diags(keep).dot(J) + sp.diags(1.0 - keep)).tocsr()
    J.sort_indices()
    B0, B1 = L.diag_blocks(J, n)
    J01, J10 = J[:n, n:].tocsr(), J[n:, :n].tocsr()
    # dt |grad u|: the size of the dropped blocks relative to the mass matrix
    ratio = abs(J01).sum() / abs(M1).sum()
    gmax = 0.0
    xy = lay.dof_coords
    print('N = %d per component, dt %.3f, CFL %.2f, diffusion number %.2f, '
          '|J01|_1 / |M|_1 = %.3f  (%.1f s)' % (
              n, dt, 0.0159 * dt / (0.6 / args.nx),
              mu / rho * dt / (0.6 / args.nx)**2, ratio, time.time() - t0),
          flush=True)
    isbc = numpy.zeros(2 * n, dtype=bool)
    isbc[bc] = True
    vd = lay.vertex_dofs
    Pm = L.p2_to_p1_prolongation(lay, play, mesh)

    def prolongation(bcmask):
        free1 = ~bcmask[vd]
        return sp.diags((~bcmask).astype(float)).dot(Pm).dot(
            sp.diags(free1.astype(float))).tocsr(), free1

    P0, f0 = prolongation(isbc[:n])
    P1, f1 = prolongation(isbc[n:])

    def coarse_of(blk, Pb, free1):
        Ac = Pb.T.dot(blk.dot(Pb)).tocsr()
        return (Ac + sp.diags((~free1).astype(float))).tocsr()

    lu0, lu1 = spla.splu(B0.tocsc()), spla.splu(B1.tocsc())
    cyc0 = L.TwoLevel(B0, P0, coarse_of(B0, P0, f0), 1, 2, 6, ratio_f=5.0,
                      ratio_c=12.0)
    cyc1 = L.TwoLevel(B1, P1, coarse_of(B1, P1, f1), 1, 2, 6, ratio_f=5.0,
                      ratio_c=12.0)
    # the coupled operator: P = diag(P0, P1), Galerkin coarse operator
    Pf = sp.block_diag([P0, P1], format='csr')
    ff = numpy.concatenate([f0, f1])
    full = L.TwoLevel(J, Pf, coarse_of(J, Pf, ff), 1, 2, 6, ratio_f=5.0,
                      ratio_c=12.0)
    # ... and with the coupling on the coarse level only
    Jbd = sp.block_diag([B0, B1], format='csr')
    half = L.TwoLevel(Jbd, Pf, coarse_of(J, Pf, ff), 1, 2, 6, ratio_f=5.0,
                      ratio_c=12.0)

    def bj(s0, s1):
        return lambda v: numpy.concatenate([s0(v[:n]), s1(v[n:])])

    def gs(s0, s1):
        def apply(v):
            x0 = s0(v[:n])
            x1 = s1(v[n:] - J10.dot(x0))
            return numpy.concatenate([x0, x1])
        return apply

    def sgs(s0, s1):
        def apply(v):
            x0 = s0(v[:n])
            x1 = s1(v[n:] - J10.dot(x0))
            x0 = s0(v[:n] - J01.dot(x1))
            return numpy.concatenate([x0, x1])
        return apply

    precs = [
        ('bj-exact', bj(lu0.solve, lu1.solve)),
        ('gs-exact', gs(lu0.solve, lu1.solve)),
        ('sgs-exact', sgs(lu0.solve, lu1.solve)),
        ('bj-cycle', bj(cyc0.solve, cyc1.solve)),
        ('gs-cycle', gs(cyc0.solve, cyc1.solve)),
        ('coarse-coupled', half.solve),
        ('full-cycle', full.solve),
        ]
    rng = numpy.random.RandomState(0)
    for rname, b in (('F', F), ('random', rng.standard_normal(2 * n) * (~isbc))):
        print('--- right-hand side: %s' % rname)
        for name, Mi in precs:
            t0 = time.time()
            x, its, hist = L.fgmres(J, b, Mi, args.rtol)
            true = numpy.linalg.norm(b - J.dot(x)) / numpy.linalg.norm(b)
            print('%-15s %4d iterations  (true rel. residual %.1e)  %.1f s' % (
                name, its, true, time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
