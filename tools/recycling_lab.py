# -*- coding: utf-8 -*-
'''
Development aid (CPU, scipy; not part of the product): would the SECOND Newton
system of a step in the vortex street profit from the Krylov space of the
first (lever i)?

Setting of tools/coupling_lab.py: the oracle's momentum residual and Jacobian
on a Karman channel in the non-dimensional regime of the headline workload,
at a synthetic street u0.  One Newton step as the product takes it:

    J0 d1 = -F(u0)         flexible GMRES(10) + the block-wise two-level cycle
                           to 1e-8, keeping Z (the preconditioned directions)
                           and C = J0 Z
    u1 = u0 + d1
    J1 d2 = -F(u1)         (a) from zero; (b) from the start Z y,
                           y = argmin |F(u1) + C y| -- the first solve's space
                           offered to the second system for k dot products
                           (J1 Z ~ J0 Z = C: the Jacobians differ by O(d1));
                           (c) the same with the exact J1 Z (k applications)

Printed: how much of |F(u1)| the space removes, and the GMRES counts.

    python tools/recycling_lab.py --nx 300 [--amp 0.008]
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

from flow_amd import karman, fem                            # noqa: E402
from flow_amd.fem.bcs import collect                        # noqa: E402
from flow_amd.fem import reference                          # noqa: E402
from oracle import fem_oracle as orc                        # noqa: E402
import oracle_harness as H                                  # noqa: E402
import precond_lab as L                                     # noqa: E402
import coupling_lab as CL                                   # noqa: E402


def fgmres_keep(A, b, M, rtol, x0=None, restart=10, maxit=200):
    '''precond_lab.fgmres with a start vector; also returns the Z vectors of
    every cycle.'''
    n = len(b)
    x = numpy.zeros(n) if x0 is None else x0.copy()
    bn = numpy.linalg.norm(b)
    its = 0
    Zall = []
    first = None
    while its < maxit:
        r = b - A.dot(x)
        beta = numpy.linalg.norm(r)
        if first is None:
            first = beta / bn
        if beta <= rtol * bn:
            break
        V = numpy.zeros((restart + 1, n))
        Z = numpy.zeros((restart, n))
        Hm = numpy.zeros((restart + 1, restart))
        V[0] = r / beta
        g = numpy.zeros(restart + 1)
        g[0] = beta
        k = 0
        for j in range(restart):
            Z[j] = M(V[j])
            w = A.dot(Z[j])
            for i in range(j + 1):
                Hm[i, j] = w.dot(V[i])
                w -= Hm[i, j] * V[i]
            Hm[j + 1, j] = numpy.linalg.norm(w)
            V[j + 1] = w / Hm[j + 1, j]
            its += 1
            k = j + 1
            y = numpy.linalg.lstsq(Hm[:j + 2, :j + 1], g[:j + 2], rcond=None)[0]
            rn = numpy.linalg.norm(g[:j + 2] - Hm[:j + 2, :j + 1].dot(y))
            if rn <= rtol * bn or its >= maxit:
                break
        x = x + Z[:k].T.dot(y)
        Zall.extend(Z[:k])
        if rn <= rtol * bn:
            break
    return x, its, numpy.array(Zall), first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=300)
    ap.add_argument('--amp', type=float, default=0.008)
    args = ap.parse_args()
    mesh = fem.karman_channel(args.nx, None, fitted=True)
    W = H.oracle_space(mesh, 2)
    P = H.oracle_space(mesh, 1)
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
    u0 = CL.street(lay.dof_coords, args.amp)
    u0[bc] = bcv
    p0 = numpy.zeros(P.N)
    zero = (reference.lattice(0), numpy.zeros((mesh.num_cells(), 1, 2)))
    M1 = orc.mass_matrix(W)
    M = sp.block_diag([M1] * 2, format='csr')
    keep = numpy.ones(2 * n)
    keep[bc] = 0.0
    isbc = numpy.zeros(2 * n, dtype=bool)
    isbc[bc] = True

    def system(ui):
        Ri, dRi = orc.momentum_rhs(W, P, ui, p0, zero, rho, mu)
        F = M.dot(ui - u0) - dt / rho * Ri
        J = (M - dt / rho * dRi).tocsr()
        F[bc] = 0.0
        J = (sp.diags(keep).dot(J) + sp.diags(1.0 - keep)).tocsr()
        J.sort_indices()
        return F, J

    vd = lay.vertex_dofs
    Pm = L.p2_to_p1_prolongation(lay, play, mesh)

    def cycle_for(J):
        B0, B1 = L.diag_blocks(J, n)
        out = []
        for blk, mask in ((B0, isbc[:n]), (B1, isbc[n:])):
            free1 = ~mask[vd]
            Pb = sp.diags((~mask).astype(float)).dot(Pm).dot(
                sp.diags(free1.astype(float))).tocsr()
            Ac = (Pb.T.dot(blk.dot(Pb)) + sp.diags((~free1).astype(float))).tocsr()
            out.append(L.TwoLevel(blk, Pb, Ac, 1, 2, 6, ratio_f=5.0, ratio_c=12.0))
        return lambda v: numpy.concatenate([out[0].solve(v[:n]),
                                            out[1].solve(v[n:])])

    t0 = time.time()
    F0, J0 = system(u0)
    prec0 = cycle_for(J0)
    d1, its1, Z, _ = fgmres_keep(J0, -F0, prec0, 1e-8)
    u1 = u0 + d1
    F1, J1 = system(u1)
    print('N = %d per component, dt %.3f; |F(u0)| = %.3e, first solve: %d '
          'applications; |F(u1)| = %.3e (= %.1e |F(u0)|)  (%.0f s)' % (
              n, dt, numpy.linalg.norm(F0), its1, numpy.linalg.norm(F1),
              numpy.linalg.norm(F1) / numpy.linalg.norm(F0), time.time() - t0),
          flush=True)
    # the product's second solve: relative target 1e-6 tol / |F1| with tol such
    # that |F1| is a few times tol (the street: 2e-10 against 1e-10)
    rtol2 = 1e-6 / 2.0
    prec1 = prec0          # (the product's preconditioner is lagged)
    C0 = numpy.array([J0.dot(z) for z in Z])
    C1 = numpy.array([J1.dot(z) for z in Z])
    print('the first solve kept %d directions; |J1 Z - J0 Z| / |J0 Z| = %.1e'
          % (len(Z), numpy.linalg.norm(C1 - C0) / numpy.linalg.norm(C0)))
    for name, C in (('(a) zero start', None), ('(b) y from J0 Z', C0),
                    ('(c) y from J1 Z', C1)):
        if C is None:
            x0 = None
        else:
            y = numpy.linalg.lstsq(C.T, -F1, rcond=None)[0]
            x0 = Z.T.dot(y)
        x, its, _, first = fgmres_keep(J1, -F1, prec1, rtol2, x0=x0)
        print('%-18s residual of the start %.3e |F(u1)|, %2d applications to '
              '%.0e' % (name, first, its, rtol2), flush=True)


if __name__ == '__main__':
    main()
