# -*- coding: utf-8 -*-
'''
TEST / BENCH INFRASTRUCTURE (never imported by flow_amd/): one pressure-
correction step on the HOST CORES with iterative solvers -- the CPU figure
bench.py's `cpu_baseline` reports beside the GPU's, MEASURED at size instead of
extrapolated from the sparse-LU oracle (oracle/fem_oracle.py, whose LU does not
finish at a million DoF in a bench's budget).  Same discretisation and the same
three sub-steps as `fem_oracle.step` (reference: flow/navier_stokes/
pressure_correction.py:468-518), assembled by the oracle's own numpy routines;
what differs is how the linear systems are solved:

  assembly             the oracle's numpy routines; the momentum residual and
                       Jacobian over chunks of the cells on `assembly_threads`
                       threads;
  tentative velocity   Newton as in the oracle; every system by GMRES(30),
                       right-preconditioned with SuperLU's incomplete LU of
                       the Newton matrix (scipy `spilu`), to 1e-6 of the Newton
                       tolerance -- one core (scipy / SuperLU are not threaded);
  pressure             CG + the smoothed-aggregation V(1,1) cycle of the GPU
                       path on the hierarchy handed in (oracle/cpu_cg.c, OpenMP,
                       all cores), or Jacobi-CG (cpu_cg.c) without a hierarchy;
  velocity correction  Jacobi-CG on the Dirichlet-eliminated vector mass matrix
                       (cpu_cg.c, OpenMP, all cores).

The fields agree with the sparse-LU oracle's to solver tolerance
(tests/test_cpu_step.py).  A restatement for timing: parity is anchored on
fem_oracle.step, not on this.
'''
import time

import numpy
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import fem_oracle as orc
from . import cpu_lib


def momentum_rhs_threaded(W, P, U, p0, f, rho, mu, threads):
    '''fem_oracle.momentum_rhs over chunks of the cells on `threads` threads
    (numpy's einsum loops release the GIL): the same residual vector and
    Jacobian, summed over the chunks.'''
    from concurrent.futures import ThreadPoolExecutor
    nc = len(W.cells)
    threads = max(1, min(int(threads), nc // 64))
    if threads == 1:
        return orc.momentum_rhs(W, P, U, p0, f, rho, mu)
    bc_cells, bc_lf = orc.boundary_facets(W)
    chunks = numpy.array_split(numpy.arange(nc), threads)

    def work(ch):
        Ws = orc.Space(W.points, W.cells[ch], W.cell_dofs[ch], W.deg, W.N)
        Ps = orc.Space(P.points, P.cells[ch], P.cell_dofs[ch], 1, P.N)
        sel = (bc_cells >= ch[0]) & (bc_cells <= ch[-1])
        Ws.bfacets = (bc_cells[sel] - ch[0], bc_lf[sel])
        return orc.momentum_rhs(Ws, Ps, U, p0, (f[0], f[1][ch]), rho, mu)
    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(work, chunks))
    R = parts[0][0].copy()
    for r, _j in parts[1:]:
        R += r
    J = parts[0][1]
    for _r, j in parts[1:]:
        J = J + j
    return R, J


def step(W, P, u0, p0, f0, f1, u_bc, p_bc, rho, mu, dt, lib, hierarchy=None,
         tol=1.0e-10, method='backward euler', fill_factor=4.0,
         assembly_threads=1):
    '''One Rotational step.  lib: the loaded oracle/liboracle_cpu.so;
    hierarchy: a cpu_lib.MgHierarchy of the Dirichlet-eliminated pressure
    matrix (None: Jacobi-CG).  -> (u1, p1, ui, info); info['seconds'] has the
    wall time per phase (assembly, ilu, gmres, pressure, correction).'''
    th_i, th_e = orc._THETA[method]
    assert th_e == 0.0, 'backward Euler only (what the drivers run)'
    sec = dict(assembly=0.0, ilu=0.0, gmres=0.0, pressure=0.0, correction=0.0)
    bc_dofs, bc_vals = u_bc
    t0 = time.perf_counter()
    M = sp.block_diag([orc.mass_matrix(W)] * 2, format='csr')
    sec['assembly'] += time.perf_counter() - t0
    ui = u0.copy()
    history, gmres_its = [], []
    for it in range(11):
        t0 = time.perf_counter()
        Ri, dRi = momentum_rhs_threaded(W, P, ui, p0, f1, rho, mu,
                                        assembly_threads)
        F = M.dot(ui - u0) - dt / rho * Ri
        F[bc_dofs] = ui[bc_dofs] - bc_vals
        nrm = numpy.linalg.norm(F)
        history.append(nrm)
        if nrm < tol or it == 10:
            sec['assembly'] += time.perf_counter() - t0
            break
        J = orc._identity_rows(M - dt / rho * dRi, bc_dofs).tocsc()
        sec['assembly'] += time.perf_counter() - t0
        t0 = time.perf_counter()
        ilu = spla.spilu(J, fill_factor=fill_factor)
        sec['ilu'] += time.perf_counter() - t0
        t0 = time.perf_counter()
        count = [0]

        def cb(_):
            count[0] += 1
        dx, flag = spla.gmres(
            J, F, rtol=max(1.0e-13, 1.0e-6 * tol / nrm), atol=0.0, restart=30,
            maxiter=40, M=spla.LinearOperator(J.shape, ilu.solve), callback=cb,
            callback_type='pr_norm')
        sec['gmres'] += time.perf_counter() - t0
        if flag != 0:
            raise RuntimeError('cpu_step: GMRES did not converge (%d)' % flag)
        gmres_its.append(count[0])
        ui = ui - dx
    if not history[-1] < tol:
        raise RuntimeError('cpu_step: Newton did not converge: %r' % history)
    # pressure (Dirichlet branch: symmetric elimination; the matrix of the
    # hierarchy is that operator)
    t0 = time.perf_counter()
    b = orc.pressure_rhs(W, P, ui, p0, 1.0, rho, mu, dt, True)
    A, b = orc.symmetric_bc(orc.stiffness_matrix(P), b, p_bc[0], p_bc[1])
    sec['assembly'] += time.perf_counter() - t0
    t0 = time.perf_counter()
    if hierarchy is not None:
        p1, pits, _res, ok = hierarchy.cg(b, tol, maxit=1000, x0=p0)
    else:
        p1, pits, _res, ok = cpu_lib.jacobi_cg(lib, A, b, tol, maxit=100000,
                                               x0=p0)
    sec['pressure'] += time.perf_counter() - t0
    if not ok:
        raise RuntimeError('cpu_step: pressure CG did not converge')
    # velocity correction: the right-hand side of fem_oracle.
    # velocity_correction, the solve by Jacobi-CG
    t0 = time.perf_counter()
    rhs, Am = _correction_system(W, P, ui, p1, p0, bc_dofs, bc_vals, rho, mu,
                                 dt, M)
    sec['assembly'] += time.perf_counter() - t0
    t0 = time.perf_counter()
    u1, cits, _res, ok = cpu_lib.jacobi_cg(lib, Am, rhs, tol, maxit=1000, x0=ui)
    sec['correction'] += time.perf_counter() - t0
    if not ok:
        raise RuntimeError('cpu_step: mass CG did not converge')
    info = dict(seconds=sec, newton_history=history, gmres_iterations=gmres_its,
                pressure_iterations=pits, correction_iterations=cits,
                pressure_solver='mg-cg' if hierarchy is not None else 'jacobi-cg')
    return u1, p1, ui, info


def _correction_system(W, P, ui, p1, p0, bc_dofs, bc_vals, rho, mu, dt, M):
    pts, w = orc.duffy_rule(4)
    phi, _ = orc.basis(W.deg, pts)
    _psi, gpsi_ref = orc.basis(1, pts)
    gpsi = P.phys_grad(gpsi_ref)
    wd = w[None, :] * numpy.abs(W.detJ)[:, None]
    phi_field = (p1 - p0)[P.cell_dofs] + mu * orc.divergence_P2_to_vertices(W, ui)
    gphi_field = numpy.einsum('ci,cqid->cqd', phi_field, gpsi)
    Fe = -dt / rho * numpy.einsum('cq,cqa,qi->cai', wd, gphi_field, phi)
    b = M.dot(ui) + orc._vec(W, Fe, 2)
    A, b = orc.symmetric_bc(M, b, bc_dofs, bc_vals)
    return b, A
