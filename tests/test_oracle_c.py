# -*- coding: utf-8 -*-
'''Pins the oracle's C restatement (oracle/cpu_cg.c: CSR MatMult + Jacobi-PCG)
against scipy and the numpy oracle's matrices.  CPU only.'''
import numpy
import scipy.sparse.linalg as spla

from flow_amd import fem
from oracle import cpu_lib, fem_oracle as orc

import oracle_harness as H


def test_c_spmv_and_cg_match_scipy():
    lib = cpu_lib.load()
    assert lib.oracle_num_threads() >= 1
    mesh = fem.karman_channel(40, 10)
    P = H.oracle_space(mesh, 1)
    A = (orc.stiffness_matrix(P) + 3.0 * orc.mass_matrix(P)).tocsr()
    A.sort_indices()
    rng = numpy.random.RandomState(0)
    x = rng.standard_normal(P.N)
    y = numpy.empty(P.N)
    lib.oracle_spmv_csr(P.N, A.indptr.astype(numpy.int32),
                        A.indices.astype(numpy.int32), A.data, x, y)
    assert numpy.allclose(y, A.dot(x), rtol=1e-14, atol=1e-14)
    b = rng.standard_normal(P.N)
    sol, its, res, ok = cpu_lib.jacobi_cg(lib, A, b, 1e-12, maxit=5000)
    assert ok and its > 0
    ref = spla.splu(A.tocsc()).solve(b)
    assert numpy.linalg.norm(sol - ref) < 1e-9 * numpy.linalg.norm(ref)
    _, its2, _, ok2 = cpu_lib.jacobi_cg(lib, A, b, 1e-12, maxit=3)
    assert not ok2 and its2 == 3


def test_c_multigrid_cg_matches_direct_solve():
    '''The like-for-like CPU baseline (V-cycle-preconditioned CG on a
    smoothed-aggregation hierarchy handed in as CSR) converges to the direct
    solution, in far fewer iterations than Jacobi-CG.'''
    import scipy.sparse as sp
    lib = cpu_lib.load()
    mesh = fem.karman_channel(60, 15)
    P = H.oracle_space(mesh, 1)
    A = (orc.stiffness_matrix(P) + 1e-3 * orc.mass_matrix(P)).tocsr()
    n = A.shape[0]
    # one level of smoothed aggregation over 3x3 vertex patches
    x = mesh.points
    h = 0.6 / 60
    ix = numpy.floor(x[:, 0] / (3 * h) + 1e-9).astype(int)
    iy = numpy.floor((x[:, 1] + 0.07) / (3 * h) + 1e-9).astype(int)
    _, agg = numpy.unique(ix * 1000 + iy, return_inverse=True)
    P0 = sp.csr_matrix((numpy.ones(n), (numpy.arange(n), agg)))
    D = A.diagonal()
    Pm = (P0 - sp.diags(0.6 / D).dot(A.dot(P0))).tocsr()
    Ac = Pm.T.dot(A.dot(Pm)).toarray()
    hier = cpu_lib.MgHierarchy(lib, [(A, D, Pm)], numpy.linalg.inv(Ac), 0.8)
    rng = numpy.random.RandomState(1)
    b = rng.standard_normal(n)
    sol, its, res, ok = hier.cg(b, 1e-12, maxit=200)
    ref = spla.splu(A.tocsc()).solve(b)
    assert ok and numpy.linalg.norm(sol - ref) < 1e-9 * numpy.linalg.norm(ref)
    _, its_j, _, _ = cpu_lib.jacobi_cg(lib, A, b, 1e-12, maxit=5000)
    assert its < its_j / 3, (its, its_j)
