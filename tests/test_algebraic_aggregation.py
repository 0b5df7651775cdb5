# -*- coding: utf-8 -*-
'''
Algebraic aggregation for the smoothed-aggregation pressure hierarchy
(`flow_aggregate_host`, flow_amd/fem/multigrid.py: aggregation='algebraic') --
aggregates from the MATRIX alone, as the reference's `hypre_amg`
(flow/navier_stokes/pressure_correction.py:331, 414-418) needs no coordinates.
CPU: the routine's invariants and the quality of a two-level SA correction
built from it (scipy).  GPU (-m gpu): CG + the V-cycle on such a hierarchy
against a direct solve and against the geometric hierarchy.
'''
import ctypes

import numpy
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from flow_amd import fem
from flow_amd.fem import multigrid

import cases


def _pressure_matrix(mesh):
    '''P1 stiffness matrix with the outlet column as Dirichlet rows (the
    oracle's assembly: no GPU needed).'''
    from oracle import fem_oracle as orc
    P = orc.Space(mesh.points, mesh.cell_vertices, mesh.cell_vertices, 1,
                  mesh.num_vertices())
    K = orc.stiffness_matrix(P).tocsr()
    isbc = mesh.points[:, 0] > mesh.points[:, 0].max() - 1e-12
    A, _ = orc.symmetric_bc(K, numpy.zeros(K.shape[0]),
                            numpy.nonzero(isbc)[0], numpy.zeros(isbc.sum()))
    return A.tocsr(), isbc


@pytest.mark.parametrize('mesh', ['structured', 'graded'])
def test_aggregates_from_the_matrix_alone(mesh):
    m = fem.karman_channel(60, 14, fitted=True) if mesh == 'structured' \
        else fem.karman_channel_graded(4.0e-3).reordered()
    A, isbc = _pressure_matrix(m)
    n = A.shape[0]
    agg, na = multigrid._algebraic_aggregates(A, ~isbc, 0.08)
    agg2, na2 = multigrid._algebraic_aggregates(A, ~isbc, 0.08)
    assert na == na2 and numpy.array_equal(agg, agg2)           # deterministic
    assert (agg[isbc] == -1).all() and (agg[~isbc] >= 0).all()
    assert agg.max() == na - 1 and len(numpy.unique(agg[~isbc])) == na
    sizes = numpy.bincount(agg[~isbc])
    # neighbourhood aggregates of a P1 triangulation: ~7 rows, none huge
    assert 4.0 < sizes.mean() < 10.0 and sizes.max() <= 16, (sizes.mean(),
                                                               sizes.max())
    # every aggregate is connected in the graph of A
    G = A.copy()
    G.data[:] = 1.0
    for a in list(range(0, na, max(1, na // 40))):
        rows = numpy.nonzero(agg == a)[0]
        sub = G[rows][:, rows]
        ncomp, _ = sp.csgraph.connected_components(sub, directed=False)
        assert ncomp == 1, (a, rows)
    # a two-level smoothed-aggregation correction built from them is a good
    # preconditioner: CG iterations a fraction of Jacobi's
    free = ~isbc
    idx = numpy.nonzero(free)[0]
    P0 = sp.csr_matrix((numpy.ones(len(idx)), (idx, agg[idx])), shape=(n, na))
    D = A.diagonal()
    v = numpy.random.RandomState(1).standard_normal(n)
    for _ in range(15):
        v = A.dot(v) / D
        lam = numpy.linalg.norm(v)
        v /= lam
    P = (P0 - sp.diags((4.0 / (3.0 * lam)) / D).dot(A.dot(P0))).tocsr()
    Ac = spla.splu((P.T.dot(A.dot(P))).tocsc())

    def two_level(r):
        x = 0.8 * r / D
        x = x + P.dot(Ac.solve(P.T.dot(r - A.dot(x))))
        return x + 0.8 * (r - A.dot(x)) / D
    b = numpy.random.RandomState(2).standard_normal(n)
    b[isbc] = 0.0
    counts = {}
    for name, M in (('jacobi', lambda r: r / D), ('sa', two_level)):
        its = [0]

        def cb(_x):
            its[0] += 1
        x, flag = spla.cg(A, b, rtol=1e-10, maxiter=5000, callback=cb,
                          M=spla.LinearOperator((n, n), M))
        assert flag == 0
        counts[name] = its[0]
    assert counts['sa'] <= 30 and 6 * counts['sa'] < counts['jacobi'], counts


def test_isolated_rows_and_dirichlet_rows():
    '''Rows without a strong coupling become singletons, rows marked not free
    stay out; theta = 0 aggregates every coupling.'''
    A = sp.csr_matrix(numpy.array([
        [2.0, -1.0, 0.0, 0.0, 0.0],
        [-1.0, 2.0, -1e-6, 0.0, 0.0],
        [0.0, -1e-6, 2.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 1.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 3.0]]))
    free = numpy.array([1, 1, 1, 0, 1], dtype=bool)
    agg, na = multigrid._algebraic_aggregates(A, free, 0.08)
    assert agg[3] == -1 and agg[0] == agg[1]
    assert len({agg[0], agg[2], agg[4]}) == 3 and na == 3
    agg0, na0 = multigrid._algebraic_aggregates(A, free, 0.0)
    assert agg0[0] == agg0[1] == agg0[2] and na0 == 2


@pytest.mark.gpu
@pytest.mark.parametrize('mesh', ['structured', 'graded'])
def test_pressure_cg_on_an_algebraic_hierarchy(hip, mesh):
    from flow_amd import device
    from flow_amd.fem import ops
    m = fem.karman_channel(240, 56, fitted=True) if mesh == 'structured' \
        else fem.karman_channel_graded(1.2e-3).reordered()
    V = fem.FunctionSpace(m, 'CG', 1)
    K = ops.assemble_stiffness(V)
    isbc = m.points[:, 0] > 0.6 - 1e-12
    A = ops.symmetric_bc_matrix(K, device.to_device(isbc.astype(numpy.uint8)))
    rng = numpy.random.RandomState(3)
    b = rng.standard_normal(V.N)
    b[isbc] = 0.0
    ref = spla.splu(A.to_scipy().tocsc()).solve(b)
    its = {}
    for kind in ('geometric', 'algebraic'):
        mg = multigrid.Multigrid(A, isbc, coarsest=300, aggregation=kind)
        assert mg.nlevels >= 3, (kind, mg.sizes)
        x = device.zeros(V.N)
        info = ops.krylov_solve('cg', A, device.to_device(b), x, rtol=1e-12,
                                maxit=500, mg=mg, check_every=1)
        assert cases.rel_l2(device.to_host(x).numpy(), ref) < 1e-7, (kind, info)
        its[kind] = info.iterations
        print(mesh, kind, mg.aggregation, mg.sizes, info.iterations)
    # the algebraic hierarchy is as good a preconditioner as the geometric one
    assert its['algebraic'] <= 40 and its['algebraic'] <= 1.6 * its['geometric'], its


@pytest.mark.gpu
def test_a_step_with_the_algebraic_hierarchy(hip):
    '''solver_parameters['pressure']['aggregation'] = 'algebraic': the same
    step to solver tolerance.'''
    import flow_amd.navier_stokes as navsto
    import large_cases
    case = large_cases.KarmanStepCase(100, 23)
    par = navsto.solver_parameters['pressure']
    u1, p1, ui = case.product_step()
    its_geo = navsto.last_step_info['pressure'].iterations
    par['aggregation'] = 'algebraic'
    par['mg_coarsest'] = 300
    try:
        u1a, p1a, uia = case.product_step()
        its_alg = navsto.last_step_info['pressure'].iterations
        assert 'mg' in navsto.last_step_info['pressure'].method
    finally:
        par['aggregation'] = 'geometric'
        par['mg_coarsest'] = 4200
    assert cases.rel_l2(u1a, u1) < 1e-8 and cases.rel_l2(p1a, p1) < 1e-8
    assert its_alg <= 2 * max(its_geo, 8), (its_alg, its_geo)
