# -*- coding: utf-8 -*-
'''
oracle/cpu_step.py -- the whole step on the host cores with iterative solvers
(bench.py's measured CPU figure) -- against the sparse-LU oracle it restates
(oracle/fem_oracle.step; reference flow/navier_stokes/pressure_correction.py:
468-518): same fields to solver tolerance.  CPU only.
'''
import numpy

import cases
import large_cases


def test_iterative_cpu_step_matches_the_sparse_lu_oracle():
    from oracle import cpu_lib, cpu_step
    lib = cpu_lib.load()
    lib.oracle_set_threads(2)
    c = large_cases.KarmanStepCase(60, 14)
    W, P = c.oracle_spaces()
    u_bc, p_bc = c.bc_data()
    u1, p1, ui, info = cpu_step.step(
        W, P, c.u0, c.p0, c.lattice(c.f0), c.lattice(c.f1), u_bc, p_bc, c.rho,
        c.mu, c.dt, lib, tol=1e-10, assembly_threads=3)
    # (the chunked assembly is the oracle's: same residual and Jacobian)
    from oracle import fem_oracle as orc
    R1, J1 = orc.momentum_rhs(W, P, c.u0, c.p0, c.lattice(c.f1), c.rho, c.mu)
    R3, J3 = cpu_step.momentum_rhs_threaded(W, P, c.u0, c.p0, c.lattice(c.f1),
                                            c.rho, c.mu, 3)
    assert abs(R1 - R3).max() <= 1e-13 * abs(R1).max()
    assert abs(J1 - J3).max() <= 1e-13 * abs(J1).max()
    u1o, p1o, uio = c.oracle_step()
    assert cases.rel_l2(ui, uio) < 1e-9
    assert cases.rel_l2(p1, p1o) < 1e-8
    assert cases.rel_l2(u1, u1o) < 1e-8
    assert info['newton_history'][-1] < 1e-10
    assert set(info['seconds']) == {'assembly', 'ilu', 'gmres', 'pressure',
                                    'correction'}
    assert all(numpy.isfinite(v) and v >= 0.0 for v in info['seconds'].values())


def test_blockwise_newton_solve_equals_the_sparse_lu():
    '''fem_oracle.solve_blockwise (what the oracle's offline goldens at 4.9 M
    and 9.87 M DoF were computed with: SuperLU cannot factor the coupled
    Newton matrix there) reaches the solution of the single sparse LU -- with
    the block LUs in fp64 ('block') and in fp32 ('block32': they only
    precondition, the true residual is driven to 1e-14 in fp64).'''
    c = large_cases.KarmanStepCase(60, 14)
    ref = c.oracle_step('crank-nicolson')
    for linear in ('block', 'block32'):
        got = c.oracle_step('crank-nicolson', linear=linear)
        for a, b in zip(got, ref):
            assert cases.rel_l2(a, b) < 1e-12, linear
