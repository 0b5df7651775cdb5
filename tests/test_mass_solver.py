# -*- coding: utf-8 -*-
'''
The mass-matrix solver (flow_mass, flow_amd/fem/mass.py): mixed-precision
defect correction with a fixed Chebyshev polynomial of D^-1 M.

CPU: what the design rests on, checked on the ORACLE's mass matrices -- the a
priori spectral bounds (Wathen) on every mesh family the suite uses, incl. the
body-fitted ones, and the algorithm itself (numpy restatement with the entries
rounded to fp16 and fp32 vectors): observed contraction below the bound the
solver's stopping test uses, stopping rule => |B r| <= rtol |x|.
GPU (-m gpu): the kernels against sparse direct solves, scalar and
two-component with identity rows, P1 and P2.
'''
import numpy
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from flow_amd import fem
from flow_amd.fem import mass as fmass
from oracle import fem_oracle as orc

import cases


def _meshes():
    return [
        ('unit-crossed-6', fem.UnitSquareMesh(6, 6, 'crossed')),
        ('rect-leftright', fem.RectangleMesh(
            fem.Point(-1.0, -0.5), fem.Point(1.5, 1.0), 7, 5, 'left/right')),
        ('karman-30-fitted', fem.karman_channel(30, 10, fitted=True)),
        ('heater-12-fitted', fem.heater_box(12, fitted=True)),
        ]


def _oracle_mass(mesh, deg):
    V = fem.FunctionSpace(mesh, 'CG', deg)
    S = orc.Space(mesh.points, mesh.cell_vertices, V.layout.cell_dofs, deg, V.N)
    return V, orc.mass_matrix(S).tocsr()


@pytest.mark.parametrize('deg', [1, 2])
def test_wathen_bounds_hold_on_the_oracles_mass_matrices(deg):
    lo, hi = fmass.WATHEN[deg]
    for name, mesh in _meshes():
        _V, M = _oracle_mass(mesh, deg)
        d = M.diagonal()
        # D^-1 M is similar to the symmetric D^-1/2 M D^-1/2
        S = sp.diags(d**-0.5).dot(M).dot(sp.diags(d**-0.5)).toarray()
        ev = numpy.linalg.eigvalsh(S)
        assert ev[0] >= lo * (1.0 - 1e-3), (name, ev[0])
        assert ev[-1] <= hi * (1.0 + 1e-3), (name, ev[-1])


def _defect_correction_numpy(M, b, x, rtol, steps, lo, hi, contraction,
                             mask=None, maxit=50):
    '''Restatement of flow_mass_solve (mass_kernels.hip) for ONE component:
    fp64 residual, the polynomial on the fp16-rounded D^-1 M with fp32 vectors,
    the stopping rule.  mask: 0 = identity row.  Returns (x, corrections,
    [|z_k|]).'''
    d = M.diagonal()
    A16 = sp.diags(1.0 / d).dot(M).tocsr()
    A16.data = A16.data.astype(numpy.float16).astype(numpy.float32)
    if mask is not None:
        keep = sp.diags(mask.astype(numpy.float32))
        A16 = (keep.dot(A16) + sp.diags(1.0 - mask.astype(numpy.float32))).tocsr()
    A16 = A16.astype(numpy.float32)
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    znorms = []
    for it in range(maxit):
        Mx = M.dot(x)
        if mask is not None:
            Mx = numpy.where(mask != 0, Mx, x)
            dinv = numpy.where(mask != 0, 1.0 / d, 1.0)
        else:
            dinv = 1.0 / d
        rho = (dinv * (b - Mx)).astype(numpy.float32)
        dk = (rho / numpy.float32(theta)).astype(numpy.float32)
        acc = dk.copy()
        rk = 1.0 / sigma
        for _ in range(steps - 1):
            rho = (rho - A16.dot(dk)).astype(numpy.float32)
            rn = 1.0 / (2.0 * sigma - rk)
            dk = (numpy.float32(rn * rk) * dk
                  + numpy.float32(2.0 * rn / delta) * rho).astype(numpy.float32)
            rk = rn
            acc = (acc + dk).astype(numpy.float32)
        x = x + acc.astype(numpy.float64)
        zn, xn = numpy.linalg.norm(acc.astype(numpy.float64)), numpy.linalg.norm(x)
        znorms.append(zn)
        if contraction * zn <= rtol * xn:
            return x, it + 1, znorms
    raise AssertionError('no convergence: %r' % znorms)


@pytest.mark.parametrize('deg', [1, 2])
def test_the_algorithm_on_the_oracles_mass_matrix(deg):
    '''Contraction per defect correction below the bound the stopping test
    uses; the accepted iterate is within rtol of the direct solve.'''
    rng = numpy.random.RandomState(5)
    mesh = fem.karman_channel(30, 10, fitted=True)
    _V, M = _oracle_mass(mesh, deg)
    n = M.shape[0]
    lo, hi = fmass.WATHEN[deg]
    lo, hi = fmass._PAD[0] * lo, fmass._PAD[1] * hi
    steps = 6
    bound = 1.5 * (fmass.chebyshev_contraction(lo, hi, steps) + fmass._FP16_TERM)
    xref = rng.standard_normal(n)
    b = M.dot(xref)
    x, its, zn = _defect_correction_numpy(M, b, numpy.zeros(n), 1e-10, steps, lo,
                                          hi, bound)
    ratios = [b_ / a_ for a_, b_ in zip(zn, zn[1:])]
    assert max(ratios) < bound, (ratios, bound)
    assert cases.rel_l2(x, xref) < 1e-10
    assert its <= 7
    # identity rows: the masked rows converge to b like the rest
    mask = (rng.uniform(size=n) > 0.1).astype(numpy.uint8)
    g = rng.standard_normal(n)
    bb = numpy.where(mask != 0, b, g)
    x0 = numpy.where(mask != 0, 0.0, g)
    x, its, zn = _defect_correction_numpy(M, bb, x0, 1e-10, steps, lo, hi, bound,
                                          mask=mask)
    D = sp.diags((mask != 0).astype(float))
    Asym = D.dot(M).dot(D) + sp.diags((mask == 0).astype(float))
    bsym = numpy.where(mask != 0, bb - M.dot(numpy.where(mask == 0, g, 0.0)), g)
    uref = spla.splu(Asym.tocsc()).solve(bsym)
    assert cases.rel_l2(x, uref) < 1e-10
    assert (x[mask == 0] == g[mask == 0]).all()


# -- GPU ------------------------------------------------------------------------
def _dev(a):
    from flow_amd import device
    return device.to_device(numpy.ascontiguousarray(a))


@pytest.mark.gpu
@pytest.mark.parametrize('packed', [False, True])
@pytest.mark.parametrize('deg', [1, 2])
def test_scalar_mass_solve_matches_direct_solve(hip, deg, packed):
    from flow_amd.fem import ops
    rng = numpy.random.RandomState(7)
    for name, mesh in [('karman-48', fem.karman_channel(48, 12)),
                       ('karman-60-fitted', fem.karman_channel(60, 14, fitted=True))]:
        V = fem.FunctionSpace(mesh, 'CG', deg)
        M = ops.assemble_mass(V)
        # (packed: fp16 value + 16-bit column offset in one word per nonzero)
        solver = fmass.MassSolver(M, M.diag_inv(), packed=packed)
        assert (solver.packed16 is not None) == packed
        Ms = M.to_scipy().tocsc()
        b = rng.standard_normal(V.N)
        ref = spla.splu(Ms).solve(b)
        x = _dev(numpy.zeros(V.N))
        info = solver.solve(_dev(b), x, 1e-12)
        assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-9, (name, info)
        assert info.iterations <= 8, info
        # from a good start: fewer corrections, same answer
        x = _dev(ref * (1.0 + 1e-6 * rng.standard_normal(V.N)))
        info2 = solver.solve(_dev(b), x, 1e-12)
        assert info2.iterations < info.iterations, (info, info2)
        assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-9, (name, info2)
        # non-convergence is an error, like the Krylov solvers'
        with pytest.raises(RuntimeError):
            solver.solve(_dev(b), _dev(numpy.zeros(V.N)), 1e-12, maxit=1)


@pytest.mark.gpu
@pytest.mark.parametrize('packed', [False, True])
def test_pair_mass_solve_with_identity_rows(hip, packed):
    '''flow_operator kind 4 (the velocity correction's system, reference
    :451-464) against the symmetrically eliminated direct solve; the count the
    device reports is exact whatever the host enqueues ahead.'''
    import torch
    from flow_amd.fem import ops
    rng = numpy.random.RandomState(21)
    mesh = fem.karman_channel(48, 12, fitted=True)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    n = lay.N
    M = ops.assemble_mass(V)
    free = (rng.uniform(size=2 * n) > 0.07).astype(numpy.uint8)
    A = ops.Matrix(lay, 4, M.vals, rowmask=_dev(free))
    solver = fmass.MassSolver(A, A.diag_inv(), packed=packed)
    Ms = M.to_scipy()
    g = rng.standard_normal(2 * n)
    b = rng.standard_normal(2 * n)
    b[free == 0] = g[free == 0]
    x0 = rng.standard_normal(2 * n)
    x0[free == 0] = g[free == 0]
    M2 = sp.block_diag([Ms, Ms], format='csr')
    D = sp.diags((free != 0).astype(float))
    Asym = D.dot(M2).dot(D) + sp.diags((free == 0).astype(float))
    bsym = numpy.where(free != 0, b - M2.dot(numpy.where(free == 0, g, 0.0)), g)
    uref = spla.splu(Asym.tocsc()).solve(bsym)
    xd = _dev(x0)
    info = solver.solve(_dev(b), xd, 1e-12)
    assert cases.rel_l2(xd.cpu().numpy(), uref) < 1e-9, info
    assert (xd.cpu().numpy()[free == 0] == g[free == 0]).all()
    assert torch.isfinite(xd).all()
    # enqueueing far more corrections than needed changes nothing
    xe = _dev(x0)
    info2 = solver.solve(_dev(b), xe, 1e-12, first_check=info.iterations + 20)
    assert info2.iterations == info.iterations
    assert (xe == xd).all()
    # masked rows that do NOT carry b on entry converge there too
    x1 = rng.standard_normal(2 * n)
    xf = _dev(x1)
    solver.solve(_dev(b), xf, 1e-12)
    assert cases.rel_l2(xf.cpu().numpy(), uref) < 1e-9


@pytest.mark.gpu
def test_contraction_watch_adapts_and_falls_back(hip):
    '''The run-time watch on the a-priori contraction (mass_scalar_kernel;
    ADVICE r5): (a) a bound that is too OPTIMISTIC does not fail the solve --
    the stopping test takes the contraction the iteration is seen to have --
    and does not stop it early either; (b) a polynomial that does not contract
    at all is given up for Jacobi-CG, the answer is still the direct solve's,
    and the solver counts the fallback.'''
    from flow_amd.fem import ops
    rng = numpy.random.RandomState(5)
    mesh = fem.karman_channel_graded(4.0e-3).reordered()     # graded: 1:4
    V = fem.FunctionSpace(mesh, 'CG', 2)
    M = ops.assemble_mass(V)
    Ms = M.to_scipy().tocsc()
    b = rng.standard_normal(V.N)
    ref = spla.splu(Ms).solve(b)
    good = fmass.MassSolver(M, M.diag_inv())
    x = _dev(numpy.zeros(V.N))
    base = good.solve(_dev(b), x, 1e-12)
    assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-9
    assert getattr(good, 'fallbacks', 0) == 0            # graded mesh: no trip
    # (a) the vouched contraction 100 x too small
    opt = fmass.MassSolver(M, M.diag_inv())
    opt.struct.contraction = 0.01 * good.struct.contraction
    x = _dev(numpy.zeros(V.N))
    info = opt.solve(_dev(b), x, 1e-12)
    assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-9, info
    assert getattr(opt, 'fallbacks', 0) == 0
    assert base.iterations - 1 <= info.iterations <= base.iterations
    # (b) a Chebyshev interval that misses most of the spectrum: no contraction
    bad = fmass.MassSolver(M, M.diag_inv())
    bad.struct.lam_max = 0.3 * good.struct.lam_max
    bad.struct.lam_min = 0.3 * good.struct.lam_min
    x = _dev(numpy.zeros(V.N))
    info = bad.solve(_dev(b), x, 1e-12)
    assert 'cg' in info.method and bad.fallbacks == 1, info
    # (Jacobi-CG from the last iterate, PETSc's test on the preconditioned
    # residual: the error against the direct solve is a few 1e-9)
    assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-7, info
    # ... and with the guard off it is an error like any non-convergence
    bad.guard = False
    with pytest.raises(RuntimeError):
        bad.solve(_dev(b), _dev(numpy.zeros(V.N)), 1e-12)
