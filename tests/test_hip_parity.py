# -*- coding: utf-8 -*-
'''
Parity of the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  All tests need a real MI355X: run with `-m gpu`.

Tolerances: the arithmetic is fp64 floating point; assembly kernels are compared
to 1e-11 relative (different summation order and quadrature rule than the
oracle, both exact for the polynomial integrands), Krylov solutions to 1e-8,
whole-step fields to 1e-7 rel-L2 (north star: 1e-6), pressures after removing
the mean in the pure Neumann case (tests/test_navier_stokes.py:347-360).
'''
import ctypes

import numpy
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from flow_amd import fem
from flow_amd.fem import ops
from oracle import fem_oracle as orc

import cases
import mms

pytestmark = pytest.mark.gpu


def _dev(a):
    from flow_amd import device
    return device.to_device(numpy.ascontiguousarray(a))


def _meshes():
    return [
        ('unit-crossed-6', fem.UnitSquareMesh(6, 6, 'crossed')),
        ('rect-leftright', fem.RectangleMesh(
            fem.Point(-1.0, -0.5), fem.Point(1.5, 1.0), 7, 5, 'left/right')),
        ('karman-24', fem.karman_channel(24, 8)),
        # the body-fitted obstacle of the bench's workload (general stretched
        # triangles, a polygonal hole with its vertices on the circle) ...
        ('karman-30-fitted', fem.karman_channel(30, 10, fitted=True)),
        # ... and of the Boussinesq driver's box
        ('heater-12-fitted', fem.heater_box(12, fitted=True)),
        ]


@pytest.mark.parametrize('deg', [1, 2])
@pytest.mark.parametrize('kind', ['stiffness', 'mass', 'lumped'])
def test_scalar_matrices(hip, deg, kind):
    for name, mesh in _meshes():
        V = fem.FunctionSpace(mesh, 'CG', deg)
        S = orc.Space(mesh.points, mesh.cell_vertices, V.layout.cell_dofs, deg,
                      V.N)
        code = {'stiffness': ops.STIFFNESS, 'mass': ops.MASS,
                'lumped': ops.LUMPED_MASS}[kind]
        A = ops.assemble_scalar_matrix(V.layout, code).to_scipy()
        ref = {'stiffness': orc.stiffness_matrix, 'mass': orc.mass_matrix,
               'lumped': orc.lumped_mass_vertex_rule}[kind](S)
        err = abs(A - ref).max() / abs(ref).max()
        assert err < 1e-12, (name, deg, kind, err)


@pytest.mark.parametrize('deg', [1, 2])
def test_spmv_and_block_operators(hip, deg):
    rng = numpy.random.RandomState(1)
    mesh = fem.karman_channel(60, 14)
    V = fem.FunctionSpace(mesh, 'CG', deg)
    lay = V.layout
    n, nnz = lay.N, lay.nnz
    planes = rng.standard_normal(4 * nnz)
    x = rng.standard_normal(2 * n)
    for kind, npl in ((0, 1), (1, 2), (2, 4)):
        A = ops.Matrix(lay, kind, _dev(planes[:npl * nnz].copy()))
        size = A.size
        xd = _dev(x[:size].copy())
        yd = _dev(numpy.zeros(size))
        A.apply(xd, yd)
        ref = A.to_scipy().dot(x[:size])
        err = abs(yd.cpu().numpy() - ref).max() / abs(ref).max()
        assert err < 1e-13, (kind, err)
        dinv = A.diag_inv().cpu().numpy()
        assert numpy.allclose(dinv, 1.0 / A.to_scipy().diagonal(), rtol=1e-14)


def test_blas1(hip):
    rng = numpy.random.RandomState(2)
    for n in (1, 63, 1000, 300001):
        x = rng.standard_normal(n)
        y = rng.standard_normal(n)
        xd, yd = _dev(x), _dev(y)
        assert abs(ops.dot(xd, yd) - x.dot(y)) <= 1e-12 * max(1.0, abs(x.dot(y)))
        assert abs(ops.vector_norm(xd, 'l2') - numpy.linalg.norm(x)) < 1e-12 * n**0.5
        assert ops.vector_norm(xd, 'linf') == abs(x).max()
        ops.axpby(0.5, xd, -2.0, yd)
        assert numpy.allclose(yd.cpu().numpy(), 0.5 * x - 2.0 * y, rtol=1e-15)
    # reductions are deterministic (no fp atomics): bitwise equal on repeat
    a = ops.dot(xd, yd)
    assert a == ops.dot(xd, yd)


def test_identity_row_operator(hip):
    '''flow_operator kind 4: one plane for both components, Dirichlet rows by
    mask.  Product, Jacobi diagonal, and CG from a start vector that carries the
    boundary values against the symmetrically eliminated system (what
    `solve(..., 'symmetric': True)` assembles, reference :451-464).'''
    import torch
    rng = numpy.random.RandomState(21)
    mesh = fem.karman_channel(48, 12)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    n = lay.N
    M = ops.assemble_mass(V)
    free = (rng.uniform(size=2 * n) > 0.07).astype(numpy.uint8)
    A = ops.Matrix(lay, 4, M.vals, rowmask=_dev(free))
    assert A.size == 2 * n
    x = rng.standard_normal(2 * n)
    y = _dev(numpy.zeros(2 * n))
    A.apply(_dev(x), y)
    Ms = M.to_scipy()
    ref = numpy.concatenate([Ms.dot(x[:n]), Ms.dot(x[n:])])
    ref = numpy.where(free != 0, ref, x)
    assert abs(y.cpu().numpy() - ref).max() <= 1e-13 * abs(ref).max()
    assert abs(A.to_scipy().dot(x) - ref).max() <= 1e-13 * abs(ref).max()
    dinv = A.diag_inv().cpu().numpy()
    dref = numpy.where(free != 0, 1.0 / numpy.tile(Ms.diagonal(), 2), 1.0)
    assert numpy.allclose(dinv, dref, rtol=1e-14)
    # the Dirichlet problem: u = g on the masked rows, (M u)_i = b_i elsewhere
    g = rng.standard_normal(2 * n)
    b = rng.standard_normal(2 * n)
    b[free == 0] = g[free == 0]
    x0 = rng.standard_normal(2 * n)
    x0[free == 0] = g[free == 0]
    xd = _dev(x0)
    info = ops.krylov_solve('cg', A, _dev(b), xd, rtol=1e-13, maxit=2000,
                            check_every=5)
    M2 = sp.block_diag([Ms, Ms], format='csr')
    D = sp.diags((free != 0).astype(float))
    Asym = D.dot(M2).dot(D) + sp.diags((free == 0).astype(float))
    bsym = numpy.where(free != 0, b - M2.dot(numpy.where(free == 0, g, 0.0)), g)
    uref = spla.splu(Asym.tocsc()).solve(bsym)
    assert cases.rel_l2(xd.cpu().numpy(), uref) < 1e-9, info
    assert (xd.cpu().numpy()[free == 0] == g[free == 0]).all()
    assert torch.isfinite(xd).all()


def test_cg_matches_direct_solve(hip):
    rng = numpy.random.RandomState(3)
    mesh = fem.karman_channel(48, 12)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    M = ops.assemble_mass(V)
    K = ops.assemble_stiffness(V)
    A = ops.Matrix(V.layout, 0, (K.vals + 50.0 * M.vals).contiguous())
    b = rng.standard_normal(V.N)
    x = _dev(numpy.zeros(V.N))
    info = ops.krylov_solve('cg', A, _dev(b), x, rtol=1e-13, maxit=5000)
    ref = spla.splu(A.to_scipy().tocsc()).solve(b)
    assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-9, info
    with pytest.raises(RuntimeError):
        ops.krylov_solve('cg', A, _dev(b), _dev(numpy.zeros(V.N)), rtol=1e-13,
                         maxit=3, check_every=2)


@pytest.mark.parametrize('neumann', [False, True])
def test_two_level_cg(hip, neumann):
    '''Jacobi + aggregate coarse space: same solution, far fewer iterations.'''
    rng = numpy.random.RandomState(14)
    mesh = fem.karman_channel(160, 37)
    V = fem.FunctionSpace(mesh, 'CG', 1)
    K = ops.assemble_stiffness(V)
    n = V.N
    if neumann:
        A = K
        isbc = None
        xs = rng.standard_normal(n)
        b = A.to_scipy().dot(xs)            # consistent right-hand side
    else:
        isbc = mesh.points[:, 0] > 0.6 - 1e-12
        A = ops.symmetric_bc_matrix(K, _dev(isbc.astype(numpy.uint8)))
        b = rng.standard_normal(n)
        b[isbc] = 0.0
    coarse = ops.CoarseSpace(A, isbc, singular=neumann, target_nc=256)
    assert coarse.nc <= 320
    assert (coarse.agg_of_host[isbc] == -1).all() if isbc is not None else True
    x1 = _dev(numpy.zeros(n))
    x2 = _dev(numpy.zeros(n))
    i1 = ops.krylov_solve('cg', A, _dev(b), x1, rtol=1e-12, maxit=20000)
    i2 = ops.krylov_solve('cg', A, _dev(b), x2, rtol=1e-12, maxit=20000,
                          coarse=coarse, check_every=5)
    assert i2.iterations * 4 < i1.iterations, (i1, i2)
    a1, a2 = x1.cpu().numpy(), x2.cpu().numpy()
    if neumann:
        a1 -= a1.mean()
        a2 -= a2.mean()
        ref = xs - xs.mean()
    else:
        ref = spla.splu(A.to_scipy().tocsc()).solve(b)
    assert cases.rel_l2(a2, ref) < 1e-7, i2
    assert cases.rel_l2(a1, ref) < 1e-7, i1
    # smoothed-aggregation multigrid V-cycle: same solution again, an order of
    # magnitude fewer iterations still, and nearly mesh independent
    from flow_amd.fem.multigrid import Multigrid
    mg = Multigrid(A, isbc, singular=neumann, coarsest=300, keep_host=True)
    assert mg.nlevels >= 3 and mg.sizes[-1] <= 300, mg.sizes
    # the device cycle (regrouped: one product with A per level) against the
    # textbook V(1,1) cycle it restates, on the host with the same hierarchy
    def textbook(l, r):
        if l == mg.nlevels - 1:
            return mg.Ainv_host.dot(r)
        Al, D, P = mg.host_levels[l]
        x = mg.omega * r / D
        x = x + P.dot(textbook(l + 1, P.T.dot(r - Al.dot(x))))
        return x + mg.omega * (r - Al.dot(x)) / D
    rv = rng.standard_normal(n)
    zv = _dev(numpy.zeros(n))
    mg.apply(_dev(rv), zv)
    # (the coarsest inverse is held in fp32 on the device)
    assert cases.rel_l2(zv.cpu().numpy(), textbook(0, rv)) < 1e-6
    x3 = _dev(numpy.zeros(n))
    i3 = ops.krylov_solve('cg', A, _dev(b), x3, rtol=1e-12, maxit=500, mg=mg,
                          check_every=1)
    assert i3.iterations <= 40 and i3.iterations * 2 < i2.iterations, (i2, i3)
    a3 = x3.cpu().numpy()
    if neumann:
        a3 -= a3.mean()
    assert cases.rel_l2(a3, ref) < 1e-7, i3
    # the V-cycle is a symmetric positive operator (CG needs that)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    mu_, mv_ = _dev(numpy.zeros(n)), _dev(numpy.zeros(n))
    mg.apply(_dev(u), mu_)
    mg.apply(_dev(v), mv_)
    s_uv = v.dot(mu_.cpu().numpy())
    s_vu = u.dot(mv_.cpu().numpy())
    assert abs(s_uv - s_vu) <= 1e-6 * (abs(s_uv) + abs(s_vu)), (s_uv, s_vu)
    assert u.dot(mu_.cpu().numpy()) > 0.0 and v.dot(mv_.cpu().numpy()) > 0.0


def test_bicgstab_matches_direct_solve(hip):
    rng = numpy.random.RandomState(4)
    mesh = fem.UnitSquareMesh(10, 10, 'crossed')
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    M = ops.assemble_mass(V).vals
    K = ops.assemble_stiffness(V).vals
    import torch
    pert = _dev(0.02 * rng.standard_normal(4 * M.numel()) * float(M.abs().max()))
    base = torch.cat([M + 0.01 * K, torch.zeros_like(M), torch.zeros_like(M),
                      M + 0.01 * K])
    A = ops.Matrix(lay, 2, (base + pert).contiguous())
    b = rng.standard_normal(2 * V.N)
    x = _dev(numpy.zeros(2 * V.N))
    info = ops.krylov_solve('bicgstab', A, _dev(b), x, rtol=1e-13, maxit=2000)
    ref = spla.splu(A.to_scipy().tocsc()).solve(b)
    assert cases.rel_l2(x.cpu().numpy(), ref) < 1e-9, info
    # GMRES on the same system: one long cycle, short cycles (restarts), a
    # nonzero start, no preconditioner
    for kw in (dict(restart=30, x_is_zero=True), dict(restart=4, x_is_zero=True),
               dict(restart=7), dict(restart=30, dinv=None, x_is_zero=True)):
        xg = _dev(numpy.zeros(2 * V.N) if kw.get('x_is_zero')
                  else rng.standard_normal(2 * V.N))
        ig = ops.krylov_solve('gmres', A, _dev(b), xg, rtol=1e-13, maxit=4000,
                              **kw)
        assert cases.rel_l2(xg.cpu().numpy(), ref) < 1e-9, (kw, ig)
    # the Arnoldi steps are enqueued ahead of the host's read-backs (the
    # stopping test runs on the device): however many the caller announces --
    # none, the exact count, far too many -- the accepted iterate and the
    # reported count are the same
    for restart in (30, 6):
        outcomes = []
        for hint in (0, None, 3, 500):
            xg = _dev(numpy.zeros(2 * V.N))
            if hint is None:
                hint = outcomes[0][0]
            ig = ops.krylov_solve('gmres', A, _dev(b), xg, rtol=1e-11,
                                  maxit=4000, restart=restart, x_is_zero=True,
                                  first_check=hint)
            outcomes.append((ig.iterations, ig.residual, xg.clone()))
        for o in outcomes[1:]:
            assert o[0] == outcomes[0][0] and o[1] == outcomes[0][1]
            assert torch.equal(o[2], outcomes[0][2])
    # fewer operator applications than BiCGStab (two per iteration) needs
    xg = _dev(numpy.zeros(2 * V.N))
    ig = ops.krylov_solve('gmres', A, _dev(b), xg, rtol=1e-13, maxit=4000,
                          restart=30, x_is_zero=True)
    assert ig.iterations <= 2 * info.iterations, (ig, info)
    with pytest.raises(RuntimeError):
        ops.krylov_solve('gmres', A, _dev(b), _dev(numpy.zeros(2 * V.N)),
                         rtol=1e-13, maxit=3, restart=30, x_is_zero=True)


def test_gmres_does_not_take_cancellation_for_convergence(hip):
    '''A nearly exact preconditioner (here: an operator within 1e-9 of the
    identity, as the Newton systems are at tiny time steps) makes the first
    Krylov vector parallel to the residual to 9 digits: the one-pass
    Gram-Schmidt estimate of h_{1,0} is then rounding noise.  The solver must
    not read that as an invariant subspace and stop (round 1 did: one
    iteration, true residual 1e-9 |b| for a requested 1e-13) but verify with
    the true residual and go on.'''
    rng = numpy.random.RandomState(9)
    mesh = fem.UnitSquareMesh(12, 12, 'crossed')
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    K = ops.assemble_stiffness(V)
    vals = 1.0e-9 * K.vals / float(K.vals.abs().max())
    vals[lay.dev('diag_idx').long()] += 1.0
    A = ops.Matrix(lay, 0, vals.contiguous())
    b = rng.standard_normal(V.N)
    x = _dev(numpy.zeros(V.N))
    info = ops.krylov_solve('gmres', A, _dev(b), x, rtol=1e-13, maxit=100,
                            restart=10, dinv=None, x_is_zero=True)
    t = _dev(numpy.zeros(V.N))
    A.apply(x, t)
    true = float((t - _dev(b)).norm())
    assert true <= 2e-13 * numpy.linalg.norm(b), (info, true)
    assert info.iterations >= 2


def test_gmres_reports_the_true_residual(hip):
    '''A cycle of flow_gmres_solve stops on the least-squares estimate of the
    residual; the iterate it returns is then verified with b - A x, and the
    norm handed back is that TRUE one -- also with a reduced-precision
    preconditioner (fp32-packed ILU(0) with an fp32 sweep vector) at
    rtol 1e-13, where the estimate alone could run ahead of the truth.'''
    from flow_amd.fem import ilu
    rng = numpy.random.RandomState(31)
    mesh = fem.UnitSquareMesh(10, 10, 'crossed')
    V = fem.FunctionSpace(mesh, 'CG', 2)
    lay = V.layout
    M = ops.assemble_mass(V).vals
    K = ops.assemble_stiffness(V).vals
    import torch
    # (the system of test_bicgstab_matches_direct_solve)
    pert = _dev(0.02 * rng.standard_normal(4 * M.numel()) * float(M.abs().max()))
    base = torch.cat([M + 0.01 * K, torch.zeros_like(M), torch.zeros_like(M),
                      M + 0.01 * K])
    A = ops.Matrix(lay, 2, (base + pert).contiguous())
    b = rng.standard_normal(2 * V.N)
    for kw in (dict(ilu=ilu.Ilu0(A, packed=True, single_vector=True)),
               dict(dinv='jacobi')):
        x = _dev(numpy.zeros(2 * V.N))
        info = ops.krylov_solve('gmres', A, _dev(b), x, rtol=1e-13, maxit=4000,
                                restart=30, x_is_zero=True, **kw)
        t = _dev(numpy.zeros(2 * V.N))
        A.apply(x, t)
        true = float((t - _dev(b)).norm())
        assert abs(info.residual - true) <= 1e-2 * true + 1e-300, (info, true)
        assert true <= 10.0 * 1e-13 * numpy.linalg.norm(b), (info, true)


@pytest.mark.parametrize('vdeg', [1, 2])
@pytest.mark.parametrize('method', ['backward euler', 'crank-nicolson',
                                    'forward euler'])
def test_momentum_residual_and_jacobian(hip, vdeg, method):
    '''K5/K6 against the oracle's restatement of _rhs_weak + derivative().'''
    from flow_amd import _hip, device
    from flow_amd.navier_stokes import pressure_correction as pc
    lib = hip
    for name, mesh in _meshes():
        case = cases.Case(mesh, vdeg=vdeg, dt=0.07, rho=1.3, mu=0.4,
                          f_degree=3, seed=5)
        W, P = case.oracle_spaces()
        rng = numpy.random.RandomState(6)
        ui = case.u0 + 0.1 * rng.standard_normal(len(case.u0))
        th_i, th_e = pc._THETA[method]
        # oracle
        Mo = sp.block_diag([orc.mass_matrix(W)] * 2, format='csr')
        Ri, dRi = orc.momentum_rhs(W, P, ui, case.p0, case.lattice(case.f1),
                                   case.rho, case.mu)
        Re, _ = orc.momentum_rhs(W, P, case.u0, case.p0,
                                 case.lattice(case.f0), case.rho, case.mu,
                                 want_jacobian=False)
        c = case.dt / case.rho
        F_ref = Mo.dot(ui - case.u0) - c * (th_i * Ri + th_e * Re)
        J_ref = Mo - c * th_i * dRi
        # product
        lay = case.W.layout
        nc = mesh.num_cells()
        f0 = fem.as_cell_coefficient(case.f0, mesh, 2)
        f1 = fem.as_cell_coefficient(case.f1, mesh, 2)
        f0s, k0 = ops.coef_struct(f0, mesh, vdeg)
        f1s, k1 = ops.coef_struct(f1, mesh, vdeg)
        prm = _hip.NsParams(case.dt, case.rho, case.mu, th_i, th_e)
        F = device.empty(2 * lay.N)
        J = ops.Matrix(lay, 2)
        buf = ops.scratch(mesh, 4 * lay.nloc**2 * nc)
        bfm = _dev(mesh.cell_bfacet_mask())
        uid, u0d, p0d = _dev(ui), _dev(case.u0), _dev(case.p0)
        _hip.check(lib.flow_assemble_momentum(
            ctypes.byref(ops.mesh_struct(mesh)),
            ctypes.byref(ops.space_struct(lay)),
            ctypes.byref(ops.space_struct(case.P.layout)), _hip.i32(bfm),
            _hip.f64(uid), _hip.f64(u0d), _hip.f64(p0d), ctypes.byref(f0s),
            ctypes.byref(f1s), ctypes.byref(prm), _hip.f64(buf), _hip.f64(F),
            _hip.f64(J.vals), J.stride, _hip.stream()
            ))
        Fh = F.cpu().numpy()
        errF = abs(Fh - F_ref).max() / abs(F_ref).max()
        assert errF < 1e-11, (name, vdeg, method, 'F', errF)
        Jh = J.to_scipy()
        errJ = abs(Jh - J_ref).max() / abs(J_ref).max()
        assert errJ < 1e-11, (name, vdeg, method, 'J', errJ)
        # the matrix-free action of the same Jacobian (flow_momentum_jvp_apply),
        # with identity rows on a set of "Dirichlet" dofs
        v = rng.standard_normal(2 * lay.N)
        bc = numpy.unique(rng.randint(0, 2 * lay.N, size=7)).astype(numpy.int32)
        Jop = ops.MomentumJacobian(case.W, bfm, uid, prm, _dev(bc))
        out = device.empty(2 * lay.N)
        Jop.apply(_dev(v), out)
        ref = J_ref.dot(v)
        ref[bc] = v[bc]
        errV = abs(out.cpu().numpy() - ref).max() / abs(ref).max()
        assert errV < 1e-12, (name, vdeg, method, 'J v', errV)


@pytest.mark.parametrize('vdeg', [1, 2])
@pytest.mark.parametrize('rotational', [False, True])
def test_pressure_and_correction_rhs(hip, vdeg, rotational):
    from flow_amd import _hip, device
    lib = hip
    for name, mesh in _meshes():
        case = cases.Case(mesh, vdeg=vdeg, dt=0.03, rho=2.0, mu=0.7, seed=7)
        W, P = case.oracle_spaces()
        rng = numpy.random.RandomState(8)
        p1 = case.p0 + 0.3 * rng.standard_normal(len(case.p0))
        b_ref = orc.pressure_rhs(W, P, case.u0, case.p0, 1.0, case.rho, case.mu,
                                 case.dt, rotational)
        nc = mesh.num_cells()
        lay = case.W.layout
        ud, p0d, p1d = _dev(case.u0), _dev(case.p0), _dev(p1)
        b = device.empty(case.P.N)
        buf = ops.scratch(mesh, 2 * lay.nloc * nc)
        ms = ops.mesh_struct(mesh)
        ws = ops.space_struct(lay)
        ps = ops.space_struct(case.P.layout)
        _hip.check(lib.flow_assemble_pressure_rhs(
            ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(ps), _hip.f64(ud),
            _hip.f64(p0d), case.rho / case.dt, case.mu, int(rotational),
            _hip.f64(buf), _hip.f64(b), _hip.stream()
            ))
        err = abs(b.cpu().numpy() - b_ref).max() / abs(b_ref).max()
        assert err < 1e-11, (name, 'pressure rhs', err)
        # correction rhs: oracle builds it inside velocity_correction; restate
        Mo = sp.block_diag([orc.mass_matrix(W)] * 2, format='csr')
        u_free = orc.velocity_correction(
            W, P, case.u0, p1, case.p0, numpy.zeros(0, dtype=int),
            numpy.zeros(0), case.rho, case.mu, case.dt, rotational
            )
        c_ref = Mo.dot(u_free)
        cvec = device.empty(2 * lay.N)
        _hip.check(lib.flow_assemble_correction_rhs(
            ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(ps), _hip.f64(ud),
            _hip.f64(p1d), _hip.f64(p0d), case.dt / case.rho, case.mu,
            int(rotational), _hip.f64(buf), _hip.f64(cvec), _hip.stream()
            ))
        err = abs(cvec.cpu().numpy() - c_ref).max() / abs(c_ref).max()
        assert err < 1e-10, (name, 'correction rhs', err)


def test_project_and_norms(hip):
    mesh = fem.UnitSquareMesh(8, 8, 'crossed')
    pb = mms.guermond2()
    W = fem.VectorFunctionSpace(mesh, 'CG', 2)
    S = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    expr = fem.Expression(lambda x: pb.u(x, 0.3), degree=5)
    uh = fem.project(expr, W)
    case = cases.Case(mesh)
    lat = case.lattice(expr)
    ref = orc.l2_project(S, lat[0], lat[1], dim=2)
    assert cases.rel_l2(uh.array(), ref) < 1e-11
    err_h = fem.errornorm(expr, uh)
    err_o = orc.l2_error(S, ref, lat[0], lat[1], dim=2)
    assert abs(err_h - err_o) < 1e-9 * max(err_o, 1e-3)
    M = orc.mass_matrix(S)
    l2 = numpy.sqrt(sum(
        ref[c * S.N:(c + 1) * S.N].dot(M.dot(ref[c * S.N:(c + 1) * S.N]))
        for c in range(2)))
    assert abs(fem.norm(uh, 'L2') - l2) < 1e-11 * l2
    assert fem.norm(uh.vector(), 'linf') == abs(uh.array()).max()


@pytest.mark.parametrize('scheme', ['chorin', 'ipcs', 'rotational'])
@pytest.mark.parametrize('vdeg', [1, 2])
def test_step_parity_neumann(hip, scheme, vdeg):
    '''Whole step vs oracle, velocity Dirichlet everywhere, Neumann pressure.'''
    mesh = fem.UnitSquareMesh(8, 8, 'crossed')
    case = cases.Case(mesh, vdeg=vdeg, dt=0.05, bc_kind='all', seed=11)
    u1o, p1o, uio = case.oracle_step(scheme)
    u1, p1, ui = case.product_step(scheme)
    W, P = case.oracle_spaces()
    Mp = orc.mass_matrix(P)
    assert cases.rel_l2(ui, uio) < 1e-7, 'tentative velocity'
    assert cases.rel_l2(cases.mean_free(p1, Mp), cases.mean_free(p1o, Mp)) < 1e-7
    assert cases.rel_l2(u1, u1o) < 1e-7


@pytest.mark.parametrize('fitted', [False, True])
@pytest.mark.parametrize('method', ['backward euler', 'crank-nicolson',
                                    'forward euler'])
def test_step_parity_channel(hip, method, fitted):
    '''Component-wise velocity conditions + pressure Dirichlet at the outlet
    (the Karman setting), Rotational scheme, on the channel-with-obstacle mesh:
    the staircase obstacle and the body-fitted one the bench runs on (stretched
    triangles; the `ds` terms of _rhs_weak, reference :135-144, act on the
    free rows of the two sides).'''
    mesh = fem.karman_channel(30, 10, fitted=fitted)
    pb = mms.guermond2()
    case = cases.Case(mesh, vdeg=2, problem=pb, dt=0.02, bc_kind='channel',
                      rho=1.5, mu=0.05, seed=12)
    u1o, p1o, uio = case.oracle_step('rotational', method)
    u1, p1, ui = case.product_step('rotational', method)
    assert cases.rel_l2(ui, uio) < 1e-7
    assert cases.rel_l2(p1, p1o) < 1e-7
    assert cases.rel_l2(u1, u1o) < 1e-7


@pytest.mark.parametrize('scheme', ['chorin', 'ipcs', 'rotational'])
def test_step_parity_fitted_box(hip, scheme):
    '''Whole step vs oracle on the body-fitted heater box (the Boussinesq
    driver's mesh): velocity Dirichlet everywhere -- also on the polygonal
    heater --, Neumann pressure.'''
    mesh = fem.heater_box(12, fitted=True)
    case = cases.Case(mesh, vdeg=2, dt=0.03, bc_kind='all', rho=1.2, mu=0.08,
                      seed=31)
    u1o, p1o, uio = case.oracle_step(scheme)
    u1, p1, ui = case.product_step(scheme)
    W, P = case.oracle_spaces()
    Mp = orc.mass_matrix(P)
    assert cases.rel_l2(ui, uio) < 1e-7, 'tentative velocity'
    assert cases.rel_l2(cases.mean_free(p1, Mp), cases.mean_free(p1o, Mp)) < 1e-7
    assert cases.rel_l2(u1, u1o) < 1e-7


def test_newton_failure_raises_runtime_error(hip):
    import flow_amd.navier_stokes as navsto
    mesh = fem.UnitSquareMesh(4, 4, 'crossed')
    case = cases.Case(mesh, vdeg=2, dt=0.05, seed=13)
    old = dict(navsto.solver_parameters['newton'])
    navsto.solver_parameters['newton']['maximum_iterations'] = 0
    try:
        with pytest.raises(RuntimeError):
            case.product_step('ipcs')
    finally:
        navsto.solver_parameters['newton'].update(old)
    with pytest.raises(AssertionError):
        import flow_amd.fem as f
        u0 = f.Function(case.W)
        p0 = f.Function(case.P)
        navsto.IPCS().step(f.Constant(-1.0), {0: u0}, p0, [], [],
                           f.Constant(1.0), f.Constant(1.0),
                           f={0: f.Constant((0, 0)), 1: f.Constant((0, 0))})


_PAIR_SNIPPET = r'''
import sys, numpy, scipy.sparse as sp
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
from flow_amd import fem, _hip, device
from flow_amd.fem import ops
from flow_amd.navier_stokes import pressure_correction as pc
from oracle import fem_oracle as orc
import cases
worst = 0.0
for mesh in (fem.karman_channel(30, 10, fitted=True), fem.heater_box(12, fitted=True)):
    for method in ('backward euler', 'crank-nicolson'):
        case = cases.Case(mesh, vdeg=2, dt=0.07, rho=1.3, mu=0.4, f_degree=3, seed=5)
        W, P = case.oracle_spaces()
        rng = numpy.random.RandomState(6)
        ui = case.u0 + 0.1 * rng.standard_normal(len(case.u0))
        th_i, th_e = pc._THETA[method]
        Mo = sp.block_diag([orc.mass_matrix(W)] * 2, format='csr')
        Ri, dRi = orc.momentum_rhs(W, P, ui, case.p0, case.lattice(case.f1), case.rho, case.mu)
        J_ref = Mo - case.dt / case.rho * th_i * dRi
        lay = case.W.layout
        prm = _hip.NsParams(case.dt, case.rho, case.mu, th_i, th_e)
        bfm = device.to_device(mesh.cell_bfacet_mask())
        v = rng.standard_normal(2 * lay.N)
        bc = numpy.unique(rng.randint(0, 2 * lay.N, size=7)).astype(numpy.int32)
        Jop = ops.MomentumJacobian(case.W, bfm, device.to_device(ui), prm, device.to_device(bc))
        out = device.empty(2 * lay.N)
        Jop.apply(device.to_device(v), out)
        ref = J_ref.dot(v); ref[bc] = v[bc]
        worst = max(worst, abs(device.to_host(out).numpy() - ref).max() / abs(ref).max())
print('PAIR_ERR %%.3e' %% worst)
'''


def test_two_lanes_per_cell_jacobian_action(hip):
    '''momentum_jvp_pair_kernel (FLOW_AMD_JVP_PAIR=1: read once per process,
    hence the child process -- a plain script, not the test runner): the same
    action as the oracle's Jacobian to 1e-12, boundary facets included.'''
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FLOW_AMD_JVP_PAIR='1')
    out = subprocess.run(
        [sys.executable, '-c', _PAIR_SNIPPET % dict(
            tests=here, root=os.path.dirname(here))],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith('PAIR_ERR')]
    assert line and float(line[0].split()[1]) < 1e-12, out.stdout.decode()
