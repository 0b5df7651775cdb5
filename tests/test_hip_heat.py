# -*- coding: utf-8 -*-
'''
Parity of the heat operator (flow_amd/heat.py, kernels K13/K14) against the CPU
oracle's restatement of flow/heat.py and flow/stabilization.py.  GPU only.
Tolerances: assembled operators 1e-11 relative (fp64, different summation order
and quadrature), solves 1e-8.
'''
import numpy
import pytest

from flow_amd import fem, heat, stabilization, time_steppers
from flow_amd.fem import ops
from flow_amd.fem.bcs import collect
from oracle import fem_oracle as orc

import cases

pytestmark = pytest.mark.gpu


class Hot(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] < 1e-12)


class Cool(fem.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary & (x[1] > 0.2 - 1e-12)


def _setup(qdeg, wdeg, speed):
    mesh = fem.heater_box(6)
    Q = fem.FunctionSpace(mesh, 'Lagrange', qdeg)
    W = fem.VectorFunctionSpace(mesh, 'Lagrange', wdeg)
    conv = fem.Function(W)
    x = W.layout.dof_coords
    # a swirling field; `speed` scales the Peclet number
    conv.set_array(speed * numpy.concatenate([
        -(x[:, 1] - 0.1) * (1.0 + x[:, 0]), (x[:, 0] - 0.05) * (1.0 + x[:, 1]**2)
        ]))
    Qo = orc.Space(mesh.points, mesh.cell_vertices, Q.layout.cell_dofs, qdeg, Q.N)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, wdeg, W.N)
    bcs = [fem.DirichletBC(Q, 320.0, Hot()), fem.DirichletBC(Q, 293.0, Cool())]
    return mesh, Q, W, conv, Qo, Wo, bcs


@pytest.mark.parametrize('qdeg,wdeg', [(2, 2), (1, 2), (1, 1)])
@pytest.mark.parametrize('supg', [False, True])
def test_heat_operators(hip, qdeg, wdeg, supg):
    kappa, rho, cp = 0.6, 998.0, 4182.0
    for speed in (1e-3, 5.0):
        mesh, Q, W, conv, Qo, Wo, bcs = _setup(qdeg, wdeg, speed)
        H = heat.Heat(Q, conv, kappa, rho, cp, bcs, fem.Constant(0.0),
                      supg_stabilization=supg)
        Mo, Ao, bo = orc.heat_operators(Qo, Wo, conv.array(), kappa, rho, cp,
                                        0.0, supg)
        Mh = H.M.to_scipy()
        Ah = H.A.to_scipy()
        assert abs(Mh - Mo).max() < 1e-11 * abs(Mo).max(), (speed, 'M')
        assert abs(Ah - Ao).max() < 1e-11 * abs(Ao).max(), (speed, 'A')
        assert abs(H.b.get_local()).max() == 0.0
        if qdeg == 2:
            # the vertex-quadrature quirk: edge rows of the lumped mass are zero
            lumped = ops.assemble_scalar_matrix(Q.layout, ops.LUMPED_MASS)
            d = lumped.to_scipy().diagonal()
            assert (d[Q.layout.edge_dofs] == 0.0).all()
            assert (d[Q.layout.vertex_dofs] > 0.0).all()


@pytest.mark.parametrize('qdeg,wdeg', [(1, 2), (2, 2)])
def test_heat_load_vector_with_source(hip, qdeg, wdeg):
    '''b = rhs(f) with a non-zero source: - int s v, and with SUPG also
    - int (s / rho_cp) tau conv.grad(v) (reference flow/heat.py:54-58, 79-86),
    against the oracle for a constant source; a source handed in as a
    degree-1 / degree-2 Expression of the same constant gives the same vector
    (the P_k interpolation per cell is exact for it).'''
    # (rho cp = 1: the SUPG part, which carries a factor 1 / (rho cp), is then
    # as large as - int s v and cannot hide in the tolerance)
    kappa, rho, cp = 1.0e-3, 1.0, 1.0
    src = 3.0
    mesh, Q, W, conv, Qo, Wo, bcs = _setup(qdeg, wdeg, 0.5)
    got = {}
    for supg in (False, True):
        _, _, bo = orc.heat_operators(Qo, Wo, conv.array(), kappa, rho, cp, src,
                                      supg)
        H = heat.Heat(Q, conv, kappa, rho, cp, bcs, fem.Constant(src),
                      supg_stabilization=supg)
        b = H.b.get_local()
        got[supg] = (b, bo)
        assert abs(b - bo).max() < 1e-11 * abs(bo).max(), supg
        for k in (1, 2):
            Hk = heat.Heat(Q, conv, kappa, rho, cp, bcs,
                           fem.Expression('s + 0.0 * x[0]', degree=k, s=src),
                           supg_stabilization=supg)
            assert abs(Hk.b.get_local() - bo).max() < 1e-11 * abs(bo).max()
    # the SUPG part on its own (it is small next to - int s v: a missing term
    # could hide in the tolerance above)
    part = got[True][0] - got[False][0]
    part_o = got[True][1] - got[False][1]
    assert abs(part_o).max() > 1e-2 * abs(got[False][1]).max()
    assert abs(part - part_o).max() < 1e-9 * abs(part_o).max()


def test_supg_tau_kernel(hip):
    mesh, Q, W, conv, Qo, Wo, _ = _setup(2, 2, 2.0)
    eps = 1.4e-4
    tau = stabilization.supg(mesh, conv, eps, 2).cell_vertex_values()
    c = conv.array()
    Cc = numpy.stack([c[W.layout.cell_dofs], c[W.N + W.layout.cell_dofs]], axis=1)
    pc = mesh.points[mesh.cell_vertices]
    ref = numpy.array([
        [orc.supg_tau(pc[k], Cc[k, :, v], eps, 2) for v in range(3)]
        for k in range(mesh.num_cells())
        ])
    assert abs(tau - ref).max() < 1e-12 * abs(ref).max()
    # zero convection -> tau = 0 (no NaN); huge tau -> RuntimeError
    zero = fem.Function(W)
    assert (stabilization.supg(mesh, zero, eps, 2).cell_vertex_values() == 0).all()
    slow = fem.Function(W)
    slow.set_array(numpy.full(2 * W.N, 1e-6))
    with pytest.raises(RuntimeError):
        stabilization.supg(mesh, slow, 1e-12, 1).cell_vertex_values()


@pytest.mark.parametrize('supg', [False, True])
def test_heat_solve_and_eval(hip, supg):
    kappa, rho, cp = 0.6, 998.0, 4182.0
    # velocities of the order of the Boussinesq run (cell Peclet number ~ 1):
    # with Jacobi the Krylov solver needs a diffusion-resolved regime (the
    # reference uses LU: 'The Krylov solver doesn't converge', heat.py:116)
    mesh, Q, W, conv, Qo, Wo, bcs = _setup(2, 2, 2.0e-5)
    H = heat.Heat(Q, conv, kappa, rho, cp, bcs, fem.Constant(0.0),
                  supg_stabilization=supg)
    Mo, Ao, bo = orc.heat_operators(Qo, Wo, conv.array(), kappa, rho, cp, 0.0,
                                    supg)
    rng = numpy.random.RandomState(0)
    u = fem.Function(Q)
    u.set_array(293.0 + rng.standard_normal(Q.N))
    ev = H.eval_alpha_M_beta_F(0.7, -0.3, u, 0.0).get_local()
    ref = 0.7 * Mo.dot(u.array()) - 0.3 * (Ao.dot(u.array()) + bo)
    assert abs(ev - ref).max() < 1e-11 * abs(ref).max()
    # one implicit Euler step = solve (M - dt A) u1 = M u0 with the BCs
    dt = 0.01
    stepper = time_steppers.ImplicitEuler(H)
    u1 = stepper.step(u, 0.0, dt)
    dofs, vals = collect(bcs, Q.N)
    ref1 = orc.heat_solve(Mo, Ao, 1.0, -dt, Mo.dot(u.array()), dofs, vals)
    # ill-conditioned (zero-mass edge rows, tiny diffusion): 1e-6 is the
    # north-star field tolerance
    assert cases.rel_l2(u1.array(), ref1) < 1e-6
