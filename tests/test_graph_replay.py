# -*- coding: utf-8 -*-
'''
Iteration bodies replayed as HIP graphs (flow_amd/csrc/graph_replay.hip,
include/flow_hip.h: flow_graph_mode): the loops of the pressure CG, of the
flexible GMRES of the Newton systems and of the mass solver issue the same
launches with the same arguments every iteration; a replayed graph must
therefore give the SAME numbers as the launches one by one -- bit for bit --,
survive the step-size controller (the step size of the matrix-free Jacobian
travels through device memory), and never outlive the buffers it was captured
over (the key is a hash of the values that reach the kernels).  The replay is
an option (FLOW_AMD_GRAPHS=1), not the default: measured slower than the
launches it replaces (csrc/graph_replay.hip).  GPU only.
'''
import numpy
import pytest

pytestmark = pytest.mark.gpu


def _problem():
    from flow_amd import karman
    import flow_amd.navier_stokes as navsto
    navsto.solver_parameters['pressure']['mg_coarsest'] = 200
    prob = karman.KarmanProblem(193, 45, mu=0.0226)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    return prob


def _run(prob, snap, steps):
    from flow_amd import device
    prob.restore(snap)
    # (the Newton preconditioner is lagged and its eigenvalue estimates are
    # warm-started from the previous build: every run builds a new one)
    for slot in ('jacobian_ilu', 'jacobian_pmg'):
        prob.W.layout._dev.pop(slot, None)
    rows = []
    for _ in range(steps):
        info = prob.step()
        rows.append((info['dt'], info['pressure'].iterations,
                     tuple(info['newton_linear_applications']),
                     info['correction'].iterations))
    return (device.to_host(prob.u0.data).numpy().copy(),
            device.to_host(prob.p0.data).numpy().copy(), rows)


def test_replayed_iterations_are_the_launched_ones_bit_for_bit(hip):
    '''From the start-up ramp (the step size doubles from step to step) onto
    the plateau: every solve of 14 steps with the graphs forced on against the
    same steps launched kernel by kernel.'''
    from flow_amd import _hip
    prob = _problem()
    snap = prob.snapshot()
    try:
        _hip.graph_mode(0)
        u_a, p_a, rows_a = _run(prob, snap, 14)
        s0 = _hip.graph_stats()
        assert s0['graphs'] == 0
        _hip.graph_mode(1)
        n0 = _hip.launch_count()
        u_b, p_b, rows_b = _run(prob, snap, 14)
        n1 = _hip.launch_count()
        s1 = _hip.graph_stats()
        # a second pass replays what the first one captured
        u_c, p_c, rows_c = _run(prob, snap, 14)
        n2 = _hip.launch_count()
        s2 = _hip.graph_stats()
    finally:
        _hip.graph_mode(0)
    assert rows_a == rows_b == rows_c
    assert len(set(r[0] for r in rows_a)) > 5        # the step size did move
    assert numpy.array_equal(u_a, u_b) and numpy.array_equal(p_a, p_b)
    assert numpy.array_equal(u_a, u_c) and numpy.array_equal(p_a, p_c)
    assert s1['replays'] > 100 and s1['nodes'] > 5 * s1['replays']
    # the second pass finds the graphs of the pressure CG and of the mass
    # solver again; those of the GMRES are captured anew, over the buffers of
    # its new preconditioner
    assert 0 < s2['captures'] - s1['captures'] < s1['captures']
    assert s2['replays'] > s1['replays'] + 300
    # a replay counts as one launch
    assert n2 - n1 < 0.6 * (s2['nodes'] - s1['nodes'] + n2 - n1)
    print('14 steps: %d launches with graphs (%d replays carrying %d kernels, '
          '%d graphs kept)' % (n1 - n0, s1['replays'], s1['nodes'], s1['graphs']))


def test_the_graphs_survive_the_step_size_controller(hip):
    '''On the plateau the controller still moves the step size a little every
    step, and the matrix-free Jacobian carries it: it travels through device
    memory, not through the captured arguments -- no capture in steady
    stepping (the lagged preconditioner is not rebuilt in ten steps).'''
    from flow_amd import _hip
    prob = _problem()
    try:
        _hip.graph_mode(1)
        prob.settle()
        for _ in range(4):
            prob.step()
        s0 = _hip.graph_stats()
        infos = [prob.step() for _ in range(10)]
        s1 = _hip.graph_stats()
    finally:
        _hip.graph_mode(0)
    assert len(set(i['dt'] for i in infos)) == 10
    assert all(len(i['newton_linear_applications']) >= 1 for i in infos)
    assert s1['captures'] == s0['captures'], (s0, s1)
    assert s1['replays'] - s0['replays'] >= 10 * 8


def test_a_repacked_operator_is_never_replayed_over_its_old_buffers(hip):
    '''Same solver, same vectors, the same pattern with its values in NEW
    buffers: the key changes with the pointers inside the operator struct, the
    loop is captured again and gives the new matrix's solution.'''
    import scipy.sparse.linalg as spla
    from flow_amd import _hip, device, fem
    from flow_amd.fem import ops
    rng = numpy.random.RandomState(3)
    mesh = fem.karman_channel(48, 12)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    M = ops.assemble_mass(V)
    K = ops.assemble_stiffness(V)
    b = rng.standard_normal(V.N)
    bd = device.to_device(b)
    xd = device.zeros(V.N)
    try:
        _hip.graph_mode(1)
        before = _hip.graph_stats()
        keep = []
        for shift in (50.0, 400.0, 50.0):
            A = ops.Matrix(V.layout, 0, (K.vals + shift * M.vals).contiguous())
            keep.append(A)            # (all alive: three sets of buffers)
            _hip.fill(xd, 0.0)
            info = ops.krylov_solve('cg', A, bd, xd, rtol=1e-13, maxit=5000)
            ref = spla.splu(A.to_scipy().tocsc()).solve(b)
            x = device.to_host(xd).numpy()
            assert numpy.linalg.norm(x - ref) < 1e-9 * numpy.linalg.norm(ref), info
        after = _hip.graph_stats()
        assert after['captures'] == before['captures'] + 3
        assert after['replays'] > before['replays'] + 30
    finally:
        _hip.graph_mode(0)
