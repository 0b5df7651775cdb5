# -*- coding: utf-8 -*-
'''
Generates the golden fixtures under tests/golden/ with the CPU oracle
(oracle/fem_oracle.py).  The reference itself cannot run offline (no dolfin),
so these vectors come from the oracle, which is pinned by the reference's
analytic known-answer tests (tests/test_oracle_pinning.py).

    python tests/golden/make_golden.py [--force] [--large [name ...]]

--large writes the fixtures ns_large_<name>.npz, bq_large_<name>.npz and
stokes_large_<name>.npz instead (tests/large_cases.py:
one Rotational step, backward Euler and Crank-Nicolson, of the Karman channel
problem at BASELINE config 2's size and on a Taylor-Hood channel of 0.76 M DoF
-- minutes of sparse LU each, run in the build container; the GPU box only
reads them).  Such a fixture holds the generator's arguments, a fingerprint of
the generated inputs and, of each output field, every 87th dof and the l2 /
max norms per component: the inputs are analytic and rebuilt where they are
needed.

Existing fixtures are left alone unless --force is given: they carry their own
mesh arrays, and the ones written before the generators numbered cells x-major
(channel_p1, channel_p2, heat_ops) would come out with the same values in a
different cell order.

Fixtures are plain data (inputs and expected outputs):
  ns_step_<name>.npz   one pressure-correction step per scheme
  heat_ops.npz         heat operators M, A (CSR data) with and without SUPG,
                       SUPG tau per cell vertex, one implicit-Euler solve
'''
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from flow_amd import fem                                   # noqa: E402
from flow_amd.fem.bcs import collect                       # noqa: E402
from oracle import fem_oracle as orc                       # noqa: E402
import cases                                               # noqa: E402
import mms                                                 # noqa: E402


FORCE = '--force' in sys.argv[1:]
LARGE = '--large' in sys.argv[1:]


def _wanted(filename):
    path = os.path.join(HERE, filename)
    if os.path.exists(path) and not FORCE:
        print('kept   ', filename)
        return None
    print('writing', filename)
    return path


def ns_fixture(name, mesh, vdeg, bc_kind, dt, rho, mu, seed):
    path = _wanted('ns_step_%s.npz' % name)
    if path is None:
        return
    case = cases.Case(mesh, vdeg=vdeg, dt=dt, bc_kind=bc_kind, rho=rho, mu=mu,
                      f_degree=2, seed=seed)
    u_bc, p_bc = case.bc_data()
    data = {
        'points': mesh.points, 'cells': mesh.cell_vertices,
        'w_cell_dofs': case.W.layout.cell_dofs,
        'p_cell_dofs': case.P.layout.cell_dofs,
        'vdeg': vdeg, 'dt': dt, 'rho': rho, 'mu': mu,
        'u0': case.u0, 'p0': case.p0,
        'u_bc_dofs': u_bc[0], 'u_bc_vals': u_bc[1],
        'p_bc_dofs': p_bc[0] if p_bc else numpy.zeros(0, dtype=numpy.int32),
        'p_bc_vals': p_bc[1] if p_bc else numpy.zeros(0),
        'f_degree': 2,
        'f0': case.lattice(case.f0)[1], 'f1': case.lattice(case.f1)[1],
        }
    for scheme, method in (('chorin', 'backward euler'),
                           ('ipcs', 'backward euler'),
                           ('rotational', 'backward euler'),
                           ('ipcs', 'crank-nicolson')):
        u1, p1, ui = case.oracle_step(scheme, method)
        key = scheme + '_' + method.replace(' ', '_').replace('-', '_')
        data[key + '_u1'] = u1
        data[key + '_p1'] = p1
        data[key + '_ui'] = ui
    numpy.savez_compressed(path, **data)


def heat_fixture():
    path = _wanted('heat_ops.npz')
    if path is None:
        return
    mesh = fem.heater_box(5)
    Q = fem.FunctionSpace(mesh, 'Lagrange', 2)
    W = fem.VectorFunctionSpace(mesh, 'Lagrange', 2)
    x = W.layout.dof_coords
    conv = 2.0e-5 * numpy.concatenate([
        -(x[:, 1] - 0.1) * (1.0 + x[:, 0]), (x[:, 0] - 0.05) * (1.0 + x[:, 1]**2)
        ])
    Qo = orc.Space(mesh.points, mesh.cell_vertices, Q.layout.cell_dofs, 2, Q.N)
    Wo = orc.Space(mesh.points, mesh.cell_vertices, W.layout.cell_dofs, 2, W.N)
    kappa, rho, cp = 0.6, 998.0, 4182.0
    data = {'points': mesh.points, 'cells': mesh.cell_vertices,
            'q_cell_dofs': Q.layout.cell_dofs, 'conv': conv,
            'kappa': kappa, 'rho': rho, 'cp': cp}
    rng = numpy.random.RandomState(3)
    u0 = 293.0 + rng.standard_normal(Q.N)
    data['u0'] = u0
    bc_dofs = numpy.nonzero(Q.layout.dof_coords[:, 1] < 1e-12)[0]
    data['bc_dofs'] = bc_dofs
    data['bc_vals'] = numpy.full(len(bc_dofs), 320.0)
    for supg in (False, True):
        M, A, b = orc.heat_operators(Qo, Wo, conv, kappa, rho, cp, 0.0, supg)
        M = M.tocsr()
        A = A.tocsr()
        tag = 'supg' if supg else 'plain'
        data['M_dense_' + tag] = M.toarray()
        data['A_dense_' + tag] = A.toarray()
        data['solve_' + tag] = orc.heat_solve(
            M, A, 1.0, -0.01, M.dot(u0), bc_dofs, data['bc_vals'])
    Cc = numpy.stack([conv[W.layout.cell_dofs],
                      conv[W.N + W.layout.cell_dofs]], axis=1)
    pc = mesh.points[mesh.cell_vertices]
    data['tau'] = numpy.array([
        [orc.supg_tau(pc[k], Cc[k, :, v], kappa, 2) for v in range(3)]
        for k in range(mesh.num_cells())])
    numpy.savez_compressed(path, **data)


def large_fixture(name):
    import time
    import large_cases
    path = _wanted('ns_large_%s.npz' % name)
    if path is None:
        return
    args = large_cases.LARGE[name]
    case = large_cases.KarmanStepCase(**args)
    # (about 30 k stored values per fixture)
    stride = large_cases.STRIDE * (3 if case.num_dofs() > 2000000 else 1)
    data = {'fingerprint': case.fingerprint(), 'stride': stride,
            'dt': case.dt, 'num_dofs': case.num_dofs()}
    for k, v in case.args.items():
        data['arg_' + k] = v
    for method in ('backward euler', 'crank-nicolson'):
        info = {}
        t0 = time.time()
        u1, p1, ui = case.oracle_step(method, info=info)
        key = method.replace(' ', '_').replace('-', '_')
        data[key + '_oracle_seconds'] = time.time() - t0
        data[key + '_newton_history'] = numpy.array(info['newton_history'])
        for fname, field, ncomp in (('ui', ui, 2), ('p1', p1, 1), ('u1', u1, 2)):
            sample, l2, linf = large_cases.summary(field, ncomp, stride)
            data['%s_%s_sample' % (key, fname)] = sample
            data['%s_%s_l2' % (key, fname)] = l2
            data['%s_%s_linf' % (key, fname)] = linf
        print('  %s: %d DoF, %s, %.0f s, Newton %s' % (
            name, case.num_dofs(), method, data[key + '_oracle_seconds'],
            ' '.join('%.2e' % r for r in info['newton_history'])), flush=True)
    numpy.savez_compressed(path, **data)


def check_block_solver(name):
    '''--check-block NAME: recompute an EXISTING ns_large fixture with the
    oracle's block-wise linear solver (fem_oracle.solve_blockwise) and print
    how far it is from what the single sparse LU stored -- what pins the solver
    of the half-size fixture to the one of all the others.'''
    import large_cases
    gold = numpy.load(os.path.join(HERE, 'ns_large_%s.npz' % name))
    args = dict(large_cases.LARGE[name],
                linear=os.environ.get('ORACLE_LINEAR', 'block'))
    case = large_cases.KarmanStepCase(**args)
    assert numpy.allclose(case.fingerprint(), gold['fingerprint'], rtol=1e-9,
                          atol=1e-12)
    stride = int(gold['stride'])
    for method in ('backward euler', 'crank-nicolson'):
        info = {}
        u1, p1, ui = case.oracle_step(method, info=info)
        key = method.replace(' ', '_').replace('-', '_')
        worst = 0.0
        for fname, field, ncomp in (('ui', ui, 2), ('p1', p1, 1), ('u1', u1, 2)):
            sample, l2, linf = large_cases.summary(field, ncomp, stride)
            gs = gold['%s_%s_sample' % (key, fname)]
            err = numpy.linalg.norm(sample - gs) / numpy.linalg.norm(gs)
            dl2 = abs(l2 - gold['%s_%s_l2' % (key, fname)]).max() / l2.max()
            worst = max(worst, err, dl2)
            print('  %s %s %s: samples rel-l2 %.2e, norms %.2e' % (
                name, method, fname, err, dl2), flush=True)
        h0, h1 = numpy.array(info['newton_history']), gold[key + '_newton_history']
        print('  Newton histories: %s | stored %s' % (
            ' '.join('%.3e' % r for r in h0), ' '.join('%.3e' % r for r in h1)),
            flush=True)
        assert worst < 1e-10, worst


def _store(data, key, fields, stride):
    import large_cases
    for fname, field, ncomp in fields:
        sample, l2, linf = large_cases.summary(field, ncomp, stride)
        data['%s%s_sample' % (key, fname)] = sample
        data['%s%s_l2' % (key, fname)] = l2
        data['%s%s_linf' % (key, fname)] = linf


def boussinesq_fixture(name):
    '''bq_large_<name>.npz: one coupled sweep (BASELINE config 4).'''
    import time
    import large_cases
    import cases
    path = _wanted('bq_large_%s.npz' % name)
    if path is None:
        return
    case = large_cases.BoussinesqSweepCase(**large_cases.LARGE_BOUSSINESQ[name])
    data = {'fingerprint': case.fingerprint(), 'stride': large_cases.STRIDE,
            'num_dofs': case.num_dofs()}
    for k, v in case.args.items():
        data['arg_' + k] = v
    info = {}
    t0 = time.time()
    theta, u, p = case.oracle_sweep(info=info)
    data['oracle_seconds'] = time.time() - t0
    data['newton_history'] = numpy.array(info['newton_history'])
    p = cases.mean_free(p, case.pressure_mass())
    _store(data, '', (('theta', theta - 293.0, 1), ('u', u, 2), ('p', p, 1)),
           large_cases.STRIDE)
    print('  bq %s: %d DoF, %.0f s, Newton %s' % (
        name, case.num_dofs(), data['oracle_seconds'],
        ' '.join('%.2e' % r for r in info['newton_history'])), flush=True)
    numpy.savez_compressed(path, **data)


def stokes_fixture(name):
    '''stokes_large_<name>.npz: the Stokes bootstrap (BASELINE config 5's
    solver).'''
    import time
    import large_cases
    path = _wanted('stokes_large_%s.npz' % name)
    if path is None:
        return
    case = large_cases.StokesChannelCase(**large_cases.LARGE_STOKES[name])
    data = {'fingerprint': case.fingerprint(), 'stride': large_cases.STRIDE,
            'num_dofs': case.num_dofs()}
    for k, v in case.args.items():
        data['arg_' + k] = v
    t0 = time.time()
    u, p = case.oracle_solve()
    data['oracle_seconds'] = time.time() - t0
    _store(data, '', (('u', u, 2), ('p', p, 1)), large_cases.STRIDE)
    print('  stokes %s: %d DoF, %.0f s' % (name, case.num_dofs(),
                                          data['oracle_seconds']), flush=True)
    numpy.savez_compressed(path, **data)


if __name__ == '__main__':
    if '--check-block' in sys.argv[1:]:
        for name in sys.argv[sys.argv.index('--check-block') + 1:]:
            check_block_solver(name)
        sys.exit(0)
    if LARGE:
        import large_cases
        asked = [a for a in sys.argv[1:] if not a.startswith('--')]
        every = not asked
        for name in sorted(large_cases.LARGE):
            if every or name in asked:
                large_fixture(name)
        for name in sorted(large_cases.LARGE_BOUSSINESQ):
            if every or name in asked:
                boussinesq_fixture(name)
        for name in sorted(large_cases.LARGE_STOKES):
            if every or name in asked:
                stokes_fixture(name)
        sys.exit(0)
    # C1: the reference's plumbing configuration (tests/test_navier_stokes.py:
    # 403-410): UnitSquareMesh(8, 8, 'crossed'), P2-P1, guermond2
    ns_fixture('c1_unit_square', fem.UnitSquareMesh(8, 8, 'crossed'), 2, 'all',
               0.5, 1.0, 1.0, 21)
    ns_fixture('channel_p2', fem.karman_channel(24, 8), 2, 'channel', 0.02, 1.5,
               0.05, 22)
    ns_fixture('channel_p1', fem.karman_channel(24, 8), 1, 'channel', 0.02, 1.5,
               0.05, 23)
    # the body-fitted obstacle of the bench's workload (stretched triangles)
    ns_fixture('channel_p2_fitted', fem.karman_channel(30, 10, fitted=True), 2,
               'channel', 0.02, 1.5, 0.05, 24)
    heat_fixture()
    print('fixtures written to', HERE)
