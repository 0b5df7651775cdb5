# -*- coding: utf-8 -*-
'''
CPU tests of the host logic (meshes, dof maps, sparsity pattern, contribution
maps, Dirichlet dof search, coefficients) and of the C-ABI library's symbol
table.  No compute call is made: that needs a GPU.
'''
import os
import sys
import re

import numpy
import pytest
import scipy.sparse as sp

from flow_amd import fem, _hip, message
from flow_amd.fem import reference
from flow_amd.fem.bcs import collect
from flow_amd.fem.space import csr_stream_rowblocks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _meshes():
    return [
        fem.UnitSquareMesh(5, 4, 'crossed'),
        fem.UnitSquareMesh(3, 3, 'left/right'),
        fem.RectangleMesh(fem.Point(-1, -1), fem.Point(1, 1), 4, 6, 'right'),
        fem.karman_channel(30, 8),
        fem.heater_box(5),
        ]


def test_mesh_topology():
    m = fem.UnitSquareMesh(5, 4, 'crossed')
    assert m.num_vertices() == 6 * 5 + 20 and m.num_cells() == 80
    # Euler: V - E + F = 1 for a simply connected planar mesh
    assert m.num_vertices() - m.num_edges() + m.num_cells() == 1
    assert len(m.bfacets) == 2 * (5 + 4)
    assert m.cell_areas().sum() == pytest.approx(1.0)
    k = fem.karman_channel(30, 8)
    # one hole
    assert k.num_vertices() - k.num_edges() + k.num_cells() == 0
    for mesh in _meshes():
        # every boundary facet is the stated local facet of its owner cell
        for e, c, lf in zip(mesh.bfacets, mesh.bfacet_cell, mesh.bfacet_local):
            assert mesh.cell_edges[c, lf] == e
            verts = set(mesh.cell_vertices[c]) - {mesh.cell_vertices[c, lf]}
            assert verts == set(mesh.edges[e])
        mask = mesh.cell_bfacet_mask()
        assert numpy.count_nonzero(mask) == len(set(mesh.bfacet_cell))
        assert mesh.hmin() <= mesh.hmax()


def test_karman_geometry_constants():
    # tests/test_karman_vortex_street.py:18-23, 35-38 of the reference
    m = fem.karman_channel(120, 28)
    p = m.points
    assert p[:, 0].min() == 0.0 and p[:, 0].max() == pytest.approx(0.6)
    assert p[:, 1].min() == pytest.approx(-0.07)
    assert p[:, 1].max() == pytest.approx(0.07)
    area = m.cell_areas().sum()
    assert area == pytest.approx(0.6 * 0.14 - numpy.pi * 0.02**2, rel=2e-3)
    # x-major numbering: x never decreases with the vertex id
    assert (numpy.diff(p[:, 0]) >= -1e-15).all()


@pytest.mark.parametrize('deg', [1, 2])
def test_dofmap_and_pattern(deg):
    rng = numpy.random.RandomState(0)
    for mesh in _meshes():
        V = fem.FunctionSpace(mesh, 'CG', deg)
        lay = V.layout
        nloc = reference.nloc(deg)
        assert lay.cell_dofs.shape == (mesh.num_cells(), nloc)
        assert sorted(set(lay.cell_dofs.ravel())) == list(range(lay.N))
        # dof coordinates agree with the per-cell lattice
        X = fem.cell_lattice_points(mesh, deg)
        assert numpy.allclose(lay.dof_coords[lay.cell_dofs], X, atol=1e-14)
        # contribution maps reproduce a COO assembly of random local tensors
        nc = mesh.num_cells()
        Ke = rng.standard_normal((nc, nloc, nloc))
        rows = numpy.repeat(lay.cell_dofs[:, :, None], nloc, axis=2)
        cols = numpy.repeat(lay.cell_dofs[:, None, :], nloc, axis=1)
        ref = sp.coo_matrix((Ke.ravel(), (rows.ravel(), cols.ravel())),
                            shape=(lay.N, lay.N)).tocsr()
        ref.sort_indices()
        rowptr, colidx = lay.pattern('rowptr'), lay.pattern('cols')
        assert numpy.array_equal(rowptr, ref.indptr)
        assert numpy.array_equal(colidx, ref.indices)
        scratch = Ke.transpose(1, 2, 0).reshape(-1)      # [ij][cell]
        cptr, csrc = lay.pattern('cptr'), lay.pattern('csrc')
        vals = numpy.add.reduceat(scratch[csrc], cptr[:-1])
        assert numpy.allclose(vals, ref.data, rtol=1e-13, atol=1e-13)
        assert numpy.array_equal(
            colidx[lay.pattern('diag_idx')], numpy.arange(lay.N))
        Fe = rng.standard_normal((nc, nloc))
        vref = numpy.zeros(lay.N)
        numpy.add.at(vref, lay.cell_dofs.ravel(), Fe.ravel())
        vptr, vsrc = lay.vmap('vptr'), lay.vmap('vsrc')
        vec = numpy.add.reduceat(Fe.T.reshape(-1)[vsrc], vptr[:-1])
        assert numpy.allclose(vec, vref, rtol=1e-13, atol=1e-13)


def test_rowblocks():
    rng = numpy.random.RandomState(1)
    lens = rng.randint(1, 40, size=5000)
    rowptr = numpy.concatenate([[0], numpy.cumsum(lens)])
    rb = csr_stream_rowblocks(rowptr)
    assert rb[0] == 0 and rb[-1] == 5000 and (numpy.diff(rb) > 0).all()
    assert (numpy.diff(rb) <= _hip.SPMV_ROWS_PER_BLOCK).all()
    assert (numpy.diff(rowptr[rb]) <= _hip.spmv_tile_nnz(0)).all()
    with pytest.raises(AssertionError):
        csr_stream_rowblocks(numpy.array([0, 5000]))


def test_dirichlet_search_and_override():
    mesh = fem.UnitSquareMesh(4, 4, 'crossed')
    W = fem.VectorFunctionSpace(mesh, 'CG', 2)
    lay = W.layout
    on_b = numpy.zeros(lay.N, dtype=bool)
    c = lay.dof_coords
    on_b[(abs(c[:, 0]) < 1e-12) | (abs(c[:, 0] - 1) < 1e-12)
         | (abs(c[:, 1]) < 1e-12) | (abs(c[:, 1] - 1) < 1e-12)] = True
    d, v = collect([fem.DirichletBC(W, (1.0, 2.0), 'on_boundary')], W.size())
    assert numpy.array_equal(d[:len(d) // 2], numpy.nonzero(on_b)[0])
    assert numpy.array_equal(d[len(d) // 2:], numpy.nonzero(on_b)[0] + lay.N)
    assert set(v[:len(d) // 2]) == {1.0} and set(v[len(d) // 2:]) == {2.0}

    class Left(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary and x[0] < 1e-12      # scalar-style condition

    # component-wise condition, later condition overrides on shared dofs
    expr = fem.Expression('3.0 + x[1]', degree=1)
    d2, v2 = collect([
        fem.DirichletBC(W, (1.0, 2.0), 'on_boundary'),
        fem.DirichletBC(W.sub(0), expr, Left()),
        ], W.size())
    assert numpy.array_equal(d2, d)
    left = numpy.nonzero(abs(c[:, 0]) < 1e-12)[0]
    pos = numpy.searchsorted(d2, left)
    assert numpy.allclose(v2[pos], 3.0 + c[left, 1])
    # corner dofs belong to the left facets too, the y-component is untouched
    assert set(v2[numpy.searchsorted(d2, left + lay.N)]) == {2.0}
    # chained comparison forces the point-by-point fallback
    class Inner(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary and 0.2 < x[1] < 0.8 and x[0] < 1e-12
    d3, _ = collect([fem.DirichletBC(W.sub(1), 0.0, Inner())], W.size())
    assert len(d3) > 0 and (d3 >= lay.N).all()
    assert (c[d3 - lay.N, 1] > 0.2 - 1e-12).all()


def test_expression_and_coefficients():
    e = fem.Expression(('sin(x[0] + t)*pow(x[1], 2)', 'cos(pi*x[0])'),
                       degree=3, t=0.5)
    x = numpy.array([[0.1, 0.7], [0.2, 0.3]])
    assert numpy.allclose(e.eval(x)[0], numpy.sin(x[0] + 0.5) * x[1]**2)
    e.t = 1.5
    assert numpy.allclose(e.eval(x)[0], numpy.sin(x[0] + 1.5) * x[1]**2)
    assert numpy.allclose(e.eval(x)[1], numpy.cos(numpy.pi * x[0]))
    assert e.value_dim() == 2
    c = fem.Constant(2.5)
    assert c.values()[0] == 2.5 and fem.scalar_value(c) == 2.5
    assert fem.scalar_value(3.0) == 3.0
    # reference matrices: rows of G sum to the integral of the test basis
    for k in range(6):
        G = reference.source_matrix(k, 2)
        assert numpy.allclose(G.sum(axis=0), [0, 0, 0, 1 / 6., 1 / 6., 1 / 6.],
                              atol=1e-14)
        assert numpy.allclose(reference.tabulate(k, reference.lattice(k)),
                              numpy.eye(reference.nloc(k)), atol=1e-10)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'flow_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|const char\*)\s+(flow_\w+)\s*\(',
                              header, flags=re.M))
    assert declared, 'no declarations parsed'
    lib = _hip.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared - {'flow_last_error'} == set(_hip.SYMBOLS), \
        declared ^ set(_hip.SYMBOLS)
    assert lib.flow_abi_version() == _hip.ABI_VERSION == 30
    assert _hip.SPMV_ROWS_PER_BLOCK == int(
        re.search(r'FLOW_SPMV_ROWS_PER_BLOCK (\d+)', header).group(1))
    assert _hip.SPMV_NNZ_PER_BLOCK == int(
        re.search(r'FLOW_SPMV_NNZ_PER_BLOCK (\d+)', header).group(1))
    assert _hip.REDUCE_WORK == int(
        re.search(r'FLOW_REDUCE_WORK (\d+)', header).group(1))


def test_graph_replay_is_off_unless_asked_for():
    '''The HIP-graph replay of the iteration bodies (csrc/graph_replay.hip) is an
    option: nothing is kept, captured or replayed by default; the switches and
    the counters answer without a GPU.'''
    import subprocess
    code = ('import sys; sys.path.insert(0, %r); from flow_amd import _hip; '
            's = _hip.graph_stats(); assert s["graphs"] == s["captures"] == 0, s; '
            '_hip.graph_mode(1, sites=3); _hip.graph_mode(2, 1000); '
            '_hip.graph_mode(0); print(_hip.graph_stats()["replays"])' % ROOT)
    env = dict(os.environ)
    env.pop('FLOW_AMD_GRAPHS', None)
    out = subprocess.run([sys.executable, '-c', code], env=env,
                         capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == '0', out.stderr
    lib = _hip.load_library()
    assert lib.flow_graph_mode(3, -1) != 0          # (modes are 0, 1, 2)


def test_xcd_tile_mapping_is_a_permutation():
    '''The XCD-aware workgroup -> tile mapping of the CSR-stream kernels (host
    copy of the device function): every tile exactly once, for any grid size;
    blocks b, b+8, ... (one XCD) get runs of consecutive tiles.'''
    lib = _hip.load_library()
    for n in list(range(1, 530)) + [767, 1024, 4286, 24685]:
        tiles = [lib.flow_xcd_tile_host(b, n) for b in range(n)]
        assert sorted(tiles) == list(range(n)), n
    n = 4286
    run = 64            # kXcdRun, flow_amd/csrc/common.h
    same_xcd = [lib.flow_xcd_tile_host(b, n) for b in range(3, 3 + 8 * run, 8)]
    assert same_xcd == list(range(same_xcd[0], same_xcd[0] + run))


def test_boundary_data_cache_follows_the_data():
    '''`collect` reuses the merged (dofs, values) arrays of a list of
    conditions only while their data are unchanged (time loops pass the same
    conditions every step): a parameter of an Expression or the value of a
    Constant changing must show up in the next call.'''
    from flow_amd.fem.bcs import collect
    mesh = fem.UnitSquareMesh(4, 4)
    V = fem.FunctionSpace(mesh, 'CG', 2)
    ramp = fem.Expression('t * (1.0 + x[0])', degree=1, t=1.0)
    const = fem.Constant(2.0)

    class Left(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[0] < 1e-12)

    class Right(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[0] > 1.0 - 1e-12)
    bcs = [fem.DirichletBC(V, ramp, Left()), fem.DirichletBC(V, const, Right())]
    d0, v0 = collect(bcs, V.N)
    d1, v1 = collect(bcs, V.N)
    assert d1 is d0 and v1 is v0                      # unchanged: same arrays
    ramp.t = 3.0
    d2, v2 = collect(bcs, V.N)
    assert numpy.array_equal(d2, d0) and not numpy.array_equal(v2, v0)
    left = V.layout.dof_coords[d2, 0] < 1e-12
    assert numpy.allclose(v2[left], 3.0) and numpy.allclose(v2[~left], 2.0)
    const.assign(5.0)
    d3, v3 = collect(bcs, V.N)
    assert numpy.allclose(v3[~left], 5.0) and numpy.allclose(v3[left], 3.0)
    # a Python callable may depend on anything: never cached
    state = {'a': 1.0}
    call = fem.Expression(lambda x: state['a'] + 0.0 * x[0], degree=1)
    bc = [fem.DirichletBC(V, call, Left())]
    _, w0 = collect(bc, V.N)
    state['a'] = 4.0
    _, w1 = collect(bc, V.N)
    assert numpy.allclose(w0, 1.0) and numpy.allclose(w1, 4.0)


def test_start_chooser_switches_on_iteration_counts():
    '''ops.StartChooser: keeps the extrapolation order that needs fewer
    iterations, tries the other one every `period`-th call.'''
    from flow_amd.fem.ops import StartChooser
    ch = StartChooser(period=4)
    cost = {1: 10, 2: 6}                 # quadratic is better
    used = []
    for _ in range(12):
        m = ch.pick()
        used.append(m)
        ch.report(m, cost[m])
    assert used.count(2) >= 9 and ch.mode == 2
    cost = {1: 6, 2: 12}                 # now solver noise: linear is better
    used = []
    for _ in range(12):
        m = ch.pick()
        used.append(m)
        ch.report(m, cost[m])
    assert ch.mode == 1 and used.count(1) >= 7


def test_no_cpu_fallback():
    from flow_amd import device
    if device.on_gpu():
        pytest.skip('GPU present')
    with pytest.raises(_hip.HipError):
        _hip.lib()
    mesh = fem.UnitSquareMesh(2, 2)
    V = fem.FunctionSpace(mesh, 'CG', 1)
    with pytest.raises(_hip.HipError):
        fem.project(fem.Constant(1.0), V)


def test_body_fitted_hole():
    '''fem.karman_channel(fitted=True): the hole's boundary vertices lie on the
    circle, no cell is inverted or badly shaped, vertices keep their x-major
    order, cells stay sorted by lowest vertex (what the strip decomposition and
    the cell kernels assume), the outer boundary does not move.'''
    import numpy
    from flow_amd import fem
    for nx, ny in ((60, 14), (240, 56)):
        m = fem.karman_channel(nx, ny, fitted=True)
        pts, cells = m.points, m.cell_vertices
        e = m.edges[m.bfacets]
        v = numpy.unique(e)
        p = pts[v]
        inner = (p[:, 0] > 1e-9) & (p[:, 0] < 0.6 - 1e-9) & \
            (p[:, 1] > -0.07 + 1e-9) & (p[:, 1] < 0.07 - 1e-9)
        hx, hy = 0.6 / nx, 0.14 / ny
        c = numpy.array([round(0.1 / hx) * hx,
                         -0.07 + round((0.01 + 0.07) / hy) * hy])
        r = numpy.hypot(*(p[inner] - c).T)
        assert inner.sum() >= 16 and abs(r - 0.02).max() < 1e-12
        outer = p[~inner]
        on_box = (abs(outer[:, 0]) < 1e-12) | (abs(outer[:, 0] - 0.6) < 1e-12) \
            | (abs(outer[:, 1] + 0.07) < 1e-12) | (abs(outer[:, 1] - 0.07) < 1e-12)
        assert on_box.all()
        q = pts[cells]
        a, b, d = q[:, 1] - q[:, 0], q[:, 2] - q[:, 0], q[:, 2] - q[:, 1]

        def ang(u, w):
            cs = (u * w).sum(1) / numpy.linalg.norm(u, axis=1) \
                / numpy.linalg.norm(w, axis=1)
            return numpy.degrees(numpy.arccos(numpy.clip(cs, -1, 1)))
        angs = numpy.stack([ang(a, b), ang(-a, d), ang(-b, -d)], 1)
        assert angs.min() > 20.0 and angs.max() < 110.0
        area = m.cell_areas() / (0.5 * hx * hy)
        assert 0.6 < area.min() and area.max() < 1.5
        assert (numpy.diff(cells.min(axis=1)) >= 0).all()
        # the fluid area: the box minus a polygon inscribed in the circle (its
        # vertices are not equally spaced: a little less than the regular one)
        nb = inner.sum()
        poly = 0.5 * nb * 0.02**2 * numpy.sin(2 * numpy.pi / nb)
        hole = 0.6 * 0.14 - m.cell_areas().sum()
        assert 0.99 * poly < hole <= poly * (1.0 + 1e-12)


def test_message(capsys):
    message.set_log_active(True)
    try:
        with message.Message('outer'):
            message.info('inner')
    finally:
        message.set_log_active(False)
    out = capsys.readouterr().out.splitlines()
    assert out == ['outer', '  inner']


def test_newton_preconditioner_ageing_rule():
    """newton_preconditioner.age (host logic): when the lagged p-multigrid
    cycle is rebuilt -- solves that stay long (smoothed count against the best
    since the rebuild, per Newton iteration), or pmg_refresh solves where they
    average pmg_refresh_min applications; never on short solves by the clock,
    never because of one long solve."""
    from types import SimpleNamespace
    from flow_amd.navier_stokes.newton_preconditioner import age
    npar = {'check_every': 1, 'pmg_refresh': 50, 'pmg_min_solves': 3,
            'pmg_refresh_min': 6.0}

    def fresh(first=9):
        pre = SimpleNamespace(stale=False)
        age(pre, 'pmg', True, 5, first, npar, it=0)
        assert not pre.stale and pre.uses == 0
        return pre

    # the second Newton iteration's shorter solves have their own yardstick,
    # counts that fluctuate by one are no ageing
    pre = fresh()
    for k in range(24):
        age(pre, 'pmg', False, 3, 6, npar, it=1)
        age(pre, 'pmg', False, 5, 9 + (k % 2), npar, it=0)
        assert not pre.stale, k
    # one long solve (a poor start vector) does not rebuild ...
    pre = fresh()
    for k in range(8):
        age(pre, 'pmg', False, 5, 13 if k == 4 else 9, npar, it=0)
        assert not pre.stale, k
    # ... solves that STAY two applications longer do, within a few solves
    pre = fresh(4)
    for k in range(6):
        age(pre, 'pmg', False, 2, 4, npar, it=0)
    assert not pre.stale
    hit = None
    for k in range(8):
        age(pre, 'pmg', False, 3, 6, npar, it=0)
        if pre.stale:
            hit = k
            break
    assert hit is not None and 1 <= hit <= 5, hit
    # the yardstick is the BEST solve since the rebuild, not the first (which
    # may have had no start vector)
    pre = fresh(12)
    for k in range(5):
        age(pre, 'pmg', False, 2, 4, npar, it=0)
    for k in range(8):
        age(pre, 'pmg', False, 3, 7, npar, it=0)
    assert pre.stale
    # long solves: refreshed after pmg_refresh of them whatever the counts do
    pre = fresh()
    for k in range(49):
        age(pre, 'pmg', False, 5, 9, npar, it=0)
    assert not pre.stale
    age(pre, 'pmg', False, 5, 9, npar, it=0)
    assert pre.stale
    # short solves (the early plateau: 4-5 applications): never by the clock
    pre = fresh(4)
    for k in range(300):
        age(pre, 'pmg', False, 2, 4 + (k % 2), npar, it=0)
    assert not pre.stale
    # the ILU(0) rule is the old one: twice the fresh iterations
    pre = SimpleNamespace(stale=False)
    age(pre, 'ilu0', True, 8, 16, npar)
    age(pre, 'ilu0', False, 16, 32, npar)
    assert not pre.stale
    age(pre, 'ilu0', False, 17, 34, npar)
    assert pre.stale
