# -*- coding: utf-8 -*-
'''
Meshes that come from a file in the numbering of whoever wrote them -- the
reference's real workload is a graded gmsh mesh read back from its cache file
(tests/test_karman_vortex_street.py:26-53).  The fast path leans on the
generators' numbering (vertices along the channel, cells by lowest vertex):
`Mesh.reordered` gives a loaded mesh the same, `io.read_mesh` applies it.

CPU: the graded Delaunay channel (fem.karman_channel_graded: own generator,
gmsh is not available), MSH round trip, what the reordering buys (bandwidth,
16-bit column offsets, strips).  GPU (-m gpu): one step on the loaded, graded
mesh against the oracle.
'''
import os

import numpy
import pytest

from flow_amd import fem, parallel
from flow_amd.fem import io
from flow_amd.fem.space import scalar_layout, csr_stream_rowblocks


def _min_angle(mesh):
    p = mesh.points[mesh.cell_vertices]

    def ang(a, b, c):
        u, v = b - a, c - a
        cosv = (u * v).sum(1) / numpy.linalg.norm(u, axis=1) \
            / numpy.linalg.norm(v, axis=1)
        return numpy.degrees(numpy.arccos(numpy.clip(cosv, -1.0, 1.0)))
    A = numpy.stack([ang(p[:, 0], p[:, 1], p[:, 2]), ang(p[:, 1], p[:, 2], p[:, 0]),
                     ang(p[:, 2], p[:, 0], p[:, 1])], axis=1)
    return A.min()


def test_graded_channel_mesh():
    m = fem.karman_channel_graded(2.5e-3)
    m2 = fem.karman_channel_graded(2.5e-3)
    assert numpy.array_equal(m.points, m2.points)            # deterministic
    assert numpy.array_equal(m.cell_vertices, m2.cell_vertices)
    assert _min_angle(m) > 10.0
    # graded: cells at the cylinder are a few times smaller than far from it
    e = m._edge_lengths().max(axis=1)
    cen = m.points[m.cell_vertices].mean(axis=1)
    near = numpy.hypot(cen[:, 0] - 0.1, cen[:, 1] - 0.01) < 0.03
    far = cen[:, 0] > 0.4
    assert numpy.median(e[far]) > 2.5 * numpy.median(e[near])
    # the domain: channel minus the 79-gon on the circle
    area = m.cell_areas().sum()
    assert abs(area - (0.6 * 0.14 - numpy.pi * 0.02**2)) < 1e-5
    # counter-clockwise cells, every vertex used, one hole
    p = m.points[m.cell_vertices]
    det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) \
        - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1])
    assert (det > 0).all()
    assert len(numpy.unique(m.cell_vertices)) == m.num_vertices()
    # Euler: V - E + F = 1 - holes  (F without the outer face)
    assert m.num_vertices() - m.num_edges() + m.num_cells() == 0


def test_reordering_a_loaded_mesh(tmp_path):
    m = fem.karman_channel_graded(1.0e-3)          # 11 k vertices, 0.1 M DoF
    path = os.path.join(str(tmp_path), 'karman.msh')
    io.write_msh(path, m, binary=True)
    raw = io.read_mesh(path, reorder=False)
    assert numpy.array_equal(raw.points, m.points)
    assert numpy.array_equal(raw.cell_vertices, m.cell_vertices)
    r = fem.Mesh(path)                              # reordered, as a driver gets it
    assert r.vertex_origin is not None and r.cell_origin is not None
    assert numpy.array_equal(r.points, m.points[r.vertex_origin])
    # data indexed by the FILE's vertices and cells round-trip through the maps
    vdata = numpy.cos(17.0 * m.points[:, 0]) + m.points[:, 1]       # per vertex
    cdata = m.points[m.cell_vertices].mean(axis=1)[:, 0]            # per cell
    assert numpy.array_equal(vdata[r.vertex_origin],
                             numpy.cos(17.0 * r.points[:, 0]) + r.points[:, 1])
    assert numpy.allclose(cdata[r.cell_origin],
                          r.points[r.cell_vertices].mean(axis=1)[:, 0],
                          rtol=0, atol=1e-15)
    assert sorted(r.cell_origin) == list(range(m.num_cells()))
    # the opt-out: the file's numbering as it is (what dolfin's Mesh(path) gives)
    keep = fem.Mesh(path, reorder=False)
    assert keep.vertex_origin is None and keep.cell_origin is None
    assert numpy.array_equal(keep.points, m.points)
    assert numpy.array_equal(keep.cell_vertices, m.cell_vertices)
    # ... and a second reordering composes the maps (still the file's ids)
    rr2 = keep.reordered().reordered()
    assert numpy.array_equal(rr2.vertex_origin, r.vertex_origin)
    assert numpy.array_equal(rr2.cell_origin, r.cell_origin)
    # the same triangles (as sets of file vertex ids)
    def key(cells):
        c = numpy.sort(cells.astype(numpy.int64), axis=1)
        return numpy.sort(c[:, 0] * 10**12 + c[:, 1] * 10**6 + c[:, 2])
    assert numpy.array_equal(key(r.vertex_origin[r.cell_vertices]),
                             key(m.cell_vertices))
    # vertices along the channel, cells by lowest vertex
    assert (numpy.diff(r.points[:, 0]) >= 0.0).all()
    assert (numpy.diff(r.cell_vertices.min(axis=1)) >= 0).all()
    # what it buys: the file's numbering couples vertices across the whole
    # mesh (boundary curves first), the reordered one within one cross-section
    assert m.bandwidth() > 0.9 * m.num_vertices()
    assert r.bandwidth() < 0.04 * m.num_vertices()
    spans = []
    for mesh in (m, r):
        lay = scalar_layout(mesh, 2)
        rowptr = lay.pattern('rowptr').astype(numpy.int64)
        cols = lay.pattern('cols').astype(numpy.int64)
        rb = csr_stream_rowblocks(rowptr, nnz_per_block=2044)
        spans.append(max(int(cols[rowptr[a]:rowptr[b]].max()
                             - cols[rowptr[a]:rowptr[b]].min())
                         for a, b in zip(rb[:-1], rb[1:])))
    # the columns a row block (2044 nonzeros) of the P2 pattern reaches: the
    # whole matrix in the file's numbering, a few cross-sections after the
    # reordering -- the 16-bit column offsets of the packed fp16 streams need
    # < 65536 (at 1 M DoF: 4 x 108 k rows in the file's numbering)
    assert spans[0] > 0.9 * scalar_layout(m, 2).N
    assert spans[1] < 0.06 * scalar_layout(r, 2).N and spans[1] < 65536
    # strips of the reordered mesh: eight ranks, deep halos included
    st = parallel.Strips(r, 8)
    for degree, depth in ((1, 2), (2, 6)):
        st.deep_blocks(scalar_layout(r, degree), depth)
    with pytest.raises(parallel.StripsTooThin):
        parallel.Strips(m, 8).blocks(scalar_layout(m, 2))
    # already ordered: nothing moves
    rr = r.reordered()
    assert numpy.array_equal(rr.points, r.points)
    assert numpy.array_equal(rr.cell_vertices, r.cell_vertices)


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['backward euler', 'crank-nicolson'])
def test_step_on_a_loaded_graded_mesh_against_the_oracle(hip, tmp_path, method):
    '''The Karman step of tests/large_cases.py on the graded Delaunay channel
    (17 k DoF), written to MSH and read back like the reference's driver reads
    gmsh's file -- viscosity scaled to a cell Peclet number ~ 1 on the coarse
    cells --, against the oracle on the same (reordered) mesh.'''
    import cases
    import large_cases
    import flow_amd.navier_stokes as navsto
    path = os.path.join(str(tmp_path), 'karman_graded.msh')
    io.write_msh(path, fem.karman_channel_graded(2.5e-3))
    mesh = fem.Mesh(path)
    case = large_cases.KarmanStepCase(mesh=mesh, mu=0.02)
    info = {}
    u1o, p1o, uio = case.oracle_step(method, info=info)
    u1, p1, ui = case.product_step(method)
    assert cases.rel_l2(ui, uio) < 1e-7
    assert cases.rel_l2(p1, p1o) < 1e-7
    assert cases.rel_l2(u1, u1o) < 1e-7
    assert len(navsto.last_step_info['newton_residuals']) == \
        len(info['newton_history'])
    # the mass solver's defect correction ran as itself (a silent fallback to
    # Jacobi-CG on a graded mesh would be a performance regression nobody sees)
    assert 'cg' not in navsto.last_step_info['correction'].method, \
        navsto.last_step_info['correction']


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['backward euler', 'crank-nicolson'])
def test_step_on_the_references_own_mesh_against_the_oracle(hip, method):
    """What the reference's driver really meshes (tests/
    test_karman_vortex_street.py:35-45: ONE characteristic length for the
    circle and the rectangle, `lcar = 5e-3` at the `__main__` setting): a
    quasi-uniform unstructured channel of 3.9 k vertices, 34.6 k DoF, at the
    driver's own viscosity (:167; cell Peclet ~20) -- the Karman step against
    the oracle on the same mesh."""
    import cases
    import large_cases
    import flow_amd.navier_stokes as navsto
    mesh = fem.karman_channel_graded(5.0e-3, lcar_far=5.0e-3).reordered()
    assert 30000 < 9 * mesh.num_vertices() < 40000
    e = mesh._edge_lengths().max(axis=1)
    assert e.max() < 2.5 * numpy.median(e)                   # not graded
    case = large_cases.KarmanStepCase(mesh=mesh, mu=0.002)
    info = {}
    u1o, p1o, uio = case.oracle_step(method, info=info)
    u1, p1, ui = case.product_step(method)
    assert cases.rel_l2(ui, uio) < 1e-7
    assert cases.rel_l2(p1, p1o) < 1e-7
    assert cases.rel_l2(u1, u1o) < 1e-7
    assert len(navsto.last_step_info['newton_residuals']) == \
        len(info['newton_history'])
    assert len(info['newton_history']) >= 3
