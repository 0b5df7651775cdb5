# -*- coding: utf-8 -*-
'''
Mesh readers/writers (gmsh MSH 2.2 ASCII and binary, DOLFIN XML) and the XDMF time-series
writer the reference's drivers use (tests/test_karman_vortex_street.py:29-53,
214-227).  CPU only.
'''
import numpy
import pytest

from flow_amd import fem
from flow_amd.fem import io


@pytest.mark.parametrize('ext', ['msh', 'msh-binary', 'xml'])
def test_mesh_round_trip(tmp_path, ext):
    mesh = fem.karman_channel(20, 6)
    path = str(tmp_path / ('mesh.' + ext.split('-')[0]))
    if ext == 'xml':
        io.write_dolfin_xml(path, mesh)
    else:
        io.write_msh(path, mesh, binary=ext.endswith('binary'))
    back = fem.Mesh(path)
    assert numpy.array_equal(back.points, mesh.points)
    assert numpy.array_equal(back.cell_vertices, mesh.cell_vertices)
    assert back.num_edges() == mesh.num_edges()


def test_msh_with_other_element_types(tmp_path):
    # a gmsh file as pygmsh writes it: points, lines and triangles mixed,
    # 1-based node ids, unused nodes (the circle centre)
    text = '''$MeshFormat
2.2 0 8
$EndMeshFormat
$Nodes
5
1 0 0 0
2 1 0 0
3 0 1 0
4 1 1 0
5 0.5 0.5 0
$EndNodes
$Elements
4
1 15 2 0 1 1
2 1 2 0 1 1 2
3 2 2 0 6 1 2 3
4 2 2 0 6 2 4 3
$EndElements
'''
    path = tmp_path / 'g.msh'
    path.write_text(text)
    mesh = io.read_mesh(str(path))
    assert mesh.num_vertices() == 4 and mesh.num_cells() == 2
    assert mesh.cell_areas().sum() == pytest.approx(1.0)


def test_binary_msh_with_other_element_types(tmp_path):
    '''The binary flavour of the same file: element blocks of points, lines and
    triangles (type, count, number of tags | id, tags, nodes), node records
    (int32 id, 3 doubles), both byte orders.'''
    nodes = [(1, 0, 0), (2, 1, 0), (3, 0, 1), (4, 1, 1), (5, 0.5, 0.5)]
    blocks = [(15, 2, [[1, 0, 1, 1]]), (1, 2, [[2, 0, 1, 1, 2]]),
              (2, 2, [[3, 0, 6, 1, 2, 3], [4, 0, 6, 2, 4, 3]])]
    for order in ('<', '>'):
        raw = b'$MeshFormat\n2.2 1 8\n' + numpy.array([1], order + 'i4').tobytes() \
            + b'\n$EndMeshFormat\n$Nodes\n5\n'
        for k, x, y in nodes:
            raw += numpy.array([k], order + 'i4').tobytes() \
                + numpy.array([x, y, 0.0], order + 'f8').tobytes()
        raw += b'\n$EndNodes\n$Elements\n4\n'
        for etype, ntags, rows in blocks:
            raw += numpy.array([etype, len(rows), ntags], order + 'i4').tobytes()
            raw += numpy.array(rows, order + 'i4').tobytes()
        raw += b'\n$EndElements\n'
        path = tmp_path / ('g%s.msh' % ('le' if order == '<' else 'be'))
        path.write_bytes(raw)
        mesh = io.read_mesh(str(path))
        assert mesh.num_vertices() == 4 and mesh.num_cells() == 2
        assert mesh.cell_areas().sum() == pytest.approx(1.0)


@pytest.mark.parametrize('heavy', ['xml', 'hdf5'])
@pytest.mark.parametrize('rewrite_mesh', [False, True])
def test_xdmf_time_series(tmp_path, heavy, rewrite_mesh):
    """The reference driver's output loop (tests/test_karman_vortex_street.py:
    214-227) with inline data and with the heavy data in an HDF5 file beside
    the XML, laid out as dolfin does."""
    import os
    import shutil
    import subprocess
    from flow_amd.fem import hdf5
    if heavy == 'hdf5' and not hdf5.available():
        pytest.skip('no HDF5 library in this environment')
    mesh = fem.UnitSquareMesh(3, 3)
    W = fem.VectorFunctionSpace(mesh, 'CG', 2)
    P = fem.FunctionSpace(mesh, 'CG', 1)
    u = fem.Function(W)
    p = fem.Function(P)
    u.rename('velocity', 'velocity')
    p.rename('pressure', 'pressure')
    path = str(tmp_path / 'out.xdmf')
    with io.XDMFFile(io.mpi_comm_world(), path) as xf:
        xf.parameters['flush_output'] = True
        xf.parameters['rewrite_function_mesh'] = rewrite_mesh
        xf.parameters['heavy_data'] = heavy
        for k, t in enumerate((0.0, 0.5)):
            x = W.layout.dof_coords
            u.set_array(numpy.concatenate([x[:, 0] + t, x[:, 1] * (k + 1)]))
            p.set_array(mesh.points[:, 0] * t)
            xf.write(u, t)
            xf.write(p, t)
            # flush_output: the files are complete after every write
            assert len(io.read_xdmf_series(path)) == k + 1
    series = io.read_xdmf_series(path)
    assert [s[0] for s in series] == [0.0, 0.5]
    vel = series[1][1]['velocity']
    assert vel.shape == (mesh.num_vertices(), 3)
    assert numpy.allclose(vel[:, 0], mesh.points[:, 0] + 0.5)
    assert numpy.allclose(vel[:, 1], mesh.points[:, 1] * 2)
    assert numpy.allclose(series[1][1]['pressure'], mesh.points[:, 0] * 0.5)
    for _t, fields in series:
        assert numpy.array_equal(fields['_topology'], mesh.cell_vertices)
        assert numpy.array_equal(fields['_geometry'], mesh.points)
    h5 = str(tmp_path / 'out.h5')
    if heavy == 'xml':
        assert not os.path.exists(h5)
        return
    with hdf5.File(h5) as f:
        assert f.exists('/Mesh/0/mesh/geometry')
        assert f.exists('/Mesh/0/mesh/topology')
        assert f.exists('/Mesh/1/mesh/topology') == rewrite_mesh
        assert f.exists('/VisualisationVector/3')
        assert not f.exists('/VisualisationVector/4')
        assert f.read('/Mesh/0/mesh/topology').dtype == numpy.int64
    # an independent reader, where the image has one
    tool = shutil.which('h5dump') or '/opt/conda/bin/h5dump'
    if os.path.exists(tool):
        head = subprocess.run([tool, '-H', h5], capture_output=True, text=True,
                              check=True).stdout
        assert 'DATASET "geometry"' in head and 'H5T_IEEE_F64LE' in head
        assert 'DATASET "topology"' in head and 'H5T_STD_I64LE' in head
        assert head.count('DATASET') == (8 if rewrite_mesh else 6)


def test_hdf5_binding_round_trip(tmp_path):
    from flow_amd.fem import hdf5
    if not hdf5.available():
        pytest.skip('no HDF5 library in this environment')
    rng = numpy.random.RandomState(0)
    arrays = {'/a/b/c': rng.standard_normal((5, 3)),
              '/a/ints': rng.randint(0, 1000, size=7).astype(numpy.int32),
              '/top': numpy.arange(4, dtype=numpy.int64)}
    path = str(tmp_path / 'x.h5')
    with hdf5.File(path, 'w') as f:
        for name, a in arrays.items():
            f.write(name, a)
        with pytest.raises(TypeError):
            f.write('/bad', numpy.zeros(3, dtype=numpy.float32))
    with hdf5.File(path) as f:
        for name, a in arrays.items():
            got = f.read(name)
            assert got.dtype == a.dtype and numpy.array_equal(got, a)
        assert not f.exists('/a/b/d')
    with pytest.raises(IOError):
        hdf5.File(str(tmp_path / 'missing.h5'))
