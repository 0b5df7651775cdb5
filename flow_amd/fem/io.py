# -*- coding: utf-8 -*-
'''
On-disk formats either side of the path: the meshes the
reference's drivers read -- gmsh `.msh` (the pygmsh cache file,
tests/test_karman_vortex_street.py:29-33) and DOLFIN XML (`Mesh('test.xml')`,
:52-53) -- and the XDMF time series they write (`XDMFFile(...).write(u0, t)`,
:214-227; tests/test_boussinesq.py:164-166, 307-309).

Own implementations (MSH 2.2 ASCII and binary, DOLFIN XML, XDMF with inline XML
data items): no meshio / h5py offline, so no HDF5-backed XDMF.  Host-side I/O
only.
'''
from __future__ import print_function

import os
import xml.etree.ElementTree as ET

import numpy

from .mesh import Mesh


def mpi_comm_world():
    '''Placeholder for dolfin's communicator argument of XDMFFile.'''
    return None


# -- meshes -------------------------------------------------------------------
def _mesh_from_gmsh(ids, xyz, tris):
    '''Compress node ids to the vertices the triangles use, keep file order.'''
    tris = numpy.asarray(tris, dtype=numpy.int64).reshape(-1, 3)
    used = numpy.unique(tris)
    lookup = -numpy.ones(int(ids.max()) + 1, dtype=numpy.int64)
    pos = {int(v): k for k, v in enumerate(ids)}
    rows = numpy.array([pos[int(v)] for v in used])
    lookup[used] = numpy.arange(len(used))
    return Mesh(numpy.asarray(xyz)[rows][:, :2], lookup[tris])


_GMSH_NODES = {1: 2, 2: 3, 3: 4, 4: 4, 5: 8, 6: 6, 7: 5, 8: 3, 9: 6, 10: 9,
               11: 10, 15: 1}     # nodes per element type (the common ones)


def _read_msh_binary(raw, start):
    '''The sections of a binary MSH 2.2 file behind the format line: `raw` is
    the file, `start` the offset of the byte behind that line.'''
    one = numpy.frombuffer(raw, dtype='<i4', count=1, offset=start)[0]
    order = '<' if one == 1 else '>'
    if numpy.frombuffer(raw, dtype=order + 'i4', count=1, offset=start)[0] != 1:
        raise ValueError('binary MSH: bad endianness marker')

    def section(name, frm):
        # sections are found SEQUENTIALLY, from the end of the one before: a
        # search over the whole file could hit the tag's bytes inside a binary
        # payload
        tag = ('$%s' % name).encode()
        at = raw.index(tag, frm) + len(tag)
        eol = raw.index(b'\n', at)            # (LF or CRLF line ends)
        at = eol + 1
        eol = raw.index(b'\n', at)
        return int(raw[at:eol].strip()), eol + 1

    # behind the format line: the 4-byte endianness marker, its line end, then
    # $EndMeshFormat
    n, at = section('Nodes', start + 4)
    rec = numpy.dtype([('id', order + 'i4'), ('x', order + 'f8', (3,))])
    nodes = numpy.frombuffer(raw, dtype=rec, count=n, offset=at)
    m, at = section('Elements', at + n * rec.itemsize)
    tris = []
    seen = 0
    while seen < m:
        etype, count, ntags = numpy.frombuffer(raw, dtype=order + 'i4', count=3,
                                               offset=at)
        at += 12
        if int(etype) not in _GMSH_NODES:
            raise ValueError('binary MSH: element type %d' % etype)
        width = 1 + int(ntags) + _GMSH_NODES[int(etype)]
        block = numpy.frombuffer(raw, dtype=order + 'i4',
                                 count=int(count) * width,
                                 offset=at).reshape(int(count), width)
        at += 4 * int(count) * width
        if int(etype) == 2:
            tris.append(block[:, 1 + int(ntags):])
        seen += int(count)
    if not tris:
        raise ValueError('binary MSH: no triangles')
    return _mesh_from_gmsh(nodes['id'].astype(numpy.int64), nodes['x'],
                           numpy.concatenate(tris))


def read_msh(path):
    '''gmsh MSH 2.2, ASCII or binary (what `pygmsh` caches; gmsh writes the
    binary flavour with `-bin`): triangles (element type 2) of a planar
    mesh.'''
    with open(path, 'rb') as fh:
        raw = fh.read()
    head = raw.index(b'$MeshFormat') + len(b'$MeshFormat')
    eol = raw.index(b'\n', head + 1)
    version = raw[head:eol].split()
    if not version[0].startswith(b'2'):
        raise ValueError('only MSH 2.x is supported (got %r)' % version)
    if version[1] == b'1':
        return _read_msh_binary(raw, eol + 1)
    lines = [ln.strip() for ln in raw.decode().splitlines()]
    i = lines.index('$Nodes')
    n = int(lines[i + 1])
    nodes = numpy.array([ln.split() for ln in lines[i + 2:i + 2 + n]],
                        dtype=float)
    ids = nodes[:, 0].astype(numpy.int64)
    i = lines.index('$Elements')
    m = int(lines[i + 1])
    tris = []
    for ln in lines[i + 2:i + 2 + m]:
        t = ln.split()
        if int(t[1]) == 2:                   # 3-node triangle
            ntags = int(t[2])
            tris.append([int(v) for v in t[3 + ntags:3 + ntags + 3]])
    return _mesh_from_gmsh(ids, nodes[:, 1:4], tris)


def write_msh(path, mesh, binary=False):
    if binary:
        return _write_msh_binary(path, mesh)
    with open(path, 'w') as fh:
        fh.write('$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n'
                 % mesh.num_vertices())
        for k, (x, y) in enumerate(mesh.points):
            fh.write('%d %.17g %.17g 0\n' % (k + 1, x, y))
        fh.write('$EndNodes\n$Elements\n%d\n' % mesh.num_cells())
        for k, c in enumerate(mesh.cell_vertices):
            fh.write('%d 2 2 0 1 %d %d %d\n' % (k + 1, c[0] + 1, c[1] + 1,
                                                 c[2] + 1))
        fh.write('$EndElements\n')


def _write_msh_binary(path, mesh):
    '''MSH 2.2 binary, little endian: the boundary edges (element type 1, two
    tags) in a block of their own BEFORE the block of triangles, as gmsh
    writes physical lines -- readers have to walk mixed blocks.'''
    nv, nc = mesh.num_vertices(), mesh.num_cells()
    rec = numpy.zeros(nv, dtype=[('id', '<i4'), ('x', '<f8', (3,))])
    rec['id'] = numpy.arange(1, nv + 1)
    rec['x'][:, :2] = mesh.points
    bedges = mesh.edges[mesh.bfacets]
    nb = len(bedges)
    lines = numpy.zeros((nb, 5), dtype='<i4')
    lines[:, 0] = numpy.arange(1, nb + 1)
    lines[:, 1] = 1
    lines[:, 2] = 1
    lines[:, 3:] = bedges + 1
    tri = numpy.zeros((nc, 6), dtype='<i4')
    tri[:, 0] = numpy.arange(nb + 1, nb + nc + 1)
    tri[:, 2] = 1
    tri[:, 3:] = mesh.cell_vertices + 1
    with open(path, 'wb') as fh:
        fh.write(b'$MeshFormat\n2.2 1 8\n')
        fh.write(numpy.array([1], dtype='<i4').tobytes())
        fh.write(b'\n$EndMeshFormat\n$Nodes\n%d\n' % nv)
        fh.write(rec.tobytes())
        fh.write(b'\n$EndNodes\n$Elements\n%d\n' % (nb + nc))
        fh.write(numpy.array([1, nb, 2], dtype='<i4').tobytes())
        fh.write(lines.tobytes())
        fh.write(numpy.array([2, nc, 2], dtype='<i4').tobytes())
        fh.write(tri.tobytes())
        fh.write(b'\n$EndElements\n')


def read_dolfin_xml(path):
    root = ET.parse(path).getroot()
    m = root.find('mesh')
    assert m.get('celltype') == 'triangle'
    verts = m.find('vertices')
    pts = numpy.zeros((int(verts.get('size')), 2))
    for v in verts:
        pts[int(v.get('index'))] = (float(v.get('x')), float(v.get('y')))
    cells_el = m.find('cells')
    cells = numpy.zeros((int(cells_el.get('size')), 3), dtype=numpy.int64)
    for c in cells_el:
        cells[int(c.get('index'))] = (int(c.get('v0')), int(c.get('v1')),
                                      int(c.get('v2')))
    return Mesh(pts, cells)


def write_dolfin_xml(path, mesh):
    with open(path, 'w') as fh:
        fh.write('<?xml version="1.0"?>\n<dolfin xmlns:dolfin='
                 '"http://fenicsproject.org">\n  <mesh celltype="triangle" '
                 'dim="2">\n    <vertices size="%d">\n' % mesh.num_vertices())
        for k, (x, y) in enumerate(mesh.points):
            fh.write('      <vertex index="%d" x="%.17g" y="%.17g"/>\n'
                     % (k, x, y))
        fh.write('    </vertices>\n    <cells size="%d">\n' % mesh.num_cells())
        for k, c in enumerate(mesh.cell_vertices):
            fh.write('      <triangle index="%d" v0="%d" v1="%d" v2="%d"/>\n'
                     % (k, c[0], c[1], c[2]))
        fh.write('    </cells>\n  </mesh>\n</dolfin>\n')


def read_mesh(path):
    '''`Mesh('test.xml')` / the `.msh` cache of the reference drivers.'''
    ext = os.path.splitext(path)[1].lower()
    if ext == '.msh':
        return read_msh(path)
    if ext == '.xml':
        return read_dolfin_xml(path)
    raise ValueError('unknown mesh format %r' % ext)


# -- XDMF time series ---------------------------------------------------------
class XDMFFile(object):
    '''`with XDMFFile(mpi_comm_world(), 'karman.xdmf') as f: f.write(u, t)`.
    Vertex values of the field are written (what dolfin's XDMFFile.write does
    for a P2 function as well); data items are inline XML.'''

    def __init__(self, comm_or_path, path=None):
        self.path = path if path is not None else comm_or_path
        self.parameters = {'flush_output': False,
                           'rewrite_function_mesh': True}
        self._steps = []          # (t, name, ncomp, values (Nv, ncomp))
        self._mesh = None

    def __enter__(self):
        return self

    def __exit__(self, tpe, value, traceback):
        self.close()
        return False

    def write(self, u, t=0.0):
        V = u.function_space()
        mesh = V.mesh()
        assert self._mesh is None or self._mesh is mesh
        self._mesh = mesh
        arr = u.array().reshape(V.dim, V.N)
        vals = arr[:, V.layout.vertex_dofs].T.copy()
        self._steps.append((float(t), u.name(), V.dim, vals))
        if self.parameters.get('flush_output'):
            self.close()

    def close(self):
        if self._mesh is None:
            return
        mesh = self._mesh
        nv, nc = mesh.num_vertices(), mesh.num_cells()
        out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0">', ' <Domain>',
               '  <Grid Name="TimeSeries" GridType="Collection" '
               'CollectionType="Temporal">']
        topo = '\n'.join(' '.join(str(v) for v in c)
                         for c in mesh.cell_vertices)
        geom = '\n'.join('%.17g %.17g' % (x, y) for x, y in mesh.points)
        times = sorted(set(s[0] for s in self._steps))
        for t in times:
            out.append('   <Grid Name="mesh" GridType="Uniform">')
            out.append('    <Time Value="%.17g"/>' % t)
            out.append('    <Topology TopologyType="Triangle" '
                       'NumberOfElements="%d"><DataItem Format="XML" '
                       'Dimensions="%d 3" NumberType="Int">' % (nc, nc))
            out.append(topo)
            out.append('    </DataItem></Topology>')
            out.append('    <Geometry GeometryType="XY"><DataItem Format="XML" '
                       'Dimensions="%d 2">' % nv)
            out.append(geom)
            out.append('    </DataItem></Geometry>')
            for (ts, name, ncomp, vals) in self._steps:
                if ts != t:
                    continue
                if ncomp == 1:
                    atype, dims, rows = 'Scalar', '%d' % nv, vals[:, 0:1]
                else:
                    atype, dims = 'Vector', '%d 3' % nv
                    rows = numpy.concatenate(
                        [vals, numpy.zeros((nv, 1))], axis=1)
                out.append('    <Attribute Name="%s" AttributeType="%s" '
                           'Center="Node"><DataItem Format="XML" '
                           'Dimensions="%s">' % (name, atype, dims))
                out.append('\n'.join(' '.join('%.17g' % v for v in r)
                                     for r in rows))
                out.append('    </DataItem></Attribute>')
            out.append('   </Grid>')
        out += ['  </Grid>', ' </Domain>', '</Xdmf>']
        with open(self.path, 'w') as fh:
            fh.write('\n'.join(out) + '\n')


def read_xdmf_series(path):
    '''[(t, {name: values})] of a file written by XDMFFile (tests).'''
    root = ET.parse(path).getroot()
    series = []
    for grid in root.iter('Grid'):
        if grid.get('GridType') != 'Uniform':
            continue
        t = float(grid.find('Time').get('Value'))
        fields = {}
        for att in grid.findall('Attribute'):
            item = att.find('DataItem')
            dims = [int(d) for d in item.get('Dimensions').split()]
            fields[att.get('Name')] = numpy.array(
                item.text.split(), dtype=float).reshape(dims)
        series.append((t, fields))
    return series
