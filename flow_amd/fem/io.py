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


def read_mesh(path, reorder=True):
    '''`Mesh('test.xml')` / the `.msh` cache of the reference drivers
    (tests/test_karman_vortex_street.py:29-33, 52-53).  reorder: a loaded mesh
    is renumbered along its longest axis like the generators' meshes
    (Mesh.reordered: what the packed streams, the strips and the cell kernels'
    coalescing lean on); `vertex_origin` maps back to the file's ids.'''
    ext = os.path.splitext(path)[1].lower()
    if ext == '.msh':
        mesh = read_msh(path)
    elif ext == '.xml':
        mesh = read_dolfin_xml(path)
    else:
        raise ValueError('unknown mesh format %r' % ext)
    return mesh.reordered() if reorder else mesh


# -- XDMF time series ---------------------------------------------------------
class XDMFFile(object):
    """`with XDMFFile(mpi_comm_world(), 'karman.xdmf') as f: f.write(u, t)`
    (reference driver: tests/test_karman_vortex_street.py:214-227).  Vertex
    values of the field are written (what dolfin's XDMFFile.write does for a
    P2 function as well).

    parameters['heavy_data']: 'hdf5' -- the arrays go to `<stem>.h5` beside the
    XML (datasets /Mesh/<k>/mesh/geometry, /Mesh/<k>/mesh/topology,
    /VisualisationVector/<k>, as dolfin lays them out), through the HDF5 C
    library (flow_amd/fem/hdf5.py) --, 'xml' -- inline data items --, or 'auto'
    (default): 'hdf5' where the library is found.  With
    'rewrite_function_mesh' False the mesh is stored once and every time step
    refers to it; 'flush_output' rewrites the XML (and flushes the HDF5 file)
    at every write, so that a run can be looked at while it lasts (in 'xml'
    mode that is the whole inline data again at every write -- the arrays are
    formatted once, the file grows quadratically: use 'hdf5' for long runs).
    A write after close() opens the heavy-data file again and adds to it.  On
    the strips of flow_amd.parallel the field is gathered and rank 0 alone
    writes.  Differences from dolfin's files: the topology is stored as
    int64 in the HDF5 file (dolfin: uint64; declared `UInt` in the XML either
    way), one `Grid` per time value holds every field written at that time,
    and only vertex values are written."""

    def __init__(self, comm_or_path, path=None):
        self.path = path if path is not None else comm_or_path
        self.parameters = {'flush_output': False,
                           'rewrite_function_mesh': True,
                           'heavy_data': 'auto'}
        self._steps = []          # (t, name, ncomp, values or dataset path)
        self._mesh = None
        self._h5 = None
        self._heavy = None        # decided at the first write
        self._mesh_sets = []      # [(geometry, topology)] dataset paths
        self._nsets = 0

    def __enter__(self):
        return self

    def __exit__(self, tpe, value, traceback):
        self.close()
        return False

    # -- heavy data ----------------------------------------------------------------
    def _h5_name(self):
        stem = self.path[:-5] if self.path.endswith('.xdmf') else self.path
        return stem + '.h5'

    def _decide(self):
        if self._heavy is None:
            from . import hdf5
            want = self.parameters.get('heavy_data', 'auto')
            if want not in ('auto', 'hdf5', 'xml'):
                raise ValueError("heavy_data: 'auto', 'hdf5' or 'xml'")
            if want == 'hdf5' and not hdf5.available():
                raise IOError('heavy_data = hdf5: no HDF5 library found '
                              '(FLOW_AMD_HDF5_LIB)')
            self._heavy = 'hdf5' if want == 'hdf5' or (
                want == 'auto' and hdf5.available()) else 'xml'
            if self._heavy == 'hdf5':
                self._h5 = hdf5.File(self._h5_name(), 'w')
        elif self._heavy == 'hdf5' and self._h5 is None:
            # a write after close(): the heavy-data file is opened again and
            # added to (the XML is rewritten from what this object remembers)
            from . import hdf5
            self._h5 = hdf5.File(self._h5_name(), 'a')
        return self._heavy

    def write(self, u, t=0.0):
        '''Vertex values of `u` at time `t` (dolfin: XDMFFile.write(u, t)).
        On the strips of flow_amd.parallel every rank calls this: the field is
        made whole (owners' rows summed over the ranks) and RANK 0 ALONE
        writes -- one file, one writer.'''
        from .. import parallel
        V = u.function_space()
        mesh = V.mesh()
        assert self._mesh is None or self._mesh is mesh
        if parallel.active():
            from .. import _hip, device
            whole = _hip.clone(u.data)
            parallel.gather_field(whole, V.layout, V.dim)
            if parallel.comm().rank != 0:
                return
            arr = device.to_host(whole).numpy().reshape(V.dim, V.N)
        else:
            arr = u.array().reshape(V.dim, V.N)
        self._mesh = mesh
        vals = arr[:, V.layout.vertex_dofs].T.copy()
        nv = mesh.num_vertices()
        if V.dim > 1:           # (XDMF vectors have three components)
            vals = numpy.concatenate([vals, numpy.zeros((nv, 1))], axis=1)
        if self._decide() == 'hdf5':
            new_time = not self._steps or self._steps[-1][0] != float(t)
            if not self._mesh_sets or (
                    new_time and self.parameters.get('rewrite_function_mesh')):
                k = len(self._mesh_sets)
                names = ('/Mesh/%d/mesh/geometry' % k,
                         '/Mesh/%d/mesh/topology' % k)
                self._h5.write(names[0], numpy.asarray(mesh.points, dtype=float))
                self._h5.write(names[1], numpy.asarray(mesh.cell_vertices,
                                                       dtype=numpy.int64))
                self._mesh_sets.append(names)
            name = '/VisualisationVector/%d' % self._nsets
            self._nsets += 1
            self._h5.write(name, vals)
            self._steps.append((float(t), u.name(), V.dim, name,
                                len(self._mesh_sets) - 1))
        else:
            # (formatted once: flush_output rewrites the XML after every write)
            text = '\n'.join(' '.join('%.17g' % v for v in r) for r in vals)
            self._steps.append((float(t), u.name(), V.dim, text, 0))
        if self.parameters.get('flush_output'):
            self._write_xml()
            if self._h5 is not None:
                self._h5.flush()

    def close(self):
        if self._mesh is None:
            return
        self._write_xml()
        if self._h5 is not None:
            self._h5.close()
            self._h5 = None

    # -- light data ----------------------------------------------------------------
    def _item(self, dims, number_type, payload):
        """A DataItem: `payload` a dataset path (heavy data in the HDF5 file)
        or the text of an inline item."""
        nt = 'NumberType="UInt"' if number_type == 'Int' else \
            'NumberType="Float" Precision="8"'
        if self._heavy == 'hdf5':
            return ('<DataItem Format="HDF" Dimensions="%s" %s>%s:%s</DataItem>'
                    % (dims, nt, os.path.basename(self._h5_name()), payload))
        return ('<DataItem Format="XML" Dimensions="%s" %s>\n%s\n    </DataItem>'
                % (dims, nt, payload))

    def _write_xml(self):
        mesh = self._mesh
        nv, nc = mesh.num_vertices(), mesh.num_cells()
        out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0">', ' <Domain>',
               '  <Grid Name="TimeSeries" GridType="Collection" '
               'CollectionType="Temporal">']
        if self._heavy == 'hdf5':
            topo = geom = None
        else:
            if getattr(self, '_mesh_text', None) is None:
                self._mesh_text = (
                    '\n'.join(' '.join(str(v) for v in c)
                              for c in mesh.cell_vertices),
                    '\n'.join('%.17g %.17g' % (x, y) for x, y in mesh.points))
            topo, geom = self._mesh_text
        times = sorted(set(s[0] for s in self._steps))
        for t in times:
            mine = [s for s in self._steps if s[0] == t]
            if self._heavy == 'hdf5':
                geom, topo = self._mesh_sets[mine[0][4]]
            out.append('   <Grid Name="mesh" GridType="Uniform">')
            out.append('    <Time Value="%.17g"/>' % t)
            out.append('    <Topology TopologyType="Triangle" '
                       'NumberOfElements="%d">%s</Topology>'
                       % (nc, self._item('%d 3' % nc, 'Int', topo)))
            out.append('    <Geometry GeometryType="XY">%s</Geometry>'
                       % self._item('%d 2' % nv, 'Float', geom))
            for (ts, name, ncomp, vals, _m) in mine:
                atype = 'Scalar' if ncomp == 1 else 'Vector'
                dims = '%d' % nv if ncomp == 1 else '%d 3' % nv
                if self._heavy == 'hdf5':
                    dims = '%d 1' % nv if ncomp == 1 else dims
                    payload = vals
                else:
                    payload = vals
                out.append('    <Attribute Name="%s" AttributeType="%s" '
                           'Center="Node">%s</Attribute>'
                           % (name, atype, self._item(dims, 'Float', payload)))
            out.append('   </Grid>')
        out += ['  </Grid>', ' </Domain>', '</Xdmf>']
        with open(self.path, 'w') as fh:
            fh.write('\n'.join(out) + '\n')


def read_xdmf_series(path):
    """[(t, {name: values})] of a file written by XDMFFile (tests); heavy data
    in an HDF5 file beside it is read through flow_amd/fem/hdf5.py."""
    root = ET.parse(path).getroot()
    files = {}

    def data(item):
        dims = [int(d) for d in item.get('Dimensions').split()]
        if item.get('Format') == 'HDF':
            fname, dset = item.text.strip().split(':', 1)
            if fname not in files:
                from . import hdf5
                files[fname] = hdf5.File(
                    os.path.join(os.path.dirname(path) or '.', fname))
            arr = files[fname].read(dset)
            assert list(arr.shape) == dims, (dset, arr.shape, dims)
            return arr
        kind = int if item.get('NumberType') in ('Int', 'UInt') else float
        return numpy.array(item.text.split(), dtype=kind).reshape(dims)

    series = []
    try:
        for grid in root.iter('Grid'):
            if grid.get('GridType') != 'Uniform':
                continue
            t = float(grid.find('Time').get('Value'))
            fields = {}
            for att in grid.findall('Attribute'):
                vals = data(att.find('DataItem'))
                if att.get('AttributeType') == 'Scalar':
                    vals = vals.reshape(-1)
                fields[att.get('Name')] = vals
            fields['_topology'] = data(grid.find('Topology').find('DataItem'))
            fields['_geometry'] = data(grid.find('Geometry').find('DataItem'))
            series.append((t, fields))
    finally:
        for f in files.values():
            f.close()
    return series
