# -*- coding: utf-8 -*-
'''
Lagrange P1/P2 function spaces on triangle meshes: dof maps, the CSR sparsity
pattern and the atomic-free "contribution maps" the HIP assembly kernels use.

Host-side setup only (numpy, once per mesh); the per-step work happens in
flow_amd/csrc.  Replaces what `FunctionSpace(mesh, 'CG', k)` /
`VectorFunctionSpace(mesh, 'CG', 2)` provide to the reference
(tests/test_navier_stokes.py:282-283, tests/test_karman_vortex_street.py:59-61).

Layout decisions:
  * vector spaces are stored component-blocked: dof (comp, i) -> comp*N + i, so
    each component is a contiguous fp64 array in HBM;
  * scalar dofs are numbered by (lowest vertex id, kind): P2 edge dofs sit next
    to their lower vertex, so the mesh's vertex ordering (x-major for the
    channel) carries over to a banded operator;
  * every matrix on a space shares ONE CSR pattern (rowptr, cols); values are
    separate arrays (the Jacobian has four value planes over the scalar
    pattern).
'''
import numpy

from . import reference

# CSR-stream tiling (must match flow_amd/csrc/la_kernels.hip)
SPMV_ROWS_PER_BLOCK = 256
SPMV_NNZ_PER_BLOCK = 1022     # LDS tile (1024) minus the alignment slack


class ScalarLayout(object):
    '''Dof map + sparsity pattern + contribution maps of a scalar P_k space.'''

    def __init__(self, mesh, degree):
        assert degree in (1, 2)
        self.mesh = mesh
        self.degree = degree
        self.nloc = reference.nloc(degree)
        nv = mesh.num_vertices()
        if degree == 1:
            self.N = nv
            self.cell_dofs = mesh.cell_vertices.copy()
            self.vertex_dofs = numpy.arange(nv, dtype=numpy.int32)
            self.edge_dofs = None
            self.dof_coords = mesh.points.copy()
        else:
            edges = mesh.edges.astype(numpy.int64)
            ne = len(edges)
            vkey = numpy.arange(nv, dtype=numpy.int64) * (nv + 1)
            ekey = edges[:, 0] * (nv + 1) + 1 + edges[:, 1]
            order = numpy.argsort(numpy.concatenate([vkey, ekey]))
            rank = numpy.empty(nv + ne, dtype=numpy.int32)
            rank[order] = numpy.arange(nv + ne, dtype=numpy.int32)
            self.N = nv + ne
            self.vertex_dofs = rank[:nv]
            self.edge_dofs = rank[nv:]
            self.cell_dofs = numpy.concatenate([
                self.vertex_dofs[mesh.cell_vertices],
                self.edge_dofs[mesh.cell_edges],
                ], axis=1).astype(numpy.int32)
            coords = numpy.empty((self.N, 2))
            coords[self.vertex_dofs] = mesh.points
            coords[self.edge_dofs] = 0.5 * (
                mesh.points[edges[:, 0]] + mesh.points[edges[:, 1]]
                )
            self.dof_coords = coords
        self.cell_dofs = numpy.ascontiguousarray(self.cell_dofs)
        self._pattern = None
        self._vmap = None
        self._dev = {}
        return

    # -- sparsity pattern and matrix contribution map ------------------------
    def _build_pattern(self):
        nloc = self.nloc
        nc = len(self.cell_dofs)
        cd = self.cell_dofs.astype(numpy.int64)
        key = (cd[:, :, None] * self.N + cd[:, None, :]).ravel()
        order = numpy.argsort(key, kind='stable')
        skey = key[order]
        del key
        new = numpy.empty(len(skey), dtype=bool)
        new[0] = True
        numpy.not_equal(skey[1:], skey[:-1], out=new[1:])
        first = numpy.nonzero(new)[0]
        ukey = skey[first]
        del skey, new
        rows = ukey // self.N
        cols = (ukey % self.N).astype(numpy.int32)
        nnz = len(ukey)
        rowptr = numpy.zeros(self.N + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(rows, minlength=self.N), out=rowptr[1:])
        cptr = numpy.concatenate([first, [len(order)]]).astype(numpy.int32)
        # scratch index of pair p = (cell, ij): ij*Nc + cell  (cell fastest)
        csrc = ((order % (nloc * nloc)) * nc + order // (nloc * nloc)).astype(
            numpy.int32
            )
        assert nloc * nloc * nc < 2**31
        # position of the diagonal entry in every row
        is_diag = cols == rows
        diag_idx = numpy.nonzero(is_diag)[0].astype(numpy.int32)
        assert len(diag_idx) == self.N
        self._pattern = {
            'rowptr': rowptr.astype(numpy.int32),
            'cols': cols,
            'nnz': nnz,
            'cptr': cptr,
            'csrc': csrc,
            'diag_idx': diag_idx,
            # CSR-stream row blocks: for operators that park one product per
            # nonzero in LDS (kinds 0, 1) and for those that park two (2, 4)
            'rowblocks': csr_stream_rowblocks(rowptr),
            'rowblocks2': csr_stream_rowblocks(
                rowptr, nnz_per_block=SPMV_NNZ_PER_BLOCK),
            }
        return

    def pattern(self, name):
        if self._pattern is None:
            self._build_pattern()
        return self._pattern[name]

    @property
    def nnz(self):
        return self.pattern('nnz')

    # -- vector contribution map ---------------------------------------------
    def _build_vmap(self):
        nc = len(self.cell_dofs)
        flat = self.cell_dofs.ravel()
        order = numpy.argsort(flat, kind='stable')
        vptr = numpy.zeros(self.N + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.bincount(flat, minlength=self.N), out=vptr[1:])
        vsrc = ((order % self.nloc) * nc + order // self.nloc).astype(
            numpy.int32
            )
        self._vmap = {'vptr': vptr.astype(numpy.int32), 'vsrc': vsrc}
        return

    def vmap(self, name):
        if self._vmap is None:
            self._build_vmap()
        return self._vmap[name]

    # -- device-resident copies (torch tensors: plumbing for HBM storage) ----
    def dev(self, name):
        from .. import device
        if name not in self._dev:
            if name == 'cell_dofs':
                # (nloc, Nc): cell fastest, coalesced for one-thread-per-cell
                arr = numpy.ascontiguousarray(self.cell_dofs.T)
            elif name in ('vptr', 'vsrc'):
                arr = self.vmap(name)
            else:
                arr = self.pattern(name)
            if name == 'cols':
                # the SpMV loads index PAIRS, the kernels of flow_pmg index
                # QUADS: readable up to three entries past nnz
                arr = numpy.concatenate([arr, numpy.zeros(4, dtype=arr.dtype)])
            self._dev[name] = device.to_device(arr)
        return self._dev[name]


def csr_stream_rowblocks(rowptr, rows_per_block=SPMV_ROWS_PER_BLOCK,
                         nnz_per_block=None):
    '''Row-block boundaries for the CSR-stream SpMV: consecutive rows are
    grouped so that a block has at most `rows_per_block` rows and at most
    `nnz_per_block` nonzeros (the LDS tile of products; default: what the
    library's scalar kernels take, flow_spmv_tile_nnz(0)).  `rowptr` may be a
    LIST of row pointers over the same rows (kernels that do the products of
    several matrices for the rows of a workgroup: flow_mg's up-sweep): every
    one of them then stays below the cap.'''
    if nnz_per_block is None:
        from .. import _hip
        try:
            nnz_per_block = _hip.spmv_tile_nnz(0)
        except (OSError, _hip.HipError):
            # host-only use (mesh IO, transfer tables, oracle comparisons)
            # without a built library: the header's constant
            nnz_per_block = _hip.SPMV_NNZ_PER_BLOCK
    rowptrs = rowptr if isinstance(rowptr, (list, tuple)) else [rowptr]
    rowptrs = [numpy.asarray(rp, dtype=numpy.int64) for rp in rowptrs]
    n = len(rowptrs[0]) - 1
    for rp in rowptrs:
        assert len(rp) == n + 1
        if n > 0:
            assert (rp[1:] - rp[:-1]).max() <= nnz_per_block, \
                'row longer than the CSR-stream LDS tile'
    blocks = [0]
    r = 0
    while r < n:
        r_next = min(r + rows_per_block, n)
        for rp in rowptrs:
            r_nnz = int(numpy.searchsorted(
                rp, rp[r] + nnz_per_block, side='right')) - 1
            r_next = min(r_next, r_nnz)
        assert r_next > r
        blocks.append(r_next)
        r = r_next
    return numpy.array(blocks, dtype=numpy.int32)


def mesh_geometry_dev(mesh):
    '''Device copies of the per-cell geometry inputs: vertex coordinates as
    two (3, Nc) SoA planes (x and y, cell fastest).'''
    from .. import device
    if 'geom_dev' not in mesh._cache:
        p = mesh.points[mesh.cell_vertices]        # (Nc, 3, 2)
        xy = numpy.ascontiguousarray(p.transpose(2, 1, 0))   # (2, 3, Nc)
        mesh._cache['geom_dev'] = device.to_device(xy)
    return mesh._cache['geom_dev']


def scalar_layout(mesh, degree):
    key = ('layout', degree)
    if key not in mesh._cache:
        mesh._cache[key] = ScalarLayout(mesh, degree)
    return mesh._cache[key]


class FunctionSpace(object):
    '''`FunctionSpace(mesh, 'CG'|'Lagrange'|'P', degree)` (scalar) or, with
    dim=2, the vector space.  `sub(i)` gives the component view used for
    component-wise Dirichlet conditions
    (tests/test_karman_vortex_street.py:194-196).'''

    def __new__(cls, mesh, family='CG', *args, **kwargs):
        # FunctionSpace(mesh, W_element * P_element) -> Taylor-Hood pair
        # (tests/test_karman_vortex_street.py:59-61)
        if isinstance(family, MixedElement):
            return MixedFunctionSpace(mesh, family)
        return super(FunctionSpace, cls).__new__(cls)

    def __init__(self, mesh, family='CG', degree=1, dim=1, _component=None,
                 _parent=None):
        if isinstance(family, FiniteElement):
            degree = family.degree()
            dim = family.dim
            family = family.family
        assert family in ('CG', 'Lagrange', 'P'), family
        assert dim in (1, 2)
        self._mesh = mesh
        self.degree = degree
        self.dim = dim
        self.layout = scalar_layout(mesh, degree)
        self.component = _component
        self.parent = _parent
        return

    def mesh(self):
        return self._mesh

    @property
    def N(self):
        return self.layout.N

    def size(self):
        return self.dim * self.layout.N

    def dim_(self):
        return self.size()

    def num_sub_spaces(self):
        return self.dim if self.dim > 1 else 0

    def sub(self, i):
        assert self.dim == 2 and i in (0, 1)
        return FunctionSpace(
            self._mesh, 'CG', self.degree, dim=1, _component=i, _parent=self
            )

    def collapse(self):
        return FunctionSpace(self._mesh, 'CG', self.degree, dim=1)

    def ufl_element(self):
        return FiniteElement('Lagrange', 'triangle', self.degree, self.dim)

    def tabulate_dof_coordinates(self):
        c = self.layout.dof_coords
        return numpy.concatenate([c] * self.dim) if self.dim > 1 else c

    def same_as(self, other):
        return (
            self._mesh is other._mesh and self.degree == other.degree
            and self.dim == other.dim
            )


class FiniteElement(object):
    def __init__(self, family, cell, degree, dim=1):
        self.family = family
        self.cell = cell
        self._degree = degree
        self.dim = dim

    def degree(self):
        return self._degree

    def __mul__(self, other):
        return MixedElement([self, other])


class MixedElement(object):
    def __init__(self, elements):
        self.elements = list(elements)


class MixedFunctionSpace(object):
    '''The velocity-pressure pair `W_element * P_element`: only what the
    reference's drivers use -- `.sub(0)` / `.sub(1)` to pose Dirichlet
    conditions and to hand the pair to `flow.stokes.solve`
    (tests/test_karman_vortex_street.py:59-63, 171-179; tests/test_stokes.py:
    137-153).  The sub-spaces are ordinary (collapsed) spaces; there is no
    monolithic mixed dof vector.'''

    def __init__(self, mesh, element):
        self._mesh = mesh
        self.spaces = [FunctionSpace(mesh, e) for e in element.elements]

    def mesh(self):
        return self._mesh

    def sub(self, i):
        return self.spaces[i]

    def num_sub_spaces(self):
        return len(self.spaces)


# pylint: disable=invalid-name
def VectorElement(family, cell, degree):
    return FiniteElement(family, cell, degree, dim=2)


def VectorFunctionSpace(mesh, family, degree):
    return FunctionSpace(mesh, family, degree, dim=2)
