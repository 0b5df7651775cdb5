# -*- coding: utf-8 -*-
'''
Triangle meshes for the host side of the path: storage, derived topology
(edges, boundary facets) and the structured generators the reference drivers
use -- `UnitSquareMesh(n, n, 'crossed'|'left/right')`
(tests/test_navier_stokes.py:82,176), `RectangleMesh(Point, Point, n, n,
'crossed')` (:144) -- plus a structured rectangle-with-circular-hole generator
that stands in for the pygmsh meshes of tests/test_karman_vortex_street.py:26-53
and tests/test_boussinesq.py:46-79 (gmsh is not available offline).
'''
import numpy


class Point(object):
    def __init__(self, *xy):
        self.xy = numpy.array(xy, dtype=float)

    def __getitem__(self, i):
        return self.xy[i]


class Mesh(object):
    '''points: (Nv, 2) float64, cells: (Nc, 3) int32 vertex ids.

    Derived (lazily):
      edges        (Ne, 2)  sorted vertex pairs
      cell_edges   (Nc, 3)  edge id opposite local vertex i
      bfacets      (Nb,)    ids of edges with exactly one incident cell
      bfacet_cell  (Nb,), bfacet_local (Nb,)  the cell and its local facet index
    '''
    def __init__(self, points, cells=None, reorder=True):
        # (ids the vertices / cells had in the file / in the mesh this one was
        # renumbered from: `reordered`; None: this numbering is the original)
        self.vertex_origin = None
        self.cell_origin = None
        if isinstance(points, str):
            # Mesh('test.xml') / Mesh('karman.msh') as the reference drivers do
            # (tests/test_karman_vortex_street.py:52-53); renumbered along the
            # longest axis like the generators' meshes (io.read_mesh) unless
            # reorder=False -- the reference's Mesh(path) keeps the file's
            # numbering; data indexed by the file's vertices or cells
            # (markers, per-cell coefficients, fields written elsewhere) are
            # carried over with vertex_origin / cell_origin
            from . import io
            loaded = io.read_mesh(points, reorder=reorder)
            points, cells = loaded.points, loaded.cell_vertices
            self.vertex_origin = loaded.vertex_origin
            self.cell_origin = loaded.cell_origin
        self.points = numpy.ascontiguousarray(points, dtype=numpy.float64)
        self.cell_vertices = numpy.ascontiguousarray(cells, dtype=numpy.int32)
        assert self.points.ndim == 2 and self.points.shape[1] == 2
        assert self.cell_vertices.ndim == 2 and self.cell_vertices.shape[1] == 3
        assert self.cell_vertices.min() >= 0
        assert self.cell_vertices.max() < len(self.points)
        self._topology = None
        self._cache = {}
        return

    # -- numbering -------------------------------------------------------------
    def reordered(self):
        '''The same mesh numbered the way the generators number theirs, which
        is what the fast path leans on: vertices sorted along the LONGEST axis
        of the bounding box (ties: the other coordinate) -- every operator is
        then banded with a bandwidth of about one cross-section of vertices,
        x[col] gathers stay in cache, the 16-bit column offsets of the packed
        streams hold, and a strip of the domain is a contiguous range of rows
        (flow_amd/parallel.py) --, cells in the order of their lowest vertex,
        so that neighbouring threads of the cell kernels touch neighbouring
        dofs and a strip is a contiguous cell range.  A mesh from a generic
        generator (gmsh numbers boundary curves first, then the interior in
        the order of its front) has none of that.  `vertex_origin[k]` /
        `cell_origin[c]` = the id vertex k / cell c had before (composed over
        repeated reorderings: always the ORIGINAL ids).'''
        p = self.points
        ext = p.max(axis=0) - p.min(axis=0)
        major = int(numpy.argmax(ext))
        order = numpy.lexsort((p[:, 1 - major], p[:, major]))
        new_id = numpy.empty(len(p), dtype=numpy.int64)
        new_id[order] = numpy.arange(len(p))
        cells = new_id[self.cell_vertices.astype(numpy.int64)]
        corder = numpy.lexsort((cells.max(axis=1), cells.min(axis=1)))
        out = Mesh(p[order], cells[corder].astype(numpy.int32))
        origin = order if self.vertex_origin is None \
            else numpy.asarray(self.vertex_origin)[order]
        out.vertex_origin = origin.astype(numpy.int64)
        corigin = corder if self.cell_origin is None \
            else numpy.asarray(self.cell_origin)[corder]
        out.cell_origin = corigin.astype(numpy.int64)
        return out

    def bandwidth(self):
        '''Largest difference of two vertex ids within a cell.'''
        c = self.cell_vertices.astype(numpy.int64)
        return int((c.max(axis=1) - c.min(axis=1)).max())

    # -- dolfin-flavoured accessors ------------------------------------------
    def num_vertices(self):
        return len(self.points)

    def num_cells(self):
        return len(self.cell_vertices)

    def coordinates(self):
        return self.points

    def cells(self):
        return self.cell_vertices

    def ufl_cell(self):
        return 'triangle'

    def _edge_lengths(self):
        p = self.points
        c = self.cell_vertices
        e = numpy.stack([
            p[c[:, 1]] - p[c[:, 2]],
            p[c[:, 0]] - p[c[:, 2]],
            p[c[:, 0]] - p[c[:, 1]],
            ], axis=1)
        return numpy.sqrt((e**2).sum(axis=2))

    def hmax(self):
        '''Largest cell diameter (= longest edge for a triangle).'''
        return float(self._edge_lengths().max())

    def hmin(self):
        return float(self._edge_lengths().max(axis=1).min())

    def cell_areas(self):
        p = self.points
        c = self.cell_vertices
        d1 = p[c[:, 1]] - p[c[:, 0]]
        d2 = p[c[:, 2]] - p[c[:, 0]]
        return 0.5 * numpy.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])

    # -- topology ------------------------------------------------------------
    def _build_topology(self):
        c = self.cell_vertices.astype(numpy.int64)
        nv = len(self.points)
        # local edge i is opposite local vertex i
        a = numpy.stack([c[:, 1], c[:, 0], c[:, 0]], axis=1)
        b = numpy.stack([c[:, 2], c[:, 2], c[:, 1]], axis=1)
        lo = numpy.minimum(a, b)
        hi = numpy.maximum(a, b)
        key = (lo * nv + hi).ravel()
        ukey, inverse, counts = numpy.unique(
            key, return_inverse=True, return_counts=True
            )
        edges = numpy.stack([ukey // nv, ukey % nv], axis=1).astype(numpy.int32)
        cell_edges = inverse.reshape(-1, 3).astype(numpy.int32)
        bmask = counts == 1
        bfacets = numpy.nonzero(bmask)[0].astype(numpy.int32)
        # owner cell / local index of each boundary facet
        flat_is_b = bmask[inverse]
        flat_idx = numpy.nonzero(flat_is_b)[0]
        order = numpy.argsort(inverse[flat_idx], kind='stable')
        flat_idx = flat_idx[order]
        assert numpy.array_equal(inverse[flat_idx], bfacets)
        self._topology = {
            'edges': edges,
            'cell_edges': cell_edges,
            'bfacets': bfacets,
            'bfacet_cell': (flat_idx // 3).astype(numpy.int32),
            'bfacet_local': (flat_idx % 3).astype(numpy.int32),
            }
        return

    def _topo(self, name):
        if self._topology is None:
            self._build_topology()
        return self._topology[name]

    @property
    def edges(self):
        return self._topo('edges')

    @property
    def cell_edges(self):
        return self._topo('cell_edges')

    @property
    def bfacets(self):
        return self._topo('bfacets')

    @property
    def bfacet_cell(self):
        return self._topo('bfacet_cell')

    @property
    def bfacet_local(self):
        return self._topo('bfacet_local')

    def num_edges(self):
        return len(self.edges)

    def cell_bfacet_mask(self):
        '''Per cell: bit i set iff local facet i lies on the boundary.'''
        mask = numpy.zeros(self.num_cells(), dtype=numpy.int32)
        numpy.bitwise_or.at(
            mask, self.bfacet_cell, (1 << self.bfacet_local).astype(numpy.int32)
            )
        return mask


def _quad_cells(nx, ny, diagonal, vid, mid=None):
    '''Triangles of a structured nx x ny quad grid.  vid(ix, iy) -> vertex id
    arrays; mid(ix, iy) -> centre-vertex ids for 'crossed'.
    '''
    ix, iy = numpy.meshgrid(numpy.arange(nx), numpy.arange(ny), indexing='xy')
    ix = ix.ravel()
    iy = iy.ravel()
    v0 = vid(ix, iy)
    v1 = vid(ix + 1, iy)
    v2 = vid(ix, iy + 1)
    v3 = vid(ix + 1, iy + 1)
    if diagonal == 'crossed':
        vm = mid(ix, iy)
        tris = numpy.stack([
            numpy.stack([v0, v1, vm], axis=1),
            numpy.stack([v0, v2, vm], axis=1),
            numpy.stack([v1, v3, vm], axis=1),
            numpy.stack([v2, v3, vm], axis=1),
            ], axis=1)
        return tris.reshape(-1, 3)
    if diagonal == 'right':
        is_right = numpy.ones(len(ix), dtype=bool)
    elif diagonal == 'left':
        is_right = numpy.zeros(len(ix), dtype=bool)
    elif diagonal == 'left/right':
        is_right = ((ix + iy) % 2) == 1
    elif diagonal == 'right/left':
        is_right = ((ix + iy) % 2) == 0
    else:
        raise ValueError('unknown diagonal %r' % diagonal)
    # right: diagonal v0-v3; left: diagonal v1-v2
    t0 = numpy.where(
        is_right[:, None],
        numpy.stack([v0, v1, v3], axis=1), numpy.stack([v0, v1, v2], axis=1)
        )
    t1 = numpy.where(
        is_right[:, None],
        numpy.stack([v0, v2, v3], axis=1), numpy.stack([v1, v2, v3], axis=1)
        )
    return numpy.stack([t0, t1], axis=1).reshape(-1, 3)


# pylint: disable=invalid-name
def RectangleMesh(p0, p1, nx, ny, diagonal='right'):
    '''Structured rectangle mesh; vertex (ix, iy) has id iy*(nx+1)+ix, the
    centre vertices of 'crossed' follow.'''
    x = numpy.linspace(p0[0], p1[0], nx + 1)
    y = numpy.linspace(p0[1], p1[1], ny + 1)
    X, Y = numpy.meshgrid(x, y, indexing='xy')
    pts = numpy.stack([X.ravel(), Y.ravel()], axis=1)

    def vid(ix, iy):
        return iy * (nx + 1) + ix

    mid = None
    if diagonal == 'crossed':
        xm = 0.5 * (x[:-1] + x[1:])
        ym = 0.5 * (y[:-1] + y[1:])
        XM, YM = numpy.meshgrid(xm, ym, indexing='xy')
        pts = numpy.concatenate(
            [pts, numpy.stack([XM.ravel(), YM.ravel()], axis=1)]
            )

        def mid(ix, iy):
            return (nx + 1) * (ny + 1) + iy * nx + ix

    cells = _quad_cells(nx, ny, diagonal, vid, mid)
    return Mesh(pts, cells)


def UnitSquareMesh(nx, ny, diagonal='right'):
    return RectangleMesh(Point(0.0, 0.0), Point(1.0, 1.0), nx, ny, diagonal)


def rectangle_with_hole(
        x0, x1, y0, y1, centre, radius, nx, ny, diagonal='right'
        ):
    '''Structured nx x ny-cell rectangle [x0,x1]x[y0,y1] with the cells whose
    centroid lies inside the circle (centre, radius) removed (staircase
    obstacle).  Vertices are numbered x-major (all vertices of one grid column
    are consecutive), so that with ny <= nx the operator bandwidth is O(ny):
    this is the layout the row-block sharding of the pressure solve assumes
    (SURVEY.md section 8e).  Geometry constants of the Karman channel:
    tests/test_karman_vortex_street.py:18-23,35-45.
    '''
    x = numpy.linspace(x0, x1, nx + 1)
    y = numpy.linspace(y0, y1, ny + 1)
    X, Y = numpy.meshgrid(x, y, indexing='ij')
    pts = numpy.stack([X.ravel(), Y.ravel()], axis=1)

    def vid(ix, iy):
        return ix * (ny + 1) + iy

    cells = _quad_cells(nx, ny, diagonal, vid)
    cen = pts[cells].mean(axis=1)
    keep = (
        (cen[:, 0] - centre[0])**2 + (cen[:, 1] - centre[1])**2 > radius**2
        )
    cells = cells[keep]
    # cells in the order of their lowest vertex: x-major like the vertices, so
    # that neighbouring threads of the cell kernels (one cell each) touch
    # neighbouring dofs, the contribution lists of a row point to neighbouring
    # scratch entries, and a strip of the domain is a contiguous cell range
    # (flow_amd/parallel.py)
    cells = cells[numpy.argsort(cells.min(axis=1), kind='stable')]
    # drop unused vertices, keeping the x-major order
    used = numpy.zeros(len(pts), dtype=bool)
    used[cells.ravel()] = True
    new_id = numpy.cumsum(used) - 1
    return Mesh(pts[used], new_id[cells])


def rectangle_with_fitted_hole(
        x0, x1, y0, y1, centre, radius, nx, ny, diagonal='right', blend=3.0,
        shrink=1.15
        ):
    '''The same structured grid with a BODY-FITTED circular hole: a square
    block of grid cells (half-side a ~ radius / shrink, centred on the grid
    vertex next to `centre`) is removed and the grid around it is pulled onto
    the circle --
    along every ray from the block's centre the point at max-norm distance m
    (the block's boundary is m = a) moves to
        d' = d * ((1 - t) * (radius / a) * (m / |d|) + t),  t = (m - a)/(A - a),
    for a <= m <= A = blend * a and stays where it is beyond: the block's
    boundary lands on the circle (with shrink = 1.15 its edge midpoints move out
    by 15 %, its corners in by 19 %), the map is monotone along rays (no cell
    folds) and the identity outside the blend zone; the quads of the zone are
    cut along the diagonal that points away from the centre.  At 2182 x 509:
    angles between 25 and 101 degrees, cell areas 0.74 .. 1.28 of the uniform
    cell's, longest edge 1.25 x the uniform diagonal.  Connectivity, x-major numbering and
    cell order are those of `rectangle_with_hole`; the boundary is a polygon
    with its vertices ON the circle (O(h^2) from it, where the staircase is
    O(h) and leaves one-cell notches).  The circle's centre moves to the
    nearest grid vertex (< h/2).'''
    hx, hy = (x1 - x0) / nx, (y1 - y0) / ny
    ic = int(round((centre[0] - x0) / hx))
    jc = int(round((centre[1] - y0) / hy))
    ka = max(1, int(round(radius / hx / shrink)))
    la = max(1, int(round(radius / hy / shrink)))
    cx, cy = x0 + ic * hx, y0 + jc * hy
    ax, ay = ka * hx, la * hy
    assert x0 < cx - blend * ax and cx + blend * ax < x1, 'blend zone in x'
    assert y0 < cy - blend * ay and cy + blend * ay < y1, 'blend zone in y'
    x = numpy.linspace(x0, x1, nx + 1)
    y = numpy.linspace(y0, y1, ny + 1)
    X, Y = numpy.meshgrid(x, y, indexing='ij')
    pts = numpy.stack([X.ravel(), Y.ravel()], axis=1)

    def vid(ix, iy):
        return ix * (ny + 1) + iy

    cells = _quad_cells(nx, ny, diagonal, vid)
    # inside the blend zone the quads are cut along the diagonal that points
    # away from the centre (the map squeezes them radially near the block's
    # corners: the other diagonal would leave 170-degree angles there)
    other = _quad_cells(nx, ny, 'left' if diagonal == 'right' else 'right', vid)
    qc = pts[cells].reshape(-1, 2, 3, 2).mean(axis=(1, 2)) - numpy.array([cx, cy])
    qm = numpy.maximum(abs(qc[:, 0]) / ax, abs(qc[:, 1]) / ay)
    radial_is_right = qc[:, 0] * qc[:, 1] > 0.0       # diagonal v0-v3 ~ (1, 1)
    want_right = radial_is_right if diagonal in ('right', 'left') else None
    if want_right is not None:
        flip = (qm < blend + 1.0) & (want_right != (diagonal == 'right'))
        cells = numpy.where(numpy.repeat(flip, 2)[:, None], other, cells)
    # the block: quads [ic-ka, ic+ka) x [jc-la, jc+la)
    cen = pts[cells].mean(axis=1)
    inside = (abs(cen[:, 0] - cx) < ax) & (abs(cen[:, 1] - cy) < ay)
    cells = cells[~inside]
    # pull the neighbourhood onto the circle
    d = pts - numpy.array([cx, cy])
    m = numpy.maximum(abs(d[:, 0]) / ax, abs(d[:, 1]) / ay)     # block: m = 1
    e = numpy.hypot(d[:, 0], d[:, 1])
    zone = (m >= 1.0 - 1.0e-12) & (m < blend)
    t = (m[zone] - 1.0) / (blend - 1.0)
    # on the block's boundary (t = 0) the point goes to distance `radius`
    on_circle = radius / e[zone]
    # in between: the boundary scaling, carried outward with the max-norm
    # distance and blended into the identity
    f = (1.0 - t) * on_circle * m[zone] + t
    pts = pts.copy()
    pts[zone] = numpy.array([cx, cy]) + d[zone] * f[:, None]
    cells = cells[numpy.argsort(cells.min(axis=1), kind='stable')]
    used = numpy.zeros(len(pts), dtype=bool)
    used[cells.ravel()] = True
    new_id = numpy.cumsum(used) - 1
    return Mesh(pts[used], new_id[cells])


def karman_channel(nx, ny=None, diagonal='right', fitted=False, length=0.6):
    '''Channel [0, 0.6] x [-0.07, 0.07] with a circular obstacle of diameter
    0.04 at (0.1, 0.01): tests/test_karman_vortex_street.py:18-23, 35-38.
    fitted: the body-fitted hole of `rectangle_with_fitted_hole` instead of the
    staircase.  length: a longer channel [0, length] (weak-scaling runs).'''
    if ny is None:
        ny = max(2, int(round(nx * 0.14 / length)))
    make = rectangle_with_fitted_hole if fitted else rectangle_with_hole
    return make(0.0, length, -0.07, 0.07, (0.1, 1.0e-2), 0.02, nx, ny, diagonal)


def heater_box(nx, ny=None, diagonal='right', fitted=False):
    '''Box [0, 0.1] x [0, 0.2] with a circular heater of radius 0.02 at
    (0.05, 0.05): tests/test_boussinesq.py:27-30, 62-64 and
    tests/test_sealed_box.py:35-40.  fitted: body-fitted heater (the blend
    zone has to fit between heater and walls: 2.4 block half-sides; needs
    nx >= 12).'''
    if ny is None:
        ny = 2 * nx
    if fitted:
        return rectangle_with_fitted_hole(
            0.0, 0.1, 0.0, 0.2, (0.05, 0.05), 0.02, nx, ny, diagonal, blend=2.4
            )
    return rectangle_with_hole(
        0.0, 0.1, 0.0, 0.2, (0.05, 0.05), 0.02, nx, ny, diagonal
        )


def heater_box_coarse(nsides=9, side_points=1):
    '''The heater box as gmsh meshes it at `lcar = 0.1` -- the setting of the
    reference's golden norms (tests/test_boussinesq.py:84-97): the box is only
    one `lcar` wide, so its short sides stay one segment and its long sides
    get `side_points` interior points; the circle, three arcs with gmsh's
    minimum of points per arc, becomes a polygon of `nsides` = 9 sides (area
    2.8925 r^2: the reference's ||theta|| = 40.2258 = 293 sqrt(|Omega|) up to
    the heating fits that to 1e-4).  Delaunay triangulation (scipy) of those
    points with the polygon cut out.  The reference's exact triangulation
    (gmsh version dependent, not stored anywhere) cannot be reproduced: this
    is the same geometry at the same resolution, for a like-with-like look at
    its goldens, not a pin.'''
    from scipy.spatial import Delaunay
    x0, x1, y0, y1 = 0.0, 0.1, 0.0, 0.2
    cx, cy, r = 0.05, 0.05, 0.02
    pts = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
    for k in range(1, side_points + 1):
        y = y0 + (y1 - y0) * k / (side_points + 1.0)
        pts += [(x0, y), (x1, y)]
    ang = 2.0 * numpy.pi * numpy.arange(nsides) / nsides
    pts += [(cx + r * numpy.cos(a), cy + r * numpy.sin(a)) for a in ang]
    pts = numpy.array(pts)
    tri = Delaunay(pts).simplices
    cen = pts[tri].mean(axis=1)
    # inside the polygon: a triangle of polygon vertices only
    first = 4 + 2 * side_points
    inside = (tri >= first).all(axis=1)
    assert (numpy.hypot(cen[inside, 0] - cx, cen[inside, 1] - cy) < r).all()
    cells = tri[~inside]
    # counter-clockwise
    p = pts[cells]
    det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) \
        - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1])
    cells[det < 0] = cells[det < 0][:, [0, 2, 1]]
    return Mesh(pts, cells.astype(numpy.int32))


def karman_channel_graded(lcar=5.0e-3, lcar_far=None, seed=0, smoothing=10):
    '''The Karman channel as the reference's driver meshes it with gmsh
    (tests/test_karman_vortex_street.py:26-53: rectangle [0, 0.6] x [-0.07,
    0.07] minus the disk of radius 0.02 at (0.1, 0.01), characteristic length
    `lcar`), without gmsh: an UNSTRUCTURED Delaunay triangulation, graded --
    cells of size `lcar` at the cylinder growing to `lcar_far` (default
    4 lcar) over a distance of 0.1 --, in the numbering it was generated in
    (boundary curves first, interior points in lattice order: nothing the fast
    path could lean on; `.reordered()` gives it that).  Deterministic for a
    given seed.  Points: the boundary curves at the local size, a thinned
    hexagonal lattice inside (density ~ 1 / h^2), `smoothing` rounds of
    Laplacian smoothing with re-triangulation.'''
    from scipy.spatial import Delaunay
    x0, x1, y0, y1 = 0.0, 0.6, -0.07, 0.07
    cx, cy, r = 0.1, 0.01, 0.02
    h0 = float(lcar)
    h1 = float(lcar_far) if lcar_far is not None else 4.0 * h0

    def size(p):
        d = numpy.hypot(p[..., 0] - cx, p[..., 1] - cy) - r
        t = numpy.clip(d / 0.1, 0.0, 1.0)
        return h0 + (h1 - h0) * t

    def walk(a, b):
        """Points from a to b (b excluded), spaced by the local size: equal
        steps of the arc length measured in units of h."""
        a, b = numpy.asarray(a, float), numpy.asarray(b, float)
        L = numpy.linalg.norm(b - a)
        t = numpy.linspace(0.0, 1.0, 4001)
        hh = size(a[None, :] + (b - a)[None, :] * t[:, None])
        m = numpy.concatenate([[0.0], numpy.cumsum(
            0.5 * (1.0 / hh[1:] + 1.0 / hh[:-1]) * (L / 4000.0))])
        n = max(1, int(round(m[-1])))
        tk = numpy.interp(numpy.arange(n) * m[-1] / n, m, t)
        return a[None, :] + (b - a)[None, :] * tk[:, None]

    corners = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
    boundary = [walk(corners[k], corners[(k + 1) % 4]) for k in range(4)]
    nc = max(12, int(round(2.0 * numpy.pi * r / h0)))
    ang = 2.0 * numpy.pi * numpy.arange(nc) / nc
    circle = numpy.stack([cx + r * numpy.cos(ang), cy + r * numpy.sin(ang)], axis=1)
    fixed = numpy.concatenate(boundary + [circle])
    nfixed = len(fixed)
    # interior: hexagonal lattice at the finest size, thinned to the local one
    rng = numpy.random.RandomState(seed)
    s0 = h0
    rows = int((y1 - y0) / (0.866 * s0)) + 1
    cols = int((x1 - x0) / s0) + 1
    J, I = numpy.meshgrid(numpy.arange(rows), numpy.arange(cols), indexing='ij')
    cand = numpy.stack([(x0 + (I + 0.5 * (J % 2)) * s0).ravel(),
                        (y0 + J * 0.866 * s0).ravel()], axis=1)
    h = size(cand)
    keep = rng.uniform(size=len(cand)) < (s0 / h)**2
    d_wall = numpy.minimum.reduce([cand[:, 0] - x0, x1 - cand[:, 0],
                                   cand[:, 1] - y0, y1 - cand[:, 1]])
    d_cyl = numpy.hypot(cand[:, 0] - cx, cand[:, 1] - cy) - r
    keep &= (d_wall > 0.6 * h) & (d_cyl > 0.6 * h)
    pts = numpy.concatenate([fixed, cand[keep]])

    def triangulate(pts):
        tri = Delaunay(pts).simplices
        cen = pts[tri].mean(axis=1)
        # inside the hole: the centroid test, and any triangle of circle points
        # only (three neighbours on the polygon: its centroid lies within
        # rounding of the polygon's edge)
        inside = numpy.hypot(cen[:, 0] - cx, cen[:, 1] - cy) < r * numpy.cos(
            numpy.pi / nc) * 0.999
        inside |= ((tri >= nfixed - nc) & (tri < nfixed)).all(axis=1)
        tri = tri[~inside]
        p = pts[tri]
        det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) \
            - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1])
        # (slivers of collinear boundary points)
        tri = tri[numpy.abs(det) > 1e-14]
        det = det[numpy.abs(det) > 1e-14]
        tri[det < 0] = tri[det < 0][:, [0, 2, 1]]
        return tri

    for _ in range(int(smoothing)):
        tri = triangulate(pts)
        # Laplacian smoothing of the interior points: mean of the neighbours
        a = numpy.concatenate([tri[:, 0], tri[:, 1], tri[:, 2],
                               tri[:, 1], tri[:, 2], tri[:, 0]])
        b = numpy.concatenate([tri[:, 1], tri[:, 2], tri[:, 0],
                               tri[:, 0], tri[:, 1], tri[:, 2]])
        acc = numpy.zeros_like(pts)
        cnt = numpy.zeros(len(pts))
        numpy.add.at(acc, a, pts[b])
        numpy.add.at(cnt, a, 1.0)
        new = acc / numpy.maximum(cnt, 1.0)[:, None]
        move = numpy.arange(len(pts)) >= nfixed
        move &= cnt > 0
        # (a point stays clear of the walls and of the cylinder by half the
        # local size: no flat triangles on the boundary)
        hn = size(new)
        ok = (numpy.minimum.reduce([new[:, 0] - x0, x1 - new[:, 0],
                                    new[:, 1] - y0, y1 - new[:, 1]]) > 0.5 * hn) \
            & (numpy.hypot(new[:, 0] - cx, new[:, 1] - cy) - r > 0.5 * hn)
        move &= ok
        pts[move] = 0.5 * pts[move] + 0.5 * new[move]
    tri = triangulate(pts)
    used = numpy.zeros(len(pts), dtype=bool)
    used[tri.ravel()] = True
    new_id = numpy.cumsum(used) - 1
    return Mesh(pts[used], new_id[tri].astype(numpy.int32))
