# -*- coding: utf-8 -*-
'''
An INDEPENDENT front end under the oracle.  Every other parity test hands the
oracle the product's own `layout.cell_dofs`, `collect(bcs)`, `Expression.eval`
and `cell_lattice_points`: a mistake in the P2 numbering, in the Dirichlet
search or in the interpolation of an expression would be common to both sides
and invisible.  Here the oracle's inputs are rebuilt from `mesh.points` and
`mesh.cell_vertices` ALONE, in plain numpy, with the FEniCS semantics SURVEY
section 8c lists (reference: tests/test_karman_vortex_street.py:128-145, 190-203
for the conditions):

  * P2 dof table: vertices first, then one dof per edge (numbered by the sorted
    key of its end points), local order = 3 vertices + the edges opposite them;
  * Dirichlet search: topological -- a boundary facet (an edge with one cell)
    belongs to a sub-domain when both its vertices and its mid point are
    inside; its dofs are its two vertex dofs and its edge dof; component-wise
    conditions touch one component; LATER conditions in the list override
    earlier ones on shared dofs;
  * `Expression(degree=2)` forcing: per cell its values at the six P2 nodes of
    that cell (the oracle integrates the per-cell interpolant).

CPU: the product's front end produces the same spaces, Dirichlet data and
forcing lattices (compared through the dof permutation that matches
coordinates).  GPU: the product's step against the oracle fed by THIS front end.
'''
import numpy
import pytest

from flow_amd import fem

import cases
import mms


# -- the independent front end (numpy only) ------------------------------------
def p2_space(points, cells):
    '''(cell_dofs (Nc, 6), N, dof coordinates, (edge end points))'''
    nv = len(points)
    c = cells.astype(numpy.int64)
    # edge e of a cell is the one opposite its vertex e
    a = numpy.stack([c[:, 1], c[:, 0], c[:, 0]], axis=1)
    b = numpy.stack([c[:, 2], c[:, 2], c[:, 1]], axis=1)
    key = numpy.minimum(a, b) * nv + numpy.maximum(a, b)
    uniq, inv = numpy.unique(key.ravel(), return_inverse=True)
    cell_dofs = numpy.concatenate([c, nv + inv.reshape(-1, 3)], axis=1)
    ends = numpy.stack([uniq // nv, uniq % nv], axis=1)
    coords = numpy.concatenate([points, 0.5 * (points[ends[:, 0]]
                                               + points[ends[:, 1]])])
    return cell_dofs, nv + len(uniq), coords, ends


def boundary_facets(points, cells):
    '''(end points (nf, 2)) of the edges that belong to exactly one cell'''
    nv = len(points)
    c = cells.astype(numpy.int64)
    a = numpy.concatenate([c[:, 1], c[:, 0], c[:, 0]])
    b = numpy.concatenate([c[:, 2], c[:, 2], c[:, 1]])
    key = numpy.minimum(a, b) * nv + numpy.maximum(a, b)
    uniq, cnt = numpy.unique(key, return_counts=True)
    on = uniq[cnt == 1]
    return numpy.stack([on // nv, on % nv], axis=1)


def dirichlet(points, cells, degree, conditions, n_scalar):
    '''conditions: [(inside(x) -> bool mask for x of shape (2, n), component or
    None (all), value(x) -> (ncomp, n))].  -> sorted (dofs, values) in the
    component-blocked numbering a * n_scalar + i; later conditions win.'''
    nv = len(points)
    facets = boundary_facets(points, cells)
    edge_dof = None
    if degree == 2:
        _, _, _, ends = p2_space(points, cells)
        edge_key = ends[:, 0] * nv + ends[:, 1]
        fkey = facets[:, 0] * nv + facets[:, 1]
        edge_dof = nv + numpy.searchsorted(edge_key, fkey)
        assert (edge_key[edge_dof - nv] == fkey).all()
    table = {}
    for inside, comp, value in conditions:
        pa, pb = points[facets[:, 0]], points[facets[:, 1]]
        marked = inside(pa.T) & inside(pb.T) & inside((0.5 * (pa + pb)).T)
        dofs = [facets[marked, 0], facets[marked, 1]]
        xs = [pa[marked], pb[marked]]
        if degree == 2:
            dofs.append(edge_dof[marked])
            xs.append(0.5 * (pa + pb)[marked])
        dofs = numpy.concatenate(dofs)
        xs = numpy.concatenate(xs)
        vals = numpy.atleast_2d(value(xs.T))
        comps = range(vals.shape[0]) if comp is None else [comp]
        for k, a in enumerate(comps):
            row = vals[k] if comp is None else vals[0]
            for d, v in zip(dofs, row):
                table[a * n_scalar + int(d)] = float(v)     # later wins
    keys = numpy.array(sorted(table), dtype=numpy.int64)
    return keys, numpy.array([table[k] for k in keys])


_P2_NODES = numpy.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0],
                         [0.5, 0.5], [0.0, 0.5], [0.5, 0.0]])


def forcing_lattice(points, cells, fun):
    '''(reference nodes, per-cell values (Nc, 6, dim)) of fun(x) -> (dim, n)'''
    p = points[cells]
    X = p[:, None, 0, :] \
        + _P2_NODES[None, :, 0, None] * (p[:, None, 1, :] - p[:, None, 0, :]) \
        + _P2_NODES[None, :, 1, None] * (p[:, None, 2, :] - p[:, None, 0, :])
    nc = len(cells)
    v = fun(X.reshape(-1, 2).T)
    return _P2_NODES, numpy.ascontiguousarray(
        v.reshape(v.shape[0], nc, 6).transpose(1, 2, 0))


def match(coords_a, coords_b):
    '''perm with coords_a[perm] == coords_b (same point sets)'''
    def order(c):
        return numpy.lexsort((numpy.round(c[:, 1], 10), numpy.round(c[:, 0], 10)))
    oa, ob = order(coords_a), order(coords_b)
    perm = numpy.empty(len(coords_b), dtype=numpy.int64)
    perm[ob] = oa
    assert abs(coords_a[perm] - coords_b).max() < 1e-12
    return perm


# -- the two cases -----------------------------------------------------------------
def _setups():
    pb = mms.guermond2()
    dt = 0.05
    yield ('channel', fem.karman_channel(30, 10, fitted=True),
           dict(dt=0.02, bc_kind='channel', rho=1.5, mu=0.05, seed=1), pb)
    yield ('square', fem.UnitSquareMesh(8, 8, 'crossed'),
           dict(dt=dt, bc_kind='all', rho=1.0, mu=1.0, seed=2), pb)


def _conditions(kind, points, pb, dt):
    eps = 1e-12
    x0, y0 = points.min(axis=0)
    x1, y1 = points.max(axis=0)
    everywhere = lambda x: numpy.ones(x.shape[1], dtype=bool)
    if kind == 'all':
        return [(everywhere, None, lambda x: pb.u(x, dt))], []
    walls = lambda x: (x[1] < y0 + eps) | (x[1] > y1 - eps)
    sides = lambda x: (x[0] < x0 + eps) | (x[0] > x1 - eps)
    right = lambda x: x[0] > x1 - eps
    zero2 = lambda x: numpy.zeros((2, x.shape[1]))
    u_cond = [(walls, None, zero2),
              (sides, 0, lambda x: pb.u(x, dt)[:1])]
    p_cond = [(right, None, lambda x: numpy.zeros((1, x.shape[1])))]
    return u_cond, p_cond


def _independent_inputs(case, kind, pb):
    m = case.mesh
    pts, cells = m.points, m.cell_vertices
    cd2, n2, x2, _ = p2_space(pts, cells)
    u_cond, p_cond = _conditions(case.bc_kind, pts, pb, case.dt)
    u_bc = dirichlet(pts, cells, 2, u_cond, n2)
    p_bc = dirichlet(pts, cells, 1, p_cond, len(pts)) if p_cond else None
    f0 = forcing_lattice(pts, cells, lambda x: pb.f(x, 0.0))
    f1 = forcing_lattice(pts, cells, lambda x: pb.f(x, case.dt))
    return cd2, n2, x2, u_bc, p_bc, f0, f1


@pytest.mark.parametrize('which', [0, 1])
def test_the_products_front_end_agrees_with_the_independent_one(which):
    kind, mesh, kw, pb = list(_setups())[which]
    case = cases.Case(mesh, vdeg=2, **kw)
    cd2, n2, x2, u_bc, p_bc, f0, f1 = _independent_inputs(case, kind, pb)
    lay = case.W.layout
    assert n2 == lay.N == mesh.num_vertices() + mesh.num_edges()
    perm = match(lay.dof_coords, x2)        # product dof of every own dof
    # the same cells own the same dofs (as sets; the local order is a
    # convention each side keeps with its own basis)
    assert numpy.array_equal(numpy.sort(perm[cd2], axis=1),
                             numpy.sort(lay.cell_dofs.astype(numpy.int64), axis=1))
    # ... and the local order IS the same convention: 3 vertices, then the
    # edges opposite them
    assert numpy.array_equal(perm[cd2], lay.cell_dofs)
    # Dirichlet data: same dofs, same values, component by component
    (d_prod, v_prod), p_prod = case.bc_data()
    own = {int((k // n2) * n2 + perm[k % n2]): v for k, v in zip(*u_bc)}
    assert sorted(own) == [int(d) for d in d_prod]
    assert abs(numpy.array([own[int(d)] for d in d_prod]) - v_prod).max() < 1e-13
    if p_bc is None:
        assert p_prod is None
    else:
        pperm = match(case.P.layout.dof_coords, mesh.points)
        assert sorted(pperm[p_bc[0]]) == sorted(int(d) for d in p_prod[0])
        assert abs(p_bc[1]).max() == 0.0 and abs(p_prod[1]).max() == 0.0
    # forcing: the per-cell interpolants agree at the six nodes (each side's
    # lattice order is its own: compare through the points they sit at)
    lat_pts, vals = case.lattice(case.f0)
    assert len(lat_pts) == 6
    for k, xi in enumerate(lat_pts):
        j = int(numpy.argmin(abs(f0[0] - xi).sum(axis=1)))
        assert abs(f0[0][j] - xi).max() < 1e-14
        assert abs(vals[:, k, :] - f0[1][:, j, :]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('which', [0, 1])
def test_step_against_the_oracle_behind_the_independent_front_end(hip, which):
    from oracle import fem_oracle as orc
    kind, mesh, kw, pb = list(_setups())[which]
    case = cases.Case(mesh, vdeg=2, **kw)
    cd2, n2, x2, u_bc, p_bc, f0, f1 = _independent_inputs(case, kind, pb)
    pts, cells = mesh.points, mesh.cell_vertices
    W = orc.Space(pts, cells, cd2, 2, n2)
    P = orc.Space(pts, cells, cells, 1, len(pts))
    perm = match(case.W.layout.dof_coords, x2)
    pperm = match(case.P.layout.dof_coords, pts)
    # the case's seeded input fields, carried into the own numbering
    u0 = numpy.concatenate([case.u0[:n2][perm], case.u0[n2:][perm]])
    p0 = case.p0[pperm]
    u1o, p1o, uio = orc.step(W, P, u0, p0, f0, f1, u_bc, p_bc, case.rho, case.mu,
                             case.dt, scheme='rotational')
    u1, p1, ui = case.product_step('rotational')
    own = lambda f: numpy.concatenate([f[:n2][perm], f[n2:][perm]])
    if p_bc is None:
        mass = orc.mass_matrix(P)
        p1o = cases.mean_free(p1o, mass)
        p1c = cases.mean_free(p1[pperm], mass)
    else:
        p1c = p1[pperm]
    errs = (cases.rel_l2(own(ui), uio), cases.rel_l2(p1c, p1o),
            cases.rel_l2(own(u1), u1o))
    print('%s: rel-L2 vs the oracle behind the independent front end: '
          'ui %.2e p1 %.2e u1 %.2e' % ((kind,) + errs))
    assert max(errs) < 1e-7, errs
