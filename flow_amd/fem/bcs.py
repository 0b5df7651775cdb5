# -*- coding: utf-8 -*-
'''
Dirichlet conditions: `SubDomain` / `DirichletBC` with the topological
(facet-based) dof search legacy DOLFIN uses by default -- a boundary facet is
marked when `inside(x, on_boundary)` holds for all of its vertices and its
midpoint; the marked dofs are the facet's vertex dofs and (P2) its edge dof.
Later conditions in a list override earlier ones on shared dofs.

Call sites in the reference: `bcs=u_bcs` / `bcs=p_bcs` in
flow/navier_stokes/pressure_correction.py:226,327,452 and `bc.apply(A, b)` in
flow/heat.py:113-114; construction in tests/test_karman_vortex_street.py:128-145,
190-203 (component-wise `W.sub(0)` conditions) and
tests/test_navier_stokes.py:305-306 ('on_boundary').
'''
import numpy

from .function import Constant, Expression, Function


class SubDomain(object):
    def inside(self, x, on_boundary):   # pragma: no cover - user overrides
        raise NotImplementedError


def _eval_inside(where, x):
    '''x: (n, 2) -> bool (n,).  Tries a vectorised call first (x[0], x[1] are
    arrays), falls back to point-by-point evaluation.'''
    n = len(x)
    if isinstance(where, str):
        assert where == 'on_boundary'
        return numpy.ones(n, dtype=bool)
    fun = where.inside if hasattr(where, 'inside') else where
    try:
        res = fun(x.T, True)
        res = numpy.broadcast_to(numpy.asarray(res, dtype=bool), (n,))
        return res.copy()
    except (ValueError, TypeError):
        return numpy.array([bool(fun(xi, True)) for xi in x], dtype=bool)


class DirichletBC(object):
    def __init__(self, V, value, where):
        self.V = V
        self.where = where
        self.value = value
        if V.component is not None:
            self.space = V.parent
            self.components = [V.component]
        else:
            self.space = V
            self.components = list(range(V.dim))
        self._facet_cache = None
        self._value_cache = (None, None)
        return

    def function_space(self):
        return self.V

    def signature(self):
        '''Hashable description of the boundary data, or None when it cannot
        be told whether they changed (Functions, Python callables): time
        loops hand the same conditions to every step, and the merged
        (dofs, values) arrays are then reused (`collect`).'''
        value = self.value
        try:
            if isinstance(value, (tuple, list, float, int)):
                return ('c', tuple(numpy.ravel(value).tolist()))
            if isinstance(value, Constant):
                return ('c', tuple(value.values().tolist()))
            if isinstance(value, Expression) and all(
                    isinstance(c, str) for c in value._codes):
                sig = ('e', tuple(value._codes),
                       tuple(sorted(value.user_parameters.items())))
                hash(sig)
                return sig
        except TypeError:
            pass
        return None

    def _scalar_dofs(self):
        '''Scalar dof ids on the marked boundary facets (sorted, unique).'''
        if self._facet_cache is None:
            mesh = self.space.mesh()
            layout = self.space.layout
            bf = mesh.bfacets
            ev = mesh.edges[bf]                         # (Nb, 2)
            p0 = mesh.points[ev[:, 0]]
            p1 = mesh.points[ev[:, 1]]
            marked = (
                _eval_inside(self.where, p0)
                & _eval_inside(self.where, p1)
                & _eval_inside(self.where, 0.5 * (p0 + p1))
                )
            ev = ev[marked]
            dofs = [layout.vertex_dofs[ev.ravel()]]
            if layout.degree == 2:
                dofs.append(layout.edge_dofs[bf[marked]])
            self._facet_cache = numpy.unique(numpy.concatenate(dofs))
        return self._facet_cache

    def dofs_and_values(self):
        '''(dofs, values) in the dof numbering of the (parent) space.'''
        sig = self.signature()
        if sig is not None and self._value_cache[0] == sig:
            return self._value_cache[1]
        out = self._dofs_and_values()
        self._value_cache = (sig, out)
        return out

    def _dofs_and_values(self):
        sd = self._scalar_dofs()
        n = self.space.N
        x = self.space.layout.dof_coords[sd]            # (m, 2)
        value = self.value
        ncomp = len(self.components)
        if isinstance(value, (tuple, list, float, int)):
            value = Constant(value)
        if isinstance(value, Constant):
            v = value.values()
            assert len(v) == ncomp, (len(v), ncomp)
            vals = numpy.repeat(v[:, None], len(sd), axis=1)
        elif isinstance(value, Expression):
            vals = value.eval(x.T)
            assert vals.shape[0] == ncomp
        elif isinstance(value, Function):
            arr = value.array()
            nv = value.function_space().N
            assert nv == n
            vals = numpy.stack([
                arr[k * nv + sd] for k in range(value.function_space().dim)
                ])
            assert vals.shape[0] == ncomp
        else:
            raise TypeError('unsupported boundary value %r' % type(value))
        dofs = numpy.concatenate([c * n + sd for c in self.components])
        return dofs.astype(numpy.int64), vals.reshape(-1).astype(float)


def collect(bcs, size):
    '''Merge a list of conditions (later ones win) into sorted unique
    (dofs int32, values fp64) arrays.'''
    if not bcs:
        return (numpy.zeros(0, dtype=numpy.int32), numpy.zeros(0))
    sigs = tuple(
        (id(bc), bc.signature() if hasattr(bc, 'signature') else None)
        for bc in bcs
        )
    cacheable = all(s[1] is not None for s in sigs)
    if cacheable:
        hit = _COLLECTED.get((sigs, size))
        if hit is not None:
            return hit[1]
    dofs_all = []
    vals_all = []
    for bc in bcs:
        d, v = bc.dofs_and_values()
        dofs_all.append(d)
        vals_all.append(v)
    d = numpy.concatenate(dofs_all)
    v = numpy.concatenate(vals_all)
    # keep the LAST occurrence of every dof
    order = numpy.argsort(d, kind='stable')
    d = d[order]
    v = v[order]
    last = numpy.ones(len(d), dtype=bool)
    last[:-1] = d[1:] != d[:-1]
    d = d[last]
    v = v[last]
    assert len(d) == 0 or (d[0] >= 0 and d[-1] < size)
    out = (d.astype(numpy.int32), v)
    if cacheable:
        if len(_COLLECTED) >= 16:
            _COLLECTED.clear()
        # the conditions are kept alive with the entry: their ids stay unique
        _COLLECTED[(sigs, size)] = (list(bcs), out)
    return out


# merged arrays of lists of conditions whose data are unchanged (see signature)
_COLLECTED = {}


class FixedDofsBC(object):
    '''Dirichlet data given directly as (dofs, values) in the numbering of the
    space (replays of recorded cases, golden fixtures).'''

    def __init__(self, V, dofs, values):
        self.V = V
        self._dofs = numpy.asarray(dofs, dtype=numpy.int64)
        self._values = numpy.asarray(values, dtype=float)
        assert self._dofs.shape == self._values.shape

    def function_space(self):
        return self.V

    def dofs_and_values(self):
        return self._dofs, self._values
