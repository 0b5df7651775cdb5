# -*- coding: utf-8 -*-
'''
Fields and coefficients for the host side of the path: `Function` (dofs live in
HBM as one fp64 array), `Constant`, `Expression`, and the conversion of any
coefficient into the per-cell P_k lattice values the HIP source-term kernel
reads.  These are the objects `step()` receives in the reference
(flow/navier_stokes/pressure_correction.py:533-542; callers
tests/test_navier_stokes.py:312-320, tests/test_karman_vortex_street.py:229-240).

FEniCS semantics restated: `Expression(..., degree=k)` is interpolated into P_k
on every cell before it is integrated.
'''
import numpy
import torch

from . import reference
from .space import FunctionSpace
from .. import _hip
from .. import device


class Constant(object):
    def __init__(self, value):
        self._values = numpy.atleast_1d(numpy.asarray(value, dtype=float))
        return

    def values(self):
        return self._values

    def value_dim(self):
        return len(self._values)

    def assign(self, value):
        if isinstance(value, Constant):
            value = value.values()
        self._values = numpy.atleast_1d(numpy.asarray(value, dtype=float))

    def __float__(self):
        assert len(self._values) == 1
        return float(self._values[0])

    def __mul__(self, other):
        return Constant(self._values * float(other))

    __rmul__ = __mul__

    def __truediv__(self, other):
        return Constant(self._values / float(other))


def scalar_value(c):
    '''float of a Constant (`.values()[0]`, cf. pressure_correction.py:488-489)
    or of a plain number (rho may be a float: tests/test_boussinesq.py:246).'''
    if hasattr(c, 'values'):
        return float(c.values()[0])
    return float(c)


_NAMESPACE = {
    'sin': numpy.sin, 'cos': numpy.cos, 'tan': numpy.tan, 'exp': numpy.exp,
    'log': numpy.log, 'sqrt': numpy.sqrt, 'pow': numpy.power,
    'tanh': numpy.tanh, 'sinh': numpy.sinh, 'cosh': numpy.cosh,
    'fabs': numpy.abs, 'abs': numpy.abs, 'pi': numpy.pi, 'M_PI': numpy.pi,
    'atan2': numpy.arctan2, 'atan': numpy.arctan,
    }


class Expression(object):
    '''`Expression(code, degree=k, **params)`.

    `code` is a Python callable f(x) (x has shape (2, n); returns an array or a
    tuple of arrays), a C-like string such as the ones `sympy.ccode` emits
    (`sin(x[0] + t)*pow(x[1], 2)`), or a tuple of those (vector valued).
    Parameters (e.g. `t`) are attributes and may be changed between calls
    (tests/test_navier_stokes.py:303-311).
    '''
    def __init__(self, code, degree=None, element=None, cell=None,
                 **params):
        assert degree is not None or element is not None
        object.__setattr__(self, 'user_parameters', dict(params))
        self.cppcode = code
        self.degree = degree if degree is not None else element.degree()
        self._element = element
        self._codes = list(code) if isinstance(code, (tuple, list)) else [code]
        self._compiled = [
            compile(c, '<expression>', 'eval') if isinstance(c, str) else c
            for c in self._codes
            ]
        return

    def __setattr__(self, name, value):
        if name in self.__dict__.get('user_parameters', {}):
            self.user_parameters[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        params = self.__dict__.get('user_parameters', {})
        if name in params:
            return params[name]
        raise AttributeError(name)

    def ufl_element(self):
        return self._element if self._element is not None else \
            _ExprElement(self.degree)

    def value_dim(self):
        if len(self._codes) == 1 and callable(self._compiled[0]):
            return self.eval(numpy.zeros((2, 1))).shape[0]
        return len(self._codes)

    def eval(self, x):
        '''x: (2, n) -> (value_dim, n).'''
        x = numpy.asarray(x, dtype=float)
        n = x.shape[1]
        ns = dict(_NAMESPACE)
        ns.update(self.user_parameters)
        ns['x'] = x
        rows = []
        for c in self._compiled:
            if callable(c):
                val = c(x, **self.user_parameters) \
                    if self.user_parameters else c(x)
            else:
                val = eval(c, {'__builtins__': {}}, ns)   # noqa: S307
            if isinstance(val, (tuple, list)) or (
                    isinstance(val, numpy.ndarray) and val.ndim == 2):
                rows.extend(numpy.broadcast_to(v, (n,)) for v in val)
            else:
                rows.append(numpy.broadcast_to(val, (n,)))
        return numpy.array(rows, dtype=float)


class _ExprElement(object):
    def __init__(self, degree):
        self._degree = degree

    def degree(self):
        return self._degree


class Vector(object):
    '''Thin view of a Function's dof array (numpy-like element access; the
    data stays on the device).'''

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return self.data.numel()

    def size(self):
        return self.data.numel()

    def get_local(self):
        return device.to_host(self.data).numpy().copy()

    def set_local(self, values):
        self.data.copy_(torch.as_tensor(
            numpy.ascontiguousarray(values, dtype=numpy.float64)
            ))
        device.synchronize()

    def __getitem__(self, idx):
        if isinstance(idx, slice) and idx == slice(None):
            return _VectorSlice(self)
        return self.get_local()[idx]

    def __setitem__(self, idx, value):
        if isinstance(value, _VectorSlice):
            value = value.vec
        if isinstance(value, Vector):
            value = value.data
        if isinstance(idx, slice) and idx == slice(None):
            if isinstance(value, torch.Tensor):
                _hip.copy(self.data, value)
            elif numpy.isscalar(value):
                _hip.fill(self.data, float(value))
            else:
                self.set_local(value)
            return
        arr = self.get_local()
        arr[idx] = value
        self.set_local(arr)

    def _binary(self, other, op):
        if isinstance(other, Vector):
            other = other.data
        return Vector(op(self.data, other))

    def __sub__(self, other):
        return self._binary(other, torch.sub)

    def __add__(self, other):
        return self._binary(other, torch.add)

    def __mul__(self, other):
        return Vector(self.data * float(other))

    __rmul__ = __mul__

    def norm(self, kind):
        from . import ops
        return ops.vector_norm(self.data, kind)


class _VectorSlice(object):
    '''Result of `vec[:]`: supports `vec[:] += alpha`, `vec[:] = other[:]`.'''

    def __init__(self, vec):
        self.vec = vec

    def __iadd__(self, alpha):
        if isinstance(alpha, (Vector, _VectorSlice)):
            alpha = alpha.vec.data if isinstance(alpha, _VectorSlice) \
                else alpha.data
            self.vec.data.add_(alpha)
        else:
            self.vec.data.add_(float(alpha))
        return self

    def __isub__(self, alpha):
        if isinstance(alpha, (Vector, _VectorSlice)):
            alpha = alpha.vec.data if isinstance(alpha, _VectorSlice) \
                else alpha.data
            self.vec.data.sub_(alpha)
        else:
            self.vec.data.sub_(float(alpha))
        return self

    def __array__(self, dtype=None, copy=None):
        return self.vec.get_local()


class Function(object):
    def __init__(self, V, data=None):
        assert isinstance(V, FunctionSpace)
        assert V.component is None, 'Function on a sub-space view'
        self._V = V
        if data is None:
            self.data = device.zeros(V.size())
        else:
            assert data.numel() == V.size() and data.dtype == torch.float64
            self.data = data
        self._name = 'f'
        return

    def function_space(self):
        return self._V

    def vector(self):
        return Vector(self.data)

    def array(self):
        return device.to_host(self.data).numpy().copy()

    def set_array(self, values):
        values = numpy.ascontiguousarray(values, dtype=numpy.float64)
        assert values.shape == (self._V.size(),)
        self.data.copy_(torch.from_numpy(values))
        device.synchronize()

    def assign(self, other):
        if isinstance(other, Function):
            assert self._V.same_as(other._V)
            _hip.copy(self.data, other.data)
        elif isinstance(other, Constant):
            vals = other.values()
            assert len(vals) == self._V.dim
            n = self._V.N
            for c in range(self._V.dim):
                _hip.fill(self.data[c * n:(c + 1) * n], float(vals[c]))
        else:
            raise TypeError('cannot assign %r' % type(other))

    def copy(self, deepcopy=True):
        assert deepcopy
        return Function(self._V, _hip.clone(self.data))

    def rename(self, name, _label=None):
        self._name = name

    def name(self):
        return self._name

    def split(self, deepcopy=True):
        '''Component functions of a vector field (copies).'''
        assert self._V.dim == 2
        n = self._V.N
        S = self._V.collapse()
        return tuple(
            Function(S, _hip.clone(self.data[c * n:(c + 1) * n]))
            for c in range(2)
            )

    def ufl_element(self):
        return self._V.ufl_element()

    def value_dim(self):
        return self._V.dim

    def __bool__(self):
        # `if p0:` is always true for a Function
        # (flow/navier_stokes/pressure_correction.py:308)
        return True

    __nonzero__ = __bool__


class NodalExpression(object):
    '''Pointwise function of P_k fields, re-interpolated into P_k on every cell
    (nodal evaluation): stands in for UFL expressions such as
    `rho(theta_prev) * g` handed to step() as `f`
    (tests/test_boussinesq.py:248-249).  `func(*nodal_values)` must accept and
    return torch tensors; `scale` is an optional constant vector factor.'''

    def __init__(self, func, args, scale=None):
        self.func = func
        self.args = list(args)
        self.scale = None if scale is None else \
            numpy.atleast_1d(numpy.asarray(scale, dtype=float))
        V = self.args[0].function_space()
        assert all(a.function_space().same_as(V) for a in self.args)
        assert V.dim == 1
        self._V = V

    def value_dim(self):
        return 1 if self.scale is None else len(self.scale)

    def __mul__(self, other):
        vals = other.values() if isinstance(other, Constant) else \
            numpy.atleast_1d(numpy.asarray(other, dtype=float))
        if self.scale is not None:
            assert len(vals) == 1
            vals = self.scale * vals[0]
        return NodalExpression(self.func, self.args, vals)

    __rmul__ = __mul__


def cell_lattice_points(mesh, k):
    '''Physical coordinates of the P_k lattice on every cell: (Nc, nl, 2).'''
    lat = reference.lattice(k)
    lam = numpy.stack(
        [1.0 - lat[:, 0] - lat[:, 1], lat[:, 0], lat[:, 1]], axis=1
        )                                           # (nl, 3)
    p = mesh.points[mesh.cell_vertices]             # (Nc, 3, 2)
    return numpy.einsum('lv,cvd->cld', lam, p)


class CellCoefficient(object):
    '''Per-cell P_k lattice values of a coefficient, device resident:
    `values` has shape (dim, nl, Nc_eff) with the cell index fastest;
    `cell_stride` is 0 for spatially constant data (Nc_eff = 1), else 1.'''

    def __init__(self, k, dim, values, cell_stride):
        self.k = k
        self.dim = dim
        self.values = values
        self.cell_stride = cell_stride
        self.nl = reference.nloc(k)


_CONSTANT_COEFFICIENTS = {}


def as_cell_coefficient(f, mesh, dim):
    '''Convert Constant / Expression / Function / NodalExpression / tuple of
    numbers to a CellCoefficient with `dim` components.'''
    if isinstance(f, CellCoefficient):
        assert f.dim == dim
        return f
    if isinstance(f, (tuple, list, float, int)):
        f = Constant(f)
    if isinstance(f, Constant):
        vals = f.values()
        assert len(vals) == dim, (len(vals), dim)
        # (uploaded once per value: an upload drains the stream, and time
        # loops pass the same forcing constants every step)
        key = (tuple(float(v) for v in vals), str(device.get()))
        if key not in _CONSTANT_COEFFICIENTS:
            if len(_CONSTANT_COEFFICIENTS) >= 64:
                _CONSTANT_COEFFICIENTS.clear()
            _CONSTANT_COEFFICIENTS[key] = device.to_device(
                vals.reshape(dim, 1, 1).copy())
        return CellCoefficient(0, dim, _CONSTANT_COEFFICIENTS[key], 0)
    if isinstance(f, Expression):
        assert f.value_dim() == dim
        k = int(f.degree)
        assert 0 <= k <= 5, 'Expression degree must be <= 5'
        X = cell_lattice_points(mesh, k)            # (Nc, nl, 2)
        nc, nl = X.shape[:2]
        vals = f.eval(X.reshape(-1, 2).T)           # (dim, Nc*nl)
        vals = vals.reshape(dim, nc, nl).transpose(0, 2, 1)
        return CellCoefficient(k, dim, device.to_device(vals), 1)
    if isinstance(f, Function):
        V = f.function_space()
        assert V.dim == dim and V.mesh() is mesh
        cd = V.layout.dev('cell_dofs').long()       # (nloc, Nc)
        n = V.N
        vals = torch.stack([f.data[c * n:(c + 1) * n][cd] for c in range(dim)])
        return CellCoefficient(V.degree, dim, vals.contiguous(), 1)
    if isinstance(f, NodalExpression):
        assert f.value_dim() == dim
        V = f._V
        cd = V.layout.dev('cell_dofs').long()
        nodal = f.func(*[a.data for a in f.args])   # (N,)
        cellv = nodal[cd]                           # (nloc, Nc)
        scale = f.scale if f.scale is not None else numpy.ones(1)
        vals = torch.stack([cellv * float(s) for s in scale])
        return CellCoefficient(V.degree, dim, vals.contiguous(), 1)
    raise TypeError('unsupported coefficient %r' % type(f))
