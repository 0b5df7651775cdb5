# -*- coding: utf-8 -*-
'''
Manufactured solutions for the temporal-order tests (helper, not a test).

Same analytic fields as the reference harness uses
(tests/test_navier_stokes.py:78-104 `problem_flat`, :136-165
`problem_guermond1`, :168-195 `problem_guermond2`); the forcing is derived
symbolically from the momentum equation
    rho (u_t + (u.grad)u) = -grad p + mu Lap u + f
and everything is lambdified to numpy callables f(x, t) with x of shape (2, n).
'''
import numpy
import sympy

MAX_DEGREE = 5      # the reference truncates Expression degrees to 5 (:21)


class Problem(object):
    def __init__(self, name, u, p, domain, diagonal, u_degree, p_degree,
                 mu=1.0, rho=1.0):
        X, Y, t = sympy.symbols('X Y t')
        self.name = name
        self.domain = domain          # ((x0, y0), (x1, y1))
        self.diagonal = diagonal
        self.mu = mu
        self.rho = rho
        self.u_degree = min(u_degree, MAX_DEGREE)
        self.p_degree = min(p_degree, MAX_DEGREE)
        self.f_degree = MAX_DEGREE
        u = [sympy.sympify(c) for c in u(X, Y, t)]
        p = sympy.sympify(p(X, Y, t))
        div = sympy.simplify(sympy.diff(u[0], X) + sympy.diff(u[1], Y))
        assert div == 0, 'manufactured velocity is not solenoidal'
        f = []
        for a, xa in enumerate((X, Y)):
            f.append(
                rho * (sympy.diff(u[a], t)
                       + u[0] * sympy.diff(u[a], X)
                       + u[1] * sympy.diff(u[a], Y))
                + sympy.diff(p, xa)
                - mu * (sympy.diff(u[a], X, 2) + sympy.diff(u[a], Y, 2))
                )
        self._u = [sympy.lambdify((X, Y, t), c, 'numpy') for c in u]
        self._p = sympy.lambdify((X, Y, t), p, 'numpy')
        self._f = [sympy.lambdify((X, Y, t), c, 'numpy') for c in f]

    @staticmethod
    def _b(val, x):
        return numpy.broadcast_to(numpy.asarray(val, dtype=float), x[0].shape)

    def u(self, x, t):
        return numpy.array([self._b(c(x[0], x[1], t), x) for c in self._u])

    def p(self, x, t):
        return self._b(self._p(x[0], x[1], t), x)[None, :]

    def f(self, x, t):
        return numpy.array([self._b(c(x[0], x[1], t), x) for c in self._f])


def flat():
    return Problem(
        'flat',
        lambda X, Y, t: (0 * X, 0 * Y),
        lambda X, Y, t: -Y,
        ((0.0, 0.0), (1.0, 1.0)), 'left/right', 1, 1
        )


def guermond1():
    pi = sympy.pi
    return Problem(
        'guermond1',
        lambda X, Y, t: (
            +pi * sympy.sin(t) * 2 * sympy.sin(pi * Y) * sympy.cos(pi * Y)
            * sympy.sin(pi * X)**2,
            -pi * sympy.sin(t) * 2 * sympy.sin(pi * X) * sympy.cos(pi * X)
            * sympy.sin(pi * Y)**2,
            ),
        lambda X, Y, t: sympy.sin(t) * sympy.cos(pi * X) * sympy.sin(pi * Y),
        ((-1.0, -1.0), (1.0, 1.0)), 'crossed', MAX_DEGREE, MAX_DEGREE
        )


def guermond2():
    return Problem(
        'guermond2',
        lambda X, Y, t: (
            sympy.sin(X + t) * sympy.sin(Y + t),
            sympy.cos(X + t) * sympy.cos(Y + t),
            ),
        lambda X, Y, t: sympy.sin(X - Y + t),
        ((0.0, 0.0), (1.0, 1.0)), 'crossed', MAX_DEGREE, MAX_DEGREE
        )


def channel():
    '''Time-dependent Poiseuille flow in the unit square, posed like the
    reference's Karman driver (tests/test_karman_vortex_street.py:190-203):
    no-slip walls top and bottom, ONLY the x-velocity prescribed on the left
    and right sides, p = 0 on the right.  The y-velocity rows on the two sides
    are free, so the exterior-facet terms of `_rhs_weak`
    (pressure_correction.py:142-143) act on them -- with a non-zero value:
    mu d_y u_x n_x -- and the scheme is only consistent with this solution if
    those terms carry the right sign and factor: after partial integration the
    form leaves the natural condition  mu d_n u_y = 0  on the free component,
    which u = (g(t) y (1 - y), 0) satisfies, as it does (u.n) u_y = 0 for the
    skew-symmetric convection term.  Velocity and pressure are exactly
    representable (P2 / P1): the errors are those of the time discretisation.'''
    return Problem(
        'channel',
        lambda X, Y, t: ((1 + sympy.sin(t)) * Y * (1 - Y), 0 * X),
        lambda X, Y, t: (2 + sympy.cos(t)) * (1 - X),
        ((0.0, 0.0), (1.0, 1.0)), 'crossed', 2, 1, mu=0.7, rho=1.3
        )
