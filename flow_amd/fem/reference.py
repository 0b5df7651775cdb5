# -*- coding: utf-8 -*-
'''
Reference-triangle tables for the host side: Lagrange P_k nodal lattices and
basis tabulation (k = 0..5), exact quadrature rules, and the constant
reference matrices G[l][i] = int_ref psi_l^(k) phi_i^(deg) that the HIP source
kernels consume.

Conventions (shared with flow_amd/csrc/fem_device.h):
  reference triangle (0,0), (1,0), (0,1); barycentric l0 = 1-xi-eta, l1 = xi,
  l2 = eta.  P2 local dof order = [v0, v1, v2, e0, e1, e2] where edge dof e_i
  sits on the edge opposite vertex i (DOLFIN/UFC ordering, which the reference
  relies on implicitly through FunctionSpace(mesh, 'CG', 2),
  tests/test_navier_stokes.py:282).
'''
import functools

import numpy


def lattice(k):
    '''Equispaced P_k lattice points (xi, eta) on the reference triangle.
    For k = 1 and k = 2 the order matches the local dof order above.
    '''
    if k == 0:
        return numpy.array([[1.0 / 3.0, 1.0 / 3.0]])
    if k == 1:
        return numpy.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    if k == 2:
        return numpy.array([
            [0.0, 0.0], [1.0, 0.0], [0.0, 1.0],
            [0.5, 0.5], [0.0, 0.5], [0.5, 0.0],
            ])
    pts = []
    for j in range(k + 1):
        for i in range(k + 1 - j):
            pts.append([i / float(k), j / float(k)])
    return numpy.array(pts)


def _monomials(k, pts):
    '''Monomial Vandermonde  xi^a eta^b, a+b <= k, and its gradients.'''
    xi = pts[:, 0]
    eta = pts[:, 1]
    cols = []
    dxi = []
    deta = []
    for b in range(k + 1):
        for a in range(k + 1 - b):
            cols.append(xi**a * eta**b)
            dxi.append(a * xi**max(a - 1, 0) * eta**b if a > 0
                       else numpy.zeros_like(xi))
            deta.append(b * xi**a * eta**max(b - 1, 0) if b > 0
                        else numpy.zeros_like(xi))
    return (numpy.array(cols).T, numpy.array(dxi).T, numpy.array(deta).T)


@functools.lru_cache(maxsize=None)
def _coefficients(k):
    V, _, _ = _monomials(k, lattice(k))
    return numpy.linalg.inv(V)


def tabulate(k, pts):
    '''Values of the P_k Lagrange basis at pts: array (npts, nbasis).'''
    pts = numpy.atleast_2d(numpy.asarray(pts, dtype=float))
    if k == 0:
        return numpy.ones((len(pts), 1))
    V, _, _ = _monomials(k, pts)
    return V.dot(_coefficients(k))


def tabulate_grad(k, pts):
    '''Reference gradients: array (npts, nbasis, 2).'''
    pts = numpy.atleast_2d(numpy.asarray(pts, dtype=float))
    if k == 0:
        return numpy.zeros((len(pts), 1, 2))
    _, dxi, deta = _monomials(k, pts)
    C = _coefficients(k)
    return numpy.stack([dxi.dot(C), deta.dot(C)], axis=-1)


@functools.lru_cache(maxsize=None)
def triangle_rule(degree):
    '''Collapsed Gauss-Jacobi rule, exact for total degree <= `degree`.
    Returns (points (nq,2), weights (nq,)), weights sum to 1/2.
    '''
    from scipy.special import roots_jacobi
    n = degree // 2 + 1
    xg, wg = numpy.polynomial.legendre.leggauss(n)
    xj, wj = roots_jacobi(n, 1.0, 0.0)
    # map [-1,1] -> [0,1]
    u = 0.5 * (xj + 1.0)     # collapsed coordinate, weight (1-u)
    wu = 0.25 * wj
    t = 0.5 * (xg + 1.0)
    wt = 0.5 * wg
    pts = []
    wts = []
    for a, wa in zip(u, wu):
        for b, wb in zip(t, wt):
            pts.append([a, b * (1.0 - a)])
            wts.append(wa * wb)
    return numpy.array(pts), numpy.array(wts)


@functools.lru_cache(maxsize=None)
def source_matrix(k, deg):
    '''G[l][i] = int_ref psi_l^(k) phi_i^(deg)  (reference area 1/2 included).
    A coefficient given by its P_k lattice values F_l on a cell contributes
    |detJ| * sum_l F_l G[l][i] to the load vector entry of local dof i.
    '''
    pts, wts = triangle_rule(k + deg)
    psi = tabulate(k, pts)
    phi = tabulate(deg, pts)
    return numpy.einsum('q,ql,qi->li', wts, psi, phi)


@functools.lru_cache(maxsize=None)
def mass_matrix(k):
    '''Reference mass matrix of P_k (reference area 1/2 included).'''
    return source_matrix(k, k)


def nloc(deg):
    return (deg + 1) * (deg + 2) // 2
