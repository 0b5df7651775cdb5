# -*- coding: utf-8 -*-
'''
TEST INFRASTRUCTURE -- CPU oracle, not part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; flow_amd/ never does.  It is a plain numpy/scipy restatement of
the reference's algorithm for the hot path (legacy-FEniCS semantics of
flow/navier_stokes/pressure_correction.py and flow/heat.py), written
independently of the product's host code: it works on plain arrays (points,
cells, cell->dof tables handed in as data), tabulates P1/P2 bases from closed
forms, integrates with a Duffy-collapsed Gauss-Legendre rule of generous degree,
assembles through scipy COO->CSR and solves every linear system with a sparse
direct factorisation (the reference: Newton + LU, pressure_correction.py:224-254;
CG+AMG to rtol 1e-10, :326-339, :419-432, :451-464).

Pinning status: the reference cannot be executed offline (dolfin is not
installed).  The oracle is pinned by the reference's own
analytic known-answer tests -- temporal orders of Chorin/IPCS/Rotational
(tests/test_navier_stokes.py:379-446), the hydrostatic rest state
(tests/test_sealed_box.py:141), the SUPG closed form
(flow/stabilization.py:116-130) -- see tests/test_oracle_pinning.py.  The
Boussinesq golden norms (tests/test_boussinesq.py:84-97) are PARITY UNPINNED
(gmsh mesh + absent `materials`/`parabolic` packages).
'''
import numpy
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# -- reference element --------------------------------------------------------
def duffy_rule(n):
    '''n x n collapsed Gauss-Legendre rule on the reference triangle, exact for
    total degree <= 2n-2.  Points (nq,2), weights (nq,) summing to 1/2.'''
    g, w = numpy.polynomial.legendre.leggauss(n)
    g = 0.5 * (g + 1.0)
    w = 0.5 * w
    A, B = numpy.meshgrid(g, g, indexing='ij')
    WA, WB = numpy.meshgrid(w, w, indexing='ij')
    xi = A.ravel()
    eta = (B * (1.0 - A)).ravel()
    wt = (WA * WB * (1.0 - A)).ravel()
    return numpy.stack([xi, eta], axis=1), wt


def bary(pts):
    pts = numpy.atleast_2d(pts)
    return numpy.stack(
        [1.0 - pts[:, 0] - pts[:, 1], pts[:, 0], pts[:, 1]], axis=1
        )


_DLAM = numpy.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])   # d lambda / d(xi,eta)
_EDGE = [(1, 2), (0, 2), (0, 1)]    # edge i is opposite vertex i


def basis(deg, pts):
    '''(values (nq, nloc), reference gradients (nq, nloc, 2)).'''
    L = bary(pts)
    if deg == 1:
        return L.copy(), numpy.broadcast_to(_DLAM, (len(L), 3, 2)).copy()
    assert deg == 2
    val = numpy.empty((len(L), 6))
    grad = numpy.empty((len(L), 6, 2))
    for i in range(3):
        val[:, i] = L[:, i] * (2.0 * L[:, i] - 1.0)
        grad[:, i, :] = (4.0 * L[:, i] - 1.0)[:, None] * _DLAM[i]
    for e, (j, k) in enumerate(_EDGE):
        val[:, 3 + e] = 4.0 * L[:, j] * L[:, k]
        grad[:, 3 + e, :] = 4.0 * (
            L[:, j, None] * _DLAM[k] + L[:, k, None] * _DLAM[j]
            )
    return val, grad


def lagrange_eval(lattice_pts, pts):
    '''Values at `pts` of the Lagrange basis with nodes `lattice_pts` (complete
    polynomial space whose dimension equals the number of nodes).'''
    nl = len(lattice_pts)
    k = int(round((numpy.sqrt(8 * nl + 1) - 3) / 2))
    assert (k + 1) * (k + 2) // 2 == nl

    def vander(p):
        return numpy.stack([
            p[:, 0]**a * p[:, 1]**b
            for b in range(k + 1) for a in range(k + 1 - b)
            ], axis=1)
    return vander(numpy.atleast_2d(pts)).dot(
        numpy.linalg.inv(vander(numpy.atleast_2d(lattice_pts)))
        )


class Space(object):
    '''Scalar P_deg space given as data.'''

    def __init__(self, points, cells, cell_dofs, deg, n_dofs):
        self.points = numpy.asarray(points, dtype=float)
        self.cells = numpy.asarray(cells, dtype=numpy.int64)
        self.cell_dofs = numpy.asarray(cell_dofs, dtype=numpy.int64)
        self.deg = deg
        self.N = int(n_dofs)
        self.nloc = self.cell_dofs.shape[1]
        p = self.points[self.cells]
        self.J = numpy.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]], axis=2)
        self.detJ = (
            self.J[:, 0, 0] * self.J[:, 1, 1] - self.J[:, 0, 1] * self.J[:, 1, 0]
            )
        self.invJ = numpy.linalg.inv(self.J)
        # physical gradients of the barycentric coordinates: (Nc, 3, 2)
        self.glam = numpy.einsum('lr,crd->cld', _DLAM, self.invJ)

    def phys_grad(self, gref):
        '''(nq, nloc, 2) reference -> (Nc, nq, nloc, 2) physical gradients.'''
        return numpy.einsum('qir,crd->cqid', gref, self.invJ)


def _coo(space_r, space_c, Ke, dim_r=1, dim_c=1):
    '''Ke: (Nc, dim_r, nr, dim_c, nc) -> CSR of size (dim_r*Nr, dim_c*Nc);
    vector dof (comp, i) -> comp*N + i.'''
    nr = space_r.nloc
    ncl = space_c.nloc
    R = (numpy.arange(dim_r)[None, :, None] * space_r.N
         + space_r.cell_dofs[:, None, :])
    C = (numpy.arange(dim_c)[None, :, None] * space_c.N
         + space_c.cell_dofs[:, None, :])
    rows = numpy.broadcast_to(
        R[:, :, :, None, None], (len(Ke), dim_r, nr, dim_c, ncl)
        )
    cols = numpy.broadcast_to(
        C[:, None, None, :, :], (len(Ke), dim_r, nr, dim_c, ncl)
        )
    A = sp.coo_matrix(
        (Ke.ravel(), (rows.ravel(), cols.ravel())),
        shape=(dim_r * space_r.N, dim_c * space_c.N)
        )
    return A.tocsr()


def _vec(space, Fe, dim=1):
    '''Fe: (Nc, dim, nloc) -> vector of size dim*N.'''
    idx = (numpy.arange(dim)[None, :, None] * space.N
           + space.cell_dofs[:, None, :])
    out = numpy.zeros(dim * space.N)
    numpy.add.at(out, idx.ravel(), Fe.ravel())
    return out


# -- basic operators ----------------------------------------------------------
def mass_matrix(S):
    pts, w = duffy_rule(4)
    phi, _ = basis(S.deg, pts)
    Ke = numpy.einsum('q,qi,qj,c->cij', w, phi, phi, numpy.abs(S.detJ))
    return _coo(S, S, Ke[:, None, :, None, :])


def stiffness_matrix(S):
    '''a2 = dot(grad(p), grad(q))*dx  (pressure_correction.py:317).'''
    pts, w = duffy_rule(3)
    _, gref = basis(S.deg, pts)
    g = S.phys_grad(gref)
    Ke = numpy.einsum('q,cqid,cqjd,c->cij', w, g, g, numpy.abs(S.detJ))
    return _coo(S, S, Ke[:, None, :, None, :])


def lumped_mass_vertex_rule(S):
    '''u*v*dx with the 'vertex' quadrature scheme (flow/heat.py:39-45): the
    rule has the three cell vertices as points, weights |T|/3.  For P2 only the
    vertex basis functions are non-zero there, so edge rows are zero.'''
    pts = numpy.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    w = numpy.full(3, 1.0 / 6.0)
    phi, _ = basis(S.deg, pts)
    Ke = numpy.einsum('q,qi,qj,c->cij', w, phi, phi, numpy.abs(S.detJ))
    return _coo(S, S, Ke[:, None, :, None, :])


def load_vector(S, lattice_pts, cell_values, dim=1):
    '''int f v dx with f given by per-cell Lagrange lattice values
    (Nc, nl, dim)  (FEniCS: Expression(degree=k) is interpolated per cell).'''
    pts, w = duffy_rule(6)
    phi, _ = basis(S.deg, pts)
    psi = lagrange_eval(lattice_pts, pts)                     # (nq, nl)
    fq = numpy.einsum('ql,cld->cqd', psi, cell_values)
    Fe = numpy.einsum('q,cqd,qi,c->cdi', w, fq, phi, numpy.abs(S.detJ))
    return _vec(S, Fe, dim)


def boundary_facets(S):
    '''(cell, local facet) of all boundary facets, from the cell-vertex table.
    (A Space that is a CHUNK of a mesh's cells -- oracle/cpu_step.py assembles
    in chunks on several threads -- carries the true boundary facets among its
    cells as `bfacets`: a chunk's own outline is not the domain's.)'''
    if getattr(S, 'bfacets', None) is not None:
        return S.bfacets
    c = S.cells
    nv = len(S.points)
    a = numpy.stack([c[:, 1], c[:, 0], c[:, 0]], axis=1)
    b = numpy.stack([c[:, 2], c[:, 2], c[:, 1]], axis=1)
    key = (numpy.minimum(a, b) * nv + numpy.maximum(a, b)).ravel()
    _, inv, cnt = numpy.unique(key, return_inverse=True, return_counts=True)
    flat = numpy.nonzero(cnt[inv] == 1)[0]
    return flat // 3, flat % 3


# -- Navier-Stokes momentum residual -----------------------------------------
def _facet_points(lf, n=3):
    '''Gauss points on local facet lf in cell reference coordinates.'''
    g, w = numpy.polynomial.legendre.leggauss(n)
    s = 0.5 * (g + 1.0)
    w = 0.5 * w
    ref = numpy.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    j, k = _EDGE[lf]
    pts = (1.0 - s)[:, None] * ref[j] + s[:, None] * ref[k]
    return pts, w


def momentum_rhs(W, P, U, p0, f, rho, mu, want_jacobian=True):
    '''R(u; v) of `_rhs_weak` (pressure_correction.py:135-144) for every test
    function, and (optionally) dR/dU.  W: velocity scalar space (deg 1|2),
    U: (2*N,), P: pressure space (deg 1), p0: (Np,), f: (lattice_pts,
    cell_values (Nc, nl, 2)).'''
    nl = W.nloc
    nc = len(W.cells)
    pts, w = duffy_rule(5)
    phi, gref = basis(W.deg, pts)
    gphi = W.phys_grad(gref)                      # (Nc, nq, nl, 2)
    psi, _ = basis(1, pts)                        # pressure basis
    wd = w[None, :] * numpy.abs(W.detJ)[:, None]  # (Nc, nq)
    Uc = numpy.stack([U[W.cell_dofs], U[W.N + W.cell_dofs]], axis=1)   # (Nc,2,nl)
    uq = numpy.einsum('cai,qi->caq', Uc, phi)
    gu = numpy.einsum('cai,cqib->cabq', Uc, gphi)  # d_b u_a
    pq = numpy.einsum('ci,qi->cq', p0[P.cell_dofs], psi)
    fl = lagrange_eval(f[0], pts)
    fq = numpy.einsum('ql,cla->caq', fl, f[1])

    # (f, v)
    R = numpy.einsum('cq,caq,qi->cai', wd, fq, phi)
    # - rho/2 [ ((grad u) u, v) - ((grad v) u, u) ]
    conv = numpy.einsum('cabq,cbq->caq', gu, uq)          # (u.grad) u_a
    ugphi = numpy.einsum('cbq,cqib->cqi', uq, gphi)       # u.grad(phi_i)
    R -= 0.5 * rho * (
        numpy.einsum('cq,caq,qi->cai', wd, conv, phi)
        - numpy.einsum('cq,cqi,caq->cai', wd, ugphi, uq)
        )
    # - (2 mu eps(u) - p0 I, eps(v))
    sym = gu + gu.transpose(0, 2, 1, 3)
    R -= mu * numpy.einsum('cq,cabq,cqib->cai', wd, sym, gphi)
    R += numpy.einsum('cq,cq,cqia->cai', wd, pq, gphi)

    Jc = None
    if want_jacobian:
        d2 = numpy.eye(2)
        Jc = numpy.zeros((nc, 2, nl, 2, nl))
        # -rho/2 [ phi_j d_c u_a phi_i + d_ac (u.grad phi_j) phi_i
        #          - phi_j d_c phi_i u_a - d_ac (u.grad phi_i) phi_j ]
        Jc -= 0.5 * rho * numpy.einsum(
            'cq,qj,cabq,qi->caibj', wd, phi, gu, phi)
        t1 = numpy.einsum('cq,cqj,qi->cij', wd, ugphi, phi)
        Jc -= 0.5 * rho * numpy.einsum('ab,cij->caibj', d2, t1)
        Jc += 0.5 * rho * numpy.einsum(
            'cq,qj,cqib,caq->caibj', wd, phi, gphi, uq)
        Jc += 0.5 * rho * numpy.einsum('ab,cji->caibj', d2, t1)
        # -mu [ d_ac grad phi_j . grad phi_i + d_a phi_j d_c phi_i ]
        K = numpy.einsum('cq,cqjd,cqid->cij', wd, gphi, gphi)
        Jc -= mu * numpy.einsum('ab,cij->caibj', d2, K)
        Jc -= mu * numpy.einsum('cq,cqja,cqib->caibj', wd, gphi, gphi)

    # exterior facets: - p0 n.v ds + mu ((grad u)^T n).v ds
    bc_cells, bc_lf = boundary_facets(W)
    for lf in range(3):
        cs = bc_cells[bc_lf == lf]
        if len(cs) == 0:
            continue
        fpts, fw = _facet_points(lf)
        fphi, fgref = basis(W.deg, fpts)
        fpsi, _ = basis(1, fpts)
        j, k = _EDGE[lf]
        pc = W.points[W.cells[cs]]                          # (n, 3, 2)
        t = pc[:, k] - pc[:, j]
        length = numpy.sqrt((t**2).sum(axis=1))
        nrm = numpy.stack([t[:, 1], -t[:, 0]], axis=1) / length[:, None]
        inward = pc[:, lf] - pc[:, j]
        sign = numpy.where((nrm * inward).sum(axis=1) > 0.0, -1.0, 1.0)
        nrm = nrm * sign[:, None]
        fg = numpy.einsum('qir,crd->cqid', fgref, W.invJ[cs])
        ws = fw[None, :] * length[:, None]
        Ucs = Uc[cs]
        gus = numpy.einsum('cai,cqib->cabq', Ucs, fg)
        pqs = numpy.einsum('ci,qi->cq', p0[P.cell_dofs[cs]], fpsi)
        Rf = -numpy.einsum('cq,cq,ca,qi->cai', ws, pqs, nrm, fphi)
        Rf += mu * numpy.einsum('cq,cbaq,cb,qi->cai', ws, gus, nrm, fphi)
        numpy.add.at(R, cs, Rf)
        if want_jacobian:
            Jf = mu * numpy.einsum('cq,cqja,cb,qi->caibj', ws, fg, nrm, fphi)
            numpy.add.at(Jc, cs, Jf)

    Rv = _vec(W, R, 2)
    Jm = _coo(W, W, Jc, 2, 2) if want_jacobian else None
    return Rv, Jm


def _identity_rows(A, dofs):
    '''`bc.apply(A)`: the rows of the Dirichlet dofs replaced by rows of the
    identity (row scaling with 0/1 + a unit diagonal: the same entries as
    assigning them one by one, in time linear in the nonzeros).'''
    keep = numpy.ones(A.shape[0])
    keep[dofs] = 0.0
    A = sp.diags(keep).dot(A.tocsr()) + sp.diags(1.0 - keep)
    A = A.tocsr()
    A.eliminate_zeros()
    return A


def solve_blockwise(A, b, n, rtol=1.0e-14, max_it=80, single=False):
    '''A x = b for a matrix of 2 x 2 blocks of n rows each -- the SAME discrete
    solution the sparse LU of the whole matrix defines (`linear='lu'`, what the
    reference's default solver computes, pressure_correction.py:224-254),
    reached by full GMRES right-preconditioned with the sparse LUs of the two
    DIAGONAL blocks (block Gauss-Seidel: the second block sees the first one's
    update), restarted on the TRUE residual until ||b - A x|| <= rtol ||b||,
    which is checked at the end.  Exists because SuperLU gives up on the
    coupled Newton matrix beyond ~2.5 M rows ("not enough memory": the index
    range of its factors) while each diagonal block has a quarter of its
    nonzeros; the coupling blocks (the reaction term of the linearisation) are
    small, a handful of iterations per digit-complete solve.  A block-diagonal
    A (the vector mass matrix of the velocity correction) is solved by the two
    LUs directly.  single: the two LUs in fp32 (half the memory -- what lets
    the full-size workload's blocks fit in 62 GB); they only precondition: the
    Krylov vectors, the operator and the residual that is checked stay fp64.'''
    A = A.tocsr()
    A00 = A[:n, :n].tocsc()
    coupled = A[:n, n:].nnz > 0
    A10, A11 = A[n:, :n].tocsr(), A[n:, n:].tocsc()
    # (column ordering: minimum degree on the pattern of A^T + A -- the blocks
    # are structurally symmetric finite-element matrices; half the fill and a
    # quarter of the time of SuperLU's default COLAMD, measured on a 0.15 M-row
    # P2 block: 8.3 against 16.9 times the nonzeros of A.  Diagonal pivots are
    # preferred: mass-dominated rows.  What the factors are worth is checked on
    # the true residual below.)
    opts = dict(permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.1)
    if single:
        lu0 = spla.splu(A00.astype(numpy.float32), **opts)
        del A00
        lu1 = spla.splu(A11.astype(numpy.float32), **opts)
        del A11
        solve0 = lambda v: lu0.solve(v.astype(numpy.float32)).astype(numpy.float64)
        solve1 = lambda v: lu1.solve(v.astype(numpy.float32)).astype(numpy.float64)
    else:
        lu0 = spla.splu(A00, **opts)
        del A00
        lu1 = spla.splu(A11, **opts)
        del A11
        solve0, solve1 = lu0.solve, lu1.solve

    def precondition(r):
        z0 = solve0(r[:n])
        z1 = solve1(r[n:] - A10.dot(z0))
        return numpy.concatenate([z0, z1])
    bn = numpy.linalg.norm(b)
    if bn == 0.0:
        return numpy.zeros_like(b)
    x = precondition(b)
    if not coupled and A10.nnz == 0:
        for _refine in range(1 if not single else 6):
            x = x + precondition(b - A.dot(x))   # (refinement on the true residual)
            if numpy.linalg.norm(b - A.dot(x)) <= 1.0e-13 * bn:
                break
        assert numpy.linalg.norm(b - A.dot(x)) <= 1.0e-12 * bn
        return x
    for _outer in range(6):
        r = b - A.dot(x)
        beta = numpy.linalg.norm(r)
        if beta <= rtol * bn:
            break
        V, Z = [r / beta], []
        H = numpy.zeros((max_it + 1, max_it))
        for j in range(max_it):
            Z.append(precondition(V[j]))
            w = A.dot(Z[j])
            for _pass in range(2):                  # (re-orthogonalised)
                for i in range(j + 1):
                    h = w.dot(V[i])
                    H[i, j] += h
                    w = w - h * V[i]
            H[j + 1, j] = numpy.linalg.norm(w)
            V.append(w / H[j + 1, j])
            g = numpy.zeros(j + 2)
            g[0] = beta
            y = numpy.linalg.lstsq(H[:j + 2, :j + 1], g, rcond=None)[0]
            est = numpy.linalg.norm(g - H[:j + 2, :j + 1].dot(y))
            if est <= 0.1 * rtol * bn or H[j + 1, j] <= 1e-300:
                break
        for yj, zj in zip(y, Z):
            x = x + yj * zj
    res = numpy.linalg.norm(b - A.dot(x))
    if not res <= 10.0 * rtol * bn:
        raise RuntimeError('solve_blockwise: true residual %.2e |b|' % (res / bn))
    return x


_THETA = {
    'forward euler': (0.0, 1.0),
    'backward euler': (1.0, 0.0),
    'crank-nicolson': (0.5, 0.5),
    }


def tentative_velocity(W, P, u0, p0, f0, f1, bc_dofs, bc_vals, method,
                       rho, mu, dt, tol=1.0e-10, max_it=10, linear='lu'):
    '''`_compute_tentative_velocity` (pressure_correction.py:147-255): Newton
    from ui = u0, exact Jacobian, LU, BCs as identity rows with residual x - g,
    converged when ||F||_2 < tol (absolute); RuntimeError otherwise.
    linear: 'lu' = one sparse LU of the Newton matrix; 'block' = the same
    solution through `solve_blockwise` (sizes SuperLU cannot factor whole);
    'block32' = that with the block LUs held in fp32 (the full-size workload).'''
    assert method in _THETA
    th_i, th_e = _THETA[method]
    M = sp.block_diag([mass_matrix(W)] * 2, format='csr')
    Re = None
    if th_e != 0.0:
        Re, _ = momentum_rhs(W, P, u0, p0, f0, rho, mu, want_jacobian=False)
    ui = u0.copy()
    history = []
    for it in range(max_it + 1):
        F = M.dot(ui - u0)
        J = M.copy()
        if th_i != 0.0:
            Ri, dRi = momentum_rhs(W, P, ui, p0, f1, rho, mu)
            F -= dt / rho * th_i * Ri
            J = J - dt / rho * th_i * dRi
        if th_e != 0.0:
            F -= dt / rho * th_e * Re
        F[bc_dofs] = ui[bc_dofs] - bc_vals
        nrm = numpy.linalg.norm(F)
        history.append(nrm)
        if nrm < tol:
            return ui, history
        if it == max_it:
            break
        Jbc = _identity_rows(J, bc_dofs)
        del J
        if th_i != 0.0:
            del dRi               # (GBs at the sizes the block-wise solve exists for)
        if linear in ('block', 'block32'):
            ui = ui - solve_blockwise(Jbc, F, W.N, single=linear == 'block32')
        else:
            ui = ui - spla.splu(Jbc.tocsc()).solve(F)
        del Jbc
    raise RuntimeError('Newton solver did not converge: %r' % history)


def divergence_P2_to_vertices(W, U):
    '''div(u) at the three vertices of every cell (it is P1 per cell for P2 u,
    constant for P1 u).  (Nc, 3).'''
    vpts = numpy.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    _, gref = basis(W.deg, vpts)
    g = W.phys_grad(gref)
    Uc = numpy.stack([U[W.cell_dofs], U[W.N + W.cell_dofs]], axis=1)
    return numpy.einsum('cai,cqia->cq', Uc, g)


def pressure_rhs(W, P, ui, p0, alpha, rho, mu, dt, rotational):
    '''L2 of pressure_correction.py:318-323.'''
    pts, w = duffy_rule(4)
    phi, gref = basis(W.deg, pts)
    gphi = W.phys_grad(gref)
    psi, gpsi_ref = basis(1, pts)
    gpsi = P.phys_grad(gpsi_ref)
    wd = w[None, :] * numpy.abs(W.detJ)[:, None]
    Uc = numpy.stack([ui[W.cell_dofs], ui[W.N + W.cell_dofs]], axis=1)
    divq = numpy.einsum('cai,cqia->cq', Uc, gphi)
    gp0 = numpy.einsum('ci,cqid->cqd', p0[P.cell_dofs], gpsi)
    Fe = -alpha * rho / dt * numpy.einsum('cq,cq,qi->ci', wd, divq, psi)
    Fe += numpy.einsum('cq,cqd,cqid->ci', wd, gp0, gpsi)
    if rotational and W.deg == 2:
        # grad(div u): second derivatives of the P2 basis (constant per cell)
        g = W.glam
        H = numpy.empty((len(W.cells), 6, 2, 2))
        for i in range(3):
            H[:, i] = 4.0 * numpy.einsum('ca,cb->cab', g[:, i], g[:, i])
        for e, (j, k) in enumerate(_EDGE):
            H[:, 3 + e] = 4.0 * (
                numpy.einsum('ca,cb->cab', g[:, j], g[:, k])
                + numpy.einsum('ca,cb->cab', g[:, k], g[:, j])
                )
        gdiv = numpy.einsum('cai,ciab->cb', Uc, H)
        Fe -= mu * numpy.einsum('cq,cb,cqib->ci', wd, gdiv, gpsi)
    return _vec(P, Fe[:, None, :])


def symmetric_bc(A, b, bc_dofs, bc_vals):
    '''`assemble_system`-style symmetric elimination.'''
    A = A.tocsr().copy()
    b = b.copy()
    xg = numpy.zeros(A.shape[0])
    xg[bc_dofs] = bc_vals
    b -= A.dot(xg)
    keep = numpy.ones(A.shape[0])
    keep[bc_dofs] = 0.0
    Dk = sp.diags(keep)
    A = Dk.dot(A).dot(Dk) + sp.diags(1.0 - keep)
    b[bc_dofs] = bc_vals
    return A.tocsr(), b


def solve_pressure(P, b, bc_dofs=None, bc_vals=None):
    '''`_compute_pressure` solve (pressure_correction.py:325-433).  With
    Dirichlet data: symmetric elimination + direct solve.  Pure Neumann: the
    singular consistent system is solved in the space orthogonal to the
    constants (bordered system); the reference's CG+AMG fixes the constant in a
    preconditioner-dependent way, so pressures are compared after removing the
    mean (tests/test_navier_stokes.py:347-360).'''
    A = stiffness_matrix(P)
    if bc_dofs is not None and len(bc_dofs) > 0:
        A, b = symmetric_bc(A, b, bc_dofs, bc_vals)
        return spla.splu(A.tocsc()).solve(b)
    n = A.shape[0]
    one = numpy.ones((n, 1))
    K = sp.bmat([[A, sp.csr_matrix(one)], [sp.csr_matrix(one.T), None]],
                format='csc')
    x = spla.splu(K).solve(numpy.concatenate([b, [0.0]]))
    return x[:n]


def velocity_correction(W, P, ui, p1, p0, bc_dofs, bc_vals, rho, mu, dt,
                        rotational, linear='lu'):
    '''`_compute_velocity_correction` (pressure_correction.py:436-465).'''
    pts, w = duffy_rule(4)
    phi, _ = basis(W.deg, pts)
    psi, gpsi_ref = basis(1, pts)
    gpsi = P.phys_grad(gpsi_ref)
    wd = w[None, :] * numpy.abs(W.detJ)[:, None]
    phi_field = (p1 - p0)[P.cell_dofs]                 # (Nc, 3)
    if rotational:
        phi_field = phi_field + mu * divergence_P2_to_vertices(W, ui)
    gphi_field = numpy.einsum('ci,cqid->cqd', phi_field, gpsi)
    Fe = -dt / rho * numpy.einsum('cq,cqa,qi->cai', wd, gphi_field, phi)
    M = sp.block_diag([mass_matrix(W)] * 2, format='csr')
    b = M.dot(ui) + _vec(W, Fe, 2)
    A, b = symmetric_bc(M, b, bc_dofs, bc_vals)
    if linear in ('block', 'block32'):
        return solve_blockwise(A, b, W.N, single=linear == 'block32')
    return spla.splu(A.tocsc()).solve(b)


def step(W, P, u0, p0, f0, f1, u_bc, p_bc, rho, mu, dt,
         scheme='ipcs', method='backward euler', info=None, linear='lu'):
    '''`_step` (pressure_correction.py:468-518) with the scheme flags of
    Chorin (:545-548), IPCS (:575-584), Rotational (:607-617).
    u_bc / p_bc: (dofs, values) tuples; p_bc None or empty -> Neumann branch.
    Returns (u1, p1, ui); a dict handed in as `info` receives the Newton
    iteration's residual norms.  linear: see `tentative_velocity`.'''
    assert dt > 0.0 and mu > 0.0
    rotational = False
    if scheme == 'chorin':
        p0 = numpy.zeros_like(p0)
        method = 'backward euler'
    elif scheme == 'rotational':
        rotational = True
    else:
        assert scheme == 'ipcs'
    ui, history = tentative_velocity(
        W, P, u0, p0, f0, f1, u_bc[0], u_bc[1], method, rho, mu, dt,
        linear=linear)
    if info is not None:
        info['newton_history'] = history
    b = pressure_rhs(W, P, ui, p0, 1.0, rho, mu, dt, rotational)
    if p_bc is not None and len(p_bc[0]) > 0:
        p1 = solve_pressure(P, b, p_bc[0], p_bc[1])
    else:
        p1 = solve_pressure(P, b)
    u1 = velocity_correction(
        W, P, ui, p1, p0, u_bc[0], u_bc[1], rho, mu, dt, rotational,
        linear=linear)
    return u1, p1, ui


# -- heat operator (flow/heat.py) ---------------------------------------------
def supg_tau(points3, conv, epsilon, p):
    '''`SupgStab::eval` (flow/stabilization.py:50-143) for one cell.
    points3: (3,2) vertex coordinates, conv: (2,) convection at the evaluation
    point.'''
    conv_norm = numpy.sqrt(conv[0]**2 + conv[1]**2)
    if conv_norm < 1.0e-10:
        return 0.0
    d1 = points3[1] - points3[0]
    d2 = points3[2] - points3[0]
    area = 0.5 * abs(d1[0] * d2[1] - d1[1] * d2[0])
    s = 0.0
    for i in range(3):
        for j in range(i + 1, 3):
            e0 = points3[i][0] - points3[j][0]
            e1 = points3[i][1] - points3[j][1]
            s += abs(e1 * conv[0] - e0 * conv[1])
    h = 4.0 * conv_norm * area / s
    Pe = 0.5 * conv_norm * h / (p * epsilon)
    if Pe > 1.0e-5:
        xi = (1.0 / numpy.tanh(Pe) - 1.0 / Pe) / Pe
    else:
        xi = 1.0 / 3.0 - Pe * Pe / 45.0 + 2.0 / 945.0 * Pe**4
    tau = h * h / 4.0 / epsilon / p * xi
    if tau > 1.0e3:
        raise RuntimeError('SUPG tau = %e > 1e3' % tau)
    return tau


def heat_operators(Q, W, conv, kappa, rho, cp, source_const=0.0, supg=False):
    '''`Heat.__init__` (flow/heat.py:20-89): returns (M, A, b).  Q: scalar
    temperature space, W: scalar space of the convection velocity, conv: (2*Nw,).
    f = -kappa grad u . grad(v/(rho cp)) - (conv . grad u) v + source v;
    A = matrix of the bilinear part, b = rhs(f) = -int source v (UFL `rhs`
    negates; immaterial for source = 0, tests/test_boussinesq.py:224).'''
    rho_cp = rho * cp
    pts, w = duffy_rule(5)
    phi, gref = basis(Q.deg, pts)
    gphi = Q.phys_grad(gref)
    wphi, _ = basis(W.deg, pts)
    wd = w[None, :] * numpy.abs(Q.detJ)[:, None]
    Cc = numpy.stack([conv[W.cell_dofs], conv[W.N + W.cell_dofs]], axis=1)
    cq = numpy.einsum('cai,qi->caq', Cc, wphi)
    Ke = -kappa / rho_cp * numpy.einsum('cq,cqjd,cqid->cij', wd, gphi, gphi)
    Ke -= numpy.einsum('cq,caq,cqja,qi->cij', wd, cq, gphi, phi)
    M = lumped_mass_vertex_rule(Q)
    be = -source_const * numpy.einsum('cq,qi->ci', wd, phi)
    if supg:
        # tau is an Expression of degree 1: evaluated at the three cell
        # vertices, interpolated linearly (flow/stabilization.py:147).
        vpts = numpy.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
        wv, _ = basis(W.deg, vpts)
        cv = numpy.einsum('cai,vi->cva', Cc, wv)             # conv at vertices
        pc = Q.points[Q.cells]
        tau_v = numpy.array([
            [supg_tau(pc[c], cv[c, v], kappa, Q.deg) for v in range(3)]
            for c in range(len(Q.cells))
            ])
        lam = bary(pts)
        tauq = numpy.einsum('cv,qv->cq', tau_v, lam)
        cgv = numpy.einsum('caq,cqia->cqi', cq, gphi)        # conv . grad v_i
        # M += u tau conv.grad(v)
        Me = numpy.einsum('cq,qj,cq,cqi->cij', wd, phi, tauq, cgv)
        M = M + _coo(Q, Q, Me[:, None, :, None, :])
        # R2 = div(kappa grad u)/rho_cp - conv.grad u + source/rho_cp
        lap = numpy.zeros((len(Q.cells), Q.nloc))
        if Q.deg == 2:
            g = Q.glam
            for i in range(3):
                lap[:, i] = 4.0 * (g[:, i] * g[:, i]).sum(axis=1)
            for e, (j, k) in enumerate(_EDGE):
                lap[:, 3 + e] = 8.0 * (g[:, j] * g[:, k]).sum(axis=1)
        cgu = cgv
        Ke += kappa / rho_cp * numpy.einsum(
            'cq,cj,cq,cqi->cij', wd, lap, tauq, cgv)
        Ke -= numpy.einsum('cq,cqj,cq,cqi->cij', wd, cgu, tauq, cgv)
        be -= source_const / rho_cp * numpy.einsum('cq,cq,cqi->ci', wd, tauq, cgv)
    A = _coo(Q, Q, Ke[:, None, :, None, :])
    b = _vec(Q, be[:, None, :])
    return M, A, b


def heat_solve(M, A, alpha, beta, b, bc_dofs, bc_vals):
    '''`Heat.solve_alpha_M_beta_F` (flow/heat.py:103-122): (alpha M + beta A) u
    = b with `bc.apply(A, b)` (row replacement), sparse LU.  The reference
    computes `right_hand_side` but solves with the raw `b` (:109-121).'''
    S = _identity_rows(alpha * M + beta * A, bc_dofs)
    b = b.copy()
    b[bc_dofs] = bc_vals
    return spla.splu(S.tocsc()).solve(b)


# -- harness utilities --------------------------------------------------------
def l2_project(S, lattice_pts, cell_values, dim=1):
    '''dolfin `project`: mass solve.'''
    M = mass_matrix(S)
    b = load_vector(S, lattice_pts, cell_values, dim)
    lu = spla.splu(M.tocsc())
    return numpy.concatenate(
        [lu.solve(b[c * S.N:(c + 1) * S.N]) for c in range(dim)]
        )


def l2_error(S, U, lattice_pts, exact_values, dim=1):
    '''`errornorm`-like L2 error: the discrete field and the per-cell Lagrange
    interpolant of the exact solution (degree of `lattice_pts`) are compared
    with a rule exact for the squared difference.'''
    pts, w = duffy_rule(7)
    phi, _ = basis(S.deg, pts)
    psi = lagrange_eval(lattice_pts, pts)
    wd = w[None, :] * numpy.abs(S.detJ)[:, None]
    err = 0.0
    for c in range(dim):
        uh = numpy.einsum('ci,qi->cq', U[c * S.N + S.cell_dofs], phi)
        ue = numpy.einsum('ql,cl->cq', psi, exact_values[:, :, c])
        err += numpy.sum(wd * (uh - ue)**2)
    return numpy.sqrt(err)


def stokes_solve(W, P, f, mu, u_bc, p_bc=None):
    '''`flow.stokes.solve` (flow/stokes.py:13-148): mixed Taylor-Hood system
        a = mu (grad u, grad v) - (p, div v) - (q, div u),  L = (f, v),
    `assemble_system(a, L, bcs)` (symmetric elimination of velocity AND pressure
    Dirichlet dofs), solved here with a sparse direct factorisation (the
    reference: GMRES + AMG to rtol tol).  f: (lattice_pts, cell_values
    (Nc, nl, 2)); u_bc / p_bc: (dofs, values).  Returns (u, p).'''
    pts, w = duffy_rule(4)
    _, gref = basis(W.deg, pts)
    gphi = W.phys_grad(gref)                      # (Nc, nq, nl, 2)
    psi, _ = basis(1, pts)
    wd = w[None, :] * numpy.abs(W.detJ)[:, None]
    # B[q_i, (a, j)] = - int q_i d_a phi_j
    Be = -numpy.einsum('cq,qi,cqja->ciaj', wd, psi, gphi)
    B = _coo(P, W, Be[:, None, :, :, :], 1, 2)
    K = stiffness_matrix(W)
    A = sp.bmat([[sp.block_diag([mu * K, mu * K]), B.T], [B, None]],
                format='csr')
    nw = 2 * W.N
    b = numpy.concatenate([load_vector(W, f[0], f[1], dim=2),
                           numpy.zeros(P.N)])
    dofs = [numpy.asarray(u_bc[0], dtype=numpy.int64)]
    vals = [numpy.asarray(u_bc[1], dtype=float)]
    if p_bc is not None and len(p_bc[0]) > 0:
        dofs.append(nw + numpy.asarray(p_bc[0], dtype=numpy.int64))
        vals.append(numpy.asarray(p_bc[1], dtype=float))
    A, b = symmetric_bc(A, b, numpy.concatenate(dofs), numpy.concatenate(vals))
    x = spla.splu(A.tocsc()).solve(b)
    return x[:nw], x[nw:]


def order_of_convergence(Dt, errors):
    '''tests/helpers.py:10-14.'''
    return numpy.array([
        numpy.log(errors[k] / errors[k + 1]) / numpy.log(Dt[k] / Dt[k + 1])
        for k in range(len(Dt) - 1)
        ])
