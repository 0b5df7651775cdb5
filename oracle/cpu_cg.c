/*
 * TEST INFRASTRUCTURE -- CPU restatement in plain C of the pressure-Poisson
 * Krylov loop (not part of the product; only tests/ and bench.py's
 * cpu_baseline leg load it).
 *
 * Restates what PETSc does for the reference inside
 * `PETScKrylovSolver('cg', prec)` / `solve(..., 'iterative', 'symmetric')`
 * (flow/navier_stokes/pressure_correction.py:326-339, 419-432): CSR MatMult
 * and the textbook preconditioned conjugate-gradient recurrence (Hestenes &
 * Stiefel; PETSc KSPCG), here with the diagonal (Jacobi) preconditioner the
 * north star prescribes.  PETSc itself is an un-vendored third-party
 * dependency, absent offline; this file is pinned against scipy in
 * tests/test_oracle_c.py.  OpenMP over rows; `cores` = omp_get_max_threads().
 */
#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* The host may be shared: use the CPUs this process may really run on (the
 * caller reads the cgroup quota / affinity mask), not every CPU it can see. */
void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void oracle_spmv_csr(int n, const int* rowptr, const int* cols,
                     const double* vals, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) s += vals[k] * x[cols[k]];
    y[i] = s;
  }
}

static double dot(int n, const double* a, const double* b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* Jacobi-PCG; work = 4*n doubles; stops when the PRECONDITIONED residual
 * ||D^-1 r||_2 <= max(rtol*||D^-1 b||_2, atol) (PETSc's KSPCG default norm) or
 * after maxit iterations.  Returns 0 if converged, 1 otherwise. */
int oracle_jacobi_cg(int n, const int* rowptr, const int* cols,
                     const double* vals, const double* dinv, const double* b,
                     double* x, double rtol, double atol, int maxit,
                     double* work, int* iters, double* resid) {
  double* r = work;
  double* z = r + n;
  double* p = z + n;
  double* q = p + n;
  oracle_spmv_csr(n, rowptr, cols, vals, x, q);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    r[i] = b[i] - q[i];
    z[i] = dinv[i] * r[i];
    p[i] = z[i];
  }
  /* q is free here: q = D^-1 b */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) q[i] = dinv[i] * b[i];
  const double target = fmax(rtol * sqrt(dot(n, q, q)), atol);
  double rz = dot(n, r, z);
  double rr = dot(n, z, z);
  int it = 0;
  while (sqrt(rr) > target && it < maxit) {
    oracle_spmv_csr(n, rowptr, cols, vals, p, q);
    const double alpha = rz / dot(n, p, q);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      x[i] += alpha * p[i];
      r[i] -= alpha * q[i];
      z[i] = dinv[i] * r[i];
    }
    const double rz_new = dot(n, r, z);
    rr = dot(n, z, z);
    const double beta = rz_new / rz;
    rz = rz_new;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    ++it;
  }
  *iters = it;
  *resid = sqrt(rr);
  return sqrt(rr) <= target ? 0 : 1;
}

/* ------------------------------------------------------------------------
 * Like-for-like CPU baseline of the pressure solve the product runs on the GPU:
 * CG preconditioned with a smoothed-aggregation multigrid V(1,1) cycle (what
 * the reference gets from 'hypre_amg', pressure_correction.py:331, 414-418),
 * on the SAME hierarchy (the level matrices A_l, prolongations P_l and their
 * transposes are handed in as CSR by the caller, who builds them with scipy),
 * stopping test in the preconditioned norm like PETSc's KSPCG default:
 * ||B r||_2 <= max(rtol ||B b||_2, atol).
 *
 * The matrices live in library-owned copies whose pages are FIRST TOUCHED by
 * the OpenMP threads that later stream them (static schedule over rows), so a
 * multi-socket host does not read everything from the NUMA node of the thread
 * that happened to create the numpy arrays.
 * ------------------------------------------------------------------------ */
#include <stdlib.h>
#include <string.h>

typedef struct {
  int n, m;          /* rows, columns */
  int* rowptr;
  int* cols;
  double* vals;
} oracle_csr;

oracle_csr* oracle_csr_create(int n, int m, const int* rowptr, const int* cols,
                              const double* vals) {
  oracle_csr* A = (oracle_csr*)malloc(sizeof(oracle_csr));
  const size_t nnz = (size_t)rowptr[n];
  A->n = n;
  A->m = m;
  A->rowptr = (int*)malloc(sizeof(int) * ((size_t)n + 1));
  A->cols = (int*)malloc(sizeof(int) * (nnz ? nnz : 1));
  A->vals = (double*)malloc(sizeof(double) * (nnz ? nnz : 1));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    A->rowptr[i] = rowptr[i];
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      A->cols[k] = cols[k];
      A->vals[k] = vals[k];
    }
  }
  A->rowptr[n] = rowptr[n];
  return A;
}

void oracle_csr_destroy(oracle_csr* A) {
  if (!A) return;
  free(A->rowptr);
  free(A->cols);
  free(A->vals);
  free(A);
}

void oracle_csr_spmv(const oracle_csr* A, const double* x, double* y) {
  oracle_spmv_csr(A->n, A->rowptr, A->cols, A->vals, x, y);
}

/* first-touched vector */
static double* vec_alloc(int n) {
  double* v = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) v[i] = 0.0;
  return v;
}

double* oracle_vec_create(int n, const double* src) {
  double* v = vec_alloc(n);
  if (src) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) v[i] = src[i];
  }
  return v;
}

void oracle_vec_read(int n, const double* v, double* dst) {
  memcpy(dst, v, sizeof(double) * (size_t)n);
}

void oracle_vec_destroy(double* v) { free(v); }

typedef struct {
  int nlevels;              /* including the dense coarsest one */
  oracle_csr** A;           /* nlevels-1 level operators */
  oracle_csr** P;           /* nlevels-1 prolongations (n_l x n_{l+1}) */
  oracle_csr** R;           /* their transposes */
  double** dinv;            /* nlevels-1 inverse diagonals */
  int nc;                   /* coarsest size */
  const double* Ainv;       /* dense nc x nc (pseudo-)inverse, row major */
  double omega;
  double **r, **x, **t;     /* level vectors (level 0: caller's) */
} oracle_mg;

/* x_l = V-cycle(r_l), zero start, damped Jacobi V(1,1) */
static void vcycle(const oracle_mg* M, int l, const double* r, double* x) {
  if (l == M->nlevels - 1) {
    const int nc = M->nc;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nc; ++i) {
      double s = 0.0;
      for (int j = 0; j < nc; ++j) s += M->Ainv[(size_t)i * nc + j] * r[j];
      x[i] = s;
    }
    return;
  }
  const int n = M->A[l]->n;
  const double w = M->omega;
  const double* dinv = M->dinv[l];
  double* t = M->t[l];
  /* pre-smoothing from zero: x = w D^-1 r ; t = r - A x */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) x[i] = w * dinv[i] * r[i];
  oracle_csr_spmv(M->A[l], x, t);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) t[i] = r[i] - t[i];
  oracle_csr_spmv(M->R[l], t, M->r[l + 1]);
  vcycle(M, l + 1, M->r[l + 1], M->x[l + 1]);
  /* x += P x_c ; post-smoothing x += w D^-1 (r - A x) */
  oracle_csr_spmv(M->P[l], M->x[l + 1], t);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) x[i] += t[i];
  oracle_csr_spmv(M->A[l], x, t);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) x[i] += w * dinv[i] * (r[i] - t[i]);
}

/* CG + V-cycle on level-0 system A[0] x = b.  Returns 0 if converged. */
int oracle_mg_cg(int nlevels, oracle_csr** A, oracle_csr** P, oracle_csr** R,
                 double** dinv, int nc, const double* Ainv, double omega,
                 const double* b, double* x, double rtol, double atol,
                 int maxit, int* iters, double* resid) {
  oracle_mg M;
  M.nlevels = nlevels;
  M.A = A;
  M.P = P;
  M.R = R;
  M.dinv = dinv;
  M.nc = nc;
  M.Ainv = Ainv;
  M.omega = omega;
  M.r = (double**)calloc((size_t)nlevels, sizeof(double*));
  M.x = (double**)calloc((size_t)nlevels, sizeof(double*));
  M.t = (double**)calloc((size_t)nlevels, sizeof(double*));
  for (int l = 0; l < nlevels; ++l) {
    const int nl = l + 1 < nlevels ? A[l]->n : nc;
    if (l > 0) {
      M.r[l] = vec_alloc(nl);
      M.x[l] = vec_alloc(nl);
    }
    if (l + 1 < nlevels) M.t[l] = vec_alloc(nl);
  }
  const int n = nlevels > 1 ? A[0]->n : nc;
  double* r = vec_alloc(n);
  double* z = vec_alloc(n);
  double* p = vec_alloc(n);
  double* q = vec_alloc(n);
  vcycle(&M, 0, b, z);
  const double target = fmax(rtol * sqrt(dot(n, z, z)), atol);
  oracle_csr_spmv(A[0], x, q);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) r[i] = b[i] - q[i];
  vcycle(&M, 0, r, z);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) p[i] = z[i];
  double rz = dot(n, r, z);
  double zz = dot(n, z, z);
  int it = 0;
  while (sqrt(zz) > target && it < maxit) {
    oracle_csr_spmv(A[0], p, q);
    const double alpha = rz / dot(n, p, q);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      x[i] += alpha * p[i];
      r[i] -= alpha * q[i];
    }
    vcycle(&M, 0, r, z);
    const double rz_new = dot(n, r, z);
    zz = dot(n, z, z);
    const double beta = rz_new / rz;
    rz = rz_new;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    ++it;
  }
  *iters = it;
  *resid = sqrt(zz);
  for (int l = 0; l < nlevels; ++l) {
    free(M.r[l]);
    free(M.x[l]);
    free(M.t[l]);
  }
  free(M.r);
  free(M.x);
  free(M.t);
  free(r);
  free(z);
  free(p);
  free(q);
  return sqrt(zz) <= target ? 0 : 1;
}
