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

/* Jacobi-PCG; work = 4*n doubles; stops when ||r||_2 <= max(rtol*||b||_2, atol)
 * or after maxit iterations.  Returns 0 if converged, 1 otherwise. */
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
  const double target = fmax(rtol * sqrt(dot(n, b, b)), atol);
  double rz = dot(n, r, z);
  double rr = dot(n, r, r);
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
    rr = dot(n, r, r);
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
