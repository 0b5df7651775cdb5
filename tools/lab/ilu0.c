/* Development aid (tools/lab): ILU(0) of a CSR matrix with sorted columns and
 * its triangular solves, for the preconditioner experiments of
 * tools/precond_lab.py (CPU, scipy).  Not part of the product. */
#include <stdlib.h>

/* in place: lu holds L (unit diagonal, strictly lower part) and U */
int ilu0_factor(int n, const int* rowptr, const int* cols, const int* diag,
                double* lu) {
  for (int i = 0; i < n; ++i) {
    const int p0 = rowptr[i], pd = diag[i], p1 = rowptr[i + 1];
    for (int p = p0; p < pd; ++p) {
      const int k = cols[p];
      const double lik = lu[p] / lu[diag[k]];
      lu[p] = lik;
      int q = diag[k] + 1;
      const int qe = rowptr[k + 1];
      for (int t = p + 1; t < p1 && q < qe; ++t) {
        const int j = cols[t];
        while (q < qe && cols[q] < j) ++q;
        if (q < qe && cols[q] == j) lu[t] -= lik * lu[q];
      }
    }
    if (lu[pd] == 0.0) return i + 1;
  }
  return 0;
}

void ilu0_solve(int n, const int* rowptr, const int* cols, const int* diag,
                const double* lu, const double* b, double* x) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int p = rowptr[i]; p < diag[i]; ++p) s -= lu[p] * x[cols[p]];
    x[i] = s;
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int p = diag[i] + 1; p < rowptr[i + 1]; ++p) s -= lu[p] * x[cols[p]];
    x[i] = s / lu[diag[i]];
  }
}
