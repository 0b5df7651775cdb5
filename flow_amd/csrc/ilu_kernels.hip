// K11: multicolour ILU(0) on gfx950 -- factorisation and the two triangular
// sweeps, one kernel launch per colour (rows of one colour are an independent
// set, so a launch has no internal dependencies).  The plan (colouring, permuted
// colour-major CSR, map back to the operator's value plane) is built once per
// pattern on the host: flow_amd/fem/ilu.py.
//
// Stands in for the sparse LU behind the reference's Newton and heat solves
// (flow/navier_stokes/pressure_correction.py:224-254, flow/heat.py:117-121) as
// the preconditioner of BiCGStab.  HBM-bound: every L/U entry (12 B) is read
// once per application; lanes own rows (rows are short: 7..23 entries).
#include "common.h"

namespace flow {

// LU <- A in the permuted numbering, rows [a, b)
__global__ void ilu_copy_kernel(int nnz, const int* __restrict__ src_pos,
                                const double* __restrict__ avals,
                                double* __restrict__ lu) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += gridDim.x * blockDim.x)
    lu[k] = avals[src_pos[k]];
}

// IKJ ILU(0) of the rows [a, b) of one colour.  Every row k < i referenced here
// has a lower colour and is final.
__global__ void ilu_factor_colour_kernel(int a, int b,
                                         const int* __restrict__ rowptr,
                                         const int* __restrict__ cols,
                                         const int* __restrict__ diag,
                                         double* __restrict__ lu) {
  const int i = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const int p0 = rowptr[i], pd = diag[i], p1 = rowptr[i + 1];
  const double d_orig = lu[pd];
  for (int p = p0; p < pd; ++p) {
    const int k = cols[p];
    const double lik = lu[p] / lu[diag[k]];
    lu[p] = lik;
    // a_ij -= l_ik * u_kj for j > k present in both rows
    int q = diag[k] + 1;
    const int qe = rowptr[k + 1];
    for (int t = p + 1; t < p1 && q < qe; ++t) {
      const int j = cols[t];
      while (q < qe && cols[q] < j) ++q;     // both lists ascend: merge
      if (q < qe && cols[q] == j) lu[t] -= lik * lu[q];
    }
  }
  // pivot guard: keep the factor usable if a pivot collapses
  const double d = lu[pd];
  if (!(fabs(d) > 1.0e-12 * fabs(d_orig))) lu[pd] = d_orig != 0.0 ? d_orig : 1.0;
}

// forward sweep: y_i = r_old(i) - sum_{k<i} L_ik y_k   (unit lower)
__global__ void ilu_forward_colour_kernel(int a, int b,
                                          const int* __restrict__ rowptr,
                                          const int* __restrict__ cols,
                                          const int* __restrict__ diag,
                                          const int* __restrict__ old_of_new,
                                          const double* __restrict__ lu,
                                          const double* __restrict__ r,
                                          double* __restrict__ y) {
  const int i = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  double s = r[old_of_new[i]];
  const int pd = diag[i];
  for (int p = rowptr[i]; p < pd; ++p) s -= lu[p] * y[cols[p]];
  y[i] = s;
}

// backward sweep (in place in y): y_i = (y_i - sum_{j>i} U_ij y_j) / U_ii,
// result scattered back to the original numbering
__global__ void ilu_backward_colour_kernel(int a, int b,
                                           const int* __restrict__ rowptr,
                                           const int* __restrict__ cols,
                                           const int* __restrict__ diag,
                                           const int* __restrict__ old_of_new,
                                           const double* __restrict__ lu,
                                           double* __restrict__ y,
                                           double* __restrict__ z) {
  const int i = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  double s = y[i];
  const int pd = diag[i];
  const int p1 = rowptr[i + 1];
  for (int p = pd + 1; p < p1; ++p) s -= lu[p] * y[cols[p]];
  s /= lu[pd];
  y[i] = s;
  z[old_of_new[i]] = s;
}

static int check_plan(const flow_ilu_plan* P) {
  FLOW_REQUIRE(P && P->n > 0 && P->nnz > 0 && P->ncolors > 0, "ilu plan sizes");
  FLOW_REQUIRE(P->color_ptr_host && P->rowptr && P->cols && P->diag &&
                   P->src_pos && P->old_of_new,
               "ilu plan pointers");
  FLOW_REQUIRE(P->color_ptr_host[0] == 0 && P->color_ptr_host[P->ncolors] == P->n,
               "ilu colour ranges");
  return FLOW_OK;
}

static int factor(const flow_ilu_plan* P, const double* avals, double* lu,
                  hipStream_t st) {
  hipLaunchKernelGGL(ilu_copy_kernel, dim3(grid_for(P->nnz)), dim3(kBlock), 0, st,
                     P->nnz, P->src_pos, avals, lu);
  for (int c = 1; c < P->ncolors; ++c) {   // colour 0 has no lower neighbours
    const int a = P->color_ptr_host[c], b = P->color_ptr_host[c + 1];
    if (b <= a) continue;
    hipLaunchKernelGGL(ilu_factor_colour_kernel, dim3((b - a + kBlock - 1) / kBlock),
                       dim3(kBlock), 0, st, a, b, P->rowptr, P->cols, P->diag, lu);
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

static int solve(const flow_ilu_plan* P, const double* lu, const double* r,
                 double* z, double* work, hipStream_t st) {
  for (int c = 0; c < P->ncolors; ++c) {
    const int a = P->color_ptr_host[c], b = P->color_ptr_host[c + 1];
    if (b <= a) continue;
    hipLaunchKernelGGL(ilu_forward_colour_kernel,
                       dim3((b - a + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a,
                       b, P->rowptr, P->cols, P->diag, P->old_of_new, lu, r, work);
  }
  for (int c = P->ncolors - 1; c >= 0; --c) {
    const int a = P->color_ptr_host[c], b = P->color_ptr_host[c + 1];
    if (b <= a) continue;
    hipLaunchKernelGGL(ilu_backward_colour_kernel,
                       dim3((b - a + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a,
                       b, P->rowptr, P->cols, P->diag, P->old_of_new, lu, work, z);
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

// out = blockdiag(LU_0, LU_1)^-1 in ; used by the BiCGStab driver
int ilu_apply(const flow_ilu* ilu, const double* in, double* out, double* work,
              hipStream_t st) {
  const flow_ilu_plan* P = ilu->plan;
  for (int k = 0; k < ilu->nblocks; ++k) {
    int rc = solve(P, ilu->lu + static_cast<size_t>(k) * P->nnz,
                   in + static_cast<size_t>(k) * P->n,
                   out + static_cast<size_t>(k) * P->n, work, st);
    if (rc) return rc;
  }
  return FLOW_OK;
}

int ilu_check(const flow_ilu* ilu, int op_size) {
  FLOW_REQUIRE(ilu && ilu->plan && ilu->lu, "ilu pointers");
  int rc = check_plan(ilu->plan);
  if (rc) return rc;
  FLOW_REQUIRE(ilu->nblocks == 1 || ilu->nblocks == 2, "ilu blocks");
  FLOW_REQUIRE(ilu->nblocks * ilu->plan->n == op_size, "ilu size");
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

extern "C" int flow_ilu0_factor(const flow_ilu_plan* plan, const double* avals,
                                double* lu, void* stream) {
  int rc = check_plan(plan);
  if (rc) return rc;
  FLOW_REQUIRE(avals && lu && avals != lu, "ilu factor pointers");
  return factor(plan, avals, lu, as_stream(stream));
}

extern "C" int flow_ilu0_solve(const flow_ilu_plan* plan, const double* lu,
                               const double* r, double* z, double* work,
                               void* stream) {
  int rc = check_plan(plan);
  if (rc) return rc;
  FLOW_REQUIRE(lu && r && z && work && r != work && z != work, "ilu solve pointers");
  return solve(plan, lu, r, z, work, as_stream(stream));
}
