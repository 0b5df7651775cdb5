// K19: two-level cycle with ILU(0) smoothing (include/flow_hip.h, flow_tl) --
// the preconditioner of the Newton and heat solves where the Chebyshev cycle of
// pmg_kernels.hip is rejected (cell Peclet numbers beyond ~3: the stand-in for
// the reference's sparse LU, flow/navier_stokes/pressure_correction.py:224-254,
// flow/heat.py:117-121).  The heavy lifting is the sweeps of ilu_kernels.hip
// and the CSR-stream products of la_kernels.hip; what lives here is the glue
// between the P2 level and the P1 level of the same mesh: fp64,
// component-blocked vectors (the ILU sweeps take and return those).
#include "common.h"

namespace flow {

namespace {

// rc[a n1 + v] = sum over the restriction list of vertex v of
//   w_k * rscale[i_k] * (r[i_k] - y[i_k]),  w = 1 for the vertex's own dof (the
// first entry), 1/2 for the dofs of the edges that end there; 0 on the
// Dirichlet rows of the P1 level.  y == nullptr: the residual is r itself (no
// pre-smoothing).  One lane per (vertex, component): the lists are 1 + ~6
// entries, neighbouring vertices have neighbouring dofs.
__global__ __launch_bounds__(kBlock) void tl_restrict_kernel(
    int n, int n1, int nb, const int* __restrict__ rptr,
    const int* __restrict__ rsrc, const double* __restrict__ r,
    const double* __restrict__ y, const double* __restrict__ rscale,
    const unsigned char* __restrict__ bc_fine,
    const unsigned char* __restrict__ bc_coarse, double* __restrict__ rc,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  const int total = nb * n1;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total;
       k += gridDim.x * blockDim.x) {
    const int a = k / n1, v = k - a * n1;
    double s = 0.0;
    if (!(bc_coarse && bc_coarse[k])) {
      const int p0 = rptr[v], p1 = rptr[v + 1];
      const size_t base = static_cast<size_t>(a) * n;
      for (int p = p0; p < p1; ++p) {
        const size_t i = base + rsrc[p];
        if (bc_fine && bc_fine[i]) continue;
        double t = r[i];
        if (y) t -= y[i];
        if (rscale) t *= rscale[i];
        s += (p == p0 ? 1.0 : 0.5) * t;
      }
    }
    rc[k] = s;
  }
}

// out[a n + i] = x[a n + i] + 1/2 (xc[a n1 + e0] + xc[a n1 + e1]) on the free
// rows, x on the Dirichlet rows; x == nullptr: no pre-smoothing, x = 0
__global__ __launch_bounds__(kBlock) void tl_prolong_kernel(
    int n, int n1, int nb, const int2* __restrict__ ends,
    const double* __restrict__ xc, const double* __restrict__ x,
    const unsigned char* __restrict__ bc_fine, double* __restrict__ out,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  const int total = nb * n;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total;
       k += gridDim.x * blockDim.x) {
    const int a = k / n, i = k - a * n;
    double v = x ? x[k] : 0.0;
    if (!(bc_fine && bc_fine[k])) {
      const int2 e = ends[i];
      const size_t base = static_cast<size_t>(a) * n1;
      v += 0.5 * (xc[base + e.x] + xc[base + e.y]);
    }
    out[k] = v;
  }
}

// out = a - b
__global__ __launch_bounds__(kBlock) void tl_sub_kernel(
    int n, const double* __restrict__ a, const double* __restrict__ b,
    double* __restrict__ out, const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += gridDim.x * blockDim.x)
    out[k] = a[k] - b[k];
}

// out = a + b
__global__ __launch_bounds__(kBlock) void tl_add_kernel(
    int n, const double* __restrict__ a, const double* __restrict__ b,
    double* __restrict__ out, const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += gridDim.x * blockDim.x)
    out[k] = a[k] + b[k];
}

}  // namespace

int tl_check(const flow_tl* T, int op_size) {
  FLOW_REQUIRE(T && T->fine && T->coarse && T->fine_op, "flow_tl pointers");
  FLOW_REQUIRE(T->fine->cycle == nullptr && T->coarse->cycle == nullptr,
               "the smoothers of a flow_tl are plain ILU(0) applications");
  FLOW_REQUIRE(T->fine->plan && T->coarse->plan, "flow_tl plans");
  const int nb = T->fine->nblocks;
  const int n = T->fine->plan->n, n1 = T->coarse->plan->n;
  FLOW_REQUIRE(T->coarse->nblocks == nb, "flow_tl: blocks of the two levels");
  int rc = ilu_check(T->fine, nb * n);
  if (rc) return rc;
  if ((rc = ilu_check(T->coarse, nb * n1))) return rc;
  FLOW_REQUIRE(nb * n == op_size, "flow_tl does not match the operator");
  if ((rc = check_operator(T->fine_op))) return rc;
  FLOW_REQUIRE(operator_size(T->fine_op) == nb * n, "flow_tl: fine operator size");
  FLOW_REQUIRE((T->pre == 0 || T->pre == 1) && (T->post == 0 || T->post == 1) &&
                   T->pre + T->post >= 1,
               "flow_tl: pre, post in {0, 1}, not both 0");
  FLOW_REQUIRE(T->coarse_sweeps >= 1 && T->coarse_sweeps <= 8,
               "flow_tl: coarse sweeps 1..8");
  if (T->coarse_sweeps > 1) {
    FLOW_REQUIRE(T->coarse_op != nullptr, "flow_tl: coarse operator");
    if ((rc = check_operator(T->coarse_op))) return rc;
    FLOW_REQUIRE(operator_size(T->coarse_op) == nb * n1,
                 "flow_tl: coarse operator size");
  }
  FLOW_REQUIRE(T->ends && T->rptr && T->rsrc && T->work, "flow_tl tables / work");
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(T->work) % 16 == 0,
               "flow_tl work must be 16-byte aligned");
  return FLOW_OK;
}

// z = M^-1 r: one cycle
int tl_apply(const flow_tl* T, const double* r, double* z, hipStream_t st,
             const double* stop) {
  const int nb = T->fine->nblocks;
  const int n = T->fine->plan->n, n1 = T->coarse->plan->n;
  const size_t N = static_cast<size_t>(nb) * n, N1 = static_cast<size_t>(nb) * n1;
  double* x = T->work;
  double* y = x + N;
  double* t = y + N;
  double* iw = t + N;
  double* rc = iw + N;
  double* xc = rc + N1;
  double* yc = xc + N1;
  double* tc = yc + N1;
  double* iwc = tc + N1;
  const int g = grid_for(static_cast<long long>(N));
  const int g1 = grid_for(static_cast<long long>(N1));
  const int Ni = static_cast<int>(N), N1i = static_cast<int>(N1);
  int rcode;
  const double* nod = nullptr;
  // pre-smoothing from zero and the residual behind it, restricted
  if (T->pre) {
    if ((rcode = ilu_apply(T->fine, r, x, iw, st, stop))) return rcode;
    if ((rcode = operator_apply(T->fine_op, x, y, st, stop))) return rcode;
  }
  hipLaunchKernelGGL(tl_restrict_kernel, dim3(g1), dim3(kBlock), 0, st, n, n1, nb,
                     T->rptr, T->rsrc, r, T->pre ? y : nod, T->rscale, T->bc_fine,
                     T->bc_coarse, rc, stop);
  // the P1 level: ILU(0) as a stationary iteration from zero
  if ((rcode = ilu_apply(T->coarse, rc, xc, iwc, st, stop))) return rcode;
  for (int s = 1; s < T->coarse_sweeps; ++s) {
    if ((rcode = operator_apply(T->coarse_op, xc, yc, st, stop))) return rcode;
    hipLaunchKernelGGL(tl_sub_kernel, dim3(g1), dim3(kBlock), 0, st, N1i, rc, yc,
                       tc, stop);
    if ((rcode = ilu_apply(T->coarse, tc, tc, iwc, st, stop))) return rcode;
    hipLaunchKernelGGL(tl_add_kernel, dim3(g1), dim3(kBlock), 0, st, N1i, xc, tc,
                       xc, stop);
  }
  // correction; without post-smoothing that is the result
  double* dst = T->post ? x : z;
  hipLaunchKernelGGL(tl_prolong_kernel, dim3(g), dim3(kBlock), 0, st, n, n1, nb,
                     reinterpret_cast<const int2*>(T->ends), xc, T->pre ? x : nod,
                     T->bc_fine, dst, stop);
  if (T->post) {
    if ((rcode = operator_apply(T->fine_op, x, y, st, stop))) return rcode;
    hipLaunchKernelGGL(tl_sub_kernel, dim3(g), dim3(kBlock), 0, st, Ni, r, y, t,
                       stop);
    if ((rcode = ilu_apply(T->fine, t, t, iw, st, stop))) return rcode;
    hipLaunchKernelGGL(tl_add_kernel, dim3(g), dim3(kBlock), 0, st, Ni, x, t, z,
                       stop);
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

extern "C" int flow_tl_apply(const flow_tl* tl, const double* r, double* z,
                             void* stream) {
  FLOW_REQUIRE(tl && r && z && r != z, "flow_tl_apply arguments");
  FLOW_REQUIRE(tl->fine && tl->fine->plan, "flow_tl fine level");
  int rc = tl_check(tl, tl->fine->nblocks * tl->fine->plan->n);
  if (rc) return rc;
  return tl_apply(tl, r, z, as_stream(stream), nullptr);
}
