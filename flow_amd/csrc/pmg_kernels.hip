// K17: p-multigrid / Chebyshev preconditioner of the Newton systems of the
// tentative velocity on gfx950 (include/flow_hip.h, flow_pmg).
//
// Stands in -- like the multicolour ILU(0) of ilu_kernels.hip -- for the sparse
// LU behind the reference's Newton solve (flow/navier_stokes/
// pressure_correction.py:224-254) as the right preconditioner of the flexible
// GMRES of la_kernels.hip.  Why another one: the ILU(0) sweeps are a chain of
// ~15 dependent launches bound by gather latency (0.30 of the HBM roofline) and
// a multicolour factorisation is a weak one (33 applications per solve at
// CFL-sized steps on the 10 M-DoF workload).  This preconditioner consists of
// CSR-stream products only:
//
//   fine level    the two diagonal blocks of the assembled P2 Jacobian, rounded
//                 to fp32 and interleaved (ONE index + one 8-byte value load per
//                 nonzero for both velocity components), smoothed with `pre` /
//                 `post` steps of the Chebyshev iteration for D^-1 A;
//   coarse level  the P1 discretisation of the same linearised operator on the
//                 same mesh (P1 is a subspace of P2: vertex dofs copy, edge
//                 dofs average their end points), `coarse` Chebyshev steps
//                 from a zero start -- at CFL-sized steps the P1 operator is
//                 mass-dominated (condition ~10) and needs no further levels.
//
// All vectors inside are fp32, both components interleaved (float2 per dof: one
// 8-byte gather per nonzero serves both blocks); an application is therefore
// not exactly linear in its input, which the flexible GMRES does not need
// (la_kernels.hip: it keeps Z_j = M^-1 V_j and updates x with it).
// Every kernel is HBM-bound: 12 B per nonzero + ~10 float2 vectors per cycle.
#include "common.h"

namespace flow {

constexpr int kPmgPairs = 2;                         // nonzero pairs per lane
constexpr int kPmgTile = 2 * kBlock * kPmgPairs;     // LDS products per workgroup
static_assert(FLOW_SPMV_NNZ_PER_BLOCK == kPmgTile - 2, "tile minus alignment slack");

__device__ __forceinline__ float2 f2(float a, float b) { return make_float2(a, b); }

// One tile of the packed stream -- rows [r0, r1) of workgroup blockIdx.x: the
// products of both blocks with the gathered vector g go through LDS, then lane
// i sums row r0 + i.  Same tiling, alignment rules and window safety as
// stream_tile_row_sum (la_kernels.hip): g is only dereferenced for the tile's
// own nonzeros.
__device__ __forceinline__ float2 pmg_tile_row_sum(
    const int* __restrict__ rowptr, const int* __restrict__ cols,
    const float2* __restrict__ vals, const int* __restrict__ rowblocks,
    const float2* __restrict__ g, float2* __restrict__ prod, int& r, int& r1) {
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int r0 = rowblocks[tile];
  r1 = rowblocks[tile + 1];
  const int k0 = rowptr[r0];
  const int k1 = rowptr[r1];
  const int ka = k0 & ~1;
  r = r0 + threadIdx.x;
  int a = 0, b = 0;
  if (r < r1) {
    a = rowptr[r] - ka;
    b = rowptr[r + 1] - ka;
  }
  const float4* __restrict__ v4p = reinterpret_cast<const float4*>(vals + ka);
  const int2* __restrict__ c2p = reinterpret_cast<const int2*>(cols + ka);
  const int npair = (k1 - ka + 1) >> 1;
  float4 v[kPmgPairs];
  int2 c[kPmgPairs];
#pragma unroll
  for (int j = 0; j < kPmgPairs; ++j) {
    const int p = threadIdx.x + j * kBlock;
    const bool ok = p < npair;
    v[j] = ok ? v4p[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    c[j] = ok ? c2p[p] : make_int2(0, 0);
  }
  const int lo = k0 - ka, hi = k1 - ka;
  const int safe = cols[k0 < k1 ? k0 : (k0 > 0 ? k0 - 1 : 0)];
  float2 g0[kPmgPairs], g1[kPmgPairs];
  if (k0 < k1) {                       // (block-uniform)
#pragma unroll
    for (int j = 0; j < kPmgPairs; ++j) {   // all gathers in flight before any use
      const int e = 2 * (threadIdx.x + j * kBlock);
      g0[j] = g[(e >= lo && e < hi) ? c[j].x : safe];
      g1[j] = g[(e + 1 < hi) ? c[j].y : safe];
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPmgPairs; ++j) g0[j] = g1[j] = f2(0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < kPmgPairs; ++j) {
    const int p = threadIdx.x + j * kBlock;
    if (p < npair) {
      prod[2 * p] = f2(v[j].x * g0[j].x, v[j].y * g0[j].y);
      prod[2 * p + 1] = f2(v[j].z * g1[j].x, v[j].w * g1[j].y);
    }
  }
  __syncthreads();
  float2 s = f2(0.f, 0.f);
  for (int k = a; k < b; ++k) {
    s.x += prod[k].x;
    s.y += prod[k].y;
  }
  return s;
}

// One product with the packed operator plus what the Chebyshev iteration does
// with it, row by row:
//   res' = res - A g
//   STEP:   d' = c1 d_own + c2 dinv res'   (d_own = nullptr: 0);   x' = x + d'
//   FINAL:  x' goes out as fp64, component-blocked (z[a*n + row]); Dirichlet
//           rows (bc != 0) return the input r32 there: their rows of the
//           Jacobian are identity rows
// res_out / d_out / x_out may be nullptr (not needed); res_out may alias
// res_in and x_out may alias x_in (row-local); g must not be written.
template <bool STEP, bool FINAL>
__global__ __launch_bounds__(kBlock) void pmg_cheb_kernel(
    int n, const int* __restrict__ rowptr, const int* __restrict__ cols,
    const float2* __restrict__ vals, const int* __restrict__ rowblocks,
    const float2* __restrict__ g, const float2* res_in, float2* res_out,
    const float2* __restrict__ dinv, const float2* __restrict__ d_own, float c1,
    float c2, float2* __restrict__ d_out, const float2* x_in, float2* x_out,
    double* __restrict__ z, const unsigned char* __restrict__ bc,
    const float2* __restrict__ r32, const double* __restrict__ stop) {
  __shared__ float2 prod[kPmgTile];
  if (stopped(stop)) return;
  int r, r1;
  const float2 s = pmg_tile_row_sum(rowptr, cols, vals, rowblocks, g, prod, r, r1);
  if (r >= r1) return;
  float2 res = res_in[r];
  res.x -= s.x;
  res.y -= s.y;
  if (res_out) res_out[r] = res;
  if (!STEP) return;
  const float2 di = dinv[r];
  float2 d = f2(c2 * di.x * res.x, c2 * di.y * res.y);
  if (d_own) {
    const float2 o = d_own[r];
    d.x += c1 * o.x;
    d.y += c1 * o.y;
  }
  if (d_out) d_out[r] = d;
  float2 x = x_in[r];
  x.x += d.x;
  x.y += d.y;
  if (FINAL) {
    double zx = x.x, zy = x.y;
    if (bc) {
      const float2 in = r32[r];
      if (bc[r]) zx = in.x;
      if (bc[n + r]) zy = in.y;
    }
    z[r] = zx;
    z[static_cast<size_t>(n) + r] = zy;
  } else {
    x_out[r] = x;
  }
}

// start of a Chebyshev run from x = 0 on the fine level: the fp64 component-
// blocked input becomes r32 (kept for the post-smoothing) and res;
// d = x = dinv res / theta
__global__ __launch_bounds__(kBlock) void pmg_init_kernel(
    int n, const double* __restrict__ r, float2* __restrict__ r32,
    float2* __restrict__ res, const float2* __restrict__ dinv, float inv_theta,
    float2* __restrict__ d, float2* __restrict__ x,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const float2 v = f2(static_cast<float>(r[i]),
                        static_cast<float>(r[static_cast<size_t>(n) + i]));
    const float2 di = dinv[i];
    const float2 d0 = f2(inv_theta * di.x * v.x, inv_theta * di.y * v.y);
    r32[i] = v;
    res[i] = v;
    d[i] = d0;
    x[i] = d0;
  }
}

// the same on the coarse level (the input is already packed; res = the input
// buffer itself)
__global__ __launch_bounds__(kBlock) void pmg_init32_kernel(
    int n, const float2* __restrict__ res, const float2* __restrict__ dinv,
    float inv_theta, float2* __restrict__ d, float2* __restrict__ x,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const float2 v = res[i];
    const float2 di = dinv[i];
    const float2 d0 = f2(inv_theta * di.x * v.x, inv_theta * di.y * v.y);
    d[i] = d0;
    x[i] = d0;
  }
}

// rc = P^T res: P1 row v collects its own P2 dof (weight 1, first in its list)
// and the edge dofs around it (weight 1/2); Dirichlet rows of the coarse
// operator get 0 (bcc: 2*n1 bytes, component-blocked)
__global__ __launch_bounds__(kBlock) void pmg_restrict_kernel(
    int n1, const int* __restrict__ rptr, const int* __restrict__ rsrc,
    const float2* __restrict__ res, const unsigned char* __restrict__ bcc,
    float2* __restrict__ rc, const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n1;
       v += gridDim.x * blockDim.x) {
    const int a = rptr[v], b = rptr[v + 1];
    float2 s = res[rsrc[a]];
    float2 e = f2(0.f, 0.f);
    for (int k = a + 1; k < b; ++k) {
      const float2 t = res[rsrc[k]];
      e.x += t.x;
      e.y += t.y;
    }
    s.x += 0.5f * e.x;
    s.y += 0.5f * e.y;
    if (bcc) {
      if (bcc[v]) s.x = 0.f;
      if (bcc[n1 + v]) s.y = 0.f;
    }
    rc[v] = s;
  }
}

// x += P xc: ends[i] = the two P1 rows a P2 dof interpolates from (a vertex dof
// names its vertex twice)
__global__ __launch_bounds__(kBlock) void pmg_prolong_kernel(
    int n, const int2* __restrict__ ends, const float2* __restrict__ xc,
    float2* __restrict__ x, const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const int2 e = ends[i];
    const float2 a = xc[e.x], b = xc[e.y];
    float2 v = x[i];
    v.x += 0.5f * (a.x + b.x);
    v.y += 0.5f * (a.y + b.y);
    x[i] = v;
  }
}

// setup: vals[k] = (a00[k], a11[k]) rounded; dinv[i] = 1 / diag
__global__ void pmg_pack_kernel(int nnz, const double* __restrict__ a00,
                                const double* __restrict__ a11,
                                float2* __restrict__ vals) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += gridDim.x * blockDim.x)
    vals[k] = f2(static_cast<float>(a00[k]), static_cast<float>(a11[k]));
}

__global__ void pmg_dinv_kernel(int n, const int* __restrict__ diag_idx,
                                const double* __restrict__ a00,
                                const double* __restrict__ a11,
                                float2* __restrict__ dinv) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const int k = diag_idx[i];
    dinv[i] = f2(static_cast<float>(1.0 / a00[k]), static_cast<float>(1.0 / a11[k]));
  }
}

// power iteration for the spectral radius of D^-1 A: w = dinv (A v) -- via
// res' = 0 - A v in the product kernel -- then |w|^2 in block partials
__global__ __launch_bounds__(kBlock) void pmg_scale_norm_kernel(
    int n, const float2* __restrict__ dinv, float2* __restrict__ w,
    double* __restrict__ partial) {
  double s = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const float2 di = dinv[i];
    float2 v = w[i];
    v.x *= -di.x;
    v.y *= -di.y;
    w[i] = v;
    s += static_cast<double>(v.x) * v.x + static_cast<double>(v.y) * v.y;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(kBlock) void pmg_scale_kernel(int n, float a,
                                                          float2* __restrict__ v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    float2 t = v[i];
    t.x *= a;
    t.y *= a;
    v[i] = t;
  }
}

__global__ __launch_bounds__(kBlock) void pmg_seed_kernel(int n,
                                                         float2* __restrict__ v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    // a fixed full-spectrum vector: no RNG on the device, no host upload
    const float t = static_cast<float>(i);
    v[i] = f2(__sinf(0.7f * t) + 0.3f, __cosf(1.3f * t) - 0.2f);
  }
}

__global__ void pmg_zero_kernel(int n, float2* __restrict__ v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    v[i] = f2(0.f, 0.f);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int check_level(const flow_pmg_level* L, const char* which) {
  FLOW_REQUIRE(L->n > 0 && L->nnz > 0 && L->nblocks > 0, which);
  FLOW_REQUIRE(L->rowptr && L->cols && L->rowblocks && L->vals && L->dinv, which);
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(L->vals) % 16 == 0,
               "packed values must be 16-byte aligned");
  FLOW_REQUIRE(L->lam_max > L->lam_min && L->lam_min > 0.0,
               "Chebyshev interval (0 < lam_min < lam_max)");
  return FLOW_OK;
}

int pmg_check(const flow_pmg* M, int op_size) {
  FLOW_REQUIRE(M != nullptr, "flow_pmg is NULL");
  int rc = check_level(&M->fine, "fine level of flow_pmg");
  if (rc) return rc;
  if ((rc = check_level(&M->coarse, "coarse level of flow_pmg"))) return rc;
  FLOW_REQUIRE(2 * M->fine.n == op_size, "flow_pmg does not match the operator");
  FLOW_REQUIRE(M->pre >= 1 && M->post >= 1 && M->coarse_steps >= 1 &&
                   M->pre <= 16 && M->post <= 16 && M->coarse_steps <= 32,
               "Chebyshev step counts");
  FLOW_REQUIRE(M->ends && M->rptr && M->rsrc && M->work, "flow_pmg pointers");
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(M->work) % 16 == 0,
               "flow_pmg work must be 16-byte aligned");
  return FLOW_OK;
}

namespace {

struct Cheb {
  double theta, delta, sigma, rho;
  Cheb(double lo, double hi)
      : theta(0.5 * (hi + lo)), delta(0.5 * (hi - lo)), sigma(theta / delta),
        rho(1.0 / sigma) {}
  float first() const { return static_cast<float>(1.0 / theta); }
  // coefficients of the next step: d' = c1 d + c2 dinv res'
  void next(float* c1, float* c2) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    *c1 = static_cast<float>(rn * rho);
    *c2 = static_cast<float>(2.0 * rn / delta);
    rho = rn;
  }
};

template <bool STEP, bool FINAL>
void launch_cheb(const flow_pmg_level* L, const float2* g, const float2* res_in,
                 float2* res_out, const float2* d_own, float c1, float c2,
                 float2* d_out, const float2* x_in, float2* x_out, double* z,
                 const unsigned char* bc, const float2* r32, const double* stop,
                 hipStream_t st) {
  hipLaunchKernelGGL((pmg_cheb_kernel<STEP, FINAL>), dim3(L->nblocks),
                     dim3(kBlock), 0, st, L->n, L->rowptr, L->cols,
                     reinterpret_cast<const float2*>(L->vals), L->rowblocks, g,
                     res_in, res_out, reinterpret_cast<const float2*>(L->dinv),
                     d_own, c1, c2, d_out, x_in, x_out, z, bc, r32, stop);
}

}  // namespace

// z = M^-1 r: one two-level cycle.  r, z: fp64, component-blocked, 2 n.
int pmg_apply(const flow_pmg* M, const double* r, double* z, hipStream_t st,
              const double* stop) {
  const flow_pmg_level* F = &M->fine;
  const flow_pmg_level* C = &M->coarse;
  const int n = F->n, n1 = C->n;
  float2* w = reinterpret_cast<float2*>(M->work);
  float2* r32 = w;
  float2* res = r32 + n;
  float2* da = res + n;
  float2* db = da + n;
  float2* xa = db + n;
  float2* xb = xa + n;
  float2* crc = xb + n;
  float2* cda = crc + n1;
  float2* cdb = cda + n1;
  float2* cx = cdb + n1;
  const float2* fdinv = reinterpret_cast<const float2*>(F->dinv);
  const float2* cdinv = reinterpret_cast<const float2*>(C->dinv);
  float2* const none = nullptr;
  double* const nod = nullptr;
  const unsigned char* const nob = nullptr;
  float c1, c2;

  // pre-smoothing from x = 0
  Cheb pre(F->lam_min, F->lam_max);
  hipLaunchKernelGGL(pmg_init_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, r,
                     r32, res, fdinv, pre.first(), da, xa, stop);
  float2 *dc = da, *dn = db;
  for (int j = 1; j < M->pre; ++j) {
    pre.next(&c1, &c2);
    launch_cheb<true, false>(F, dc, res, res, dc, c1, c2, dn, xa, xa, nod, nob,
                             none, stop, st);
    float2* t = dc;
    dc = dn;
    dn = t;
  }
  // residual behind the last correction, restricted
  launch_cheb<false, false>(F, dc, res, res, none, 0.f, 0.f, none, none, none, nod,
                            nob, none, stop, st);
  hipLaunchKernelGGL(pmg_restrict_kernel, dim3(grid_for(n1)), dim3(kBlock), 0, st,
                     n1, M->rptr, M->rsrc, res, M->bc_coarse, crc, stop);
  // coarse level: Chebyshev from zero
  Cheb co(C->lam_min, C->lam_max);
  hipLaunchKernelGGL(pmg_init32_kernel, dim3(grid_for(n1)), dim3(kBlock), 0, st,
                     n1, crc, cdinv, co.first(), cda, cx, stop);
  float2 *cc = cda, *cn = cdb;
  for (int j = 1; j < M->coarse_steps; ++j) {
    co.next(&c1, &c2);
    launch_cheb<true, false>(C, cc, crc, crc, cc, c1, c2, cn, cx, cx, nod, nob,
                             none, stop, st);
    float2* t = cc;
    cc = cn;
    cn = t;
  }
  hipLaunchKernelGGL(pmg_prolong_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n,
                     reinterpret_cast<const int2*>(M->ends), cx, xa, stop);
  // post-smoothing: the first step needs the residual of the corrected x
  Cheb post(F->lam_min, F->lam_max);
  dc = da;
  dn = db;
  if (M->post == 1) {
    launch_cheb<true, true>(F, xa, r32, none, none, 0.f, post.first(), none, xa,
                            none, z, M->bc_fine, r32, stop, st);
  } else {
    launch_cheb<true, false>(F, xa, r32, res, none, 0.f, post.first(), dc, xa, xb,
                             nod, nob, none, stop, st);
    for (int j = 1; j < M->post; ++j) {
      post.next(&c1, &c2);
      if (j + 1 == M->post)
        launch_cheb<true, true>(F, dc, res, none, dc, c1, c2, none, xb, none, z,
                                M->bc_fine, r32, stop, st);
      else
        launch_cheb<true, false>(F, dc, res, res, dc, c1, c2, dn, xb, xb, nod, nob,
                                 none, stop, st);
      float2* t = dc;
      dc = dn;
      dn = t;
    }
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

extern "C" int flow_pmg_pack(int n, int nnz, const int* diag_idx,
                             const double* a00, const double* a11, float* vals,
                             float* dinv, void* stream) {
  FLOW_REQUIRE(n > 0 && nnz > 0 && diag_idx && a00 && a11 && vals && dinv,
               "flow_pmg_pack arguments");
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(vals) % 16 == 0,
               "packed values must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(pmg_pack_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, st, nnz,
                     a00, a11, reinterpret_cast<float2*>(vals));
  hipLaunchKernelGGL(pmg_dinv_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n,
                     diag_idx, a00, a11, reinterpret_cast<float2*>(dinv));
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_pmg_lambda_max(const flow_pmg_level* L, int iterations,
                                   float* work, double* dwork,
                                   double* result_host, void* stream) {
  FLOW_REQUIRE(L && L->n > 0 && L->rowptr && L->cols && L->rowblocks && L->vals &&
                   L->dinv && work && dwork && result_host && iterations >= 2,
               "flow_pmg_lambda_max arguments");
  hipStream_t st = as_stream(stream);
  const int n = L->n;
  float2* v = reinterpret_cast<float2*>(work);
  float2* w = v + n;
  float2* zero = w + n;
  float2* const none = nullptr;
  const int g = grid_for(n);
  const int gr = grid_for(n, kBlock, kRedBlocks);
  hipLaunchKernelGGL(pmg_seed_kernel, dim3(g), dim3(kBlock), 0, st, n, v);
  hipLaunchKernelGGL(pmg_zero_kernel, dim3(g), dim3(kBlock), 0, st, n, zero);
  double lam = 0.0;
  for (int it = 0; it < iterations; ++it) {
    // w = 0 - A v, then w = -dinv w = D^-1 A v and |w|^2
    hipLaunchKernelGGL((pmg_cheb_kernel<false, false>), dim3(L->nblocks),
                       dim3(kBlock), 0, st, n, L->rowptr, L->cols,
                       reinterpret_cast<const float2*>(L->vals), L->rowblocks, v,
                       zero, w, reinterpret_cast<const float2*>(L->dinv), none,
                       0.f, 0.f, none, none, none, static_cast<double*>(nullptr),
                       static_cast<const unsigned char*>(nullptr), none,
                       static_cast<const double*>(nullptr));
    hipLaunchKernelGGL(pmg_scale_norm_kernel, dim3(gr), dim3(kBlock), 0, st, n,
                       reinterpret_cast<const float2*>(L->dinv), w, dwork);
    FLOW_CHECK_LAUNCH();
    double nrm2 = 0.0;
    int rc = flow::sum_partials_host(dwork, gr, &nrm2, st);
    if (rc) return rc;
    FLOW_REQUIRE(nrm2 == nrm2 && nrm2 > 0.0, "power iteration broke down");
    // |v| = 1 on entry (after the first pass): the growth is the estimate
    const double nw = sqrt(nrm2);
    if (it > 0) lam = nw;
    hipLaunchKernelGGL(pmg_scale_kernel, dim3(g), dim3(kBlock), 0, st, n,
                       static_cast<float>(1.0 / nw), w);
    float2* t = v;
    v = w;
    w = t;
  }
  *result_host = lam;
  return FLOW_OK;
}

extern "C" int flow_pmg_apply(const flow_pmg* M, const double* r, double* z,
                              void* stream) {
  FLOW_REQUIRE(M != nullptr && r && z, "flow_pmg_apply arguments");
  int rc = pmg_check(M, 2 * M->fine.n);
  if (rc) return rc;
  return pmg_apply(M, r, z, as_stream(stream), nullptr);
}
