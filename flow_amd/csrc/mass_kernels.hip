// K18: mass-matrix solves by mixed-precision defect correction on gfx950
// (include/flow_hip.h, flow_mass).
//
// Replaces, for the velocity correction (flow/navier_stokes/
// pressure_correction.py:436-465: CG + hypre_amg on the vector mass system) and
// the callers' L2 projections (tests/test_karman_vortex_street.py:262-267), the
// Jacobi-CG of la_kernels.hip: 11-14 iterations of [fp64 product 12 B/nnz +
// 5-vector update + scalar kernel] become 3-4 defect corrections of [one fp64
// product whose epilogue leaves the scaled residual in fp32 + k-1 products with
// an fp16 copy of D^-1 M at 6 B/nnz, no dot products, one launch each].  The
// spectrum of D^-1 M is known a priori (Wathen), so the Chebyshev polynomial
// is fixed and its contraction is a bound, not an estimate.
//
// All kernels are HBM-bound CSR-stream products (csr_stream.h; the fp16 tile
// is the one of pmg_kernels.hip with ONE half per nonzero instead of a half2:
// the mass matrix is the same for both velocity components).
#include "common.h"
#include "csr_stream.h"
#include "csr_stream16.h"

#include <hip/hip_fp16.h>

namespace flow {

// ---------------------------------------------------------------------------
// fp64 residual with the fp32 epilogue: rho0 = D^-1 (b - A x)
// ---------------------------------------------------------------------------
// scalar operator (kind 0)
__global__ __launch_bounds__(kBlock) void mass_residual_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ cols,
    const double* __restrict__ vals, const int* __restrict__ rowblocks,
    const double* __restrict__ x, const double* __restrict__ b,
    const double* __restrict__ dinv, float* __restrict__ rho0,
    const double* __restrict__ stop) {
  __shared__ double prod[kTile];
  if (stopped(stop)) return;
  int r, r1;
  double bi = 0.0, di = 0.0;
  auto early = [&](int row, bool has) {
    if (has) {
      bi = b[row];
      di = dinv[row];
    }
  };
  const double s =
      stream_tile_row_sum(rowptr, cols, vals, rowblocks, x, prod, r, r1, early);
  if (r < r1) rho0[r] = static_cast<float>(di * (bi - s));
}

// one plane, both components, identity rows by mask (kind 4); x: component
// stride xs, b / dinv / mask: component stride n
__global__ __launch_bounds__(kBlock) void mass_residual_pair_kernel(
    int n, const int* __restrict__ rowptr, const int* __restrict__ cols,
    const double* __restrict__ vals, const int* __restrict__ rowblocks,
    const unsigned char* __restrict__ mask, const double* __restrict__ x, int xs,
    const double* __restrict__ b, const double* __restrict__ dinv,
    float2* __restrict__ rho0, const double* __restrict__ stop) {
  __shared__ double2 prod[kTile2];
  if (stopped(stop)) return;
  int r, r1;
  double b0 = 0.0, b1 = 0.0, d0 = 0.0, d1 = 0.0, x0 = 0.0, x1 = 0.0;
  bool m0 = true, m1 = true;
  auto early = [&](int row, bool has) {
    if (!has) return;
    b0 = b[row];
    b1 = b[static_cast<size_t>(n) + row];
    d0 = dinv[row];
    d1 = dinv[static_cast<size_t>(n) + row];
    m0 = mask[row] != 0;
    m1 = mask[n + row] != 0;
    x0 = x[row];
    x1 = x[xs + row];
  };
  const double2 s = stream_tile_pair_row_sum(rowptr, cols, vals, rowblocks, x, xs,
                                             prod, r, r1, early);
  if (r < r1) {
    const double s0 = m0 ? s.x : x0;
    const double s1 = m1 ? s.y : x1;
    rho0[r] = make_float2(static_cast<float>(d0 * (b0 - s0)),
                          static_cast<float>(d1 * (b1 - s1)));
  }
}

// the first defect of an INCREMENT solve (delta = 0): rho0 = D^-1 g, no product
// (rows [r0, r1) of n; rho0 is indexed by global row)
__global__ void mass_rho0_kernel(int n, int ncomp, const double* __restrict__ g,
                                 const double* __restrict__ dinv,
                                 float* __restrict__ rho0,
                                 const double* __restrict__ stop, int r0 = 0,
                                 int r1 = 0x7fffffff) {
  if (stopped(stop)) return;
  if (r1 > n) r1 = n;
  for (int i = r0 + blockIdx.x * blockDim.x + threadIdx.x; i < r1;
       i += gridDim.x * blockDim.x)
    for (int a = 0; a < ncomp; ++a) {
      const size_t k = static_cast<size_t>(a) * n + i;
      rho0[static_cast<size_t>(i) * ncomp + a] = static_cast<float>(dinv[k] * g[k]);
    }
}

// ---------------------------------------------------------------------------
// Chebyshev steps on the fp16 copy:  s = (D^-1 A) g  row by row, then
//   MODE 0  first product (g = rho0 = rho_in, d_0 = c0 rho0 folded in):
//             rho_1 = rho0 - c0 s ; d_1 = c1 c0 rho0 + c2 rho_1 ;
//             rho_out = rho_1 ; d_out = d_1 ; acc = c0 rho0 + d_1
//   MODE 1  step (g = d):  rho' = rho_in - s ; d' = c1 g_own + c2 rho' ;
//             rho_out = rho' (may alias rho_in: row-local) ; d_out = d' (NOT g) ;
//             acc += d'
//   MODE 2  last step: z = acc + d' ;  x += z in fp64 (component stride xs) ;
//             the workgroup's shares of z.z and x.x -> zz_part / xx_part
// Identity rows (mask, 0 = identity; nullptr: none): s = the row's own g.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void apply_mask(const unsigned char* mask, int n, int r,
                                           float own, float& s) {
  if (mask && !mask[r]) s = own;
}
__device__ __forceinline__ void apply_mask(const unsigned char* mask, int n, int r,
                                           float2 own, float2& s) {
  if (mask) {
    if (!mask[r]) s.x = own.x;
    if (!mask[n + r]) s.y = own.y;
  }
}
__device__ __forceinline__ float vaxpby(float a, float x, float b, float y) {
  return a * x + b * y;
}
__device__ __forceinline__ float2 vaxpby(float a, float2 x, float b, float2 y) {
  return make_float2(a * x.x + b * y.x, a * x.y + b * y.y);
}
// x += z, returns (z.z, y.y) of the row, y = x [+ base: x is then the increment
// of a solve around `base`, and the norm in the stopping test is the whole
// solution's].  The old x and base values are loaded EARLY (XRow, below), with
// the tile's own loads.
struct XRow {
  double x0 = 0.0, x1 = 0.0, b0 = 0.0, b1 = 0.0;
};
__device__ __forceinline__ void load_xrow(XRow& q, const double* x,
                                          const double* base, int xs, int r,
                                          float) {
  q.x0 = x[r];
  if (base) q.b0 = base[r];
}
__device__ __forceinline__ void load_xrow(XRow& q, const double* x,
                                          const double* base, int xs, int r,
                                          float2) {
  q.x0 = x[r];
  q.x1 = x[xs + r];
  if (base) {
    q.b0 = base[r];
    q.b1 = base[xs + r];
  }
}
__device__ __forceinline__ double2 add_to_x(double* x, const XRow& q, int xs,
                                           int r, float z) {
  const double xn = q.x0 + static_cast<double>(z);
  x[r] = xn;
  const double y = q.b0 + xn;
  return make_double2(static_cast<double>(z) * z, y * y);
}
__device__ __forceinline__ double2 add_to_x(double* x, const XRow& q, int xs,
                                           int r, float2 z) {
  const double x0 = q.x0 + static_cast<double>(z.x);
  const double x1 = q.x1 + static_cast<double>(z.y);
  x[r] = x0;
  x[xs + r] = x1;
  const double y0 = q.b0 + x0;
  const double y1 = q.b1 + x1;
  return make_double2(static_cast<double>(z.x) * z.x + static_cast<double>(z.y) * z.y,
                      y0 * y0 + y1 * y1);
}

template <class V, int MODE, bool PACKED>
__global__ __launch_bounds__(kBlock) void mass_cheb_kernel(
    int n, const int* __restrict__ rowptr, const int* __restrict__ cols,
    const void* __restrict__ vals, const int* __restrict__ rowblocks,
    const unsigned char* __restrict__ mask, const V* __restrict__ g,
    const V* rho_in, V* rho_out, float c0, float c1, float c2,
    V* __restrict__ d_out, V* __restrict__ acc, double* __restrict__ x,
    const double* __restrict__ xbase, int xs,
    double* __restrict__ zz_part, double* __restrict__ xx_part,
    const double* __restrict__ stop, int own_lo = 0, int own_hi = 0x7fffffff) {
  __shared__ V prod[kMassTile];
  if (stopped(stop)) return;
  // (PACKED: `vals` is the packed stream, `cols` the tiles' base columns)
  int r, r1;
  // the epilogue's operands, loaded as soon as the lane knows its row
  V own, rho_old, acc_old;
  vzero(own);
  vzero(rho_old);
  vzero(acc_old);
  XRow xr;
  auto early = [&](int row, bool has) {
    if (!has) return;
    own = g[row];
    if (MODE >= 1) {
      rho_old = rho_in[row];
      acc_old = acc[row];
    }
    if (MODE == 2) load_xrow(xr, x, xbase, xs, row, V());
  };
  V s = PACKED
            ? mass_tile_row_sum_packed<V>(rowptr,
                                          static_cast<const unsigned*>(vals), cols,
                                          rowblocks, g, prod, r, r1, early)
            : mass_tile_row_sum<V>(rowptr, cols,
                                   static_cast<const __half*>(vals), rowblocks, g,
                                   prod, r, r1, early);
  double2 dots = make_double2(0.0, 0.0);
  if (r < r1) {
    apply_mask(mask, n, r, own, s);
    if (MODE == 0) {
      const V rho = vaxpby(1.f, own, -c0, s);
      const V d = vaxpby(c1 * c0, own, c2, rho);
      rho_out[r] = rho;
      d_out[r] = d;
      acc[r] = vaxpby(c0, own, 1.f, d);
    } else {
      const V rho = vaxpby(1.f, rho_old, -1.f, s);
      const V d = vaxpby(c1, own, c2, rho);
      if (MODE == 1) {
        rho_out[r] = rho;
        d_out[r] = d;
        acc[r] = vaxpby(1.f, acc_old, 1.f, d);
      } else {
        dots = add_to_x(x, xr, xs, r, vaxpby(1.f, acc_old, 1.f, d));
        // (K15: the sums only count the rank's OWN rows; the ghost rows it
        // also advances are counted by their owners)
        if (r < own_lo || r >= own_hi) dots = make_double2(0.0, 0.0);
      }
    }
  }
  if (MODE == 2) {
    const double zz = block_sum(dots.x);
    const double xx = block_sum(dots.y);
    if (threadIdx.x == 0) {
      zz_part[blockIdx.x] = zz;
      xx_part[blockIdx.x] = xx;
    }
  }
}

// One workgroup: z.z and x.x from the partials of the last product; the
// stopping test  contraction |z_k| <= max(rtol |x_{k+1}|, atol)  (x already
// holds x_{k+1} = x_k + z_k, the iterate the test accepts: |B r_{k+1}| <=
// |I - B M| |z_k|); S[kConvIt] = corrections applied, S[kRes2] = z.z
constexpr int kMassScalarBlock = 1024;
__global__ __launch_bounds__(kMassScalarBlock) void mass_scalar_kernel(
    int nparts, const double* __restrict__ zz_part,
    const double* __restrict__ xx_part, double contraction2, double rtol2,
    double atol2, double* __restrict__ S) {
  __shared__ double wsum[2][kMassScalarBlock / 64];
  if (stopped(S + kDone)) return;
  double z0 = 0.0, z1 = 0.0, x0 = 0.0, x1 = 0.0;
  int i = threadIdx.x;
  for (; i + kMassScalarBlock < nparts; i += 2 * kMassScalarBlock) {
    z0 += load_scalar(zz_part + i);
    z1 += load_scalar(zz_part + i + kMassScalarBlock);
    x0 += load_scalar(xx_part + i);
    x1 += load_scalar(xx_part + i + kMassScalarBlock);
  }
  for (; i < nparts; i += kMassScalarBlock) {
    z0 += load_scalar(zz_part + i);
    x0 += load_scalar(xx_part + i);
  }
  double zz = z0 + z1, xx = x0 + x1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    zz += __shfl_down(zz, off, 64);
    xx += __shfl_down(xx, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    wsum[0][threadIdx.x >> 6] = zz;
    wsum[1][threadIdx.x >> 6] = xx;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  zz = xx = 0.0;
  for (int w = 0; w < kMassScalarBlock / 64; ++w) {
    zz += wsum[0][w];
    xx += wsum[1][w];
  }
  const double k = load_scalar(S + kIter) + 1.0;
  const bool nan = !(zz == zz) || !(xx == xx);
  // The stopping test rests on the contraction of I - B M: a priori a bound
  // on its SPECTRAL RADIUS (the scaled element matrix), which the 2-norm of a
  // step may exceed transiently -- I - B M is a polynomial in D^-1 M, only
  // similar to a symmetric matrix, and on a graded mesh (lumped diagonal
  // spread ~16) the similarity is far from an isometry.  So the test uses
  // what the iteration is SEEN to do where that is worse than the bound:
  // z_{k+1} = (I - B M) z_k, observed ratio |z_{k+1}| / |z_k|.  Only an
  // iteration that does not contract at all (ratio >= 1 above the floor fp64
  // leaves in the defect: not the mass matrix of straight P1 / P2 triangles?)
  // is given up: S[kDone] = 3, the caller falls back to Jacobi-CG.
  const double prev = load_scalar(S + kGamma);
  const double seen2 = (k > 1.0 && prev > 0.0) ? zz / prev : 0.0;
  const double c2 = fmax(contraction2, fmin(seen2, 1.0));
  if (nan || c2 * zz <= fmax(rtol2 * xx, atol2)) {
    store_scalar(S + kConvIt, k);
    store_scalar(S + kDone, nan ? 2.0 : 1.0);
  } else if (k > 1.0 && seen2 >= 1.0 && zz > 1.0e-26 * xx) {
    store_scalar(S + kConvIt, k);
    store_scalar(S + kTmp, sqrt(seen2));
    store_scalar(S + kDone, 3.0);
  }
  store_scalar(S + kGamma, zz);
  store_scalar(S + kIter, k);
  store_scalar(S + kRes2, zz);
  store_scalar(S + kB2, xx);
}

// setup: vals16[k] = half(vals[k] / diagonal of row(k)); a lane per row
__global__ void mass_pack_kernel(int n, const int* __restrict__ rowptr,
                                 const int* __restrict__ diag_idx,
                                 const double* __restrict__ vals,
                                 __half* __restrict__ vals16) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const double inv = 1.0 / vals[diag_idx[i]];
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      vals16[k] = __float2half_rn(static_cast<float>(vals[k] * inv));
  }
}

// setup of the packed stream: a workgroup per tile finds the tile's lowest
// column (cbase) and writes (offset << 16 | fp16 bits) per nonzero; *overflow
// is set when an offset does not fit in 16 bits (the caller then keeps the
// plain stream)
__global__ __launch_bounds__(kBlock) void mass_pack16_kernel(
    const int* __restrict__ rowblocks, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const int* __restrict__ diag_idx,
    const double* __restrict__ vals, int* __restrict__ cbase,
    unsigned* __restrict__ packed, int* __restrict__ overflow) {
  __shared__ int wmin[kBlock / 64];
  const int tile = blockIdx.x;
  const int r0 = rowblocks[tile], r1 = rowblocks[tile + 1];
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  int m = 0x7fffffff;
  for (int k = k0 + threadIdx.x; k < k1; k += kBlock) m = min(m, cols[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = m;
  __syncthreads();
  int base = wmin[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w) base = min(base, wmin[w]);
  if (k0 >= k1) base = 0;
  if (threadIdx.x == 0) cbase[tile] = base;
  const int r = r0 + threadIdx.x;
  if (r < r1) {
    const double inv = 1.0 / vals[diag_idx[r]];
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int off = cols[k] - base;
      if (off > 0xffff) atomicOr(overflow, 1);
      const unsigned short h =
          __half_as_ushort(__float2half_rn(static_cast<float>(vals[k] * inv)));
      packed[k] = (static_cast<unsigned>(off & 0xffff) << 16) | h;
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

struct Cheb {
  double theta, delta, sigma, rho;
  Cheb(double lo, double hi)
      : theta(0.5 * (hi + lo)), delta(0.5 * (hi - lo)), sigma(theta / delta),
        rho(1.0 / sigma) {}
  float first() const { return static_cast<float>(1.0 / theta); }
  // coefficients of the next step: d' = c1 d + c2 rho'
  void next(float* c1, float* c2) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    *c1 = static_cast<float>(rn * rho);
    *c2 = static_cast<float>(2.0 * rn / delta);
    rho = rn;
  }
};

// tiles of the products of one correction: by default every product runs over
// M->rowblocks16; K15: product j over its own, shrinking row range
struct MassTiles {
  const int* rowblocks[16];
  int nblocks[16];
  int shift;            // the fp32 vectors are windows starting at this row
  int own_lo, own_hi;   // rows whose sums count
};

template <class V, bool PACKED>
int correction(const flow_mass* M, const MassTiles& T, double* x,
               const double* xbase, double* zz_part, double* xx_part,
               const double* stop, hipStream_t st) {
  const flow_operator* A = M->A;
  const int n = A->n;
  // (PACKED: the packed stream and the tiles' base columns take the places of
  // the fp16 values and the column indices)
  const void* v16 = PACKED ? M->packed16 : M->vals16;
  const int* cols = PACKED ? M->cbase16 : A->cols;
  // (the vectors are indexed by global row: windows are shifted)
  V* w = reinterpret_cast<V*>(M->work16) - T.shift;
  const size_t len = static_cast<size_t>(M->work16_rows > 0 ? M->work16_rows : n);
  V* rho0 = w;
  V* rho = rho0 + len;
  V* d[2] = {rho + len, rho + 2 * len};
  V* acc = rho + 3 * len;
  const dim3 blk(kBlock);
  Cheb ch(M->lam_min, M->lam_max);
  const float c0 = ch.first();
  float c1, c2;
  V* const none = nullptr;
  double* const nod = nullptr;
  ch.next(&c1, &c2);
  hipLaunchKernelGGL((mass_cheb_kernel<V, 0, PACKED>), dim3(T.nblocks[0]), blk, 0,
                     st, n, A->rowptr, cols, v16, T.rowblocks[0], A->rowmask, rho0,
                     rho0, rho, c0, c1, c2, d[0], acc, nod, nod, n, nod, nod, stop,
                     0, 0x7fffffff);
  const int products = M->steps - 1;
  for (int j = 1; j + 1 < products; ++j) {
    ch.next(&c1, &c2);
    hipLaunchKernelGGL((mass_cheb_kernel<V, 1, PACKED>), dim3(T.nblocks[j]), blk,
                       0, st, n, A->rowptr, cols, v16, T.rowblocks[j], A->rowmask,
                       d[(j - 1) & 1], rho, rho, 0.f, c1, c2, d[j & 1], acc, nod,
                       nod, n, nod, nod, stop, 0, 0x7fffffff);
  }
  ch.next(&c1, &c2);
  hipLaunchKernelGGL((mass_cheb_kernel<V, 2, PACKED>),
                     dim3(T.nblocks[products - 1]), blk, 0, st, n, A->rowptr, cols,
                     v16, T.rowblocks[products - 1], A->rowmask,
                     d[(products - 2) & 1], rho, none, 0.f, c1, c2, none, acc, x,
                     xbase, n, zz_part, xx_part, stop, T.own_lo, T.own_hi);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

static MassTiles default_tiles(const flow_mass* M) {
  MassTiles T;
  for (int j = 0; j < 16; ++j) {
    T.rowblocks[j] = M->rowblocks16;
    T.nblocks[j] = M->nblocks16;
  }
  T.shift = 0;
  T.own_lo = 0;
  T.own_hi = 0x7fffffff;
  return T;
}

}  // namespace

static int mass_check(const flow_mass* M) {
  FLOW_REQUIRE(M != nullptr && M->A != nullptr, "flow_mass is NULL");
  int rc = check_operator(M->A);
  if (rc) return rc;
  FLOW_REQUIRE(M->A->kind == 0 || M->A->kind == 4,
               "flow_mass: operator kind 0 (scalar) or 4 (one plane, two "
               "components)");
  FLOW_REQUIRE(M->dinv && M->rowblocks16 && (M->vals16 || M->packed16) &&
                   M->work16 && M->nblocks16 > 0,
               "flow_mass pointers");
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(M->vals16) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(M->packed16) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(M->A->cols) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(M->work16) % 16 == 0,
               "flow_mass: vals16 / packed16, cols and work16 must be 16-byte "
               "aligned");
  FLOW_REQUIRE(M->packed16 == nullptr || M->cbase16 != nullptr,
               "flow_mass: the packed stream needs the tiles' base columns");
  FLOW_REQUIRE(M->lam_max > M->lam_min && M->lam_min > 0.0,
               "Chebyshev interval (0 < lam_min < lam_max)");
  FLOW_REQUIRE(M->steps >= 3 && M->steps <= 16, "Chebyshev steps (3..16)");
  FLOW_REQUIRE(M->contraction > 0.0 && M->contraction <= 1.0,
               "contraction bound in (0, 1]");
  return FLOW_OK;
}

// xbase != nullptr: the INCREMENT form -- b is the defect g = b' - M xbase of
// the start xbase, x the increment; x_is_zero: it is zero on entry, the first
// correction then needs no product with M
static int mass_solve(const flow_mass* M, const double* b, double* x,
                      const double* xbase, bool x_is_zero, double rtol,
                      double atol, int maxit, int first_check, double* work,
                      int* iters_host, double* resid_host, hipStream_t st) {
  const flow_operator* A = M->A;
  double* S = work + 3 * kRedBlocks;
  double* zz_part = work + FLOW_REDUCE_WORK;
  double* xx_part = zz_part + M->nblocks16;
  const double* stop = S + kDone;
  const double c2 = M->contraction * M->contraction;
  int rc;
  if ((rc = fill(kNumSlots, 0.0, S, st))) return rc;
  auto one = [&]() -> int {
    if (x_is_zero) {
      hipLaunchKernelGGL(mass_rho0_kernel, dim3(grid_for(A->n)), dim3(kBlock), 0,
                         st, A->n, A->kind == 4 ? 2 : 1, b, M->dinv, M->work16,
                         stop);
      x_is_zero = false;
    } else if (A->kind == 4) {
      hipLaunchKernelGGL(mass_residual_pair_kernel, dim3(A->nblocks), dim3(kBlock),
                         0, st, A->n, A->rowptr, A->cols, A->vals[0], A->rowblocks,
                         A->rowmask, x, A->n, b, M->dinv,
                         reinterpret_cast<float2*>(M->work16), stop);
    } else {
      hipLaunchKernelGGL(mass_residual_kernel, dim3(A->nblocks), dim3(kBlock), 0, st,
                         A->rowptr, A->cols, A->vals[0], A->rowblocks, x, b, M->dinv,
                         M->work16, stop);
    }
    const MassTiles T = default_tiles(M);
    if (A->kind == 4) {
      rc = M->packed16 ? correction<float2, true>(M, T, x, xbase, zz_part, xx_part,
                                                  stop, st)
                       : correction<float2, false>(M, T, x, xbase, zz_part,
                                                   xx_part, stop, st);
    } else {
      rc = M->packed16 ? correction<float, true>(M, T, x, xbase, zz_part, xx_part,
                                                 stop, st)
                       : correction<float, false>(M, T, x, xbase, zz_part, xx_part,
                                                  stop, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(mass_scalar_kernel, dim3(1), dim3(kMassScalarBlock), 0, st,
                       M->nblocks16, zz_part, xx_part, c2, rtol * rtol, atol * atol,
                       S);
    FLOW_CHECK_LAUNCH();
    return FLOW_OK;
  };
  // one correction as a HIP graph where the launch rate bounds it
  // (graph_replay.hip): two bodies -- from x = 0, and the general one
  const bool replay = replay_wanted(A->n, kReplayMass);
  KeyHash base;
  if (replay) {
    base.pod(0x6d73ull);      // "ms"
    key_operator(base, A);
    base.obj(M).pod(b).pod(x).pod(xbase).pod(work).pod(rtol).pod(atol);
  }
  auto one_replayed = [&]() -> int {
    if (!replay) return one();
    KeyHash key = base;
    key.pod(x_is_zero);
    hipGraphExec_t graph = nullptr;
    int nodes = 0, r;
    const bool was_zero = x_is_zero;
    if ((r = replay_prepare(key.h, kReplayMass, st, one, &graph, &nodes))) return r;
    x_is_zero = was_zero;     // (a capture has run `one` without launching)
    if (!graph) return one();
    x_is_zero = false;
    return replay_launch(graph, nodes, st);
  };
  double state[kNumSlots];
  int launched = 0;
  while (true) {
    const int batch = (launched == 0 && first_check > 0) ? first_check : 1;
    const int todo = (maxit - launched < batch) ? maxit - launched : batch;
    for (int k = 0; k < todo; ++k)
      if ((rc = one_replayed())) return rc;
    launched += todo;
    if ((rc = read_state(S, state, st))) return rc;
    const double zz = state[kRes2];
    if (state[kDone] == 2.0 || !(zz == zz)) {
      *iters_host = static_cast<int>(state[kConvIt]);
      *resid_host = zz;
      set_error("mass solve broke down (NaN) at correction %d", *iters_host);
      return FLOW_NOT_CONVERGED;
    }
    if (state[kDone] == 1.0) {
      *iters_host = static_cast<int>(state[kConvIt]);
      *resid_host = sqrt(zz);
      return FLOW_OK;
    }
    if (state[kDone] == 3.0) {
      *iters_host = static_cast<int>(state[kConvIt]);
      *resid_host = sqrt(zz);
      set_error("mass solve: correction %d contracted by %.3f (no "
                "contraction; the vouched bound is %.3f: not the mass matrix "
                "of straight P1/P2 triangles?)", *iters_host, state[kTmp],
                M->contraction);
      return FLOW_NOT_CONVERGED;
    }
    if (launched >= maxit) {
      *iters_host = launched;
      *resid_host = sqrt(zz);
      set_error("mass solve did not converge in %d defect corrections: |z| = "
                "%.3e, |x| = %.3e", launched, sqrt(zz), sqrt(state[kB2]));
      return FLOW_NOT_CONVERGED;
    }
  }
}


// ---------------------------------------------------------------------------
// K15: the same solver on the strips of a node (flow_shard_mass_solve).
// The polynomial of one correction reaches steps - 1 matrix hops: with the
// scaled defect rho0 known on a ghost zone `steps` vertex columns deep, the
// rank computes its products on shrinking row ranges -- product j on the rows
// within steps - 1 - j hops of its own -- and ends with x advanced on its own
// rows AND its first ghost layer, without communication inside the correction.
// ONE collective per correction carries the deep halo of rho0 and, riding
// along, the norms that decide about the correction before.  Rows computed
// redundantly by two ranks come out bitwise equal: the same inputs, the same
// order of summation per row.
// ---------------------------------------------------------------------------
constexpr int kMassPackBlock = 1024;

// block 0: sums of the partial lists of the last product (z.z, x.x over the
// rank's own rows) -> buf[0], buf[1] (first: zeros); blocks >= 1: the halo
// slots of rho0 at buf + 4 (own boundary rows as doubles, zeros elsewhere)
__global__ __launch_bounds__(kMassPackBlock) void shard_mass_pack_kernel(
    flow_rows R, int ncomp, int first, int nparts,
    const double* __restrict__ zz_part, const double* __restrict__ xx_part,
    const float* __restrict__ rho0, double* __restrict__ buf,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  if (blockIdx.x > 0) {
    const int total = ncomp * R.nhalo;
    for (int t = (blockIdx.x - 1) * blockDim.x + threadIdx.x; t < total;
         t += (gridDim.x - 1) * blockDim.x) {
      const int a = t / R.nhalo, k = t - a * R.nhalo;
      double v = 0.0;
#pragma unroll
      for (int sd = 0; sd < 2; ++sd)
        if (k >= R.send_slot[sd] && k < R.send_slot[sd] + R.send_len[sd])
          v = static_cast<double>(
              rho0[static_cast<size_t>(R.send_row[sd] + (k - R.send_slot[sd])) *
                       ncomp + a]);
      buf[4 + t] = v;
    }
    return;
  }
  __shared__ double wsum[2][kMassPackBlock / 64];
  double zz = 0.0, xx = 0.0;
  if (!first)
    for (int i = threadIdx.x; i < nparts; i += kMassPackBlock) {
      zz += load_scalar(zz_part + i);
      xx += load_scalar(xx_part + i);
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    zz += __shfl_down(zz, off, 64);
    xx += __shfl_down(xx, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    wsum[0][threadIdx.x >> 6] = zz;
    wsum[1][threadIdx.x >> 6] = xx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    zz = xx = 0.0;
    for (int w = 0; w < kMassPackBlock / 64; ++w) {
      zz += wsum[0][w];
      xx += wsum[1][w];
    }
    buf[0] = zz;
    buf[1] = xx;
    buf[2] = 0.0;
    buf[3] = 0.0;
  }
}

// ghost rows of rho0 <- the neighbours' slots; thread 0 of block 0: the verdict
// on the correction BEFORE (its norms have just been summed over the ranks):
// `applied` corrections are in x when it passes
__global__ void shard_mass_unpack_kernel(flow_rows R, int ncomp, int first,
                                         int applied, double contraction2,
                                         double rtol2, double atol2,
                                         const double* __restrict__ buf,
                                         float* __restrict__ rho0,
                                         double* __restrict__ S) {
  if (stopped(S + kDone)) return;
  const int per = R.recv_len[0] + R.recv_len[1];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ncomp * per;
       t += gridDim.x * blockDim.x) {
    const int a = t / per, k = t - a * per;
    const int sd = k < R.recv_len[0] ? 0 : 1;
    const int j = sd == 0 ? k : k - R.recv_len[0];
    rho0[static_cast<size_t>(R.recv_row[sd] + j) * ncomp + a] = static_cast<float>(
        load_scalar(buf + 4 + static_cast<size_t>(a) * R.nhalo + R.recv_slot[sd] + j));
  }
  if (blockIdx.x != 0 || threadIdx.x != 0 || first) return;
  const double zz = load_scalar(buf), xx = load_scalar(buf + 1);
  const bool nan = !(zz == zz) || !(xx == xx);
  // (the observed contraction where it is worse than the a-priori bound, and
  // the verdict on an iteration that does not contract: mass_scalar_kernel;
  // the sums are the same on every rank, so is the verdict)
  const double prev = load_scalar(S + kGamma);
  const double seen2 = (applied > 1 && prev > 0.0) ? zz / prev : 0.0;
  const double c2 = fmax(contraction2, fmin(seen2, 1.0));
  if (nan || c2 * zz <= fmax(rtol2 * xx, atol2)) {
    store_scalar(S + kConvIt, static_cast<double>(applied));
    store_scalar(S + kDone, nan ? 2.0 : 1.0);
  } else if (applied > 1 && seen2 >= 1.0 && zz > 1.0e-26 * xx) {
    store_scalar(S + kConvIt, static_cast<double>(applied));
    store_scalar(S + kTmp, sqrt(seen2));
    store_scalar(S + kDone, 3.0);
  }
  store_scalar(S + kGamma, zz);
  store_scalar(S + kIter, static_cast<double>(applied));
  store_scalar(S + kRes2, zz);
  store_scalar(S + kB2, xx);
}

static int shard_mass_solve(const flow_comm* C, const flow_rows* R,
                            const flow_mass* M, const flow_mass_strips* L,
                            const double* b, double* x, const double* xbase,
                            bool x_is_zero, double rtol, double atol, int maxit,
                            int first_check, double* work, int* iters_host,
                            double* resid_host, hipStream_t st) {
  const flow_operator* A = M->A;
  const int ncomp = A->kind == 4 ? 2 : 1;
  const int products = M->steps - 1;
  double* S = work + 3 * kRedBlocks;
  double* zz_part = work + FLOW_REDUCE_WORK;
  const int nlast = L->nblocks16[products - 1];
  double* xx_part = zz_part + nlast;
  const double* stop = S + kDone;
  const double c2 = M->contraction * M->contraction;
  MassTiles T;
  for (int j = 0; j < products; ++j) {
    T.rowblocks[j] = L->rowblocks16[j];
    T.nblocks[j] = L->nblocks16[j];
  }
  T.shift = R->e0;
  T.own_lo = R->r0;
  T.own_hi = R->r1;
  float* rho0 = M->work16 - static_cast<ptrdiff_t>(R->e0) * ncomp;
  const int count = 4 + ncomp * R->nhalo;
  const int gp = 1 + grid_for(ncomp * R->nhalo > 0 ? ncomp * R->nhalo : 1,
                              kMassPackBlock, 64);
  const int per = R->recv_len[0] + R->recv_len[1];
  const int gs = grid_for(ncomp * per > 0 ? ncomp * per : 1);
  int rc;
  if ((rc = fill(kNumSlots, 0.0, S, st))) return rc;
  int k = 0;                       // corrections enqueued so far
  // one pass: the defect on the own rows, the collective (deep halo of rho0 +
  // the norms of correction k - 1), the verdict on k - 1, correction k
  auto one = [&]() -> int {
    if (x_is_zero && k == 0) {
      hipLaunchKernelGGL(mass_rho0_kernel, dim3(grid_for(R->r1 - R->r0)),
                         dim3(kBlock), 0, st, A->n, ncomp, b, M->dinv, rho0, stop,
                         R->r0, R->r1);
    } else if (A->kind == 4) {
      hipLaunchKernelGGL(mass_residual_pair_kernel, dim3(A->nblocks), dim3(kBlock),
                         0, st, A->n, A->rowptr, A->cols, A->vals[0], A->rowblocks,
                         A->rowmask, x, A->n, b, M->dinv,
                         reinterpret_cast<float2*>(rho0), stop);
    } else {
      hipLaunchKernelGGL(mass_residual_kernel, dim3(A->nblocks), dim3(kBlock), 0, st,
                         A->rowptr, A->cols, A->vals[0], A->rowblocks, x, b, M->dinv,
                         rho0, stop);
    }
    hipLaunchKernelGGL(shard_mass_pack_kernel, dim3(gp), dim3(kMassPackBlock), 0,
                       st, *R, ncomp, k == 0 ? 1 : 0, nlast, zz_part, xx_part, rho0,
                       C->buf, stop);
    FLOW_CHECK_LAUNCH();
    // (every rank issues the same sequence of collectives, whatever the flag)
    (void)count;
    if ((rc = exchange_halo(C, R, ncomp, 4, 4, st))) return rc;
    hipLaunchKernelGGL(shard_mass_unpack_kernel, dim3(gs), dim3(kBlock), 0, st, *R,
                       ncomp, k == 0 ? 1 : 0, k, c2, rtol * rtol, atol * atol,
                       C->buf, rho0, S);
    FLOW_CHECK_LAUNCH();
    if (A->kind == 4)
      rc = correction<float2, false>(M, T, x, xbase, zz_part, xx_part, stop, st);
    else
      rc = correction<float, false>(M, T, x, xbase, zz_part, xx_part, stop, st);
    ++k;
    return rc;
  };
  double state[kNumSlots];
  bool started = false;
  while (true) {
    // the verdict on correction j comes with pass j + 1: `first_check`
    // corrections (what the previous call needed) need one pass more
    int todo = (!started && first_check > 0) ? first_check + 1 : 1;
    if (!started && first_check == 0) todo = 2;
    started = true;
    if (k + todo > maxit + 1) todo = maxit + 1 - k;
    for (int i = 0; i < todo; ++i)
      if ((rc = one())) return rc;
    if ((rc = read_state(S, state, st))) return rc;
    const double zz = state[kRes2];
    if (state[kDone] == 2.0 || !(zz == zz)) {
      *iters_host = static_cast<int>(state[kConvIt]);
      *resid_host = zz;
      set_error("sharded mass solve broke down (NaN) at correction %d",
                *iters_host);
      return FLOW_NOT_CONVERGED;
    }
    if (state[kDone] == 1.0) {
      *iters_host = static_cast<int>(state[kConvIt]);
      *resid_host = sqrt(zz);
      return FLOW_OK;
    }
    if (state[kDone] == 3.0) {
      *iters_host = static_cast<int>(state[kConvIt]);
      *resid_host = sqrt(zz);
      set_error("sharded mass solve: correction %d contracted by %.3f (no "
                "contraction; the vouched bound is %.3f)", *iters_host,
                state[kTmp], M->contraction);
      return FLOW_NOT_CONVERGED;
    }
    if (k > maxit) {
      *iters_host = maxit;
      *resid_host = sqrt(zz);
      set_error("sharded mass solve did not converge in %d defect corrections: "
                "|z| = %.3e, |x| = %.3e", maxit, sqrt(zz), sqrt(state[kB2]));
      return FLOW_NOT_CONVERGED;
    }
  }
}

}  // namespace flow

using namespace flow;

extern "C" int flow_mass_pack(int n, const int* rowptr, const int* diag_idx,
                              const double* vals, void* vals16, void* stream) {
  FLOW_REQUIRE(n > 0 && rowptr && diag_idx && vals && vals16,
               "flow_mass_pack arguments");
  hipLaunchKernelGGL(mass_pack_kernel, dim3(grid_for(n)), dim3(kBlock), 0,
                     as_stream(stream), n, rowptr, diag_idx, vals,
                     static_cast<__half*>(vals16));
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_mass_pack16(int n, int nblocks16, const int* rowblocks16,
                                const int* rowptr, const int* cols,
                                const int* diag_idx, const double* vals,
                                int* cbase16, void* packed16, int* overflow_dev,
                                void* stream) {
  FLOW_REQUIRE(n > 0 && nblocks16 > 0 && rowblocks16 && rowptr && cols &&
                   diag_idx && vals && cbase16 && packed16 && overflow_dev,
               "flow_mass_pack16 arguments");
  hipLaunchKernelGGL(mass_pack16_kernel, dim3(nblocks16), dim3(kBlock), 0,
                     as_stream(stream), rowblocks16, rowptr, cols, diag_idx, vals,
                     cbase16, static_cast<unsigned*>(packed16), overflow_dev);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_mass_solve(const flow_mass* M, const double* b, double* x,
                               double rtol, double atol, int maxit,
                               int first_check, double* work, size_t work_len,
                               int* iters_host, double* resid_host,
                               void* stream) {
  int rc = mass_check(M);
  if (rc) return rc;
  FLOW_REQUIRE(b && x && work && iters_host && resid_host, "solver pointers");
  FLOW_REQUIRE(rtol >= 0.0 && atol >= 0.0 && maxit >= 1 && first_check >= 0,
               "solver tolerances");
  FLOW_REQUIRE(work_len >= FLOW_REDUCE_WORK + 2 * static_cast<size_t>(M->nblocks16),
               "mass solve workspace too small");
  return mass_solve(M, b, x, nullptr, false, rtol, atol, maxit, first_check, work,
                    iters_host, resid_host, as_stream(stream));
}

// y = a + b  (the solution of an increment solve; b == nullptr: y = a)
__global__ void mass_sum_kernel(int n, const double* __restrict__ a,
                                const double* __restrict__ b,
                                double* __restrict__ y) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    y[i] = b ? a[i] + b[i] : a[i];
}

extern "C" int flow_mass_solve_increment(const flow_mass* M, const double* g,
                                         const double* xbase,
                                         const double* delta0, double* x,
                                         double rtol, double atol, int maxit,
                                         int first_check, double* work,
                                         size_t work_len, int* iters_host,
                                         double* resid_host, void* stream) {
  int rc = mass_check(M);
  if (rc) return rc;
  FLOW_REQUIRE(g && xbase && x && work && iters_host && resid_host,
               "solver pointers");
  FLOW_REQUIRE(rtol >= 0.0 && atol >= 0.0 && maxit >= 1 && first_check >= 0,
               "solver tolerances");
  const size_t N = M->A->kind == 4 ? 2 * static_cast<size_t>(M->A->n) : M->A->n;
  const size_t head = FLOW_REDUCE_WORK + 2 * static_cast<size_t>(M->nblocks16);
  FLOW_REQUIRE(work_len >= head + (head & 1) + N, "mass solve workspace too small");
  hipStream_t st = as_stream(stream);
  double* delta = work + head + (head & 1);
  if (delta0) {
    hipLaunchKernelGGL(mass_sum_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st,
                       static_cast<int>(N), delta0, static_cast<const double*>(nullptr),
                       delta);
    FLOW_CHECK_LAUNCH();
  } else if ((rc = fill(static_cast<int>(N), 0.0, delta, st))) {
    return rc;
  }
  rc = mass_solve(M, g, delta, xbase, delta0 == nullptr, rtol, atol, maxit,
                  first_check, work, iters_host, resid_host, st);
  // (the increment found so far is added also when the solve did not converge,
  // like the in-place solve leaves its last iterate)
  hipLaunchKernelGGL(mass_sum_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st,
                     static_cast<int>(N), xbase, delta, x);
  FLOW_CHECK_LAUNCH();
  return rc;
}

// y[i] = a[i] + b[i] on the rows [lo, hi) of ncomp components (stride n)
__global__ void mass_sum_rows_kernel(int n, int ncomp, int lo, int hi,
                                     const double* __restrict__ a,
                                     const double* __restrict__ b,
                                     double* __restrict__ y) {
  const int m = hi - lo;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ncomp * m;
       t += gridDim.x * blockDim.x) {
    const int c = t / m, i = lo + (t - c * m);
    const size_t k = static_cast<size_t>(c) * n + i;
    y[k] = a[k] + b[k];
  }
}

extern "C" int flow_shard_mass_solve(
    const flow_comm* comm, const flow_rows* rows, const flow_mass* M,
    const flow_mass_strips* levels, const double* b, const double* xbase,
    const double* delta0, double* x, double rtol, double atol, int maxit,
    int first_check, double* work, size_t work_len, int* iters_host,
    double* resid_host, void* stream) {
  int rc = mass_check(M);
  if (rc) return rc;
  if ((rc = check_rows(rows))) return rc;
  FLOW_REQUIRE(M->vals16 != nullptr,
               "sharded mass solve: the plain fp16 stream (vals16)");
  FLOW_REQUIRE(levels && b && x && work && iters_host && resid_host,
               "solver pointers");
  FLOW_REQUIRE(rtol >= 0.0 && atol >= 0.0 && maxit >= 1 && first_check >= 0,
               "solver tolerances");
  const flow_operator* A = M->A;
  FLOW_REQUIRE(A->n == rows->n, "operator / row ranges");
  const int ncomp = A->kind == 4 ? 2 : 1;
  const int products = M->steps - 1;
  FLOW_REQUIRE(levels->nlevels == products,
               "one set of row blocks per product of a correction");
  for (int j = 0; j < products; ++j)
    FLOW_REQUIRE(levels->rowblocks16[j] && levels->nblocks16[j] > 0,
                 "row blocks of a product");
  FLOW_REQUIRE(M->work16_rows == rows->e1 - rows->e0,
               "work16 must cover the rank's window [e0, e1)");
  if ((rc = check_comm(comm, 4 + static_cast<long long>(ncomp) * rows->nhalo)))
    return rc;
  const size_t N = static_cast<size_t>(ncomp) * A->n;
  const size_t head = FLOW_REDUCE_WORK +
                      2 * static_cast<size_t>(levels->nblocks16[products - 1]);
  FLOW_REQUIRE(work_len >= head + (head & 1) + (xbase ? N : 0),
               "sharded mass solve workspace too small");
  hipStream_t st = as_stream(stream);
  if (!xbase) {
    FLOW_REQUIRE(delta0 == nullptr, "delta0 belongs to the increment form");
    return shard_mass_solve(comm, rows, M, levels, b, x, nullptr, false, rtol,
                            atol, maxit, first_check, work, iters_host,
                            resid_host, st);
  }
  // increment form: delta in a vector of its own (global length: the kernels
  // index by global row; valid on the own + first ghost rows)
  double* delta = work + head + (head & 1);
  if (delta0) {
    hipLaunchKernelGGL(mass_sum_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st,
                       static_cast<int>(N), delta0,
                       static_cast<const double*>(nullptr), delta);
    FLOW_CHECK_LAUNCH();
  } else if ((rc = fill(static_cast<int>(N), 0.0, delta, st))) {
    return rc;
  }
  rc = shard_mass_solve(comm, rows, M, levels, b, delta, xbase,
                        delta0 == nullptr, rtol, atol, maxit, first_check, work,
                        iters_host, resid_host, st);
  // x = xbase + delta where the last product ran (own + first ghost rows)
  const int lo = levels->row_lo_last, hi = levels->row_hi_last;
  hipLaunchKernelGGL(mass_sum_rows_kernel, dim3(grid_for(ncomp * (hi - lo))),
                     dim3(kBlock), 0, st, A->n, ncomp, lo, hi, xbase, delta, x);
  FLOW_CHECK_LAUNCH();
  return rc;
}
