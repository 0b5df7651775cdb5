// K17: p-multigrid / Chebyshev preconditioner of the Newton systems of the
// tentative velocity on gfx950 (include/flow_hip.h, flow_pmg).
//
// Stands in -- like the multicolour ILU(0) of ilu_kernels.hip -- for the sparse
// LU behind the reference's Newton solve (flow/navier_stokes/
// pressure_correction.py:224-254) as the right preconditioner of the flexible
// GMRES of la_kernels.hip.  Why another one: the ILU(0) sweeps are a chain of
// ~15 dependent launches bound by gather latency (0.30 of the HBM roofline) and
// a multicolour factorisation is a weak one (33 applications per solve at
// CFL-sized steps on the 10 M-DoF workload; this one: 13).  It consists of
// CSR-stream products only:
//
//   fine level    the two diagonal blocks of the assembled P2 Jacobian, row-
//                 scaled with their diagonals (D^-1 A: entries of size <= ~1),
//                 rounded to fp16 and interleaved -- ONE index + one 4-byte
//                 value load per nonzero serves both velocity components, 8 B
//                 per nonzero where the fp64 Jacobian planes have 20 --,
//                 smoothed with `pre` / `post` steps of the Chebyshev iteration;
//   coarse level  the P1 discretisation of the same linearised operator on the
//                 same mesh (P1 is a subspace of P2: vertex dofs copy, edge
//                 dofs average their end points), `coarse_steps` Chebyshev
//                 steps from a zero start -- at CFL-sized steps the P1 operator
//                 is mass-dominated (condition ~10) and needs no further levels.
//
// A smoother does not need its matrix to more than three digits (tools/
// precond_lab.py: the same GMRES counts with fp64, fp32 and fp16 entries), and
// the Krylov method around it is flexible: all vectors inside are fp32, both
// components interleaved (float2 per dof: one 8-byte gather per nonzero).  The
// iteration runs on the SCALED residual rho = D^-1 (r - A x):
//   d_0 = rho_0 / theta;  rho_{k+1} = rho_k - (D^-1 A) d_k;
//   d_{k+1} = c1 d_k + c2 rho_{k+1};  x = sum_k d_k
// so that no kernel but the first reads the diagonal.  Every kernel is
// HBM-bound: 8 B per nonzero + a few float2 vectors.
#include "common.h"
#include "csr_stream16.h"

#include <hip/hip_fp16.h>

namespace flow {

constexpr int kPmgQuads = 2;                      // quads of nonzeros per lane
constexpr int kPmgTile = kBlock * 4 * kPmgQuads;  // LDS products per workgroup
static_assert(FLOW_PMG_NNZ_PER_BLOCK == kPmgTile - 4,
              "tile minus alignment slack (base aligned down to a multiple of 4)");
static_assert(kPmgTile == kMassTile, "one tile shape for both fp16 streams");

__device__ __forceinline__ float2 f2(float a, float b) { return make_float2(a, b); }

struct Half2x4 {           // four nonzeros: (block 0, block 1) each, 16 bytes
  __half2 v[4];
};
static_assert(sizeof(Half2x4) == 16, "packed quad");

// One tile of the packed stream -- rows [r0, r1) of workgroup blockIdx.x (at
// most kBlock rows, kPmgTile - 4 nonzeros): every lane loads kPmgQuads 16-byte
// quads of values and of column indices (the tile base is aligned down to a
// multiple of four nonzeros), all of them and all gathers behind them in flight
// before the first use -- the kernel is bound by the chain of dependent loads
// of a tile (row blocks -> row pointers -> indices -> gathers), so a tile
// carries as many bytes as the LDS products of a workgroup allow (16 KB: still
// eight workgroups per CU) --, the products of both blocks with the gathered
// vector g go through LDS, then lane i sums row r0 + i.  Window-safe like
// stream_tile_row_sum (la_kernels.hip): g is only dereferenced for the tile's
// own nonzeros (slack and idle lanes gather the tile's first column).
// C16: the column indices are 16-bit offsets from the tile's lowest column
// (cols16 / cbase of flow_pmg_level: 6 B per nonzero instead of 8).
// early(r, has_row): called as soon as the lane knows its row -- the caller
// issues the loads of its epilogue there, so that they travel with the tile's
// own loads instead of forming one more link of the dependent chain behind
// the row sum (a workgroup lives ~6 us, a link of the chain is ~1 us of it).
template <bool C16, class Early>
__device__ __forceinline__ float2 pmg_tile_row_sum(
    const int* __restrict__ rowptr, const void* __restrict__ cols_any,
    const int* __restrict__ cbase,
    const __half2* __restrict__ vals, const int* __restrict__ rowblocks,
    const float2* __restrict__ g, float2* __restrict__ prod, int& r, int& r1,
    Early early) {
  const int* __restrict__ cols = static_cast<const int*>(cols_any);
  const unsigned short* __restrict__ cols16 =
      static_cast<const unsigned short*>(cols_any);
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int r0 = rowblocks[tile];
  r1 = rowblocks[tile + 1];
  const int base = C16 ? cbase[tile] : 0;
  const int k0 = rowptr[r0];
  const int k1 = rowptr[r1];
  const int ka = k0 & ~3;
  r = r0 + threadIdx.x;
  early(r, r < r1);
  int a = 0, b = 0;
  if (r < r1) {
    a = rowptr[r] - ka;
    b = rowptr[r + 1] - ka;
  }
  const int lo = k0 - ka, hi = k1 - ka;          // hi <= kPmgTile - 1
  const Half2x4* __restrict__ vq = reinterpret_cast<const Half2x4*>(vals + ka);
  const int4* __restrict__ cq = reinterpret_cast<const int4*>(cols + ka);
  const ushort4* __restrict__ cq16 =
      reinterpret_cast<const ushort4*>(cols16 + ka);
  Half2x4 v[kPmgQuads];
  int4 c[kPmgQuads];
#pragma unroll
  for (int q = 0; q < kPmgQuads; ++q) {
    const int p = threadIdx.x + q * kBlock;
    c[q] = make_int4(0, 0, 0, 0);
    if (4 * p < hi) {
      v[q] = vq[p];
      if (C16) {
        const ushort4 u = cq16[p];
        c[q] = make_int4(base + u.x, base + u.y, base + u.z, base + u.w);
      } else {
        c[q] = cq[p];
      }
    }
  }
  if (k0 < k1) {                                   // (block-uniform)
    const int safe = C16 ? base + cols16[k0] : cols[k0];
    float2 gg[kPmgQuads][4];
#pragma unroll
    for (int q = 0; q < kPmgQuads; ++q) {          // all gathers in flight
      const int e0 = 4 * (threadIdx.x + q * kBlock);
      const int cc[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j;
        gg[q][j] = g[(e >= lo && e < hi) ? cc[j] : safe];
      }
    }
#pragma unroll
    for (int q = 0; q < kPmgQuads; ++q) {
      const int e0 = 4 * (threadIdx.x + q * kBlock);
      if (e0 < hi) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float2 w = __half22float2(v[q].v[j]);
          prod[e0 + j] = f2(w.x * gg[q][j].x, w.y * gg[q][j].y);
        }
      }
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
// with it, row by row:   rho' = rho_in - (D^-1 A) g
//   MODE 0  residual only:  rho_out = rho', times d_own[row] when that is given
//                           (the diagonal: the unscaled residual r - A x)
//   MODE 1  step:           d' = c1 d_own + c2 rho'  (d_own = nullptr: 0);
//                           rho_out = rho' (nullptr: not needed), d_out = d',
//                           x += d' (nullptr: the caller sums the d's later)
//   MODE 2  last step:      z = x + d_own + d_extra + d'  as fp64, component-
//                           blocked (z[a*n + row]); Dirichlet rows (bc != 0)
//                           return the input r there: identity rows of J
// rho_out may alias rho_in (row-local); g must not be written.
// FMT 0: half2 values + int32 columns; 1: half2 values + 16-bit column offsets;
// 2: ONE plane for both components as the packed stream of csr_stream16.h
// (`vals`: a 32-bit word per nonzero, fp16 value | 16-bit column offset) --
// rows flagged in idrows (component-blocked, stride n) are identity rows of
// their component.
// zs: component stride of z and rin (n; 0 in scalar mode -- both components
// then carry the same numbers and z has n entries)
template <int MODE, int FMT>
__global__ __launch_bounds__(kBlock) void pmg_cheb_kernel(
    int n, int zs, const int* __restrict__ rowptr, const void* __restrict__ cols,
    const int* __restrict__ cbase,
    const void* __restrict__ vals, const unsigned char* __restrict__ idrows,
    const int* __restrict__ rowblocks,
    const float2* __restrict__ g, const float2* rho_in, float2* rho_out,
    const float2* __restrict__ d_own, float c1, float c2,
    float2* __restrict__ d_out, float2* __restrict__ x,
    const float2* __restrict__ d_extra, double* __restrict__ z,
    const unsigned char* __restrict__ bc, const double* __restrict__ rin,
    const double* __restrict__ stop) {
  __shared__ float2 prod[kPmgTile];
  if (stopped(stop)) return;
  int r, r1;
  float2 s;
  // the epilogue's operands, loaded as soon as the row is known
  float2 rho = f2(0.f, 0.f), own = f2(0.f, 0.f), acc = f2(0.f, 0.f),
         extra = f2(0.f, 0.f);
  auto early = [&](int row, bool has) {
    if (!has) return;
    rho = rho_in[row];
    if (d_own) own = d_own[row];
    if (MODE >= 1 && x) acc = x[row];
    if (MODE == 2 && d_extra) extra = d_extra[row];
  };
  if (FMT == 2) {
    s = mass_tile_row_sum_packed<float2>(rowptr,
                                         static_cast<const unsigned*>(vals),
                                         cbase, rowblocks, g, prod, r, r1, early);
    if (r < r1 && idrows) {
      if (idrows[r]) s.x = g[r].x;
      if (idrows[static_cast<size_t>(n) + r]) s.y = g[r].y;
    }
  } else {
    s = pmg_tile_row_sum<FMT == 1>(rowptr, cols, cbase,
                                   static_cast<const __half2*>(vals), rowblocks,
                                   g, prod, r, r1, early);
  }
  if (r >= r1) return;
  rho.x -= s.x;
  rho.y -= s.y;
  if (MODE == 0) {
    if (d_own) {
      rho.x *= own.x;
      rho.y *= own.y;
    }
    rho_out[r] = rho;
    return;
  }
  float2 d = f2(c2 * rho.x, c2 * rho.y);
  if (d_own) {
    d.x += c1 * own.x;
    d.y += c1 * own.y;
  }
  if (MODE == 1) {
    if (rho_out) rho_out[r] = rho;
    d_out[r] = d;
    if (x) {
      acc.x += d.x;
      acc.y += d.y;
      x[r] = acc;
    }
    return;
  }
  acc.x += own.x + d.x + extra.x;
  acc.y += own.y + d.y + extra.y;
  double zx = acc.x, zy = acc.y;
  if (bc) {
    if (bc[r]) zx = rin[r];
    if (bc[n + r]) zy = rin[static_cast<size_t>(zs) + r];
  }
  z[r] = zx;
  if (zs) z[static_cast<size_t>(zs) + r] = zy;
}

// start of the cycle: the fp64 component-blocked input becomes the scaled
// residual rho0 = D^-1 r (kept for the post-smoothing), rho = rho0 and
// d0 = rho0 / theta
__global__ __launch_bounds__(kBlock) void pmg_init_kernel(
    int n, int rs, const double* __restrict__ r, const float2* __restrict__ dinv,
    float inv_theta, float2* __restrict__ rho0, float2* __restrict__ rho,
    float2* __restrict__ d, const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const float2 di = dinv[i];
    const float2 v = f2(static_cast<float>(r[i]) * di.x,
                        static_cast<float>(r[static_cast<size_t>(rs) + i]) * di.y);
    rho0[i] = v;
    rho[i] = v;
    d[i] = f2(inv_theta * v.x, inv_theta * v.y);
  }
}

// coarse residual, scaled: rhoc = Dc^-1 P^T res -- P1 row v collects its own P2
// dof (weight 1, first in its list) and the edge dofs around it (weight 1/2);
// Dirichlet rows of the coarse operator get 0 (bcc: 2*n1 bytes, component-
// blocked); xc = dc0 = rhoc / theta_c starts the coarse iteration
__global__ __launch_bounds__(kBlock) void pmg_restrict_kernel(
    int n1, const int* __restrict__ rptr, const int* __restrict__ rsrc,
    const float2* __restrict__ res, const float2* __restrict__ dinv_c,
    const unsigned char* __restrict__ bcc, float inv_theta_c,
    float2* __restrict__ rhoc, float2* __restrict__ dc, float2* __restrict__ xc,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  constexpr int kAhead = 8;      // list entries in flight (a vertex has ~7)
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n1;
       v += gridDim.x * blockDim.x) {
    const int a = rptr[v], b = rptr[v + 1];
    int idx[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) idx[k] = rsrc[a + k < b ? a + k : a];
    float2 t[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) t[k] = res[idx[k]];
    float2 s = t[0];
    float2 e = f2(0.f, 0.f);
#pragma unroll
    for (int k = 1; k < kAhead; ++k)
      if (a + k < b) {
        e.x += t[k].x;
        e.y += t[k].y;
      }
    for (int k = a + kAhead; k < b; ++k) {
      const float2 u = res[rsrc[k]];
      e.x += u.x;
      e.y += u.y;
    }
    const float2 di = dinv_c[v];
    s.x = (s.x + 0.5f * e.x) * di.x;
    s.y = (s.y + 0.5f * e.y) * di.y;
    if (bcc) {
      if (bcc[v]) s.x = 0.f;
      if (bcc[n1 + v]) s.y = 0.f;
    }
    rhoc[v] = s;
    const float2 d0 = f2(inv_theta_c * s.x, inv_theta_c * s.y);
    dc[v] = d0;
    xc[v] = d0;
  }
}

// x = d_a + d_b + d_c + P (sum of the coarse corrections): ends[i] = the two P1
// rows a P2 dof interpolates from (a vertex dof names its vertex twice);
// nullptr terms are skipped
__global__ __launch_bounds__(kBlock) void pmg_prolong_kernel(
    int n, const int2* __restrict__ ends, const float2* __restrict__ da,
    const float2* __restrict__ db, const float2* __restrict__ dc,
    const float2* __restrict__ xc, float2* __restrict__ x,
    const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const int2 e = ends[i];
    const float2 a = xc[e.x], b = xc[e.y];
    float2 v = da[i];
    if (db) {
      const float2 t = db[i];
      v.x += t.x;
      v.y += t.y;
    }
    if (dc) {
      const float2 t = dc[i];
      v.x += t.x;
      v.y += t.y;
    }
    v.x += 0.5f * (a.x + b.x);
    v.y += 0.5f * (a.y + b.y);
    x[i] = v;
  }
}

// setup: vals[k] = half2((a00, a11)[k] / their diagonals of row(k)) -- 0 where
// keep[k] == 0 (couplings that leave a rank's diagonal block) --; diag / dinv
// per row.  A lane per row (setup only: once per refactorisation).
__global__ void pmg_pack_kernel(int n, const int* __restrict__ rowptr,
                                const int* __restrict__ diag_idx,
                                const double* __restrict__ a00,
                                const double* __restrict__ a11,
                                const unsigned char* __restrict__ keep,
                                __half2* __restrict__ vals,
                                float2* __restrict__ diag,
                                float2* __restrict__ dinv) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const int kd = diag_idx[i];
    const double d0 = a00[kd], d1 = a11[kd];
    const double i0 = 1.0 / d0, i1 = 1.0 / d1;
    diag[i] = f2(static_cast<float>(d0), static_cast<float>(d1));
    dinv[i] = f2(static_cast<float>(i0), static_cast<float>(i1));
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const bool in = keep == nullptr || keep[k] != 0;
      vals[k] = in ? __floats2half2_rn(static_cast<float>(a00[k] * i0),
                                       static_cast<float>(a11[k] * i1))
                   : __floats2half2_rn(0.f, 0.f);
    }
  }
}

// setup of the ONE-plane level, step 1: idrows[a n + i] = 1 where row i of block
// a is an identity row (no off-diagonal entry inside the block: a Dirichlet
// dof of component a)
__global__ void pmg_idrows_kernel(int n, const int* __restrict__ rowptr,
                                  const int* __restrict__ diag_idx,
                                  const double* __restrict__ a00,
                                  const double* __restrict__ a11,
                                  const unsigned char* __restrict__ keep,
                                  unsigned char* __restrict__ idrows) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const int kd = diag_idx[i];
    bool id0 = true, id1 = true;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      if (k == kd || (keep != nullptr && keep[k] == 0)) continue;
      if (a00[k] != 0.0) id0 = false;
      if (a11[k] != 0.0) id1 = false;
    }
    idrows[i] = id0 ? 1 : 0;
    idrows[static_cast<size_t>(n) + i] = id1 ? 1 : 0;
  }
}

// step 2: the packed stream (fp16 value | 16-bit column offset from cbase[tile],
// csr_stream16.h).  The two diagonal blocks of the Jacobian differ by the
// reaction term of the Newton linearisation, +-(du_a/dx_a) M -- of the size of
// the off-diagonal blocks the cycle drops anyway, and their mean is the Oseen
// operator (div u ~ 0): ONE plane serves both components (tools/ab_bench.sh:
// the same GMRES counts).  Entry (i, j) = mean over the components in which
// neither row i nor column j is a Dirichlet dof, over the mean diagonal of the
// free components of row i; identity rows are applied by flag (idrows), their
// diag / dinv entries are the blocks' own.  A workgroup per tile.
__global__ __launch_bounds__(kBlock) void pmg_pack1_kernel(
    int n, const int* __restrict__ rowblocks, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const int* __restrict__ diag_idx,
    const double* __restrict__ a00, const double* __restrict__ a11,
    const unsigned char* __restrict__ keep,
    const unsigned char* __restrict__ idrows, const int* __restrict__ cbase,
    unsigned* __restrict__ packed, float2* __restrict__ diag,
    float2* __restrict__ dinv) {
  const int tile = blockIdx.x;
  const int r0 = rowblocks[tile], r1 = rowblocks[tile + 1];
  const int base = cbase[tile];
  const int i = r0 + threadIdx.x;
  if (i >= r1) return;
  const int kd = diag_idx[i];
  const bool f0 = idrows[i] == 0, f1 = idrows[static_cast<size_t>(n) + i] == 0;
  const double e0 = a00[kd], e1 = a11[kd];
  const double d = f0 && f1 ? 0.5 * (e0 + e1) : (f0 ? e0 : (f1 ? e1 : 1.0));
  const double inv = 1.0 / d;
  const double d0 = f0 ? d : e0, d1 = f1 ? d : e1;
  diag[i] = f2(static_cast<float>(d0), static_cast<float>(d1));
  dinv[i] = f2(static_cast<float>(1.0 / d0), static_cast<float>(1.0 / d1));
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    const int j = cols[k];
    double v = 0.0;
    if (keep == nullptr || keep[k] != 0) {
      const bool u0 = f0 && idrows[j] == 0;
      const bool u1 = f1 && idrows[static_cast<size_t>(n) + j] == 0;
      if (u0 && u1)
        v = 0.5 * (a00[k] + a11[k]);
      else if (u0)
        v = a00[k];
      else if (u1)
        v = a11[k];
    }
    const unsigned short h =
        __half_as_ushort(__float2half_rn(static_cast<float>(v * inv)));
    packed[k] = (static_cast<unsigned>((j - base) & 0xffff) << 16) | h;
  }
}

// setup: 16-bit column offsets from each tile's lowest column (a workgroup per
// tile); *overflow set when one does not fit
__global__ __launch_bounds__(kBlock) void pmg_cols16_kernel(
    const int* __restrict__ rowblocks, const int* __restrict__ rowptr,
    const int* __restrict__ cols, int* __restrict__ cbase,
    unsigned short* __restrict__ cols16, int* __restrict__ overflow) {
  __shared__ int wmin[kBlock / 64];
  const int tile = blockIdx.x;
  const int k0 = rowptr[rowblocks[tile]], k1 = rowptr[rowblocks[tile + 1]];
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
  for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
    const int off = cols[k] - base;
    if (off > 0xffff) atomicOr(overflow, 1);
    cols16[k] = static_cast<unsigned short>(off & 0xffff);
  }
}

// power iteration for the spectral radius of D^-1 A: |w|^2 in block partials
// (w = -(D^-1 A) v comes out of the product kernel with rho_in = 0)
__global__ __launch_bounds__(kBlock) void pmg_norm_kernel(
    int n, const float2* __restrict__ w, double* __restrict__ partial) {
  double s = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const float2 v = w[i];
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

// the norm out of the block partials, on the device (one workgroup): out[0] =
// |w|, so that the power iteration runs without a read-back per step
__global__ __launch_bounds__(kBlock) void pmg_norm_finish_kernel(
    int nparts, const double* __restrict__ partial, double* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) s += load_scalar(partial + i);
  s = block_sum(s);
  if (threadIdx.x == 0) out[0] = sqrt(s);
}

// v *= 1 / *nrm (the norm the launch before left on the device)
__global__ __launch_bounds__(kBlock) void pmg_scale_dev_kernel(
    int n, const double* __restrict__ nrm, float2* __restrict__ v) {
  const float a = static_cast<float>(1.0 / load_scalar(nrm));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    float2 t = v[i];
    t.x *= a;
    t.y *= a;
    v[i] = t;
  }
}

__global__ void pmg_copy_kernel(int n, const float2* __restrict__ src,
                                float2* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    dst[i] = src[i];
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
  FLOW_REQUIRE(L->rowptr && L->cols && L->rowblocks && (L->vals || L->packed) &&
                   L->diag && L->dinv,
               which);
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(L->vals) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(L->packed) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(L->cols) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(L->cols16) % 16 == 0,
               "packed values and column indices must be 16-byte aligned");
  FLOW_REQUIRE((L->cols16 == nullptr && L->packed == nullptr) ||
                   L->cbase != nullptr,
               "16-bit column offsets need the tiles' base columns");
  FLOW_REQUIRE(L->packed == nullptr || L->idrows != nullptr,
               "the one-plane stream needs the identity-row flags");
  FLOW_REQUIRE(L->lam_max > L->lam_min && L->lam_min > 0.0,
               "Chebyshev interval (0 < lam_min < lam_max)");
  return FLOW_OK;
}

int pmg_check(const flow_pmg* M, int op_size) {
  FLOW_REQUIRE(M != nullptr, "flow_pmg is NULL");
  int rc = check_level(&M->fine, "fine level of flow_pmg");
  if (rc) return rc;
  if ((rc = check_level(&M->coarse, "coarse level of flow_pmg"))) return rc;
  FLOW_REQUIRE((M->scalar ? 1 : 2) * M->fine.n == op_size,
               "flow_pmg does not match the operator");
  FLOW_REQUIRE(M->pre >= 1 && M->post >= 1 && M->coarse_steps >= 1 &&
                   M->pre <= 3 && M->post <= 3 && M->coarse_steps <= 64,
               "Chebyshev step counts (pre, post: 1..3; coarse: 1..64)");
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
  // coefficients of the next step: d' = c1 d + c2 rho'
  void next(float* c1, float* c2) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    *c1 = static_cast<float>(rn * rho);
    *c2 = static_cast<float>(2.0 * rn / delta);
    rho = rn;
  }
};

template <int MODE>
void launch_cheb(const flow_pmg_level* L, const float2* g, const float2* rho_in,
                 float2* rho_out, const float2* d_own, float c1, float c2,
                 float2* d_out, float2* x, const float2* d_extra, double* z,
                 const unsigned char* bc, const double* rin, const double* stop,
                 hipStream_t st, int zstride = -1) {
  const int zs = zstride < 0 ? L->n : zstride;
  if (L->packed)
    hipLaunchKernelGGL((pmg_cheb_kernel<MODE, 2>), dim3(L->nblocks), dim3(kBlock),
                       0, st, L->n, zs, L->rowptr, static_cast<const void*>(nullptr),
                       L->cbase, L->packed, L->idrows, L->rowblocks, g, rho_in,
                       rho_out, d_own, c1, c2, d_out, x, d_extra, z, bc, rin,
                       stop);
  else if (L->cols16)
    hipLaunchKernelGGL((pmg_cheb_kernel<MODE, 1>), dim3(L->nblocks), dim3(kBlock),
                       0, st, L->n, zs, L->rowptr,
                       static_cast<const void*>(L->cols16), L->cbase, L->vals,
                       static_cast<const unsigned char*>(nullptr), L->rowblocks,
                       g, rho_in, rho_out, d_own, c1, c2, d_out, x, d_extra, z,
                       bc, rin, stop);
  else
    hipLaunchKernelGGL((pmg_cheb_kernel<MODE, 0>), dim3(L->nblocks), dim3(kBlock),
                       0, st, L->n, zs, L->rowptr, static_cast<const void*>(L->cols),
                       L->cbase, L->vals,
                       static_cast<const unsigned char*>(nullptr), L->rowblocks,
                       g, rho_in, rho_out, d_own, c1, c2, d_out, x, d_extra, z,
                       bc, rin, stop);
}

}  // namespace

// z = M^-1 r: one two-level cycle.  r, z: fp64, component-blocked, 2 n.
int pmg_apply(const flow_pmg* M, const double* r, double* z, hipStream_t st,
              const double* stop) {
  const flow_pmg_level* F = &M->fine;
  const flow_pmg_level* C = &M->coarse;
  const int n = F->n, n1 = C->n;
  float2* w = reinterpret_cast<float2*>(M->work);
  float2* rho0 = w;
  float2* rho = rho0 + n;
  float2* d[3] = {rho + n, rho + 2 * static_cast<size_t>(n),
                  rho + 3 * static_cast<size_t>(n)};
  float2* x = rho + 4 * static_cast<size_t>(n);
  float2* crho = x + n;
  float2* cd[2] = {crho + n1, crho + 2 * static_cast<size_t>(n1)};
  float2* cx = crho + 3 * static_cast<size_t>(n1);
  const float2* fdiag = reinterpret_cast<const float2*>(F->diag);
  const float2* fdinv = reinterpret_cast<const float2*>(F->dinv);
  const float2* cdinv = reinterpret_cast<const float2*>(C->dinv);
  float2* const none = nullptr;
  double* const nod = nullptr;
  const unsigned char* const nob = nullptr;
  float c1, c2;

  // pre-smoothing from x = 0: d[0 .. pre-1]
  Cheb pre(F->lam_min, F->lam_max);
  // (scalar mode: ONE fp64 component in and out, carried in both lanes)
  const int zs = M->scalar ? 0 : n;
  hipLaunchKernelGGL(pmg_init_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, zs,
                     r, fdinv, pre.first(), rho0, rho, d[0], stop);
  for (int j = 1; j < M->pre; ++j) {
    pre.next(&c1, &c2);
    launch_cheb<1>(F, d[j - 1], rho, rho, d[j - 1], c1, c2, d[j], none, none, nod,
                   nob, nod, stop, st);
  }
  // residual behind the last correction (unscaled: times the diagonal),
  // restricted
  launch_cheb<0>(F, d[M->pre - 1], rho, rho, fdiag, 0.f, 0.f, none, none, none, nod,
                 nob, nod, stop, st);
  Cheb co(C->lam_min, C->lam_max);
  hipLaunchKernelGGL(pmg_restrict_kernel, dim3(grid_for(n1)), dim3(kBlock), 0, st,
                     n1, M->rptr, M->rsrc, rho, cdinv, M->bc_coarse, co.first(),
                     crho, cd[0], cx, stop);
  // coarse level: Chebyshev from zero, xc = sum of its corrections
  for (int j = 1; j < M->coarse_steps; ++j) {
    co.next(&c1, &c2);
    float2* dn = cd[j & 1];
    const float2* dc = cd[(j - 1) & 1];
    launch_cheb<1>(C, dc, crho, crho, dc, c1, c2, dn, cx, none, nod, nob, nod,
                   stop, st);
  }
  hipLaunchKernelGGL(pmg_prolong_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n,
                     reinterpret_cast<const int2*>(M->ends), d[0],
                     M->pre > 1 ? d[1] : none, M->pre > 2 ? d[2] : none, cx, x,
                     stop);
  // post-smoothing: rho = rho0 - (D^-1 A) x, then `post` steps; the last one
  // writes z = x + all its corrections
  Cheb post(F->lam_min, F->lam_max);
  if (M->post == 1) {
    launch_cheb<2>(F, x, rho0, none, none, 0.f, post.first(), none, x, none, z,
                   M->bc_fine, r, stop, st, zs);
  } else {
    launch_cheb<1>(F, x, rho0, rho, none, 0.f, post.first(), d[0], none, none, nod,
                   nob, nod, stop, st);
    for (int j = 1; j < M->post; ++j) {
      post.next(&c1, &c2);
      if (j + 1 == M->post)
        launch_cheb<2>(F, d[j - 1], rho, none, d[j - 1], c1, c2, none, x,
                       j == 2 ? d[0] : none, z, M->bc_fine, r, stop, st, zs);
      else
        launch_cheb<1>(F, d[j - 1], rho, rho, d[j - 1], c1, c2, d[j], none, none,
                       nod, nob, nod, stop, st);
    }
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

extern "C" int flow_pmg_pack(int n, int nnz, const int* rowptr,
                             const int* diag_idx, const double* a00,
                             const double* a11, const unsigned char* keep,
                             void* vals, float* diag, float* dinv, void* stream) {
  FLOW_REQUIRE(n > 0 && nnz > 0 && rowptr && diag_idx && a00 && a11 && vals &&
                   diag && dinv,
               "flow_pmg_pack arguments");
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(vals) % 16 == 0,
               "packed values must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(pmg_pack_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n,
                     rowptr, diag_idx, a00, a11, keep,
                     reinterpret_cast<__half2*>(vals),
                     reinterpret_cast<float2*>(diag),
                     reinterpret_cast<float2*>(dinv));
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_pmg_pack1(int n, int nnz, int nblocks, const int* rowblocks,
                              const int* rowptr, const int* cols,
                              const int* diag_idx, const double* a00,
                              const double* a11, const unsigned char* keep,
                              const int* cbase, unsigned char* idrows,
                              void* packed, float* diag, float* dinv,
                              void* stream) {
  FLOW_REQUIRE(n > 0 && nnz > 0 && nblocks > 0 && rowblocks && rowptr && cols &&
                   diag_idx && a00 && a11 && cbase && idrows && packed && diag &&
                   dinv,
               "flow_pmg_pack1 arguments");
  FLOW_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0,
               "the packed stream must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(pmg_idrows_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n,
                     rowptr, diag_idx, a00, a11, keep, idrows);
  hipLaunchKernelGGL(pmg_pack1_kernel, dim3(nblocks), dim3(kBlock), 0, st, n,
                     rowblocks, rowptr, cols, diag_idx, a00, a11, keep, idrows,
                     cbase, static_cast<unsigned*>(packed),
                     reinterpret_cast<float2*>(diag),
                     reinterpret_cast<float2*>(dinv));
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_pmg_cols16(int nblocks, const int* rowblocks,
                               const int* rowptr, const int* cols, int* cbase,
                               void* cols16, int* overflow_dev, void* stream) {
  FLOW_REQUIRE(nblocks > 0 && rowblocks && rowptr && cols && cbase && cols16 &&
                   overflow_dev,
               "flow_pmg_cols16 arguments");
  hipLaunchKernelGGL(pmg_cols16_kernel, dim3(nblocks), dim3(kBlock), 0,
                     as_stream(stream), rowblocks, rowptr, cols, cbase,
                     static_cast<unsigned short*>(cols16), overflow_dev);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_pmg_lambda_max(const flow_pmg_level* L, int iterations,
                                   float* work, double* dwork, float* start,
                                   double* result_host, void* stream) {
  FLOW_REQUIRE(L && L->n > 0 && L->rowptr && L->cols && L->rowblocks &&
                   (L->vals || L->packed) && work && dwork && result_host &&
                   iterations >= 2,
               "flow_pmg_lambda_max arguments");
  hipStream_t st = as_stream(stream);
  const int n = L->n;
  float2* v = reinterpret_cast<float2*>(work);
  float2* w = v + n;
  float2* zero = w + n;
  float2* const none = nullptr;
  float2* const keep = reinterpret_cast<float2*>(start);
  const int g = grid_for(n);
  const int gr = grid_for(n, kBlock, kRedBlocks);
  double* nrm = dwork + kRedBlocks;    // (behind the partials; S: 3 kRedBlocks)
  // start: the fixed full-spectrum vector, or the iterate the previous call
  // left (flagged by a nonzero first entry of its norm slot: `start` is
  // zero-initialised by the caller)
  hipLaunchKernelGGL(pmg_seed_kernel, dim3(g), dim3(kBlock), 0, st, n, v);
  hipLaunchKernelGGL(pmg_zero_kernel, dim3(g), dim3(kBlock), 0, st, n, zero);
  if (keep) {
    // |keep|^2 == 0: never written -> stay with the seed
    hipLaunchKernelGGL(pmg_norm_kernel, dim3(gr), dim3(kBlock), 0, st, n, keep,
                       dwork);
    FLOW_CHECK_LAUNCH();
    double have = 0.0;
    int rc = flow::sum_partials_host(dwork, gr, &have, st);
    if (rc) return rc;
    if (have == have && have > 0.0)
      hipLaunchKernelGGL(pmg_copy_kernel, dim3(g), dim3(kBlock), 0, st, n, keep, v);
  }
  double lam = 0.0;
  for (int it = 0; it < iterations; ++it) {
    // w = 0 - (D^-1 A) v, |w| and w / |w|; |v| = 1 on entry (after the first
    // pass): the growth is the estimate.  Only the last pass reads its norm
    // back; the others normalise on the device.
    launch_cheb<0>(L, v, zero, w, none, 0.f, 0.f, none, none, none,
                   static_cast<double*>(nullptr),
                   static_cast<const unsigned char*>(nullptr),
                   static_cast<const double*>(nullptr),
                   static_cast<const double*>(nullptr), st);
    hipLaunchKernelGGL(pmg_norm_kernel, dim3(gr), dim3(kBlock), 0, st, n, w, dwork);
    if (it + 1 < iterations) {
      hipLaunchKernelGGL(pmg_norm_finish_kernel, dim3(1), dim3(kBlock), 0, st, gr,
                         dwork, nrm);
      hipLaunchKernelGGL(pmg_scale_dev_kernel, dim3(g), dim3(kBlock), 0, st, n,
                         nrm, w);
    } else {
      FLOW_CHECK_LAUNCH();
      double nrm2 = 0.0;
      int rc = flow::sum_partials_host(dwork, gr, &nrm2, st);
      if (rc) return rc;
      FLOW_REQUIRE(nrm2 == nrm2 && nrm2 > 0.0, "power iteration broke down");
      lam = sqrt(nrm2);
      hipLaunchKernelGGL(pmg_scale_kernel, dim3(g), dim3(kBlock), 0, st, n,
                         static_cast<float>(1.0 / lam), w);
    }
    float2* t = v;
    v = w;
    w = t;
  }
  if (keep)
    hipLaunchKernelGGL(pmg_copy_kernel, dim3(g), dim3(kBlock), 0, st, n, v, keep);
  FLOW_CHECK_LAUNCH();
  *result_host = lam;
  return FLOW_OK;
}

extern "C" int flow_pmg_apply(const flow_pmg* M, const double* r, double* z,
                              void* stream) {
  FLOW_REQUIRE(M != nullptr && r && z, "flow_pmg_apply arguments");
  int rc = pmg_check(M, (M->scalar ? 1 : 2) * M->fine.n);
  if (rc) return rc;
  return pmg_apply(M, r, z, as_stream(stream), nullptr);
}
