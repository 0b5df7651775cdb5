// Development harness (not part of the product): SpMV kernel variants timed
// against each other on the real pressure / mass matrices.  The winner is ported
// into flow_amd/csrc/la_kernels.hip.  Build: make -C tools ; run: tools/spmv_tune.py
#include <hip/hip_runtime.h>
#include <cstdio>

#define KB 256
#define TILE 2048

// V0: the shipped kernel (reference point)
__global__ __launch_bounds__(KB) void v0(int n, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[TILE];
  const int r0 = rb[blockIdx.x], r1 = rb[blockIdx.x + 1];
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  for (int k = k0 + threadIdx.x; k < k1; k += KB) prod[k - k0] = vals[k] * x[cols[k]];
  __syncthreads();
  const int r = r0 + threadIdx.x;
  if (r < r1) {
    const int a = rowptr[r] - k0, b = rowptr[r + 1] - k0;
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    y[r] = s;
  }
}

// V1: all loads of a thread issued before the first use (8 per thread)
__global__ __launch_bounds__(KB) void v1(int n, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[TILE];
  const int r0 = rb[blockIdx.x], r1 = rb[blockIdx.x + 1];
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  const int r = r0 + threadIdx.x;
  int a = 0, b = 0;
  if (r < r1) { a = rowptr[r] - k0; b = rowptr[r + 1] - k0; }
  double v[8]; int c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + threadIdx.x + j * KB;
    const bool ok = k < k1;
    v[j] = ok ? vals[k] : 0.0;
    c[j] = ok ? cols[k] : 0;
  }
  double xv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xv[j] = x[c[j]];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = threadIdx.x + j * KB;
    if (k0 + k < k1) prod[k] = v[j] * xv[j];
  }
  __syncthreads();
  if (r < r1) {
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    y[r] = s;
  }
}

// V2: 16-byte loads of vals (double2) and 8-byte loads of cols (int2): each
// thread owns 2 consecutive nonzeros per step; tile base aligned down to even
__global__ __launch_bounds__(KB) void v2(int n, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[TILE + 2];
  const int r0 = rb[blockIdx.x], r1 = rb[blockIdx.x + 1];
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  const int ka = k0 & ~1;                 // aligned base (may include 1 foreign nnz)
  const int r = r0 + threadIdx.x;
  int a = 0, b = 0;
  if (r < r1) { a = rowptr[r] - ka; b = rowptr[r + 1] - ka; }
  const double2* __restrict__ v2p = reinterpret_cast<const double2*>(vals + ka);
  const int2* __restrict__ c2p = reinterpret_cast<const int2*>(cols + ka);
  const int npair = (k1 - ka + 1) >> 1;   // pairs to process (tail may overrun by 1)
  double2 v[4]; int2 c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = threadIdx.x + j * KB;
    const bool ok = p < npair;
    v[j] = ok ? v2p[p] : make_double2(0.0, 0.0);
    c[j] = ok ? c2p[p] : make_int2(0, 0);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = threadIdx.x + j * KB;
    if (p < npair) {
      // the possible overrun element (k1) has a valid col index (< n) unless
      // it is past nnz: guarded by the host (padding), products unused
      const double x0 = x[c[j].x];
      const double x1 = x[c[j].y];
      prod[2 * p] = v[j].x * x0;
      prod[2 * p + 1] = v[j].y * x1;
    }
  }
  __syncthreads();
  if (r < r1) {
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    y[r] = s;
  }
}

// V3: sub-wave per row (8 lanes), no LDS
__global__ __launch_bounds__(KB) void v3(int n, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  const int t = blockIdx.x * KB + threadIdx.x;
  const int r = t >> 3;
  const int l = t & 7;
  double s = 0.0;
  if (r < n) {
    const int a = rowptr[r], b = rowptr[r + 1];
    for (int k = a + l; k < b; k += 8) s += vals[k] * x[cols[k]];
  }
  s += __shfl_down(s, 4, 8);
  s += __shfl_down(s, 2, 8);
  s += __shfl_down(s, 1, 8);
  if (r < n && l == 0) y[r] = s;
}

// V4: 512 threads, 4096-nnz tile, 512 rows per block, loads before use
__global__ __launch_bounds__(512) void v4(int n, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[4096];
  const int r0 = rb[blockIdx.x], r1 = rb[blockIdx.x + 1];
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  const int r = r0 + threadIdx.x;
  int a = 0, b = 0;
  if (r < r1) { a = rowptr[r] - k0; b = rowptr[r + 1] - k0; }
  double v[8]; int c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + threadIdx.x + j * 512;
    const bool ok = k < k1;
    v[j] = ok ? vals[k] : 0.0;
    c[j] = ok ? cols[k] : 0;
  }
  double xv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xv[j] = x[c[j]];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = threadIdx.x + j * 512;
    if (k0 + k < k1) prod[k] = v[j] * xv[j];
  }
  __syncthreads();
  if (r < r1) {
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    y[r] = s;
  }
}

// V5: like V1 but nontemporal loads for the streamed matrix (vals/cols)
__global__ __launch_bounds__(KB) void v5(int n, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[TILE];
  const int r0 = rb[blockIdx.x], r1 = rb[blockIdx.x + 1];
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  const int r = r0 + threadIdx.x;
  int a = 0, b = 0;
  if (r < r1) { a = rowptr[r] - k0; b = rowptr[r + 1] - k0; }
  double v[8]; int c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + threadIdx.x + j * KB;
    const bool ok = k < k1;
    v[j] = ok ? __builtin_nontemporal_load(vals + k) : 0.0;
    c[j] = ok ? __builtin_nontemporal_load(cols + k) : 0;
  }
  double xv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xv[j] = x[c[j]];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = threadIdx.x + j * KB;
    if (k0 + k < k1) prod[k] = v[j] * xv[j];
  }
  __syncthreads();
  if (r < r1) {
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    __builtin_nontemporal_store(s, y + r);
  }
}


// V6/V7/V8: templated V2 -- 16-B value loads; NT threads, PAIRS pairs per thread
// (tile = 2*NT*PAIRS nonzeros, host limits blocks to tile-2 nnz)
// XCD-aware tile mapping of the product (flow_amd/csrc/common.h, xcd_tile)
template <int RUN>
__device__ __forceinline__ int xcd_map(int b, int nwg) {
  constexpr int G = 8 * RUN;
  const int full = (nwg / G) * G;
  if (b < full) {
    const int g = b / G, i = b - g * G;
    return g * G + (i & 7) * RUN + (i >> 3);
  }
  const int nt = nwg - full, i = b - full;
  const int q = nt >> 3, r = nt & 7, xcd = i & 7;
  return full + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (i >> 3);
}

template <int NT, int PAIRS, bool PERSIST, int RUN = 0>
__global__ __launch_bounds__(NT) void v2t(int nblocks, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const int* __restrict__ rb, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double prod[2 * NT * PAIRS];
  for (int blk0 = blockIdx.x; blk0 < nblocks; blk0 += gridDim.x) {
    const int blk = RUN > 0 ? xcd_map<(RUN > 0 ? RUN : 1)>(blk0, nblocks) : blk0;
    const int r0 = rb[blk], r1 = rb[blk + 1];
    const int k0 = rowptr[r0], k1 = rowptr[r1];
    const int ka = k0 & ~1;
    const int r = r0 + threadIdx.x;
    int a = 0, b = 0;
    if (r < r1) { a = rowptr[r] - ka; b = rowptr[r + 1] - ka; }
    const double2* __restrict__ v2p = reinterpret_cast<const double2*>(vals + ka);
    const int2* __restrict__ c2p = reinterpret_cast<const int2*>(cols + ka);
    const int npair = (k1 - ka + 1) >> 1;
    double2 v[PAIRS]; int2 c[PAIRS];
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const int p = threadIdx.x + j * NT;
      const bool ok = p < npair;
      v[j] = ok ? v2p[p] : make_double2(0.0, 0.0);
      c[j] = ok ? c2p[p] : make_int2(0, 0);
    }
    double x0[PAIRS], x1[PAIRS];
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) { x0[j] = x[c[j].x]; x1[j] = x[c[j].y]; }
    if (PERSIST) __syncthreads();     // previous tile fully consumed
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const int p = threadIdx.x + j * NT;
      if (p < npair) {
        prod[2 * p] = v[j].x * x0[j];
        prod[2 * p + 1] = v[j].y * x1[j];
      }
    }
    __syncthreads();
    if (r < r1) {
      double s = 0.0;
      for (int k = a; k < b; ++k) s += prod[k];
      y[r] = s;
    }
    if (!PERSIST) break;
  }
}

extern "C" int spmv_variant(int variant, int n, int nblocks, const int* rowptr,
                            const int* cols, const double* vals, const int* rb,
                            const double* x, double* y, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (variant) {
    case 0: hipLaunchKernelGGL(v0, dim3(nblocks), dim3(KB), 0, st, n, rowptr, cols, vals, rb, x, y); break;
    case 1: hipLaunchKernelGGL(v1, dim3(nblocks), dim3(KB), 0, st, n, rowptr, cols, vals, rb, x, y); break;
    case 2: hipLaunchKernelGGL(v2, dim3(nblocks), dim3(KB), 0, st, n, rowptr, cols, vals, rb, x, y); break;
    case 3: hipLaunchKernelGGL(v3, dim3((n * 8 + KB - 1) / KB), dim3(KB), 0, st, n, rowptr, cols, vals, rb, x, y); break;
    case 4: hipLaunchKernelGGL(v4, dim3(nblocks), dim3(512), 0, st, n, rowptr, cols, vals, rb, x, y); break;
    case 5: hipLaunchKernelGGL(v5, dim3(nblocks), dim3(KB), 0, st, n, rowptr, cols, vals, rb, x, y); break;
    case 6: hipLaunchKernelGGL((v2t<512, 4, false>), dim3(nblocks), dim3(512), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 7: hipLaunchKernelGGL((v2t<256, 4, true>), dim3(nblocks < 2048 ? nblocks : 2048), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 8: hipLaunchKernelGGL((v2t<256, 4, false>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 9: hipLaunchKernelGGL((v2t<256, 8, false>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 10: hipLaunchKernelGGL((v2t<128, 4, false>), dim3(nblocks), dim3(128), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 11: hipLaunchKernelGGL((v2t<256, 4, false, 32>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 12: hipLaunchKernelGGL((v2t<512, 4, false, 16>), dim3(nblocks), dim3(512), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 13: hipLaunchKernelGGL((v2t<256, 8, false, 16>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 14: hipLaunchKernelGGL((v2t<128, 4, false, 64>), dim3(nblocks), dim3(128), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 15: hipLaunchKernelGGL((v2t<256, 2, false, 64>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 16: hipLaunchKernelGGL((v2t<256, 4, false, 64>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 17: hipLaunchKernelGGL((v2t<256, 4, false, 128>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 18: hipLaunchKernelGGL((v2t<256, 3, false, 64>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 19: hipLaunchKernelGGL((v2t<512, 2, false, 32>), dim3(nblocks), dim3(512), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 20: hipLaunchKernelGGL((v2t<256, 2, false, 32>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 21: hipLaunchKernelGGL((v2t<128, 2, false, 128>), dim3(nblocks), dim3(128), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 22: hipLaunchKernelGGL((v2t<256, 1, false, 128>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    case 23: hipLaunchKernelGGL((v2t<256, 2, false, 128>), dim3(nblocks), dim3(256), 0, st, nblocks, rowptr, cols, vals, rb, x, y); break;
    default: return 2;
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
