// fp16 CSR-stream tiles with ONE value per nonzero (a plane that serves one
// component -- float vectors -- or both -- float2 vectors: one 8-byte gather
// per nonzero), shared by the mass solver (mass_kernels.hip) and the one-plane
// levels of the p-multigrid (pmg_kernels.hip).  gfx950 only.
#pragma once
#include "common.h"

#include <hip/hip_fp16.h>

namespace flow {

constexpr int kMassQuads = 2;                       // quads of nonzeros per lane
constexpr int kMassTile = kBlock * 4 * kMassQuads;  // LDS products per workgroup
static_assert(FLOW_PMG_NNZ_PER_BLOCK == kMassTile - 4,
              "tile minus alignment slack (base aligned down to a multiple of 4)");

struct Half4 {              // four nonzeros, 8 bytes
  __half v[4];
};
static_assert(sizeof(Half4) == 8, "packed quad");

// fp32 vectors: one float per dof (scalar systems) or the two components
// interleaved (float2: one 8-byte gather per nonzero serves both)
__device__ __forceinline__ float vscale(float w, float g) { return w * g; }
__device__ __forceinline__ float2 vscale(float w, float2 g) {
  return make_float2(w * g.x, w * g.y);
}
__device__ __forceinline__ void vadd(float& s, float p) { s += p; }
__device__ __forceinline__ void vadd(float2& s, float2 p) {
  s.x += p.x;
  s.y += p.y;
}
__device__ __forceinline__ void vzero(float& s) { s = 0.f; }
__device__ __forceinline__ void vzero(float2& s) { s = make_float2(0.f, 0.f); }

// One tile of the fp16 stream -- rows [r0, r1) of workgroup blockIdx.x (at most
// kBlock rows, kMassTile - 4 nonzeros): every lane loads kMassQuads quads of
// values (8 B) and of column indices (16 B) from a base aligned down to a
// multiple of four nonzeros, all of them and all gathers behind them in flight
// before the first use; products through LDS, lane i sums row r0 + i.
// Window-safe like stream_tile_row_sum: g is only dereferenced for the tile's
// own nonzeros (slack and idle lanes gather the tile's first column).
// early(row, has_row) is called as soon as the lane knows its row: the caller
// issues its epilogue's loads there, so that they travel with the tile's own
// loads instead of adding a link to the chain of dependent loads.
// (NoEarly: common.h)
template <class V, class Early = NoEarly>
__device__ __forceinline__ V mass_tile_row_sum(
    const int* __restrict__ rowptr, const int* __restrict__ cols,
    const __half* __restrict__ vals, const int* __restrict__ rowblocks,
    const V* __restrict__ g, V* __restrict__ prod, int& r, int& r1,
    Early early = Early()) {
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int r0 = rowblocks[tile];
  r1 = rowblocks[tile + 1];
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
  const int lo = k0 - ka, hi = k1 - ka;          // hi <= kMassTile - 1
  const Half4* __restrict__ vq = reinterpret_cast<const Half4*>(vals + ka);
  const int4* __restrict__ cq = reinterpret_cast<const int4*>(cols + ka);
  Half4 v[kMassQuads];
  int4 c[kMassQuads];
#pragma unroll
  for (int q = 0; q < kMassQuads; ++q) {
    const int p = threadIdx.x + q * kBlock;
    c[q] = make_int4(0, 0, 0, 0);
    if (4 * p < hi) {
      v[q] = vq[p];
      c[q] = cq[p];
    }
  }
  if (k0 < k1) {                                   // (block-uniform)
    const int safe = cols[k0];
    V gg[kMassQuads][4];
#pragma unroll
    for (int q = 0; q < kMassQuads; ++q) {          // all gathers in flight
      const int e0 = 4 * (threadIdx.x + q * kBlock);
      const int cc[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j;
        gg[q][j] = g[(e >= lo && e < hi) ? cc[j] : safe];
      }
    }
#pragma unroll
    for (int q = 0; q < kMassQuads; ++q) {
      const int e0 = 4 * (threadIdx.x + q * kBlock);
      if (e0 < hi) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          prod[e0 + j] = vscale(__half2float(v[q].v[j]), gg[q][j]);
      }
    }
  }
  __syncthreads();
  V s;
  vzero(s);
  for (int k = a; k < b; ++k) vadd(s, prod[k]);
  return s;
}

// The same tile from the PACKED stream: one 32-bit word per nonzero -- the fp16
// value in the low half, the column as a 16-bit offset from the tile's lowest
// column (cbase[tile]) in the high half -- so a quad of nonzeros is ONE 16-byte
// load (4 B per nonzero instead of 6, half the stream-load instructions).
// Possible whenever a tile's columns span < 65536 (any banded numbering; the
// host checks and falls back to the plain stream otherwise).
template <class V, class Early = NoEarly>
__device__ __forceinline__ V mass_tile_row_sum_packed(
    const int* __restrict__ rowptr, const unsigned* __restrict__ packed,
    const int* __restrict__ cbase, const int* __restrict__ rowblocks,
    const V* __restrict__ g, V* __restrict__ prod, int& r, int& r1,
    Early early = Early()) {
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int r0 = rowblocks[tile];
  r1 = rowblocks[tile + 1];
  const int base = cbase[tile];
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
  const int lo = k0 - ka, hi = k1 - ka;
  const uint4* __restrict__ pq = reinterpret_cast<const uint4*>(packed + ka);
  uint4 w[kMassQuads];
#pragma unroll
  for (int q = 0; q < kMassQuads; ++q) {
    const int p = threadIdx.x + q * kBlock;
    w[q] = make_uint4(0u, 0u, 0u, 0u);
    if (4 * p < hi) w[q] = pq[p];
  }
  if (k0 < k1) {                                   // (block-uniform)
    const int safe = base + static_cast<int>(packed[k0] >> 16);
    V gg[kMassQuads][4];
#pragma unroll
    for (int q = 0; q < kMassQuads; ++q) {          // all gathers in flight
      const int e0 = 4 * (threadIdx.x + q * kBlock);
      const unsigned ww[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j;
        gg[q][j] = g[(e >= lo && e < hi) ? base + static_cast<int>(ww[j] >> 16)
                                         : safe];
      }
    }
#pragma unroll
    for (int q = 0; q < kMassQuads; ++q) {
      const int e0 = 4 * (threadIdx.x + q * kBlock);
      if (e0 < hi) {
        const unsigned ww[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          prod[e0 + j] = vscale(
              __half2float(__ushort_as_half(static_cast<unsigned short>(
                  ww[j] & 0xffffu))),
              gg[q][j]);
      }
    }
  }
  __syncthreads();
  V s;
  vzero(s);
  for (int k = a; k < b; ++k) vadd(s, prod[k]);
  return s;
}

}  // namespace flow
