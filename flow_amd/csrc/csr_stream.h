// CSR-stream tiles of the fp64 SpMV kernels (la_kernels.hip) and of the kernels
// that carry a product with an epilogue of their own (mass_kernels.hip): a
// 256-thread workgroup owns <= 256 consecutive rows holding <= kTile - 2
// nonzeros, streams their values / column indices fully coalesced (every lane
// busy, whatever the row lengths), parks the products in LDS and then sums one
// row per lane.  gfx950 only.
#pragma once
#include "common.h"

namespace flow {

// Tiles of the kernels that park ONE product per nonzero in LDS (operator kinds
// 0 and 1, the multigrid level kernels): kPairs index pairs per lane; tiles of
// the kernels that park two (kinds 2 and 4): kPairs2.
#ifndef FLOW_SPMV_PAIRS
#define FLOW_SPMV_PAIRS 2
#endif
constexpr int kPairs = FLOW_SPMV_PAIRS;         // nonzero pairs per lane
constexpr int kTile = 2 * kBlock * kPairs;      // LDS products per workgroup
constexpr int kPairs2 = 2;
constexpr int kTile2 = 2 * kBlock * kPairs2;
static_assert(FLOW_SPMV_ROWS_PER_BLOCK == kBlock, "one lane per row");
static_assert(FLOW_SPMV_NNZ_PER_BLOCK == kTile2 - 2, "tile minus alignment slack");

// The rows [r0, r1) of one workgroup (<= kBlock rows, <= kTile - 2 nonzeros): the
// products go through LDS (prod, kTile doubles), then lane i sums row r0 + i.
// Returns that row's sum (0 for a lane without a row).
template <class Early = NoEarly>
__device__ __forceinline__ double stream_rows_sum(
    int r0, int r1, const int* __restrict__ rowptr, const int* __restrict__ cols,
    const double* __restrict__ vals, const double* __restrict__ x,
    double* __restrict__ prod, Early early = Early()) {
  const int k0 = rowptr[r0];
  const int k1 = rowptr[r1];
  // 16-byte value loads / 8-byte index loads: every lane owns PAIRS pairs of
  // consecutive nonzeros; the tile base is aligned down to an even index (value
  // planes start 16-B aligned and the host caps a block at kTile-2 nonzeros).
  const int ka = k0 & ~1;
  const int r = r0 + threadIdx.x;
  early(r, r < r1);
  int a = 0, b = 0;
  if (r < r1) {
    a = rowptr[r] - ka;
    b = rowptr[r + 1] - ka;
  }
  const double2* __restrict__ v2p = reinterpret_cast<const double2*>(vals + ka);
  const int2* __restrict__ c2p = reinterpret_cast<const int2*>(cols + ka);
  const int npair = (k1 - ka + 1) >> 1;   // a trailing odd element reads one
                                          // entry of the next tile (unused)
  double2 v[kPairs];
  int2 c[kPairs];
#pragma unroll
  for (int j = 0; j < kPairs; ++j) {
    const int p = threadIdx.x + j * kBlock;
    const bool ok = p < npair;
    v[j] = ok ? v2p[p] : make_double2(0.0, 0.0);
    c[j] = ok ? c2p[p] : make_int2(0, 0);
  }
  // x is only gathered for the tile's OWN nonzeros [k0, k1): the alignment
  // slack before k0, the odd element behind k1 (a column of another row, or
  // the padding 0 behind the last nonzero) and the idle lanes must not be
  // dereferenced -- x may be a window of a larger vector (row-sharded solves
  // pass x shifted to global row numbering: x[0] is then far outside it)
  // Those entries gather the tile's first column instead (an index select,
  // the loads themselves stay unconditional and all in flight).
  const int lo = k0 - ka, hi = k1 - ka;
  const int safe = cols[k0 < k1 ? k0 : (k0 > 0 ? k0 - 1 : 0)];
  double x0[kPairs], x1[kPairs];
  if (k0 < k1) {                       // (block-uniform)
#pragma unroll
    for (int j = 0; j < kPairs; ++j) {   // all gathers in flight before any use
      const int e = 2 * (threadIdx.x + j * kBlock);
      x0[j] = x[(e >= lo && e < hi) ? c[j].x : safe];
      x1[j] = x[(e + 1 < hi) ? c[j].y : safe];
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPairs; ++j) x0[j] = x1[j] = 0.0;
  }
#pragma unroll
  for (int j = 0; j < kPairs; ++j) {
    const int p = threadIdx.x + j * kBlock;
    if (p < npair) {
      prod[2 * p] = v[j].x * x0[j];
      prod[2 * p + 1] = v[j].y * x1[j];
    }
  }
  __syncthreads();
  double s = 0.0;
  for (int k = a; k < b; ++k) s += prod[k];
  return s;
}

// One tile of the CSR stream -- the rows [r0, r1) of workgroup blockIdx.x (XCD-
// aware tile mapping, common.h).  Returns the lane's row sum; r / r1 tell the
// caller whether the lane has a row.
template <class Early = NoEarly>
__device__ __forceinline__ double stream_tile_row_sum(
    const int* __restrict__ rowptr, const int* __restrict__ cols,
    const double* __restrict__ vals, const int* __restrict__ rowblocks,
    const double* __restrict__ x, double* __restrict__ prod, int& r, int& r1,
    Early early = Early()) {
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int r0 = rowblocks[tile];
  r1 = rowblocks[tile + 1];
  r = r0 + threadIdx.x;
  return stream_rows_sum(r0, r1, rowptr, cols, vals, x, prod, early);
}

// The same tile with ONE value plane applied to both components of a
// component-blocked vector (component stride xs): the two components' products
// side by side as double2 in LDS (prod: kTile2 double2, 16 KB -- with the
// 1022-nonzero tile that still leaves the 8 workgroups per CU the wave limit
// allows), both row sums out of ONE pass over the segment.  A block of a square
// operator is never empty.  Returns (sum of component 0, sum of component 1).
template <class Early = NoEarly>
__device__ __forceinline__ double2 stream_tile_pair_row_sum(
    const int* __restrict__ rowptr, const int* __restrict__ cols,
    const double* __restrict__ vals, const int* __restrict__ rowblocks,
    const double* __restrict__ x, int xs, double2* __restrict__ prod, int& r,
    int& r1, Early early = Early()) {
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int r0 = rowblocks[tile];
  r1 = rowblocks[tile + 1];
  const int k0 = rowptr[r0];
  const int k1 = rowptr[r1];
  const int ka = k0 & ~1;
  r = r0 + threadIdx.x;
  early(r, r < r1);
  int a = 0, b = 0;
  if (r < r1) {
    a = rowptr[r] - ka;
    b = rowptr[r + 1] - ka;
  }
  const double2* __restrict__ v2p = reinterpret_cast<const double2*>(vals + ka);
  const int2* __restrict__ c2p = reinterpret_cast<const int2*>(cols + ka);
  const int npair = (k1 - ka + 1) >> 1;
  double2 v[kPairs2];
  int2 c[kPairs2];
#pragma unroll
  for (int j = 0; j < kPairs2; ++j) {
    const int p = threadIdx.x + j * kBlock;
    const bool ok = p < npair;
    v[j] = ok ? v2p[p] : make_double2(0.0, 0.0);
    c[j] = ok ? c2p[p] : make_int2(0, 0);
  }
  // (only the tile's own columns are dereferenced: see stream_tile_row_sum)
  const int lo = k0 - ka, hi = k1 - ka;
  const int safe = cols[k0];
  double xa[kPairs2], xb[kPairs2], ua[kPairs2], ub[kPairs2];
#pragma unroll
  for (int j = 0; j < kPairs2; ++j) {   // all gathers in flight before any use
    const int e = 2 * (threadIdx.x + j * kBlock);
    const int cx = (e >= lo && e < hi) ? c[j].x : safe;
    const int cy = (e + 1 < hi) ? c[j].y : safe;
    xa[j] = x[cx];
    xb[j] = x[cy];
    ua[j] = x[xs + cx];
    ub[j] = x[xs + cy];
  }
#pragma unroll
  for (int j = 0; j < kPairs2; ++j) {
    const int p = threadIdx.x + j * kBlock;
    if (p < npair) {
      prod[2 * p] = make_double2(v[j].x * xa[j], v[j].x * ua[j]);
      prod[2 * p + 1] = make_double2(v[j].y * xb[j], v[j].y * ub[j]);
    }
  }
  __syncthreads();
  double s0 = 0.0, s1 = 0.0;
  for (int k = a; k < b; ++k) {
    const double2 q = prod[k];
    s0 += q.x;
    s1 += q.y;
  }
  return make_double2(s0, s1);
}

}  // namespace flow
