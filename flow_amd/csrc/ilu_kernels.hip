// K11: multicolour ILU(0) on gfx950.  The plan (colouring, colour-major permuted
// CSR, strictly-lower / strictly-upper sweep streams in sliced-ELL form, map
// back to the operator's value plane) is built once per pattern on the host:
// flow_amd/fem/ilu.py.
//
// Stands in for the sparse LU behind the reference's Newton and heat solves
// (flow/navier_stokes/pressure_correction.py:224-254, flow/heat.py:117-121) as
// the preconditioner of BiCGStab.
//
//  * factorisation: one launch per colour, a lane per row (IKJ with a sorted
//    merge); runs once per (re)factorisation;
//  * sweeps: one launch per colour and sweep.  A wavefront owns a slice of 64
//    rows stored column-major (sliced ELL, rows sorted by length inside the
//    colour so the padding stays small): lane i reads entry k of its own row at
//    off + 64 k + i -- every load coalesced -- and keeps the row sum in
//    registers.  No LDS, no barrier, four entries per lane in flight: the
//    launches are short (a tenth of the triangle each), so what counts is the
//    length of the dependent-load chain of a wavefront, not only the bytes.
//    Both diagonal blocks of a two-field operator go through the same launch
//    (blockIdx.y).
// HBM-bound: 12 B per factor entry, block and application (packed streams,
// below: 12 B for two blocks).
#include "common.h"
#include <type_traits>

namespace flow {

constexpr int kSlice = FLOW_ILU_SLICE;
static_assert(kSlice == 64, "one wavefront per slice");

__global__ void ilu_copy_kernel(int nnz, const int* __restrict__ src_pos,
                                const double* __restrict__ avals,
                                double* __restrict__ lu) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += gridDim.x * blockDim.x)
    lu[k] = avals[src_pos[k]];
}

// IKJ ILU(0) of the rows [a, b) of one colour.  Every row k < i referenced here
// has a lower colour and is final.
// NB blocks (the two velocity blocks share the pattern) are factored by the
// same lane: the index traffic of the sorted merge is paid once.
template <int NB>
__global__ void ilu_factor_colour_kernel(int a, int b,
                                         const int* __restrict__ rowptr,
                                         const int* __restrict__ cols,
                                         const int* __restrict__ diag,
                                         double* __restrict__ lu,
                                         size_t lu_size) {
  const int i = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const int p0 = rowptr[i], pd = diag[i], p1 = rowptr[i + 1];
  double d_orig[NB];
#pragma unroll
  for (int m = 0; m < NB; ++m) d_orig[m] = lu[m * lu_size + pd];
  for (int p = p0; p < pd; ++p) {
    const int k = cols[p];
    const int dk = diag[k];
    double lik[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      lik[m] = lu[m * lu_size + p] / lu[m * lu_size + dk];
      lu[m * lu_size + p] = lik[m];
    }
    // a_ij -= l_ik * u_kj for j > k present in both rows (both lists ascend)
    int q = dk + 1;
    const int qe = rowptr[k + 1];
    for (int t = p + 1; t < p1 && q < qe; ++t) {
      const int j = cols[t];
      while (q < qe && cols[q] < j) ++q;
      if (q < qe && cols[q] == j) {
#pragma unroll
        for (int m = 0; m < NB; ++m)
          lu[m * lu_size + t] -= lik[m] * lu[m * lu_size + q];
      }
    }
  }
  // pivot guard: keep the factor usable if a pivot collapses
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    const double d = lu[m * lu_size + pd];
    if (!(fabs(d) > 1.0e-12 * fabs(d_orig[m])))
      lu[m * lu_size + pd] = d_orig[m] != 0.0 ? d_orig[m] : 1.0;
  }
}

// Same factorisation with 8 lanes per row: every lane keeps up to kSlots of the
// row's entries (column + NB values) in registers, the L entries are
// eliminated one after the other (wave-uniform within the 8-lane group), the
// multiplier is broadcast with a shuffle, and each lane locates its own
// columns in row k's U part by bisection -- ~4 dependent loads per L entry
// instead of a ~13-step sequential merge, and 8x more loads in flight.
constexpr int kSub = 8;
constexpr int kSlots = 6;     // rows up to 48 entries; longer rows: scalar kernel

template <int NB>
__global__ __launch_bounds__(kBlock) void ilu_factor_colour_sub_kernel(
    int a, int b, const int* __restrict__ rowptr, const int* __restrict__ cols,
    const int* __restrict__ diag, double* __restrict__ lu, size_t lu_size) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = a + gid / kSub;
  const int sub = threadIdx.x % kSub;
  const bool live = i < b;
  const int ii = live ? i : b - 1;
  const int p0 = rowptr[ii], pd = diag[ii], p1 = rowptr[ii + 1];
  int col[kSlots];
  double val[kSlots][NB];
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    const int t = p0 + sub + s * kSub;
    col[s] = t < p1 ? cols[t] : -1;
#pragma unroll
    for (int m = 0; m < NB; ++m) val[s][m] = t < p1 ? lu[m * lu_size + t] : 0.0;
  }
  double d_orig[NB];
#pragma unroll
  for (int m = 0; m < NB; ++m) d_orig[m] = lu[m * lu_size + pd];

  for (int p = p0; p < pd; ++p) {            // uniform over the 8 lanes of a row
    const int k = cols[p];
    const int dk = diag[k];
    const int qb = dk + 1, qe = rowptr[k + 1];
    const int owner = (p - p0) % kSub, slot = (p - p0) / kSub;
    double lik[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      double mine = 0.0;
#pragma unroll
      for (int s = 0; s < kSlots; ++s)
        if (s == slot) mine = val[s][m];
      mine /= lu[m * lu_size + dk];
      lik[m] = __shfl(mine, owner, kSub);
    }
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      if (s == slot && sub == owner) {
#pragma unroll
        for (int m = 0; m < NB; ++m) val[s][m] = lik[m];
      }
      const int j = col[s];
      if (j > k) {
        // bisection for j in cols[qb, qe)
        int lo = qb, hi = qe;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (cols[mid] < j) lo = mid + 1; else hi = mid;
        }
        if (lo < qe && cols[lo] == j) {
#pragma unroll
          for (int m = 0; m < NB; ++m) val[s][m] -= lik[m] * lu[m * lu_size + lo];
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    const int t = p0 + sub + s * kSub;
    if (t < p1) {
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        double v = val[s][m];
        // pivot guard: keep the factor usable if a pivot collapses
        if (t == pd && !(fabs(v) > 1.0e-12 * fabs(d_orig[m])))
          v = d_orig[m] != 0.0 ? d_orig[m] : 1.0;
        lu[m * lu_size + t] = v;
      }
    }
  }
}

// split the combined factor into the two streams and the inverse pivots
__global__ void ilu_split_kernel(int n, int nnz_l, int nnz_u,
                                 const int* __restrict__ l_pos,
                                 const int* __restrict__ u_pos,
                                 const int* __restrict__ diag,
                                 const double* __restrict__ lu,
                                 double* __restrict__ lvals,
                                 double* __restrict__ uvals,
                                 double* __restrict__ dinv) {
  const int total = nnz_l + nnz_u + n;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total;
       k += gridDim.x * blockDim.x) {
    if (k < nnz_l) {
      const int p = l_pos[k];
      lvals[k] = p >= 0 ? lu[p] : 0.0;
    } else if (k < nnz_l + nnz_u) {
      const int p = u_pos[k - nnz_l];
      uvals[k - nnz_l] = p >= 0 ? lu[p] : 0.0;
    } else {
      dinv[k - nnz_l - nnz_u] = 1.0 / lu[diag[k - nnz_l - nnz_u]];
    }
  }
}

// The sweeps run in the permuted (colour-major) numbering on ONE buffer y:
//   ilu_permute_kernel   y[new(o)] = r[o]     (both loops run over the ORIGINAL
//   ilu_unpermute_kernel z[o] = y[new(o)]      index o: the vector in original
// numbering is touched coalesced, the scattered side lands in lines that the
// neighbouring lanes complete within the same few wavefronts.)  Gathering
// r[old(i)] inside the sweeps instead would drag the whole of r through every
// one of the ~10 colour launches: a colour owns 1 in 10 entries of each line.
__global__ void ilu_permute_kernel(int n, int nblocks,
                                   const int* __restrict__ new_of_old,
                                   const double* __restrict__ r,
                                   double* __restrict__ y,
                                   const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n;
       o += gridDim.x * blockDim.x) {
    const int i = new_of_old[o];
    y[i] = r[o];
    if (nblocks > 1) y[n + i] = r[n + o];
  }
}

__global__ void ilu_unpermute_kernel(int n, int nblocks,
                                     const int* __restrict__ new_of_old,
                                     const double* __restrict__ y,
                                     double* __restrict__ z,
                                     const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n;
       o += gridDim.x * blockDim.x) {
    const int i = new_of_old[o];
    z[o] = y[i];
    if (nblocks > 1) z[n + o] = y[n + i];
  }
}

// One colour of one sweep; a wavefront per slice (slice_off / slice_row are
// pre-offset to the colour's first slice), rows of the colour end at row_end.
//   FWD:  y_i = y_i - sum_k L_ik y_k
//   BWD:  y_i = (y_i - sum_j U_ij y_j) / U_ii
// blockIdx.y selects the diagonal block: values at vals + blk*lu_size, y at
// + blk*n.
template <bool BWD>
__global__ __launch_bounds__(kBlock) void ilu_sweep_kernel(
    int n, size_t lu_size, int nslices, int row_end,
    const int* __restrict__ slice_off, const int* __restrict__ slice_row,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const double* __restrict__ dinv, double* __restrict__ y,
    const double* __restrict__ stop) {
  // the done flag is only looked at before the store: a sweep launch is short
  // (the latency of three dependent loads), an early exit on the flag would
  // put a fourth in front of them
  const double halt = stop ? load_scalar(stop) : 0.0;
  // (XCD-aware: neighbouring slices gather from neighbouring windows of y)
  const int sl = xcd_tile(blockIdx.x, gridDim.x) * (kBlock / kSlice) +
                 (threadIdx.x >> 6);
  if (sl >= nslices) return;
  const int lane = threadIdx.x & 63;
  const size_t blk = blockIdx.y;
  vals += blk * lu_size;
  if (BWD) dinv += blk * lu_size;
  y += blk * n;
  const int off = slice_off[sl];
  const int width = (slice_off[sl + 1] - off) >> 6;
  const int row = slice_row[sl] + lane;
  const bool live = row < row_end;
  // what the row's final update needs is requested up front
  double rhs = 0.0, di = 1.0;
  if (live) {
    rhs = y[row];
    if (BWD) di = dinv[row];
  }
  const double* __restrict__ v = vals + off + lane;
  const int* __restrict__ c = cols + off + lane;
  // batches of kBatch entries per lane: all index/value loads of a batch are
  // issued before the first gather, all gathers before the first use (width is
  // wavefront-uniform, so the guards are scalar branches)
  constexpr int kBatch = 12;
  double s = 0.0;
  for (int k0 = 0; k0 < width; k0 += kBatch) {
    int cc[kBatch];
    double vv[kBatch], yy[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const bool ok = k0 + j < width;
      cc[j] = ok ? c[(k0 + j) * kSlice] : 0;
      vv[j] = ok ? v[(k0 + j) * kSlice] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j) yy[j] = (k0 + j < width) ? y[cc[j]] : 0.0;
#pragma unroll
    for (int j = 0; j < kBatch; ++j) s += vv[j] * yy[j];
  }
  if (live && halt == 0.0) y[row] = (rhs - s) * di;
}

// ---- packed sweep streams ---------------------------------------------------
// The sweeps are bound by the bytes of the factor: 12 B per entry, block and
// application above.  A preconditioner does not need its entries to 16 digits:
// with flow_ilu.packed the two streams are kept a second time as fp32, the
// NB blocks of an entry interleaved (one index + one NB*4-byte value load per
// entry, 12 B for both velocity blocks together instead of 24), and the sweep
// vector is interleaved the same way (one NB*8-byte gather per entry).  All
// arithmetic stays fp64: the sweeps apply the exact inverse of the ROUNDED
// factors -- a fixed linear operator, as good a preconditioner as the unrounded
// one (ILU(0) itself is off by far more than 6e-8), and the Krylov method
// around it converges to the same tolerance.
// F32 (flow_ilu.single_vector): the sweep vector is kept in fp32 as well -- 8
// instead of 16 B per gather, 20 instead of 28 B per factor entry all told; the
// row sums still accumulate in fp64, every finished row is rounded once.  The
// application is then no longer exactly linear in its input (rounding), which
// only a FLEXIBLE Krylov method may use: the GMRES of la_kernels.hip is one --
// it keeps Z_j = M^-1 V_j, multiplies THAT by A and updates x with it, so the
// Arnoldi relation A Z = V H holds whatever produced Z_j.
template <int NB>
struct PackT;
template <>
struct PackT<1> {
  using val = float;
  using vec = double;
  using vec32 = float;
};
template <>
struct PackT<2> {
  using val = float2;
  using vec = double2;
  using vec32 = float2;
};
template <int NB, bool F32>
struct SweepVec {
  using type = typename PackT<NB>::vec;
};
template <int NB>
struct SweepVec<NB, true> {
  using type = typename PackT<NB>::vec32;
};
__device__ __forceinline__ double widen(double v) { return v; }
__device__ __forceinline__ double widen(float v) { return v; }
__device__ __forceinline__ double2 widen(double2 v) { return v; }
__device__ __forceinline__ double2 widen(float2 v) { return make_double2(v.x, v.y); }
__device__ __forceinline__ void narrow(double& o, double v) { o = v; }
__device__ __forceinline__ void narrow(float& o, double v) { o = static_cast<float>(v); }
__device__ __forceinline__ void narrow(double2& o, double2 v) { o = v; }
__device__ __forceinline__ void narrow(float2& o, double2 v) {
  o = make_float2(static_cast<float>(v.x), static_cast<float>(v.y));
}
__device__ __forceinline__ void fma_pack(double& s, float v, double y) { s += v * y; }
__device__ __forceinline__ void fma_pack(double2& s, float2 v, double2 y) {
  s.x += v.x * y.x;
  s.y += v.y * y.y;
}
__device__ __forceinline__ double zero_of(double) { return 0.0; }
__device__ __forceinline__ double2 zero_of(double2) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ float fzero_of(float) { return 0.f; }
__device__ __forceinline__ float2 fzero_of(float2) { return make_float2(0.f, 0.f); }

template <int NB>
__global__ void ilu_pack_kernel(int count, size_t lu_size,
                                const double* __restrict__ vals,
                                float* __restrict__ packed) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count;
       k += gridDim.x * blockDim.x) {
#pragma unroll
    for (int m = 0; m < NB; ++m)
      packed[static_cast<size_t>(k) * NB + m] = static_cast<float>(vals[m * lu_size + k]);
  }
}

template <int NB, typename T>
__global__ void ilu_permute_packed_kernel(int n, const int* __restrict__ new_of_old,
                                          const double* __restrict__ r,
                                          T* __restrict__ y,
                                          const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n;
       o += gridDim.x * blockDim.x) {
    const size_t i = new_of_old[o];
#pragma unroll
    for (int m = 0; m < NB; ++m)
      y[i * NB + m] = static_cast<T>(r[static_cast<size_t>(m) * n + o]);
  }
}

template <int NB, typename T>
__global__ void ilu_unpermute_packed_kernel(int n, const int* __restrict__ new_of_old,
                                            const T* __restrict__ y,
                                            double* __restrict__ z,
                                            const double* __restrict__ stop) {
  if (stopped(stop)) return;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n;
       o += gridDim.x * blockDim.x) {
    const size_t i = new_of_old[o];
#pragma unroll
    for (int m = 0; m < NB; ++m) z[static_cast<size_t>(m) * n + o] = y[i * NB + m];
  }
}

// as ilu_sweep_kernel, all NB blocks in one lane
template <bool BWD, int NB, bool F32>
__global__ __launch_bounds__(kBlock) void ilu_sweep_packed_kernel(
    size_t lu_size, int nslices, int row_end, const int* __restrict__ slice_off,
    const int* __restrict__ slice_row, const int* __restrict__ cols,
    const typename PackT<NB>::val* __restrict__ vals,
    const double* __restrict__ dinv,
    typename SweepVec<NB, F32>::type* __restrict__ y,
    const double* __restrict__ stop) {
  using V = typename PackT<NB>::val;
  using Y = typename PackT<NB>::vec;            // arithmetic: fp64
  const double halt = stop ? load_scalar(stop) : 0.0;   // (see ilu_sweep_kernel)
  const int sl = xcd_tile(blockIdx.x, gridDim.x) * (kBlock / kSlice) +
                 (threadIdx.x >> 6);
  if (sl >= nslices) return;
  const int lane = threadIdx.x & 63;
  const int off = slice_off[sl];
  const int width = (slice_off[sl + 1] - off) >> 6;
  const int row = slice_row[sl] + lane;
  const bool live = row < row_end;
  Y rhs = zero_of(Y());
  double di[NB];
#pragma unroll
  for (int m = 0; m < NB; ++m) di[m] = 1.0;
  if (live) {
    rhs = widen(y[row]);
    if (BWD) {
#pragma unroll
      for (int m = 0; m < NB; ++m) di[m] = dinv[m * lu_size + row];
    }
  }
  const V* __restrict__ v = vals + off + lane;
  const int* __restrict__ c = cols + off + lane;
  // batches of 4 (12 in the fp64 kernel): an entry costs NB*3 registers here,
  // and with 8560 slices per colour what counts is that ALL of them are
  // resident at once (8 waves per SIMD) -- measured per application at
  // 2 x 4.3 M rows: batch 12: 449 us, 8: 374, 6: 355, 4: 346 (fp64 streams: 432)
  constexpr int kBatch = 4;
  Y s = zero_of(Y());
  for (int k0 = 0; k0 < width; k0 += kBatch) {
    int cc[kBatch];
    V vv[kBatch];
    Y yy[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const bool ok = k0 + j < width;
      cc[j] = ok ? c[(k0 + j) * kSlice] : 0;
      vv[j] = ok ? v[(k0 + j) * kSlice] : fzero_of(V());
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
      yy[j] = (k0 + j < width) ? widen(y[cc[j]]) : zero_of(Y());
#pragma unroll
    for (int j = 0; j < kBatch; ++j) fma_pack(s, vv[j], yy[j]);
  }
  if (live && halt == 0.0) {
    if constexpr (NB == 1) {
      narrow(y[row], (rhs - s) * di[0]);
    } else {
      narrow(y[row], make_double2((rhs.x - s.x) * di[0], (rhs.y - s.y) * di[1]));
    }
  }
}

static int check_plan(const flow_ilu_plan* P) {
  FLOW_REQUIRE(P && P->n > 0 && P->nnz > 0 && P->ncolors > 0, "ilu plan sizes");
  FLOW_REQUIRE(P->color_ptr_host && P->slice_ptr_host, "ilu plan host arrays");
  FLOW_REQUIRE(P->rowptr && P->cols && P->diag && P->src_pos && P->old_of_new &&
                   P->new_of_old && P->slice_row && P->l_slice_off && P->l_cols && P->l_pos &&
                   P->u_slice_off && P->u_cols && P->u_pos,
               "ilu plan pointers");
  FLOW_REQUIRE(P->color_ptr_host[0] == 0 && P->color_ptr_host[P->ncolors] == P->n,
               "ilu colour ranges");
  FLOW_REQUIRE(P->slice_ptr_host[0] == 0 &&
                   P->slice_ptr_host[P->ncolors] == P->nslices,
               "ilu slice ranges");
  for (int c = 0; c < P->ncolors; ++c) {
    const int rows = P->color_ptr_host[c + 1] - P->color_ptr_host[c];
    const int sl = P->slice_ptr_host[c + 1] - P->slice_ptr_host[c];
    FLOW_REQUIRE(rows >= 0 && sl == (rows + kSlice - 1) / kSlice,
                 "ilu slices per colour");
  }
  FLOW_REQUIRE(P->nnz_l % kSlice == 0 && P->nnz_u % kSlice == 0,
               "ilu stream padding");
  FLOW_REQUIRE((P->off_l & 1) == 0 && (P->off_u & 1) == 0 && (P->off_d & 1) == 0 &&
                   (P->lu_size & 1) == 0 && P->off_l >= P->nnz &&
                   P->off_u >= P->off_l + P->nnz_l && P->off_d >= P->off_u + P->nnz_u &&
                   P->lu_size >= P->off_d + P->n,
               "ilu buffer layout");
  return FLOW_OK;
}

static int factor(const flow_ilu_plan* P, int nblocks, const double* avals0,
                  const double* avals1, double* lu, hipStream_t st) {
  FLOW_REQUIRE((reinterpret_cast<size_t>(lu) & 15) == 0, "lu must be 16-B aligned");
  const size_t lus = static_cast<size_t>(P->lu_size);
  for (int m = 0; m < nblocks; ++m)
    hipLaunchKernelGGL(ilu_copy_kernel, dim3(grid_for(P->nnz)), dim3(kBlock), 0,
                       st, P->nnz, P->src_pos, m == 0 ? avals0 : avals1,
                       lu + m * lus);
  for (int c = 1; c < P->ncolors; ++c) {   // colour 0 has no lower neighbours
    const int a = P->color_ptr_host[c], b = P->color_ptr_host[c + 1];
    if (b <= a) continue;
    if (P->max_row <= kSub * kSlots) {
      // 8 lanes per row
      const dim3 grid(((b - a) * kSub + kBlock - 1) / kBlock);
      if (nblocks == 1)
        hipLaunchKernelGGL((ilu_factor_colour_sub_kernel<1>), grid, dim3(kBlock),
                           0, st, a, b, P->rowptr, P->cols, P->diag, lu, lus);
      else
        hipLaunchKernelGGL((ilu_factor_colour_sub_kernel<2>), grid, dim3(kBlock),
                           0, st, a, b, P->rowptr, P->cols, P->diag, lu, lus);
      continue;
    }
    const dim3 grid((b - a + kBlock - 1) / kBlock);
    if (nblocks == 1)
      hipLaunchKernelGGL((ilu_factor_colour_kernel<1>), grid, dim3(kBlock), 0, st,
                         a, b, P->rowptr, P->cols, P->diag, lu, lus);
    else
      hipLaunchKernelGGL((ilu_factor_colour_kernel<2>), grid, dim3(kBlock), 0, st,
                         a, b, P->rowptr, P->cols, P->diag, lu, lus);
  }
  for (int m = 0; m < nblocks; ++m) {
    double* base = lu + m * lus;
    hipLaunchKernelGGL(ilu_split_kernel,
                       dim3(grid_for(P->nnz_l + P->nnz_u + P->n)), dim3(kBlock),
                       0, st, P->n, P->nnz_l, P->nnz_u, P->l_pos, P->u_pos,
                       P->diag, base, base + P->off_l, base + P->off_u,
                       base + P->off_d);
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

template <int NB, bool F32>
static int apply_packed(const flow_ilu* ilu, const double* in, double* out,
                        double* work, hipStream_t st, const double* stop) {
  using V = typename PackT<NB>::val;
  using Y = typename SweepVec<NB, F32>::type;
  using T = std::conditional_t<F32, float, double>;
  const flow_ilu_plan* P = ilu->plan;
  const size_t lus = static_cast<size_t>(P->lu_size);
  constexpr int per_block = kBlock / kSlice;
  const V* lvals = reinterpret_cast<const V*>(ilu->packed);
  const V* uvals = lvals + P->nnz_l;
  Y* y = reinterpret_cast<Y*>(work);
  hipLaunchKernelGGL((ilu_permute_packed_kernel<NB, T>), dim3(grid_for(P->n)),
                     dim3(kBlock), 0, st, P->n, P->new_of_old, in,
                     reinterpret_cast<T*>(work), stop);
  for (int c = 1; c < P->ncolors; ++c) {   // colour 0: y = r already
    const int s0 = P->slice_ptr_host[c];
    const int ns = P->slice_ptr_host[c + 1] - s0;
    if (ns <= 0) continue;
    hipLaunchKernelGGL((ilu_sweep_packed_kernel<false, NB, F32>),
                       dim3((ns + per_block - 1) / per_block), dim3(kBlock), 0,
                       st, lus, ns, P->color_ptr_host[c + 1], P->l_slice_off + s0,
                       P->slice_row + s0, P->l_cols, lvals,
                       static_cast<const double*>(nullptr), y, stop);
  }
  for (int c = P->ncolors - 1; c >= 0; --c) {
    const int s0 = P->slice_ptr_host[c];
    const int ns = P->slice_ptr_host[c + 1] - s0;
    if (ns <= 0) continue;
    hipLaunchKernelGGL((ilu_sweep_packed_kernel<true, NB, F32>),
                       dim3((ns + per_block - 1) / per_block), dim3(kBlock), 0,
                       st, lus, ns, P->color_ptr_host[c + 1], P->u_slice_off + s0,
                       P->slice_row + s0, P->u_cols, uvals, ilu->lu + P->off_d, y,
                       stop);
  }
  hipLaunchKernelGGL((ilu_unpermute_packed_kernel<NB, T>), dim3(grid_for(P->n)),
                     dim3(kBlock), 0, st, P->n, P->new_of_old,
                     reinterpret_cast<const T*>(work), out, stop);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

// out = blockdiag(LU_0[, LU_1])^-1 in ; work: nblocks * n doubles
int ilu_apply(const flow_ilu* ilu, const double* in, double* out, double* work,
              hipStream_t st, const double* stop) {
  // (flow_ilu.cycle: this preconditioner is the two-level cycle whose fine
  // smoother these sweeps are -- tl_kernels.hip)
  if (ilu->cycle)
    return tl_apply(static_cast<const flow_tl*>(ilu->cycle), in, out, st, stop);
  const flow_ilu_plan* P = ilu->plan;
  const size_t lus = static_cast<size_t>(P->lu_size);
  constexpr int per_block = kBlock / kSlice;
  if (ilu->packed) {
    if (ilu->single_vector) {
      if (ilu->nblocks == 1)
        return apply_packed<1, true>(ilu, in, out, work, st, stop);
      return apply_packed<2, true>(ilu, in, out, work, st, stop);
    }
    if (ilu->nblocks == 1)
      return apply_packed<1, false>(ilu, in, out, work, st, stop);
    return apply_packed<2, false>(ilu, in, out, work, st, stop);
  }
  hipLaunchKernelGGL(ilu_permute_kernel, dim3(grid_for(P->n)), dim3(kBlock), 0,
                     st, P->n, ilu->nblocks, P->new_of_old, in, work, stop);
  for (int c = 1; c < P->ncolors; ++c) {   // colour 0: y = r already
    const int s0 = P->slice_ptr_host[c];
    const int ns = P->slice_ptr_host[c + 1] - s0;
    if (ns <= 0) continue;
    hipLaunchKernelGGL((ilu_sweep_kernel<false>),
                       dim3((ns + per_block - 1) / per_block, ilu->nblocks),
                       dim3(kBlock), 0, st, P->n, lus, ns,
                       P->color_ptr_host[c + 1], P->l_slice_off + s0,
                       P->slice_row + s0, P->l_cols, ilu->lu + P->off_l,
                       static_cast<const double*>(nullptr), work, stop);
  }
  for (int c = P->ncolors - 1; c >= 0; --c) {
    const int s0 = P->slice_ptr_host[c];
    const int ns = P->slice_ptr_host[c + 1] - s0;
    if (ns <= 0) continue;
    hipLaunchKernelGGL((ilu_sweep_kernel<true>),
                       dim3((ns + per_block - 1) / per_block, ilu->nblocks),
                       dim3(kBlock), 0, st, P->n, lus, ns,
                       P->color_ptr_host[c + 1], P->u_slice_off + s0,
                       P->slice_row + s0, P->u_cols, ilu->lu + P->off_u,
                       ilu->lu + P->off_d, work, stop);
  }
  hipLaunchKernelGGL(ilu_unpermute_kernel, dim3(grid_for(P->n)), dim3(kBlock), 0,
                     st, P->n, ilu->nblocks, P->new_of_old, work, out, stop);
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

int ilu_check(const flow_ilu* ilu, int op_size) {
  FLOW_REQUIRE(ilu && ilu->plan && ilu->lu, "ilu pointers");
  int rc = check_plan(ilu->plan);
  if (rc) return rc;
  FLOW_REQUIRE(ilu->nblocks == 1 || ilu->nblocks == 2, "ilu blocks");
  FLOW_REQUIRE(ilu->nblocks * ilu->plan->n == op_size, "ilu size");
  FLOW_REQUIRE((reinterpret_cast<size_t>(ilu->lu) & 15) == 0, "lu alignment");
  FLOW_REQUIRE((reinterpret_cast<size_t>(ilu->packed) & 15) == 0,
               "packed alignment");
  FLOW_REQUIRE(!ilu->single_vector || ilu->packed != nullptr,
               "single_vector needs the packed streams");
  if (ilu->cycle) return tl_check(static_cast<const flow_tl*>(ilu->cycle), op_size);
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

// First-fit greedy colouring in the given (mesh) order -- HOST routine, setup
// only.  On a mesh numbered along its structure this yields a locally periodic
// colour pattern: consecutive rows of one colour have their k-th neighbours in
// the same colour at consecutive positions, so the gathers of a sweep
// wavefront fall into a handful of cache lines (a randomised colouring
// scatters them over ~64).
extern "C" int flow_color_greedy_host(int n, const int* rowptr, const int* cols,
                                      int* colour, int* ncolors) {
  FLOW_REQUIRE(n > 0 && rowptr && cols && colour && ncolors, "colouring");
  int nc = 0;
  for (int v = 0; v < n; ++v) colour[v] = -1;
  for (int v = 0; v < n; ++v) {
    unsigned long long used = 0ull;
    for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
      const int u = cols[k];
      FLOW_REQUIRE(u >= 0 && u < n, "colouring: column out of range");
      if (u != v && colour[u] >= 0) used |= 1ull << colour[u];
    }
    const unsigned long long freebits = ~used;
    FLOW_REQUIRE((freebits & ((1ull << 63) - 1)) != 0ull,
                 "colouring: more than 63 colours");
    const int c = __builtin_ctzll(freebits);
    colour[v] = c;
    if (c + 1 > nc) nc = c + 1;
  }
  *ncolors = nc;
  return FLOW_OK;
}

// Iterated greedy (Culberson) on a proper colouring -- HOST routine, setup only.
extern "C" int flow_color_iterate_host(int n, const int* rowptr, const int* cols,
                                       int* colour, int* ncolors, int rounds) {
  FLOW_REQUIRE(n > 0 && rowptr && cols && colour && ncolors && *ncolors >= 1 &&
                   *ncolors <= 63 && rounds >= 0,
               "iterated colouring arguments");
  std::vector<int> order(n), next(n), count, start;
  int nc = *ncolors;
  for (int r = 0; r < rounds; ++r) {
    // the classes in this pass's order
    count.assign(nc, 0);
    for (int v = 0; v < n; ++v) {
      FLOW_REQUIRE(colour[v] >= 0 && colour[v] < nc, "iterated colouring: colour");
      ++count[colour[v]];
    }
    std::vector<int> classes(nc);
    for (int c = 0; c < nc; ++c) classes[c] = c;
    if (r % 2 == 0)
      std::reverse(classes.begin(), classes.end());
    else
      std::stable_sort(classes.begin(), classes.end(),
                       [&](int a, int b) { return count[a] > count[b]; });
    start.assign(nc + 1, 0);
    std::vector<int> rank_of(nc);
    for (int k = 0; k < nc; ++k) rank_of[classes[k]] = k;
    for (int k = 0; k < nc; ++k) start[k + 1] = start[k] + count[classes[k]];
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int v = 0; v < n; ++v) order[fill[rank_of[colour[v]]]++] = v;
    // first fit in that order
    for (int v = 0; v < n; ++v) next[v] = -1;
    int nn = 0;
    for (int k = 0; k < n; ++k) {
      const int v = order[k];
      unsigned long long used = 0ull;
      for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) {
        const int u = cols[p];
        if (u != v && next[u] >= 0) used |= 1ull << next[u];
      }
      const int c = __builtin_ctzll(~used);
      next[v] = c;
      if (c + 1 > nn) nn = c + 1;
    }
    if (nn <= nc) {            // (never more: kept for the next pass)
      for (int v = 0; v < n; ++v) colour[v] = next[v];
      nc = nn;
    }
  }
  *ncolors = nc;
  return FLOW_OK;
}

// Algebraic aggregation for the smoothed-aggregation hierarchy -- HOST routine,
// setup only (include/flow_hip.h).
extern "C" int flow_aggregate_host(int n, const int* rowptr, const int* cols,
                                   const double* vals, double theta,
                                   const unsigned char* free_rows, int* agg,
                                   int* naggregates) {
  FLOW_REQUIRE(n > 0 && rowptr && cols && vals && agg && naggregates && theta >= 0.0,
               "aggregation arguments");
  std::vector<double> diag(n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      FLOW_REQUIRE(cols[k] >= 0 && cols[k] < n, "aggregation: column out of range");
      if (cols[k] == i) diag[i] = fabs(vals[k]);
    }
  auto is_free = [&](int i) { return free_rows == nullptr || free_rows[i] != 0; };
  auto strong = [&](int i, int k) {
    const int j = cols[k];
    return j != i && is_free(j) &&
           fabs(vals[k]) >= theta * sqrt(diag[i] * diag[j]) && vals[k] != 0.0;
  };
  for (int i = 0; i < n; ++i) agg[i] = -1;
  int na = 0;
  // pass 1: roots whose whole strong neighbourhood is still untouched
  for (int i = 0; i < n; ++i) {
    if (!is_free(i) || agg[i] >= 0) continue;
    bool untouched = true, any = false;
    for (int k = rowptr[i]; k < rowptr[i + 1] && untouched; ++k)
      if (strong(i, k)) {
        any = true;
        if (agg[cols[k]] >= 0) untouched = false;
      }
    if (!untouched || !any) continue;
    agg[i] = na;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (strong(i, k)) agg[cols[k]] = na;
    ++na;
  }
  // pass 2: join the aggregate (of pass 1) of the strongest coupling
  std::vector<int> joined(n, -1);
  for (int i = 0; i < n; ++i) {
    if (!is_free(i) || agg[i] >= 0) continue;
    double best = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (strong(i, k) && agg[cols[k]] >= 0 && fabs(vals[k]) > best) {
        best = fabs(vals[k]);
        joined[i] = agg[cols[k]];
      }
  }
  for (int i = 0; i < n; ++i)
    if (joined[i] >= 0) agg[i] = joined[i];
  // pass 3: the rest, with their unaggregated strong neighbours
  for (int i = 0; i < n; ++i) {
    if (!is_free(i) || agg[i] >= 0) continue;
    agg[i] = na;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (strong(i, k) && agg[cols[k]] < 0) agg[cols[k]] = na;
    ++na;
  }
  *naggregates = na;
  return FLOW_OK;
}

extern "C" int flow_ilu0_factor(const flow_ilu_plan* plan, int nblocks,
                                const double* avals0, const double* avals1,
                                double* lu, void* stream) {
  int rc = check_plan(plan);
  if (rc) return rc;
  FLOW_REQUIRE(nblocks == 1 || nblocks == 2, "ilu blocks");
  FLOW_REQUIRE(avals0 && lu && (nblocks == 1 || avals1), "ilu factor pointers");
  return factor(plan, nblocks, avals0, avals1, lu, as_stream(stream));
}

extern "C" int flow_ilu0_pack(const flow_ilu* ilu, float* packed, void* stream) {
  FLOW_REQUIRE(ilu && ilu->plan && packed, "ilu pack pointers");
  int rc = ilu_check(ilu, ilu->nblocks * ilu->plan->n);
  if (rc) return rc;
  FLOW_REQUIRE((reinterpret_cast<size_t>(packed) & 15) == 0, "packed alignment");
  const flow_ilu_plan* P = ilu->plan;
  // the L and U streams lie back to back in the packed buffer; in lu they are
  // separated by alignment padding
  const size_t lus = static_cast<size_t>(P->lu_size);
  hipStream_t st = as_stream(stream);
  for (int part = 0; part < 2; ++part) {
    const int count = part == 0 ? P->nnz_l : P->nnz_u;
    const double* src = ilu->lu + (part == 0 ? P->off_l : P->off_u);
    float* dst = packed + (part == 0 ? 0 : static_cast<size_t>(P->nnz_l) * ilu->nblocks);
    if (ilu->nblocks == 1)
      hipLaunchKernelGGL((ilu_pack_kernel<1>), dim3(grid_for(count)), dim3(kBlock),
                         0, st, count, lus, src, dst);
    else
      hipLaunchKernelGGL((ilu_pack_kernel<2>), dim3(grid_for(count)), dim3(kBlock),
                         0, st, count, lus, src, dst);
  }
  FLOW_CHECK_LAUNCH();
  return FLOW_OK;
}

extern "C" int flow_ilu0_solve(const flow_ilu* ilu, const double* r, double* z,
                               double* work, void* stream) {
  FLOW_REQUIRE(ilu && ilu->plan, "ilu");
  int rc = ilu_check(ilu, ilu->nblocks * ilu->plan->n);
  if (rc) return rc;
  FLOW_REQUIRE(r && z && work && r != work && z != work, "ilu solve pointers");
  return ilu_apply(ilu, r, z, work, as_stream(stream));
}
