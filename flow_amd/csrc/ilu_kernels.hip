// K11: multicolour ILU(0) on gfx950.  The plan (colouring, colour-major permuted
// CSR, strictly-lower / strictly-upper streams with CSR-stream row blocks per
// colour, map back to the operator's value plane) is built once per pattern on
// the host: flow_amd/fem/ilu.py.
//
// Stands in for the sparse LU behind the reference's Newton and heat solves
// (flow/navier_stokes/pressure_correction.py:224-254, flow/heat.py:117-121) as
// the preconditioner of BiCGStab.
//
//  * factorisation: one launch per colour, a lane per row (IKJ with a sorted
//    merge); runs once per (re)factorisation;
//  * sweeps: one launch per colour and sweep; rows of a colour are contiguous in
//    the permuted numbering, so each launch streams its slice of the L (or U)
//    triangle exactly once with the SpMV's LDS-tiled structure (16-B value
//    loads, products parked in LDS, one lane per row for the segmented sum) and
//    applies the sweep update in the epilogue.  Both diagonal blocks of a
//    two-field operator go through the same launch (blockIdx.y).
// HBM-bound: 12 B per factor entry and application.
#include "common.h"

namespace flow {

constexpr int kPairsI = 4;
constexpr int kTileI = 2 * kBlock * kPairsI;
static_assert(FLOW_SPMV_NNZ_PER_BLOCK == kTileI - 2, "ILU sweeps reuse the SpMV tiling");

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
    if (k < nnz_l) lvals[k] = lu[l_pos[k]];
    else if (k < nnz_l + nnz_u) uvals[k - nnz_l] = lu[u_pos[k - nnz_l]];
    else dinv[k - nnz_l - nnz_u] = 1.0 / lu[diag[k - nnz_l - nnz_u]];
  }
}

// One colour of one sweep, CSR-stream style.
//   FWD:  y_i = r[old(i)] - sum_k L_ik y_k
//   BWD:  y_i = (y_i - sum_j U_ij y_j) / U_ii ;  z[old(i)] = y_i
// blockIdx.y selects the diagonal block: values at vals + blk*lu_size, vectors
// at + blk*n.
template <bool BWD>
__global__ __launch_bounds__(kBlock) void ilu_sweep_kernel(
    int n, size_t lu_size, const int* __restrict__ rowptr,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const double* __restrict__ dinv, const int* __restrict__ rowblocks,
    const int* __restrict__ old_of_new, const double* __restrict__ r,
    double* __restrict__ y, double* __restrict__ z) {
  __shared__ double prod[kTileI];
  const size_t blk = blockIdx.y;
  vals += blk * lu_size;
  if (BWD) dinv += blk * lu_size;
  y += blk * n;
  const int r0 = rowblocks[blockIdx.x];
  const int r1 = rowblocks[blockIdx.x + 1];
  const int k0 = rowptr[r0];
  const int k1 = rowptr[r1];
  const int ka = k0 & ~1;
  const int row = r0 + threadIdx.x;
  int a = 0, b = 0, o = 0;
  // everything the row's final update needs is requested up front, so that
  // these (dependent) loads are in flight while the triangle streams in
  double rhs = 0.0, di = 0.0;
  if (row < r1) {
    a = rowptr[row] - ka;
    b = rowptr[row + 1] - ka;
    o = old_of_new[row];
    if (!BWD) {
      rhs = r[blk * n + o];
    } else {
      rhs = y[row];
      di = dinv[row];
    }
  }
  const double2* __restrict__ v2p = reinterpret_cast<const double2*>(vals + ka);
  const int2* __restrict__ c2p = reinterpret_cast<const int2*>(cols + ka);
  const int npair = (k1 - ka + 1) >> 1;
#pragma unroll
  for (int j = 0; j < kPairsI; ++j) {
    const int p = threadIdx.x + j * kBlock;
    if (p < npair) {
      const double2 v = v2p[p];
      const int2 c = c2p[p];
      // the possible extra element past k1 belongs to a later row / the pad
      // (column 0): its product is never summed
      prod[2 * p] = v.x * y[c.x];
      prod[2 * p + 1] = v.y * y[c.y];
    }
  }
  __syncthreads();
  if (row < r1) {
    double s = 0.0;
    for (int k = a; k < b; ++k) s += prod[k];
    if (!BWD) {
      y[row] = rhs - s;
    } else {
      const double yi = (rhs - s) * di;
      y[row] = yi;
      z[blk * n + o] = yi;
    }
  }
}

static int check_plan(const flow_ilu_plan* P) {
  FLOW_REQUIRE(P && P->n > 0 && P->nnz > 0 && P->ncolors > 0, "ilu plan sizes");
  FLOW_REQUIRE(P->color_ptr_host && P->l_rbptr_host && P->u_rbptr_host,
               "ilu plan host arrays");
  FLOW_REQUIRE(P->rowptr && P->cols && P->diag && P->src_pos && P->old_of_new &&
                   P->l_rowptr && P->l_cols && P->l_pos && P->l_rowblocks &&
                   P->u_rowptr && P->u_cols && P->u_pos && P->u_rowblocks,
               "ilu plan pointers");
  FLOW_REQUIRE(P->color_ptr_host[0] == 0 && P->color_ptr_host[P->ncolors] == P->n,
               "ilu colour ranges");
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

// out = blockdiag(LU_0[, LU_1])^-1 in ; work: nblocks * n doubles
int ilu_apply(const flow_ilu* ilu, const double* in, double* out, double* work,
              hipStream_t st) {
  const flow_ilu_plan* P = ilu->plan;
  const size_t lus = static_cast<size_t>(P->lu_size);
  for (int c = 0; c < P->ncolors; ++c) {
    const int nb = P->l_rbptr_host[c + 1] - P->l_rbptr_host[c];
    if (nb <= 0) continue;
    hipLaunchKernelGGL((ilu_sweep_kernel<false>), dim3(nb, ilu->nblocks),
                       dim3(kBlock), 0, st, P->n, lus, P->l_rowptr, P->l_cols,
                       ilu->lu + P->off_l, static_cast<const double*>(nullptr),
                       P->l_rowblocks + P->l_rbptr_host[c], P->old_of_new, in, work,
                       static_cast<double*>(nullptr));
  }
  for (int c = P->ncolors - 1; c >= 0; --c) {
    const int nb = P->u_rbptr_host[c + 1] - P->u_rbptr_host[c];
    if (nb <= 0) continue;
    hipLaunchKernelGGL((ilu_sweep_kernel<true>), dim3(nb, ilu->nblocks),
                       dim3(kBlock), 0, st, P->n, lus, P->u_rowptr, P->u_cols,
                       ilu->lu + P->off_u, ilu->lu + P->off_d,
                       P->u_rowblocks + P->u_rbptr_host[c], P->old_of_new, in, work,
                       out);
  }
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
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

extern "C" int flow_ilu0_factor(const flow_ilu_plan* plan, int nblocks,
                                const double* avals0, const double* avals1,
                                double* lu, void* stream) {
  int rc = check_plan(plan);
  if (rc) return rc;
  FLOW_REQUIRE(nblocks == 1 || nblocks == 2, "ilu blocks");
  FLOW_REQUIRE(avals0 && lu && (nblocks == 1 || avals1), "ilu factor pointers");
  return factor(plan, nblocks, avals0, avals1, lu, as_stream(stream));
}

extern "C" int flow_ilu0_solve(const flow_ilu* ilu, const double* r, double* z,
                               double* work, void* stream) {
  FLOW_REQUIRE(ilu && ilu->plan, "ilu");
  int rc = ilu_check(ilu, ilu->nblocks * ilu->plan->n);
  if (rc) return rc;
  FLOW_REQUIRE(r && z && work && r != work && z != work, "ilu solve pointers");
  return ilu_apply(ilu, r, z, work, as_stream(stream));
}
