// Shared host-side helpers of libflow_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <vector>

#include "../../include/flow_hip.h"

// Every kernel launch of the library is counted (flow_launch_count: what a time
// step costs in launches is what bounds it where the kernels are small -- the
// 8-GPU rank): the launch macro of hip_runtime.h with one increment in front.
namespace flow {
extern unsigned long long g_launches;
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock,        \
                           streamId, ...)                                       \
  do {                                                                          \
    ++flow::g_launches;                                                         \
    (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(     \
        __VA_ARGS__);                                                           \
  } while (0)

namespace flow {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// ilu_kernels.hip
// stop: a solver's sticky "done" flag (la_kernels.hip) -- every kernel returns
// at once when it is set
int ilu_apply(const flow_ilu* ilu, const double* in, double* out, double* work,
              hipStream_t st, const double* stop = nullptr);
int ilu_check(const flow_ilu* ilu, int op_size);

// pmg_kernels.hip: the p-multigrid / Chebyshev preconditioner (flow_pmg)
int pmg_apply(const flow_pmg* M, const double* r, double* z, hipStream_t st,
              const double* stop = nullptr);
int pmg_check(const flow_pmg* M, int op_size);
// tl_kernels.hip: the two-level cycle with ILU(0) smoothing (flow_tl; reached
// through flow_ilu.cycle by ilu_apply / ilu_check)
int tl_apply(const flow_tl* T, const double* r, double* z, hipStream_t st,
             const double* stop = nullptr);
int tl_check(const flow_tl* T, int op_size);
// la_kernels.hip
// y = A x for any operator kind (vectors of operator_size(A) entries)
int operator_apply(const flow_operator* A, const double* x, double* y,
                   hipStream_t st, const double* stop = nullptr);
int operator_size(const flow_operator* A);
int sum_partials_host(double* work, int nparts, double* host, hipStream_t st);
int check_operator(const flow_operator* A);
int fill(int n, double value, double* y, hipStream_t st);
// Krylov scalar slots in HBM (a solver's S = work + 3*kRedBlocks); read_state:
// all of them -> host with ONE synchronisation (through the mailbox)
// The CSR-stream tile functions call early(row, has_row) as soon as a lane knows
// its row: a kernel issues its epilogue's loads there (NoEarly: nothing), so
// that they travel with the tile's own loads instead of adding a link to the
// chain of dependent loads behind the row sum.
struct NoEarly {
  __device__ __forceinline__ void operator()(int, bool) const {}
};

enum Slot {
  kGamma = 0, kAlpha, kBeta, kRes2, kB2, kRho, kOmega, kRhoNew, kTmp,
  kBreak, kTarget2, kDone, kConvIt, kIter, kNumSlots = 16
};
int read_state(const double* S, double* host, hipStream_t st);
// K15 (la_kernels.hip): argument checks and the one collective
int check_comm(const flow_comm* c, long long need);
int check_rows(const flow_rows* R);
int exchange(const flow_comm* c, int count);
// [sums | ... | halo at hoff]: see la_kernels.hip
int exchange_halo(const flow_comm* C, const flow_rows* R, int ncomp, int sum_count,
                  int hoff, hipStream_t st);

// assembly_kernels.hip: the matrix-free operator (flow_operator kind 3)
int momentum_jvp_check(const flow_momentum_jvp* J);
// prm_dev: the step-size dependent factors in device memory (three doubles
// written by momentum_jvp_params) instead of J->prm by value -- what lets a
// captured iteration body outlive a change of the step size
int momentum_jvp_apply(const flow_momentum_jvp* J, const double* v, double* out,
                       hipStream_t st, int v_stride = 0, int out_stride = 0,
                       const double* stop = nullptr,
                       const double* prm_dev = nullptr);
int momentum_jvp_params(const flow_momentum_jvp* J, double* dst, hipStream_t st);

// graph_replay.hip: iteration bodies as HIP graphs (OFF unless FLOW_AMD_GRAPHS
// / flow_graph_mode says otherwise: measured slower than the launches they
// replace, see there).  A solver hashes every VALUE that determines the
// launches of one iteration (KeyHash), asks replay_prepare for the instantiated
// graph -- on a miss the body is captured once: it must only enqueue on `st`
// -- and replays it with replay_launch; with *exec == nullptr (capture
// unavailable, a loop whose bodies never repeat) it launches the body itself.
// replay_wanted(rows, site): FLOW_AMD_GRAPHS unset / 0 never, 1 always, auto:
// systems of up to kReplayAutoRows rows (FLOW_AMD_GRAPH_ROWS).
constexpr long long kReplayAutoRows = 1500000;
struct KeyHash {
  unsigned long long h = 1469598103934665603ull;
  void bytes(const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
  }
  template <class T> KeyHash& pod(const T& v) {
    bytes(&v, sizeof(T));
    return *this;
  }
  template <class T> KeyHash& obj(const T* p) {   // a struct by value, or "none"
    const unsigned char tag = p ? 1 : 0;
    bytes(&tag, 1);
    if (p) bytes(p, sizeof(T));
    return *this;
  }
};
enum ReplaySite { kReplayCg = 1, kReplayGmres = 2, kReplayMass = 4 };
bool replay_wanted(long long rows, int site);
int replay_prepare(unsigned long long key, int site, hipStream_t st,
                   const std::function<int()>& body, hipGraphExec_t* exec,
                   int* nodes);
int replay_launch(hipGraphExec_t exec, int nodes, hipStream_t st);
// la_kernels.hip; skip_prm: without flow_momentum_jvp.prm (passed through
// device memory)
void key_operator(KeyHash& k, const flow_operator* A, bool skip_prm = false);

#define FLOW_CHECK_HIP(expr)                                                 \
  do {                                                                       \
    hipError_t err_ = (expr);                                                \
    if (err_ != hipSuccess) {                                                \
      flow::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err_), \
                      __FILE__, __LINE__);                                   \
      return FLOW_HIP_ERROR;                                                 \
    }                                                                        \
  } while (0)

#define FLOW_REQUIRE(cond, msg)                                  \
  do {                                                           \
    if (!(cond)) {                                               \
      flow::set_error("invalid argument: %s (%s)", msg, #cond);  \
      return FLOW_INVALID;                                       \
    }                                                            \
  } while (0)

#define FLOW_CHECK_LAUNCH() FLOW_CHECK_HIP(hipGetLastError())

constexpr int kBlock = 256;        // 4 wavefronts of 64
constexpr int kRedBlocks = 1024;   // partial sums per reduction
constexpr int kMaxGrid = 2048;     // 256 CUs x 8 blocks: memory-bound grid cap

inline int grid_for(long long n, int per_block = kBlock, int cap = kMaxGrid) {
  long long g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

// Scalars that one kernel writes and the next one reads, on the same addresses
// iteration after iteration -- the solver scalars (alpha, beta, omega, ...),
// the sums a one-block finisher reads, the head of the all-reduce buffer of the
// sharded CG -- are read with SYSTEM-scope loads, which go past the scalar
// cache, the L1s and the per-XCD L2s.  Measured on MI355X (ROCm 7.2): with
// plain loads of `S[kAlpha]` (wave-uniform, so the compiler uses the scalar
// cache) some workgroups of a kernel launched right behind the one-thread
// kernel that updates alpha / beta / omega still saw the PREVIOUS iteration's
// values.  BiCGStab converged all the same -- on twice the iterations, and
// along a path that depended on where the allocator had put the buffers
// (tools/debug_placement.py).  Agent scope was enough in the experiment; system
// scope costs the same here.  Per-lane (vector) loads of the bigger vectors
// were never seen stale.
__device__ __forceinline__ double load_scalar(const double* p) {
  return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_scalar(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a solver's sticky "done" flag (la_kernels.hip: convergence is decided on the
// device); nullptr: none
__device__ __forceinline__ bool stopped(const double* stop) {
  return stop != nullptr && load_scalar(stop) != 0.0;
}

// block-wide sum, result valid in thread 0 (blockDim.x == 256)
// Workgroups are dealt round-robin to the 8 XCDs (observed on MI355X, not
// promised by HIP: a speed matter only): the blocks b, b+8, b+16, ... share an
// XCD and its 4 MB L2.  Give them contiguous runs of kXcdRun tiles, so that the
// x[col] windows of successive tiles of a banded matrix overlap in THAT L2
// instead of being fetched once per XCD -- but keep the eight runs next to each
// other (groups of 8 kXcdRun tiles), so that the chip still streams ONE region
// of HBM at a time (one run per XCD over the whole matrix costs an
// HBM-resident 4 GB matrix 12 % of its bandwidth; measured, tools/
// spmv_stress.py).  Bijective for any workgroup count.
constexpr int kXcdRun = 64;
__host__ __device__ __forceinline__ int xcd_tile(int b, int nwg) {
  constexpr int kGroup = 8 * kXcdRun;
  const int full = (nwg / kGroup) * kGroup;
  if (b < full) {
    const int g = b / kGroup, i = b - g * kGroup;
    return g * kGroup + (i & 7) * kXcdRun + (i >> 3);
  }
  // the last, partial group: runs of q or q+1 tiles
  const int nt = nwg - full, i = b - full;
  const int q = nt >> 3, r = nt & 7, xcd = i & 7;
  return full + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) +
         (i >> 3);
}

__device__ inline double block_sum(double v) {
  __shared__ double wave_part[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __syncthreads();   // protect wave_part across repeated calls
  if (lane == 0) wave_part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) v = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3];
  return v;
}

// the same for a kernel that reduces exactly once: no barrier to protect the
// partials of an earlier call (the CSR-stream kernels run one of these per
// 1022-nonzero tile, where a barrier is a measurable share of the tile's time)
__device__ inline double block_sum_once(double v) {
  __shared__ double wave_part_once[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) wave_part_once[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    v = wave_part_once[0] + wave_part_once[1] + wave_part_once[2] +
        wave_part_once[3];
  return v;
}

__device__ inline double block_max(double v) {
  __shared__ double wave_part_m[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) wave_part_m[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    v = fmax(fmax(wave_part_m[0], wave_part_m[1]), fmax(wave_part_m[2], wave_part_m[3]));
  return v;
}

}  // namespace flow
