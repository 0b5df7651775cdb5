// Development: what does replaying a captured chain of dependent kernels
// (hipGraphLaunch) cost on MI355X / ROCm 7.2 against launching the same chain
// kernel by kernel?  The chain imitates one V-cycle-CG iteration at the proxy
// size: `len` dependent kernels, a few of them one workgroup (scalar kernels),
// the others `grid` workgroups streaming `n` doubles.
// Second line per case: the solver's pattern -- ONE chain, then a read-back.
//   hipcc --offload-arch=gfx950 -O3 graph_launch.hip -o graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                          \
      return 1;                                                               \
    }                                                                         \
  } while (0)

__global__ void stream_kernel(int n, const double* __restrict__ a,
                              double* __restrict__ b, const double* s) {
  if (s[1] != 0.0) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    b[i] = a[i] * 1.0000001 + s[0];
}
__global__ void scalar_kernel(double* s, const double* a) {
  if (threadIdx.x == 0) s[0] = a[0] * 1e-30;
}

static double now() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

static void body(int len, int n, int grid, double* a, double* b, double* s,
                 hipStream_t st) {
  for (int k = 0; k < len; ++k) {
    if (k % 5 == 4)
      hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(64), 0, st, s, a);
    else
      hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, st, n,
                         (k & 1) ? b : a, (k & 1) ? a : b, s);
  }
}

int main() {
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int nmax = 1 << 22;
  double *a, *b, *s;
  CHECK(hipMalloc(&a, sizeof(double) * nmax));
  CHECK(hipMalloc(&b, sizeof(double) * nmax));
  CHECK(hipMalloc(&s, sizeof(double) * 8));
  CHECK(hipMemset(a, 0, sizeof(double) * nmax));
  CHECK(hipMemset(b, 0, sizeof(double) * nmax));
  CHECK(hipMemset(s, 0, sizeof(double) * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 200;
  for (int n : {1 << 12, 1 << 17, 1 << 20, 1 << 22}) {
    for (int len : {8, 15, 30}) {
      const int grid = (n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256;
      // plain launches
      float ms_plain = 0, ms_graph = 0;
      double host_plain = 0, host_graph = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipStreamSynchronize(st));
        const double t0 = now();
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) body(len, n, grid, a, b, s, st);
        CHECK(hipEventRecord(e1, st));
        host_plain = now() - t0;
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms_plain, e0, e1));
      }
      // captured once, replayed
      hipGraph_t g;
      hipGraphExec_t ge;
      const double tc = now();
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      body(len, n, grid, a, b, s, st);
      CHECK(hipStreamEndCapture(st, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      const double capture = now() - tc;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipStreamSynchronize(st));
        const double t0 = now();
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e1, st));
        host_graph = now() - t0;
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms_graph, e0, e1));
      }
      // the solver's pattern: a chain, then a read-back (synchronisation)
      double sync_plain = 0, sync_graph = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < iters; ++i) {
          body(len, n, grid, a, b, s, st);
          CHECK(hipStreamSynchronize(st));
        }
        sync_plain = now() - t0;
        t0 = now();
        for (int i = 0; i < iters; ++i) {
          CHECK(hipGraphLaunch(ge, st));
          CHECK(hipStreamSynchronize(st));
        }
        sync_graph = now() - t0;
      }
      printf("            chain + synchronise: plain %7.2f us, graph %7.2f us\n",
             sync_plain * 1e6 / iters, sync_graph * 1e6 / iters);
      CHECK(hipGraphExecDestroy(ge));
      CHECK(hipGraphDestroy(g));
      printf("n %8d  chain %2d: plain %7.2f us/chain (%5.2f per kernel; host "
             "%7.2f)   graph %7.2f us/chain (%5.2f per kernel; host %7.2f)   "
             "capture+instantiate %.0f us\n",
             n, len, ms_plain * 1e3 / iters, ms_plain * 1e3 / iters / len,
             host_plain * 1e6 / iters, ms_graph * 1e3 / iters,
             ms_graph * 1e3 / iters / len, host_graph * 1e6 / iters,
             capture * 1e6);
    }
  }
  return 0;
}
