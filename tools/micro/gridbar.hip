// Development (round 6): what does a HAND-ROLLED grid barrier cost on MI355X --
// one atomic counter at agent scope, thread 0 of every workgroup adds and spins
// -- with the data exchanged between workgroups (a) through plain loads /
// stores + __threadfence() or (b) through agent-scope (L2-bypassing) accesses?
// Against the dependent kernel launch it would replace (one per colour of an
// ILU sweep: ~10 us at 1 M DoF).  tools/micro/gridsync.hip measured the
// cooperative-groups grid.sync() at 30 us for 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gridbar.hip -o tools/micro/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned nblocks,
                                             unsigned& target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nblocks;
    __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <
               target &&
           ++spins < (1 << 22))
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// MODE 0: plain accesses + __threadfence() around the barrier
// MODE 1: agent-scope relaxed atomics for the exchanged vector
template <int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(int rounds, int n, double* y,
                                                    unsigned* count, int* bad) {
  unsigned target = 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // a "far" partner: another workgroup, another XCD
  const int j = (i + 977 * 256 + 3) % n;
  double v = 1.0;
  for (int r = 0; r < rounds; ++r) {
    const double mine = v + r;
    if (MODE == 0) {
      y[i] = mine;
      __threadfence();
    } else {
      __hip_atomic_store(y + i, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(count, gridDim.x, target);
    double other;
    if (MODE == 0) {
      __threadfence();
      other = y[j];
    } else {
      other = __hip_atomic_load(y + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every thread runs the same recurrence: the partner's value of this round
    // is known
    if (other != mine) atomicAdd(bad, 1);
    v = 0.5 * (v + other - r) + 0.25;
    grid_barrier(count, gridDim.x, target);   // (WAR: before the next store)
  }
  y[i] = v;
}

__global__ void tiny_kernel(int n, double* y, int r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (i + 977 * 256 + 3) % n;
  y[i] = 0.5 * (y[i] + y[j]) + r;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  hipStream_t st;
  hipStreamCreate(&st);
  for (int per_cu : {1, 2, 4}) {
    const int grid = prop.multiProcessorCount * per_cu;
    const int n = grid * 256;
    double* y;
    unsigned* count;
    int* bad;
    hipMalloc(&y, sizeof(double) * n);
    hipMalloc(&count, sizeof(unsigned));
    hipMalloc(&bad, sizeof(int));
    const int rounds = 500;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipMemsetAsync(y, 0, sizeof(double) * n, st);
        hipMemsetAsync(count, 0, sizeof(unsigned), st);
        hipMemsetAsync(bad, 0, sizeof(int), st);
        hipEventRecord(e0, st);
        if (mode == 0)
          hipLaunchKernelGGL(sweep_kernel<0>, dim3(grid), dim3(256), 0, st, rounds,
                             n, y, count, bad);
        else
          hipLaunchKernelGGL(sweep_kernel<1>, dim3(grid), dim3(256), 0, st, rounds,
                             n, y, count, bad);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        int hbad = -1;
        hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost);
        if (rep)
          printf("grid %5d (%d per CU), %s: %.2f us per barrier (2 per round), "
                 "stale or missing reads: %d\n",
                 grid, per_cu,
                 mode == 0 ? "plain + __threadfence()" : "agent-scope accesses  ",
                 ms * 1e3 / (2 * rounds), hbad);
      }
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, st);
      for (int r = 0; r < rounds; ++r)
        hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, st, n, y, r);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep)
        printf("grid %5d: %.2f us per dependent tiny kernel\n", grid,
               ms * 1e3 / rounds);
    }
    hipFree(y);
    hipFree(count);
    hipFree(bad);
  }
  return 0;
}
