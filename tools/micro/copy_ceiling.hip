// Development aid: which plain copy kernel reaches the box's HBM ceiling?
//   hipcc --offload-arch=gfx950 -O3 -o copy_ceiling copy_ceiling.hip && ./copy_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void copy_k(size_t n2, const double2* __restrict__ s,
                                              double2* __restrict__ d) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) d[i + u * stride] = v[u];
  }
  for (; i < n2; i += stride) d[i] = s[i];
}

template <int U>
__global__ __launch_bounds__(256) void copy_nt_k(size_t n2, const double2* __restrict__ s,
                                                 double2* __restrict__ d) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u].x = __builtin_nontemporal_load(&s[i + u * stride].x);
      v[u].y = __builtin_nontemporal_load(&s[i + u * stride].y);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      __builtin_nontemporal_store(v[u].x, &d[i + u * stride].x);
      __builtin_nontemporal_store(v[u].y, &d[i + u * stride].y);
    }
  }
  for (; i < n2; i += stride) d[i] = s[i];
}

template <int U>
__global__ __launch_bounds__(256) void read_k(size_t n2, const double2* __restrict__ s,
                                              double* __restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 12345.678) out[0] = acc;
}

int main() {
  const size_t bytes = size_t(1) << 30;
  const size_t n2 = bytes / 16;
  double2 *s, *d;
  double* o;
  hipMalloc(&s, bytes);
  hipMalloc(&d, bytes);
  hipMalloc(&o, 64);
  hipMemset(s, 1, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto time = [&](const char* name, auto launch, double traffic) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.1f us  %7.1f GB/s\n", name, ms / 20 * 1e3, traffic / (ms / 20 * 1e-3) / 1e9);
  };
  for (int grid : {2048, 4096, 8192, 16384}) {
    printf("grid %d\n", grid);
    time("copy U=1", [&] { hipLaunchKernelGGL(copy_k<1>, dim3(grid), dim3(256), 0, 0, n2, s, d); }, 2.0 * bytes);
    time("copy U=2", [&] { hipLaunchKernelGGL(copy_k<2>, dim3(grid), dim3(256), 0, 0, n2, s, d); }, 2.0 * bytes);
    time("copy U=4", [&] { hipLaunchKernelGGL(copy_k<4>, dim3(grid), dim3(256), 0, 0, n2, s, d); }, 2.0 * bytes);
    time("copy nt U=4", [&] { hipLaunchKernelGGL(copy_nt_k<4>, dim3(grid), dim3(256), 0, 0, n2, s, d); }, 2.0 * bytes);
    time("read U=4", [&] { hipLaunchKernelGGL(read_k<4>, dim3(grid), dim3(256), 0, 0, n2, s, o); }, 1.0 * bytes);
    time("read U=8", [&] { hipLaunchKernelGGL(read_k<8>, dim3(grid), dim3(256), 0, 0, n2, s, o); }, 1.0 * bytes);
  }
  return 0;
}
