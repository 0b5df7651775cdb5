// Development: what ONE workgroup of 1024 threads streams (VERDICT r4 item 5b:
// "one 1024-thread workgroup for the multigrid levels >= 2 + the coarse gemv").
// At the proxy size those levels hold the 1.5 k x 1.5 k fp32 coarse inverse
// (9 MB per application) and a level-2 operator of ~4 MB: 13 MB per V-cycle.
//   hipcc --offload-arch=gfx950 -O3 one_workgroup.hip -o one_workgroup
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void read_kernel(size_t n16, const float4* __restrict__ src,
                                                    float* __restrict__ out) {
  float acc = 0.f;
  // every lane keeps 4 independent 16-byte loads in flight
  size_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride],
                 d = src[i + 3 * stride];
    acc += a.x + b.y + c.z + d.w;
  }
  for (; i < n16; i += stride) acc += src[i].x;
  if (acc == 12345.678f) out[0] = acc;      // (never: keeps the loads alive)
}

int main() {
  const size_t bytes = 13u << 20;
  float4* src;
  float* out;
  hipMalloc(&src, bytes);
  hipMalloc(&out, 64);
  hipMemset(src, 0, bytes);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {1, 2, 4, 8, 32, 256, 2048}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      for (int k = 0; k < 20; ++k)
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(1024), 0, st, bytes / 16, src, out);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%4d workgroup(s) of 1024 threads over 13 MB: %7.1f us per pass = %6.1f GB/s\n",
           grid, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) * 1e-9);
  }
  return 0;
}
