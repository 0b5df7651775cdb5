// Development: cost of a cooperative-groups grid barrier on MI355X against the
// cost of a kernel boundary (decides whether fusing the 17 launches of an ILU
// application into one persistent kernel can pay).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void sync_kernel(int nsync, double* y) {
  cg::grid_group grid = cg::this_grid();
  double v = y[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < nsync; ++i) {
    y[blockIdx.x * blockDim.x + threadIdx.x] = v + i;
    grid.sync();
    v += y[(blockIdx.x * blockDim.x + threadIdx.x + 977) % (gridDim.x * blockDim.x)];
  }
  y[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void tiny_kernel(double* y, int i) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  y[k] = y[(k + 977) % (gridDim.x * blockDim.x)] + i;
}
int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sync_kernel, 256, 0);
  printf("CUs %d, blocks/CU %d, cooperative %d\n", prop.multiProcessorCount, per_cu, prop.cooperativeLaunch);
  hipStream_t st;
  hipStreamCreate(&st);
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    if (blocks_per_cu > per_cu) continue;
    const int grid = prop.multiProcessorCount * blocks_per_cu;
    double* y;
    hipMalloc(&y, sizeof(double) * grid * 256);
    hipMemset(y, 0, sizeof(double) * grid * 256);
    int nsync = 200;
    void* args[] = {&nsync, &y};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, st);
      hipError_t err = hipLaunchCooperativeKernel((void*)sync_kernel, dim3(grid), dim3(256), args, 0, st);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("grid %5d: %s  %.2f us per grid.sync\n", grid, hipGetErrorString(err), ms * 1e3 / nsync);
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, st);
      for (int i = 0; i < nsync; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, st, y, i);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("grid %5d: %.2f us per dependent tiny kernel\n", grid, ms * 1e3 / nsync);
    }
    hipFree(y);
  }
  return 0;
}
