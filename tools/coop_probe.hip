// coop_probe — developer microbenchmark: what does a cooperative launch with grid-wide barriers cost
// next to the same work as separate launches?
//   hipcc -O3 --offload-arch=gfx950 -o tools/coop_probe tools/coop_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__global__ void __launch_bounds__(1024) k_phase(int* p, int v) {
  if (threadIdx.x == 0) p[blockIdx.x] += v;
}
__global__ void __launch_bounds__(1024) k_coop(int* p) {
  cg::grid_group g = cg::this_grid();
  if (threadIdx.x == 0) p[blockIdx.x] += 1;
  g.sync();
  if (threadIdx.x == 0) p[blockIdx.x] += 2;
  g.sync();
  if (threadIdx.x == 0) p[blockIdx.x] += 3;
}
int main() {
  int* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256, lds = 140 * 1024;
  CK(hipFuncSetAttribute((const void*)k_phase, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)k_coop, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) {
      hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(1024), lds, 0, p, 1);
      hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(1024), lds, 0, p, 2);
      hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(1024), lds, 0, p, 3);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("three launches:        %.2f us per triple\n", ms * 10);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) {
      void* args[] = {&p};
      CK(hipLaunchCooperativeKernel((const void*)k_coop, dim3(blocks), dim3(1024), args, lds, 0));
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one cooperative launch with two grid barriers: %.2f us\n", ms * 10);
  }
  return 0;
}
