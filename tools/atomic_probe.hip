// atomic_probe — developer microbenchmark: rate of scattered 64-bit global atomic adds into a small table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)
__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
template <int MODE>
__global__ void __launch_bounds__(256) k_atom(unsigned long long* tab, unsigned int* tab32, int C, long long m) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long h = mix((unsigned long long)i * 0x9E3779B97F4A7C15ull);
    const int c = (int)(h % (unsigned long long)C);
    if (MODE == 0) atomicAdd(&tab[c], h & 0xffffff);
    if (MODE == 1) atomicAdd(&tab32[c], (unsigned int)(h & 0xffff));
    if (MODE == 2) __hip_atomic_fetch_add(&tab[c], h & 0xffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) tab[c] = h;   // plain scattered store, for scale
    if (MODE == 4) atomicAdd(&tab[(size_t)(blockIdx.x & 7) * C + c], h & 0xffffff);   // one table per XCD label
    if (MODE == 5) { if ((blockIdx.x & 7) == 0) atomicAdd(&tab[c], h & 0xffffff); }   // one XCD does an eighth of the adds
    if (MODE == 6) {   // one table per XCD by the hardware's own id
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
      if (xcc != (blockIdx.x & 7)) tab32[(8u << 20) - 1] = 1;
      atomicAdd(&tab[(size_t)xcc * C + c], h & 0xffffff);
    }
    if (MODE == 7) atomicAdd(&tab32[(size_t)(blockIdx.x & 7) * C + c], (unsigned int)(h & 0xffff));
  }
}
template <int MODE> void run(const char* name, unsigned long long* tab, unsigned int* t32, int C, long long m) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_atom<MODE>, dim3(2048), dim3(256), 0, 0, tab, t32, C, m);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_atom<MODE>, dim3(2048), dim3(256), 0, 0, tab, t32, C, m);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  printf("%-40s C=%8d  %.3f ms  %.1f G adds/s\n", name, C, ms, m / ms / 1e6);
}
int main() {
  const long long m = 25000000;
  unsigned long long* tab; unsigned int* t32;
  CK(hipMalloc(&tab, 8ull << 20 << 3)); CK(hipMalloc(&t32, 8ull << 20 << 2));
  CK(hipMemset(tab, 0, 8ull << 20 << 3)); CK(hipMemset(t32, 0, 8ull << 20 << 2));
  for (int C : {100000, 1000000, 8000000}) {
    run<0>("u64 atomicAdd (agent scope)", tab, t32, C, m);
    run<1>("u32 atomicAdd (agent scope)", tab, t32, C, m);
    run<2>("u64 atomic add (workgroup scope)", tab, t32, C, m);
    run<3>("plain 8-byte scattered store", tab, t32, C, m);
    if (C <= 1000000) {
      run<4>("u64 atomicAdd, table per XCD label", tab, t32, C, m);
      run<6>("u64 atomicAdd, table per XCC_ID", tab, t32, C, m);
      run<7>("u32 atomicAdd, table per XCD label", tab, t32, C, m);
    }
    run<5>("u64 atomicAdd, 1/8 of adds, one XCD", tab, t32, C, m);
  }
  unsigned int flag = 0;
  CK(hipMemcpy(&flag, t32 + (8u << 20) - 1, 4, hipMemcpyDeviceToHost));
  printf("blockIdx %% 8 == XCC_ID for every workgroup: %s\n", flag ? "NO" : "yes");
  return 0;
}
