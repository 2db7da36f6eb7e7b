// scatter_probe — developer microbenchmark: can ONE pass move 100 M shuffled 8-byte records into their final
// buckets?  Every workgroup owns a private region per bucket (no global atomics: cursors live in LDS), reads
// 12 bytes per line like the order-restoring path's first pass and appends each record to its bucket's region.
// What decides the rate is whether the partly written cache lines at the head of (workgroups x buckets) regions
// are put together on the chip (L2, Infinity Cache) before they reach HBM.
//   scatter_probe [n_buckets] [workgroups] [million lines]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)
__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 load_stream16(const int* p) {
  const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
  return make_int4(v.x, v.y, v.z, v.w);
}
constexpr int kRound = 4096;

__global__ void k_fill(int* ids, long long n, int n_buckets) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    ids[i] = (int)(mix((unsigned long long)i * 0x9E3779B97F4A7C15ull) % (unsigned long long)n_buckets);
}

// MODE 0: every record stored where its bucket's cursor points (64 lanes, up to 64 different lines per store)
// MODE 1: the round is first put in bucket order in LDS, then written run by run (consecutive threads, consecutive
//         records of a run)
template <int MODE>
__global__ void __launch_bounds__(1024) k_scatter(const int* ids, const int* b, const int* c, long long n,
                                                  int n_buckets, unsigned int cap, unsigned long long* out,
                                                  unsigned int* dropped) {
  extern __shared__ unsigned int smem[];
  unsigned int* cur = smem;                                  // [n_buckets] records written so far
  unsigned int* cnt = cur + n_buckets;                       // [n_buckets + 1] MODE 1: this round's count, then start
  unsigned long long* s_rec = reinterpret_cast<unsigned long long*>(cnt + n_buckets + 2);
  unsigned short* s_bin = reinterpret_cast<unsigned short*>(s_rec + kRound);
  for (int i = threadIdx.x; i < n_buckets; i += 1024) { cur[i] = 0; cnt[i] = 0; }
  __syncthreads();
  const long long rounds = n / kRound;
  unsigned int drop = 0;
  for (long long r = blockIdx.x; r < rounds; r += gridDim.x) {
    const long long l0 = r * kRound + threadIdx.x * 4;
    const int4 k = load_stream16(ids + l0), x = load_stream16(b + l0), y = load_stream16(c + l0);
    const int key[4] = {k.x, k.y, k.z, k.w};
    const int xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned int pos = atomicAdd(&cur[key[i]], 1u);
        const unsigned long long rec = ((unsigned long long)(l0 + i) << 24) ^ (unsigned long long)(unsigned)(xv[i] * yv[i]);
        if (pos < cap) out[((size_t)key[i] * gridDim.x + blockIdx.x) * cap + pos] = rec; else ++drop;
      }
    } else {
      unsigned int pos[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pos[i] = atomicAdd(&cnt[key[i]], 1u);
      __syncthreads();
      // exclusive scan of the round's counts (n_buckets <= 2048: two per thread), in place
      __shared__ unsigned int s_w[16];
      const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
      const unsigned int c0 = 2 * t < n_buckets ? cnt[2 * t] : 0, c1 = 2 * t + 1 < n_buckets ? cnt[2 * t + 1] : 0;
      unsigned int incl = c0 + c1;
      for (int d = 1; d < 64; d <<= 1) { const unsigned int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
      if (lane == 63) s_w[wave] = incl;
      __syncthreads();
      unsigned int off = incl - (c0 + c1);
      for (int w = 0; w < wave; ++w) off += s_w[w];
      if (2 * t < n_buckets) cnt[2 * t] = off;
      if (2 * t + 1 < n_buckets) cnt[2 * t + 1] = off + c0;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned int p = cnt[key[i]] + pos[i];
        s_rec[p] = ((unsigned long long)(l0 + i) << 24) ^ (unsigned long long)(unsigned)(xv[i] * yv[i]);
        s_bin[p] = (unsigned short)key[i];
      }
      __syncthreads();
      for (int e = t; e < kRound; e += 1024) {
        const unsigned int bk = s_bin[e];
        const unsigned int p = cur[bk] + ((unsigned)e - cnt[bk]);
        if (p < cap) out[((size_t)bk * gridDim.x + blockIdx.x) * cap + p] = s_rec[e]; else ++drop;
      }
      __syncthreads();
      if (2 * t < n_buckets) { cur[2 * t] += c0; cnt[2 * t] = 0; }
      if (2 * t + 1 < n_buckets) { cur[2 * t + 1] += c1; cnt[2 * t + 1] = 0; }
      __syncthreads();
    }
  }
  if (drop) atomicAdd(dropped, drop);
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 1526;
  const int g = argc > 2 ? atoi(argv[2]) : 256;
  const long long n = (argc > 3 ? atoll(argv[3]) : 100) * 1000000ll / kRound * kRound;
  if (nb < 1 || nb > 2048 || g < 1) { printf("bad arguments\n"); return 1; }
  int *ids, *b, *c; unsigned long long* out; unsigned int* dropped;
  const unsigned int cap = (unsigned int)((n * 3 / 2 / ((long long)nb * g) + 64) & ~15ll);
  const size_t out_n = (size_t)nb * g * cap;
  CK(hipMalloc(&ids, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
  CK(hipMalloc(&out, out_n * 8)); CK(hipMalloc(&dropped, 4)); CK(hipMemset(dropped, 0, 4));
  CK(hipMemset(b, 1, n * 4)); CK(hipMemset(c, 2, n * 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, ids, n, nb);
  CK(hipDeviceSynchronize());
  printf("buckets %d, workgroups %d, lines %lld, region of %u records per (bucket, workgroup), output %.2f GB\n", nb, g, n, cap, out_n * 8 / 1e9);
  const size_t lds = (size_t)(2 * nb + 2) * 4 + (size_t)kRound * 10;
  for (int mode = 0; mode < 2; ++mode) {
    auto kern = mode == 0 ? k_scatter<0> : k_scatter<1>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(g), dim3(1024), lds, 0, ids, b, c, n, nb, cap, out, dropped);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(g), dim3(1024), lds, 0, ids, b, c, n, nb, cap, out, dropped);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    unsigned int d = 0; CK(hipMemcpy(&d, dropped, 4, hipMemcpyDeviceToHost));
    printf("%s: %.3f ms  (%.2f TB/s over 12 B read + 8 B written per line; %u records dropped)\n",
           mode == 0 ? "stored where the cursor points" : "put in bucket order in LDS first", ms, n * 20.0 / ms / 1e9, d);
  }
  return 0;
}
