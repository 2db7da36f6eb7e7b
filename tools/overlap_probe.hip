// overlap_probe — developer microbenchmark (not part of the product).
//   (1) what the line-pass traffic costs when the 8-byte-per-order stores leave in bursts that every
//       workgroup of the chip issues at the same time (phase taken from the 100 MHz real-time counter),
//       against bursts per workgroup and against trickling stores;
//   (2) what a line-pass-shaped kernel on G1 workgroups and a customer-pass-shaped kernel (R sibling
//       workgroups per stream) on G2 workgroups cost when they run at the same time on two streams.
//   hipcc -O3 --offload-arch=gfx950 -o tools/overlap_probe tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

constexpr int kTile = 256;
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ntload4(const int* p) {
  const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
  return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int fold(int4 a, int4 b, int4 c) {
  return (a.x ^ b.y) + (a.z ^ c.w) + (b.x ^ c.y) + (a.w + b.w + c.x) + (a.y ^ b.z ^ c.z);
}

// Line-pass shape: every wave owns a contiguous run of tiles, one tile ahead, 8 B per lane per tile staged in
// LDS (cap tiles per wave) and written out
//   FLUSH 0: every tile (no staging)
//   FLUSH 1: when the staging is full (per wave)
//   FLUSH 2: when the real-time counter enters a new period of 2^shift ticks (10 ns each), or when full
//   FLUSH 3: as 2, and the whole workgroup flushes together behind a barrier (uniform trip counts)
//   FLUSH 4: every `cap` tiles, whole workgroup behind a barrier
template <int FLUSH>
__global__ void __launch_bounds__(1024) k_lines_like(const int* __restrict__ a, const int* __restrict__ b,
                                                     const int* __restrict__ c, long long n_tiles,
                                                     long long* __restrict__ out, int cap, int shift, int* sink) {
  extern __shared__ long long s_buf[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long long n_waves = (long long)gridDim.x * wpb;
  const long long g = (long long)blockIdx.x * wpb + wave;
  const long long per = (n_tiles + n_waves - 1) / n_waves;
  const long long tb = g * per;
  long long te = tb + per;
  long long* my = s_buf + (size_t)wave * cap * 64;
  int acc = 0, held = 0;
  long long first = tb;
  unsigned long long phase = __builtin_amdgcn_s_memrealtime() >> shift;
  long long off = (tb < n_tiles ? tb : 0) * kTile + lane * 4;
  int4 xa = ntload4(a + off), xb = ntload4(b + off), xc = ntload4(c + off);
  for (long long t = tb; t < te; ++t) {
    long long tn = t + 1; if (tn >= n_tiles) tn = n_tiles - 1;
    const long long offn = tn * kTile + lane * 4;
    const int4 ya = ntload4(a + offn), yb = ntload4(b + offn), yc = ntload4(c + offn);
    const int f = fold(xa, xb, xc);
    acc += f;
    if (FLUSH == 0) { if (t < n_tiles) out[t * 64 + lane] = f; }
    else { my[held * 64 + lane] = f; ++held; }
    xa = ya; xb = yb; xc = yc;
    bool go = false;
    if (FLUSH == 1) go = held == cap;
    if (FLUSH == 2 || FLUSH == 3) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime() >> shift;
      go = held == cap || now != phase;
      phase = now;
    }
    if (FLUSH == 4) go = held == cap;
    if (FLUSH == 3) go = __syncthreads_or(go);
    if (FLUSH == 4 && go) __syncthreads();
    if (go) {
      for (int i = 0; i < held; ++i) { const long long tc = first + i; if (tc < n_tiles) out[tc * 64 + lane] = my[i * 64 + lane]; }
      first += held; held = 0;
    }
  }
  if (FLUSH != 0) for (int i = 0; i < held; ++i) { const long long tc = first + i; if (tc < n_tiles) out[tc * 64 + lane] = my[i * 64 + lane]; }
  if (acc == 0x7fffffff) *sink = acc;
}

// Round-robin tiles: tile t belongs to wave t mod (waves of the grid); 8 B per lane per tile written
//   MODE 0: by the wave itself, right away, at out[t * 64 + ragged + lane] (ragged != 0: the 512 bytes of a
//           tile start in the middle of a cache line, as the order totals of a real tile do)
//   MODE 1: staged in LDS by the 16 waves of the workgroup (their tiles are consecutive), and after a
//           barrier written by the whole workgroup as one 8 KB piece; two staging buffers take turns
template <int MODE>
__global__ void __launch_bounds__(1024) k_tiles_like(const int* __restrict__ a, const int* __restrict__ b,
                                                     const int* __restrict__ c, long long n_tiles,
                                                     long long* __restrict__ out, int ragged, int* sink) {
  __shared__ long long s_buf[2][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long long n_waves = (long long)gridDim.x * wpb;
  const long long g = (long long)blockIdx.x * wpb + wave;
  const long long rounds = (n_tiles + n_waves - 1) / n_waves;
  int acc = 0;
  long long t = g;
  long long off = (t < n_tiles ? t : 0) * kTile + lane * 4;
  int4 xa = ntload4(a + off), xb = ntload4(b + off), xc = ntload4(c + off);
  for (long long r = 0; r < rounds; ++r, t += n_waves) {
    long long tn = t + n_waves; if (tn >= n_tiles) tn = n_tiles - 1;
    const long long offn = tn * kTile + lane * 4;
    const int4 ya = ntload4(a + offn), yb = ntload4(b + offn), yc = ntload4(c + offn);
    const int f = fold(xa, xb, xc);
    acc += f;
    if (MODE == 0) {
      if (t < n_tiles) out[t * 64 + ragged + lane] = f;
    } else {
      s_buf[r & 1][wave * 64 + lane] = f;
      __syncthreads();
      const long long t0 = t - wave;                       // the workgroup's first tile of this round
      const long long e = t0 * 64 + ragged + threadIdx.x;
      if (t0 + wave < n_tiles) out[e] = s_buf[r & 1][threadIdx.x];
    }
    xa = ya; xb = yb; xc = yc;
  }
  if (acc == 0x7fffffff) *sink = acc;
}

template <int MODE>
void run_tiles(const char* name, int ragged, const int* a, const int* b, const int* c, long long n, long long* out, int* sink) {
  auto kern = k_tiles_like<MODE>;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, a, b, c, n / kTile, out, ragged, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, a, b, c, n / kTile, out, ragged, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  printf("%-52s ragged %d: %.3f ms  %.2f TB/s\n", name, ragged, ms, 14.0 * n / ms / 1e9);
  fflush(stdout);
}

// Customer-pass shape: R sibling workgroups (same blockIdx % 8) read the same stream of orders (4 B + 8 B per
// order), waves of a workgroup take tiles round-robin, DEPTH tiles in flight.
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_sib(const int* __restrict__ c, const long long* __restrict__ tot,
                                              long long n_tiles, int R, int n_streams, int* sink) {
  extern __shared__ long long s_buf[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = x + 8 * (j / R);
  const long long per = (n_tiles + n_streams - 1) / n_streams;
  const long long t0 = s * per;
  long long t1 = t0 + per; if (t1 > n_tiles) t1 = n_tiles;
  long long t = t0 + wave;
  if (t >= t1) return;
  const long long last = t + ((t1 - 1 - t) / wpb) * wpb;
  int4 rc[DEPTH]; longlong2 ra[DEPTH], rb[DEPTH];
  auto ld = [&](int d, long long tt) {
    if (tt > last) tt = last;
    rc[d] = *reinterpret_cast<const int4*>(c + tt * kTile + lane * 4);
    ra[d] = *reinterpret_cast<const longlong2*>(tot + tt * kTile + lane * 4);
    rb[d] = *reinterpret_cast<const longlong2*>(tot + tt * kTile + lane * 4 + 2);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ld(d, t + (long long)d * wpb);
  int acc = 0;
  for (; t <= last; t += (long long)DEPTH * wpb) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int4 cc = rc[d]; const longlong2 a2 = ra[d], b2 = rb[d];
      ld(d, t + (long long)(d + DEPTH) * wpb);
      acc += cc.x ^ cc.y ^ cc.z ^ cc.w ^ (int)a2.x ^ (int)a2.y ^ (int)b2.x ^ (int)b2.y;
    }
  }
  if (acc == 0x7fffffff) { *sink = acc; s_buf[threadIdx.x] = acc; }
}

static float timed(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

template <int FLUSH>
void run_lines(const char* name, int blocks, int cap, int shift, const int* a, const int* b, const int* c, long long n,
               long long* out, int* sink) {
  auto kern = k_lines_like<FLUSH>;
  const size_t lds = 144 * 1024;           // one workgroup per CU, as the real kernel
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, a, b, c, n / kTile, out, cap, shift, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, a, b, c, n / kTile, out, cap, shift, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  const float ms = timed(e0, e1) / 10;
  printf("%-44s blocks %3d cap %2d shift %2d: %.3f ms  %.2f TB/s\n", name, blocks, cap, shift, ms, 14.0 * n / ms / 1e9);
  fflush(stdout);
}

void run_overlap(int g1, int r, int g2, const int* a, const int* b, const int* c, long long n, long long* out,
                 const long long* tot2, int* sink) {
  auto k1 = k_lines_like<4>;
  auto k2 = k_sib<4>;
  const size_t lds1 = 144 * 1024, lds2 = 133 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2, f1, f2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&f2));
  const long long m = n / 4;
  const int n_streams = g2 / r;
  float best_all = 1e9f, k1ms = 0, k2ms = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s1));
    CK(hipStreamWaitEvent(s2, e0, 0));
    CK(hipEventRecord(f1, s1));
    hipLaunchKernelGGL(k1, dim3(g1), dim3(1024), lds1, s1, a, b, c, n / kTile, out, 2, 10, sink);
    CK(hipEventRecord(e1, s1));
    if (g2 > 0) {
      CK(hipEventRecord(f2, s2));
      hipLaunchKernelGGL(k2, dim3(g2), dim3(1024), lds2, s2, a, tot2, m / kTile, r, n_streams, sink);
      CK(hipEventRecord(e2, s2));
      CK(hipStreamWaitEvent(s1, e2, 0));
    }
    CK(hipEventRecord(e2, s1));
    CK(hipEventSynchronize(e2));
    const float all = timed(e0, e2);
    if (rep >= 1 && all < best_all) { best_all = all; k1ms = timed(f1, e1); k2ms = 0; }
  }
  (void)k2ms;
  printf("overlap: lines-like on %3d WGs + siblings R=%d on %3d WGs: %.3f ms in all (lines-like alone %.3f ms)\n", g1, r, g2,
         best_all, k1ms);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 100000000ll;
  int *a, *b, *c, *sink;
  long long *out, *tot2;
  CK(hipMalloc(&a, n * 4 + 4096)); CK(hipMalloc(&b, n * 4 + 4096)); CK(hipMalloc(&c, n * 4 + 4096));
  CK(hipMalloc(&out, n / 4 * 8 + 4096)); CK(hipMalloc(&tot2, n / 4 * 8 + 4096)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4)); CK(hipMemset(tot2, 3, n / 4 * 8));
  if (argc > 2) {
    for (int rg : {0, 3}) { run_tiles<0>("round-robin tiles, each wave stores its tile", rg, a, b, c, n, out, sink);
                            run_tiles<1>("round-robin tiles, workgroup stores 8 KB per round", rg, a, b, c, n, out, sink); }
    run_lines<4>("run per wave, workgroup burst every 2 tiles", 256, 2, 0, a, b, c, n, out, sink);
    return 0;
  }
  run_lines<0>("store every tile", 256, 1, 0, a, b, c, n, out, sink);
  for (int cap : {2, 4, 8}) run_lines<1>("flush when full, per wave", 256, cap, 0, a, b, c, n, out, sink);
  for (int cap : {2, 4, 8}) run_lines<4>("flush every cap tiles, workgroup barrier", 256, cap, 0, a, b, c, n, out, sink);
  for (int shift : {8, 9, 10, 11}) run_lines<2>("flush on clock period, per wave", 256, 8, shift, a, b, c, n, out, sink);
  for (int shift : {8, 9, 10, 11}) run_lines<3>("flush on clock period, workgroup", 256, 8, shift, a, b, c, n, out, sink);
  for (int g1 : {256, 240, 224, 208, 192, 176, 160}) run_lines<4>("fewer workgroups", g1, 2, 0, a, b, c, n, out, sink);
  run_overlap(256, 3, 0, a, b, c, n, out, tot2, sink);
  run_overlap(208, 3, 48, a, b, c, n, out, tot2, sink);
  run_overlap(192, 3, 48, a, b, c, n, out, tot2, sink);
  run_overlap(184, 3, 72, a, b, c, n, out, tot2, sink);
  run_overlap(160, 3, 96, a, b, c, n, out, tot2, sink);
  run_overlap(184, 1, 72, a, b, c, n, out, tot2, sink);
  return 0;
}
