// stream_probe — developer microbenchmark: how fast can three int32 columns be streamed from HBM
// under different wave->address mappings and prefetch depths?  Sets the ceiling for k_lines.
//   hipcc -O3 --offload-arch=gfx950 -o tools/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

// MODE 0: every wave owns a contiguous run of tiles (k_lines mapping); DEPTH tiles in flight.
// MODE 1: every workgroup owns a contiguous run; its waves take tiles round-robin.
// MODE 2: grid-stride over tiles (wave-granular), all waves of the chip interleaved.
template <int MODE, int DEPTH, int WRITE>
__global__ void __launch_bounds__(1024) k_probe(const int* __restrict__ a, const int* __restrict__ b,
                                                const int* __restrict__ c, long long n_tiles,
                                                long long* __restrict__ out, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long long n_waves = (long long)gridDim.x * wpb;
  const long long g = (long long)blockIdx.x * wpb + wave;
  long long t_begin, t_end, t_step;
  if (MODE == 0) {
    const long long per = (n_tiles + n_waves - 1) / n_waves;
    t_begin = g * per; t_end = t_begin + per; t_step = 1;
  } else if (MODE == 1) {
    const long long per = (n_tiles + gridDim.x - 1) / gridDim.x;
    t_begin = (long long)blockIdx.x * per + wave; t_end = (long long)(blockIdx.x + 1) * per; t_step = wpb;
  } else if (MODE == 2) {
    t_begin = g; t_end = n_tiles; t_step = n_waves;
  } else {
    t_begin = 0; t_end = 0; t_step = 1;
  }
  if (t_end > n_tiles) t_end = n_tiles;
  int acc = 0;
  if (MODE >= 200) {
    constexpr int B = (MODE >= 200 && MODE < 220) ? MODE - 200 : 1;
    __shared__ long long s_buf[16 * B * 64];
    long long* my = s_buf + wave * B * 64;
    const long long per = (n_tiles + n_waves - 1) / n_waves;
    const long long tb = g * per;
    long long te = tb + per; if (te > n_tiles) te = n_tiles;
    if (WRITE == 13) te = tb + per;     // all waves run the same trip count (barriers inside)
    long long off = (tb < n_tiles ? tb : 0) * kTile + lane * 4;
    int4 xa = ntload4(a + off), xb = ntload4(b + off), xc = ntload4(c + off);
    for (long long t0 = tb; t0 < te; t0 += B) {
#pragma unroll 1
      for (int i = 0; i < B; ++i) {
        const long long tc = t0 + i;
        long long tn = tc + 1; if (tn >= n_tiles) tn = n_tiles - 1;
        const long long offn = tn * kTile + lane * 4;
        const int4 ya = ntload4(a + offn), yb = ntload4(b + offn), yc = ntload4(c + offn);
        const int f = fold(xa, xb, xc);
        acc += f;
        my[i * 64 + lane] = f;
        xa = ya; xb = yb; xc = yc;
      }
      if (WRITE == 13) __syncthreads();
      for (int i = 0; i < B; ++i) { const long long tc = t0 + i; if (tc < n_tiles && tc < tb + per) out[tc * 64 + lane] = my[i * 64 + lane]; }
      if (WRITE == 13) __syncthreads();
    }
    if (acc == 0x7fffffff) *sink = acc;
    return;
  }
  if (MODE >= 100) {
    constexpr int S = MODE >= 100 ? MODE - 100 : 1;                 // tiles per chunk; chunk k belongs to wave k % n_waves
    const long long n_chunks = (n_tiles + S - 1) / S;
    for (long long k = g; k < n_chunks; k += n_waves) {
      const long long tb = k * S;
      long long te = tb + S; if (te > n_tiles) te = n_tiles;
      long long off = tb * kTile + lane * 4;
      int4 xa = ntload4(a + off), xb = ntload4(b + off), xc = ntload4(c + off);
      for (long long tc = tb; tc < te; ++tc) {
        const long long offn = (tc + 1 < te ? tc + 1 : tc) * kTile + lane * 4;
        const int4 ya = ntload4(a + offn), yb = ntload4(b + offn), yc = ntload4(c + offn);
        const int f = fold(xa, xb, xc);
        acc += f;
        if (WRITE == 9) out[tc * 64 + lane] = f;
        xa = ya; xb = yb; xc = yc;
      }
    }
    if (acc == 0x7fffffff) *sink = acc;
    return;
  }
  int4 ra[DEPTH], rb[DEPTH], rc[DEPTH];
  long long t = t_begin;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    long long tt = t + d * t_step;
    if (tt >= t_end) tt = t_begin < t_end ? t_begin : 0;
    const long long off = tt * kTile + lane * 4;
    ra[d] = *reinterpret_cast<const int4*>(a + off);
    rb[d] = *reinterpret_cast<const int4*>(b + off);
    rc[d] = *reinterpret_cast<const int4*>(c + off);
  }
  for (; t < t_end; t += DEPTH * t_step) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long long tc = t + d * t_step;
      const int4 ca = ra[d], cb = rb[d], cc = rc[d];
      long long tn = tc + DEPTH * t_step;
      if (tn >= t_end) tn = t_begin;
      const long long off = tn * kTile + lane * 4;
      if (WRITE >= 9) {
        ra[d] = ntload4(a + off);
        rb[d] = ntload4(b + off);
        rc[d] = ntload4(c + off);
      } else {
        ra[d] = *reinterpret_cast<const int4*>(a + off);
        rb[d] = *reinterpret_cast<const int4*>(b + off);
        rc[d] = *reinterpret_cast<const int4*>(c + off);
      }
      if (tc < t_end) {
        const int f = fold(ca, cb, cc);
        acc += f;
        if (WRITE == 1) out[tc * 64 + lane] = f;       // 8 B per 4 lines, like the order totals
        if (WRITE == 2) __builtin_nontemporal_store((long long)f, &out[tc * 64 + lane]);
        if (WRITE == 3) __hip_atomic_store(&out[tc * 64 + lane], (long long)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (WRITE == 4) out[((tc * 64) & 0x3ffff) + lane] = f;          // 2 MiB window: stays in cache
        if (WRITE == 5) acc += (int)__builtin_nontemporal_load(&out[tc * 64 + lane]);   // read instead of write
        if (WRITE == 6) acc += (int)out[tc * 64 + lane];
        if (WRITE == 9) out[tc * 64 + lane] = f;
        if (WRITE == 10) __builtin_nontemporal_store((long long)f, &out[tc * 64 + lane]);
        if (WRITE == 11) { /* nt loads, no store */ }
        if (WRITE == 7) { if (tc & 1) { longlong2 w; w.x = f; w.y = acc; *reinterpret_cast<longlong2*>(&out[(tc - 1) * 64 + lane * 2]) = w; } }   // 16 B/lane every 2nd tile
        if (WRITE == 8) { if ((tc & 3) == 3) { longlong2 w; w.x = f; w.y = acc; *reinterpret_cast<longlong2*>(&out[(tc - 3) * 64 + lane * 4]) = w; w.x = acc; *reinterpret_cast<longlong2*>(&out[(tc - 3) * 64 + lane * 4 + 2]) = w; } }   // 32 B/lane every 4th tile
      }
    }
  }
  if (WRITE == 12) {   // phase 2: all writes at the end
    for (long long tc = t_begin; tc < t_end; tc += t_step) out[tc * 64 + lane] = acc + (int)tc;
  }
  if (acc == 0x7fffffff) *sink = acc;
}


// Four input streams: three int32 columns and one int64 column (the line-price column of the lifecycle
// path), k_lines mapping (a contiguous run of tiles per wave), one tile ahead, 8 B written per 4 lines.
__global__ void __launch_bounds__(1024) k_probe4(const int* __restrict__ a, const int* __restrict__ b,
                                                 const int* __restrict__ c, const int* __restrict__ d8,
                                                 long long n_tiles, long long* __restrict__ out, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long long n_waves = (long long)gridDim.x * wpb;
  const long long g = (long long)blockIdx.x * wpb + wave;
  const long long per = (n_tiles + n_waves - 1) / n_waves;
  const long long tb = g * per;
  long long te = tb + per; if (te > n_tiles) te = n_tiles;
  if (tb >= te) return;
  int acc = 0;
  long long off = tb * kTile + lane * 4;
  int4 xa = ntload4(a + off), xb = ntload4(b + off), xc = ntload4(c + off);
  int4 xd = ntload4(d8 + tb * kTile * 2 + lane * 4), xe = ntload4(d8 + tb * kTile * 2 + 256 + lane * 4);
  for (long long t = tb; t < te; ++t) {
    const long long tn = t + 1 < te ? t + 1 : t;
    const long long offn = tn * kTile + lane * 4;
    const int4 ya = ntload4(a + offn), yb = ntload4(b + offn), yc = ntload4(c + offn);
    const int4 yd = ntload4(d8 + tn * kTile * 2 + lane * 4), ye = ntload4(d8 + tn * kTile * 2 + 256 + lane * 4);
    const int f = fold(xa, xb, xc) + (xd.x ^ xd.z) + (xe.y ^ xe.w);
    acc += f;
    out[t * 64 + lane] = f;
    xa = ya; xb = yb; xc = yc; xd = yd; xe = ye;
  }
  if (acc == 0x7fffffff) *sink = acc;
}

__global__ void __launch_bounds__(256) k_fill(int4* dst, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) dst[i] = make_int4(1, 2, 3, (int)i);
}
__global__ void __launch_bounds__(256) k_copy(const int4* __restrict__ src, int4* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_copy8(const int2* __restrict__ src, int2* __restrict__ dst, long long n2) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// k_customers read pattern: R sibling workgroups (same blockIdx % 8) read the same stream of orders
// (4 B + 8 B per order), waves of a workgroup take tiles round-robin, DEPTH tiles in flight.
template <int DEPTH, bool NT>
__global__ void __launch_bounds__(1024) k_sib(const int* __restrict__ c, const long long* __restrict__ tot,
                                              long long n_tiles, int R, int n_streams, int* sink) {
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
    if (NT) {
      rc[d] = ntload4(c + tt * kTile + lane * 4);
      const int4 u = ntload4(reinterpret_cast<const int*>(tot + tt * kTile + lane * 4));
      const int4 w = ntload4(reinterpret_cast<const int*>(tot + tt * kTile + lane * 4 + 2));
      ra[d].x = u.x; ra[d].y = u.z; rb[d].x = w.x; rb[d].y = w.z;
    } else {
      rc[d] = *reinterpret_cast<const int4*>(c + tt * kTile + lane * 4);
      ra[d] = *reinterpret_cast<const longlong2*>(tot + tt * kTile + lane * 4);
      rb[d] = *reinterpret_cast<const longlong2*>(tot + tt * kTile + lane * 4 + 2);
    }
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
  if (acc == 0x7fffffff) *sink = acc;
}

template <int DEPTH, bool NT>
void run_sib(int R, int blocks, int threads, const int* c, const long long* tot, long long m, int* sink) {
  const long long n_tiles = m / kTile;
  const int n_streams = blocks / R;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_sib<DEPTH, NT>), dim3(blocks), dim3(threads), 0, 0, c, tot, n_tiles, R, n_streams, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_sib<DEPTH, NT>), dim3(blocks), dim3(threads), 0, 0, c, tot, n_tiles, R, n_streams, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  printf("siblings R=%d depth=%d nt=%d blocks %d x %d: %.3f ms  (%.2f TB/s of unique bytes, %.2f TB/s through L1)\n", R, DEPTH, (int)NT,
         blocks, threads, ms, 12.0 * m / ms / 1e9, 12.0 * m * R / ms / 1e9);
}

template <int MODE, int DEPTH, int WRITE>
void run(const char* name, int blocks, int threads, const int* a, const int* b, const int* c, long long n,
         long long* out, int* sink) {
  const long long n_tiles = n / kTile;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_probe<MODE, DEPTH, WRITE>), dim3(blocks), dim3(threads), 0, 0, a, b, c, n_tiles, out, sink);
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe<MODE, DEPTH, WRITE>), dim3(blocks), dim3(threads), 0, 0, a, b, c, n_tiles, out, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = 12.0 * n + ((WRITE && WRITE != 4 && WRITE != 11) ? 2.0 * n : 0.0);
  printf("%-34s blocks %4d x %4d  %.3f ms  %.2f TB/s\n", name, blocks, threads, ms, bytes / ms / 1e9);
}

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 100000000ll;
  int *a, *b, *c, *sink;
  long long* out;
  CK(hipMalloc(&a, n * 4 + 4096)); CK(hipMalloc(&b, n * 4 + 4096)); CK(hipMalloc(&c, n * 4 + 4096));
  CK(hipMalloc(&out, n / 4 * 8 + 4096)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4));
  {
    const long long m = n / 4;
    for (int R : {1, 2, 4}) { run_sib<1, false>(R, 256, 1024, a, out, m, sink); run_sib<4, false>(R, 256, 1024, a, out, m, sink); run_sib<4, true>(R, 256, 1024, a, out, m, sink); }
    run_sib<4, false>(3, 240, 1024, a, out, m, sink);
    run_sib<1, false>(3, 240, 1024, a, out, m, sink);
    run_sib<2, false>(4, 256, 512, a, out, m, sink);
    run_sib<4, false>(4, 512, 512, a, out, m, sink);
    run_sib<4, false>(8, 512, 512, a, out, m, sink);
  }
  if (argc > 2 && argv[2][0] == '4') {
    int* d8;
    CK(hipMalloc(&d8, n * 8 + 4096)); CK(hipMemset(d8, 5, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_probe4, dim3(256), dim3(1024), 0, 0, a, b, c, d8, n / kTile, out, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("four streams (3 x int32 + int64 per line) + 8 B per 4 lines written: %.3f ms  %.2f TB/s\n", ms, 22.0 * n / ms / 1e9);
    }
    return 0;
  }
  if (argc > 2) return 0;
#define RUN(M, D, W, BL, TH) run<M, D, W>(#M "/" #D " write-mode " #W, BL, TH, a, b, c, n, out, sink);
  RUN(0, 1, 0, 256, 1024) RUN(0, 1, 1, 256, 1024) RUN(0, 1, 2, 256, 1024) RUN(0, 1, 3, 256, 1024)
  RUN(0, 1, 4, 256, 1024) RUN(0, 1, 5, 256, 1024) RUN(0, 1, 6, 256, 1024)
  RUN(0, 2, 1, 256, 1024) RUN(0, 2, 2, 256, 1024) RUN(0, 2, 3, 256, 1024) RUN(0, 2, 4, 256, 1024) RUN(0, 2, 6, 256, 1024)
  RUN(2, 1, 1, 256, 1024) RUN(2, 1, 2, 256, 1024) RUN(2, 1, 3, 256, 1024) RUN(2, 2, 2, 2048, 256)
  RUN(208, 1, 9, 256, 1024) RUN(216, 1, 9, 256, 1024) RUN(208, 1, 13, 256, 1024) RUN(216, 1, 13, 256, 1024) RUN(204, 1, 13, 256, 1024) RUN(202, 1, 13, 256, 1024) RUN(204, 1, 9, 256, 1024)
  RUN(104, 1, 9, 256, 1024) RUN(116, 1, 9, 256, 1024) RUN(164, 1, 9, 256, 1024) RUN(116, 1, 11, 256, 1024) RUN(164, 1, 11, 256, 1024) RUN(2, 2, 9, 256, 1024) RUN(2, 1, 9, 512, 1024) RUN(2, 1, 11, 256, 1024)
  RUN(1, 1, 9, 256, 1024) RUN(1, 1, 11, 256, 1024) RUN(1, 2, 9, 256, 1024) RUN(0, 1, 9, 256, 1024) RUN(0, 1, 10, 256, 1024) RUN(0, 1, 11, 256, 1024) RUN(0, 1, 12, 256, 1024) RUN(0, 2, 9, 256, 1024) RUN(0, 2, 10, 256, 1024) RUN(2, 1, 9, 256, 1024)
  // write-only and copy calibration: 16 B per lane
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long words = 3 * n / 4;    // int4 elements in 1.2 GB... use arrays a (src) b (dst)
    for (int variant = 0; variant < 3; ++variant) {
      float ms;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) {
          if (variant == 0) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (int4*)a, n / 4);
          if (variant == 1) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const int4*)a, (int4*)b, n / 4);
          if (variant == 2) hipLaunchKernelGGL(k_copy8, dim3(2048), dim3(256), 0, 0, (const int2*)a, (int2*)b, n / 2);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      ms /= 10;
      const double bytes = variant == 0 ? 4.0 * n : 8.0 * n;
      printf("%s  %.3f ms  %.2f TB/s\n", variant == 0 ? "fill 16B/lane (write only)" : variant == 1 ? "copy 16B/lane" : "copy 8B/lane", ms, bytes / ms / 1e9);
    }
    (void)words;
  }
  return 0;
}
