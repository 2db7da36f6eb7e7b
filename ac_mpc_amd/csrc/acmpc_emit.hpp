// Text emit: per-order totals and per-product stock deltas -> "update ..." statements or CSV rows,
// written on the device.  Shapes follow consultasSQL::UpdateSQL (library/consulSQL.php:43-44) and the
// quoting of process/updatePedido.php:11 in the reference.
//
// Three launches, like the ingest: row lengths per chunk, a scan of the chunk sizes, then the write
// pass: every thread formats one row into LDS at its scanned offset and the workgroup copies the batch
// to HBM with 16-byte stores (LDS and HBM positions are kept congruent modulo 16).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acmpc {

constexpr int kEmitThreads = 256;
constexpr int kEmitChunkRows = 4096;      // rows per chunk = 16 batches of one row per thread
constexpr int kEmitMaxRow = 104;          // longest row of any kind and style, rounded up
constexpr uint32_t kStEmitCapacity = 32u;

constexpr int kEmitTotals = 0, kEmitStock = 1;
constexpr int kEmitSql = 0, kEmitCsv = 1, kEmitDisplay = 2;

struct EmitArgs {
  const long long* value;               // order_total_cents[n_rows] or product_qty[n_rows]
  int64_t n_rows;
  int64_t n_chunks;
  long long order_id_base;
  const unsigned long long* codes;      // stock rows: four 64-bit words per product
  long long* chunk_bytes;               // [n_chunks + 1]: sizes, then exclusive offsets
  unsigned char* text;
  int64_t capacity;
  long long* n_bytes;
  uint32_t* status;
};

__device__ __forceinline__ int emit_digits32(uint32_t v) {
  return 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) +
         (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}

__device__ __forceinline__ int emit_digits(unsigned long long v) {
  if (v <= 0xffffffffull) return emit_digits32((uint32_t)v);
  int n = 10;
  unsigned long long t = 10000000000ull;
  while (n < 20 && v >= t) { ++n; t *= 10ull; }    // t wraps after 10^19, where n is already 20
  return n;
}

// Writes the decimal digits of v so that the last one lands at s[end - 1].
__device__ __forceinline__ void emit_put_dec(unsigned char* s, int end, unsigned long long v) {
  while (v > 0xffffffffull) {
    const unsigned long long q = v / 10ull;
    s[--end] = (unsigned char)('0' + (uint32_t)(v - q * 10ull));
    v = q;
  }
  uint32_t w = (uint32_t)v;
  do {
    const uint32_t q = w / 10u;
    s[--end] = (unsigned char)('0' + (w - q * 10u));
    w = q;
  } while (w);
}

template <int N>
__device__ __forceinline__ int emit_put_lit(unsigned char* s, int p, const char (&lit)[N]) {
#pragma unroll
  for (int i = 0; i < N - 1; ++i) s[p + i] = (unsigned char)lit[i];
  return p + N - 1;
}

__device__ __forceinline__ unsigned long long emit_abs(long long v) {
  return v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
}

// [-]units.cc of an amount in cents
__device__ __forceinline__ int emit_amount_len(long long cents) {
  const unsigned long long m = emit_abs(cents);
  if (m <= 0xffffffffull)                                    // (the common case without a 64-bit division)
    return (cents < 0) + emit_digits32(__umulhi((uint32_t)m, 1374389535u) >> 5) + 3;
  return (cents < 0) + emit_digits(m / 100ull) + 3;
}

__device__ __forceinline__ int emit_put_amount(unsigned char* s, int p, long long cents) {
  const unsigned long long m = emit_abs(cents);
  const unsigned long long units = m / 100ull;
  const uint32_t frac = (uint32_t)(m - units * 100ull);
  if (cents < 0) s[p++] = '-';
  const int nd = emit_digits(units);
  emit_put_dec(s, p + nd, units);
  p += nd;
  s[p] = '.';
  s[p + 1] = (unsigned char)('0' + frac / 10u);
  s[p + 2] = (unsigned char)('0' + frac % 10u);
  return p + 3;
}

// [-]u,uuu,uuu.cc: number_format($suma, 2) of process/carrito.php:19 (comma between groups of three
// digits, '.' before the cents) for an amount held exactly in cents
__device__ __forceinline__ int emit_grouped_len(long long cents) {
  const int nd = emit_digits(emit_abs(cents) / 100ull);
  return (cents < 0) + nd + (nd - 1) / 3 + 3;
}

__device__ __forceinline__ int emit_put_grouped(unsigned char* s, int p, long long cents) {
  const unsigned long long m = emit_abs(cents);
  unsigned long long units = m / 100ull;
  const uint32_t frac = (uint32_t)(m - units * 100ull);
  if (cents < 0) s[p++] = '-';
  const int nd = emit_digits(units);
  int end = p + nd + (nd - 1) / 3;                       // one past the last digit of the units
  const int after = end;
  for (int k = 0; k < nd; ++k) {
    if (k && k % 3 == 0) s[--end] = ',';
    const unsigned long long q = units / 10ull;
    s[--end] = (unsigned char)('0' + (uint32_t)(units - q * 10ull));
    units = q;
  }
  s[after] = '.';
  s[after + 1] = (unsigned char)('0' + frac / 10u);
  s[after + 2] = (unsigned char)('0' + frac % 10u);
  return after + 3;
}

__device__ __forceinline__ int emit_int_len(long long v) { return (v < 0) + emit_digits(emit_abs(v)); }

__device__ __forceinline__ int emit_put_int(unsigned char* s, int p, long long v) {
  if (v < 0) s[p++] = '-';
  const unsigned long long m = emit_abs(v);
  const int nd = emit_digits(m);
  emit_put_dec(s, p + nd, m);
  return p + nd;
}

// Bytes of a NUL-padded code held in four little-endian words (codes have no inner NUL).
__device__ __forceinline__ int emit_code_len(const unsigned long long* k) {
  int len = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned long long x = k[w];
    if (x) len = 8 * w + 8 - (__clzll((long long)x) >> 3);
  }
  return len;
}

__device__ __forceinline__ int emit_put_code(unsigned char* s, int p, const unsigned long long* k, int len) {
  for (int i = 0; i < len; ++i) s[p + i] = (unsigned char)(k[i >> 3] >> (8 * (i & 7)));
  return p + len;
}

// Row formatters: len() and put() must agree byte for byte.
template <int Kind, int Style>
struct EmitRow;

template <>
struct EmitRow<kEmitTotals, kEmitSql> {
  static __device__ __forceinline__ int len(const EmitArgs& a, int64_t r) {
    return 29 + emit_amount_len(a.value[r]) + 19 + emit_int_len(a.order_id_base + r) + 3;
  }
  static __device__ __forceinline__ void put(const EmitArgs& a, int64_t r, unsigned char* s, int p) {
    p = emit_put_lit(s, p, "update venta set TotalPagar='");
    p = emit_put_amount(s, p, a.value[r]);
    p = emit_put_lit(s, p, "' where NumPedido='");
    p = emit_put_int(s, p, a.order_id_base + r);
    emit_put_lit(s, p, "';\n");
  }
};

template <>
struct EmitRow<kEmitTotals, kEmitCsv> {
  static __device__ __forceinline__ int len(const EmitArgs& a, int64_t r) {
    return emit_int_len(a.order_id_base + r) + 1 + emit_amount_len(a.value[r]) + 1;
  }
  static __device__ __forceinline__ void put(const EmitArgs& a, int64_t r, unsigned char* s, int p) {
    p = emit_put_int(s, p, a.order_id_base + r);
    s[p++] = ',';
    p = emit_put_amount(s, p, a.value[r]);
    s[p] = '\n';
  }
};

// The fragment process/carrito.php:19 echoes for a cart, one line per order:
//   <tr><td>Subtotal</td><td>$2,200.00</td></tr>
template <>
struct EmitRow<kEmitTotals, kEmitDisplay> {
  static __device__ __forceinline__ int len(const EmitArgs& a, int64_t r) { return 26 + emit_grouped_len(a.value[r]) + 11; }
  static __device__ __forceinline__ void put(const EmitArgs& a, int64_t r, unsigned char* s, int p) {
    p = emit_put_lit(s, p, "<tr><td>Subtotal</td><td>$");
    p = emit_put_grouped(s, p, a.value[r]);
    emit_put_lit(s, p, "</td></tr>\n");
  }
};

template <>
struct EmitRow<kEmitStock, kEmitSql> {
  static __device__ __forceinline__ int len(const EmitArgs& a, int64_t r) {
    const long long d = a.value[r];
    if (d == 0) return 0;
    return 32 + emit_digits(emit_abs(d)) + 19 + emit_code_len(a.codes + 4 * r) + 3;
  }
  static __device__ __forceinline__ void put(const EmitArgs& a, int64_t r, unsigned char* s, int p) {
    const long long d = a.value[r];
    if (d == 0) return;
    unsigned long long k[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) k[w] = a.codes[4 * r + w];
    p = emit_put_lit(s, p, "update producto set Stock=Stock");
    s[p++] = d < 0 ? '+' : '-';
    const unsigned long long m = emit_abs(d);
    const int nd = emit_digits(m);
    emit_put_dec(s, p + nd, m);
    p += nd;
    p = emit_put_lit(s, p, " where CodigoProd='");
    p = emit_put_code(s, p, k, emit_code_len(k));
    emit_put_lit(s, p, "';\n");
  }
};

template <>
struct EmitRow<kEmitStock, kEmitCsv> {
  static __device__ __forceinline__ int len(const EmitArgs& a, int64_t r) {
    const long long d = a.value[r];
    if (d == 0) return 0;
    return emit_code_len(a.codes + 4 * r) + 1 + emit_int_len(d) + 1;
  }
  static __device__ __forceinline__ void put(const EmitArgs& a, int64_t r, unsigned char* s, int p) {
    const long long d = a.value[r];
    if (d == 0) return;
    unsigned long long k[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) k[w] = a.codes[4 * r + w];
    p = emit_put_code(s, p, k, emit_code_len(k));
    s[p++] = ',';
    p = emit_put_int(s, p, d);
    s[p] = '\n';
  }
};

// ---------------------------------------------------------------------------------------------
// Word-wide formatting of the common totals row: an amount of 0 .. 2^32 - 1 cents and an order id below 10^8.
// The row is put together as three byte strings with a fixed layout each (the units and the id are made
// left-aligned by scaling them to eight digits first, so nothing in a string moves with a digit count; only
// where the second and third string start does), and every string is OR-ed into the zeroed LDS batch a word
// at a time, shifted to its byte position: a word shared by two neighbouring rows takes both.  Any other row
// (negative or larger amount, longer id) sends its whole wave through the byte-wise formatter above.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t emit_w4(const char* s, int i) {
  return (uint32_t)(unsigned char)s[i] | ((uint32_t)(unsigned char)s[i + 1] << 8) |
         ((uint32_t)(unsigned char)s[i + 2] << 16) | ((uint32_t)(unsigned char)s[i + 3] << 24);
}

// x < 10000 -> its four decimal digits as ASCII, most significant in the lowest byte.
__device__ __forceinline__ uint32_t emit_ascii4(uint32_t x) {
  const uint32_t a = (x * 5243u) >> 19;                     // x / 100 (exact below 10000)
  const uint32_t y2 = a | ((x - a * 100u) << 16);           // two numbers below 100 side by side
  const uint32_t t2 = ((y2 * 205u) >> 11) & 0x000F000Fu;    // their tens (y * 205 >> 11 == y / 10 below 100)
  const uint32_t o2 = y2 - t2 * 10u;                        // their ones
  return (t2 | (o2 << 8)) + 0x30303030u;
}

// The low n bytes of a word (n <= 0: none, n >= 4: all).
__device__ __forceinline__ uint32_t emit_keep(int n) {
  n = min(max(n, 0), 4);
  return (uint32_t)((1ull << (8 * n)) - 1ull);
}

// v < 10^8 -> its nd decimal digits left-aligned in two words (bytes past the digits are zero).
struct EmitDigits { uint32_t w0, w1; int nd; };
__device__ __forceinline__ EmitDigits emit_left_digits(uint32_t v) {
  int nd = 8;
  if (v < 10000u) { v *= 10000u; nd -= 4; }
  if (v < 1000000u) { v *= 100u; nd -= 2; }
  if (v < 10000000u) { v *= 10u; nd -= 1; }                 // (0 ends here as 0 with one digit)
  const uint32_t hi = __umulhi(v, 3518437209u) >> 13;       // v / 10000 (exact below 2^32)
  const uint32_t lo = v - hi * 10000u;
  EmitDigits d;
  d.nd = nd;
  d.w0 = emit_ascii4(hi) & emit_keep(nd);
  d.w1 = emit_ascii4(lo) & emit_keep(nd - 4);
  return d;
}

// OR the byte string held in w[0..N) (zero past its end) into the LDS text at byte `pos`: N + 1 words.
// (`s` has at least four bytes of slack in front of position 0.)
template <int N>
__device__ __forceinline__ void emit_or_words(unsigned char* s, int pos, const uint32_t (&w)[N]) {
  // word i of the shifted string is {w[i], w[i-1]} >> 8 * ((4 - pos) & 3); with pos a multiple of 4 that is
  // w[i-1], one word late: the words then start one word earlier
  uint32_t* dst = reinterpret_cast<uint32_t*>(s) + ((pos + 3) >> 2) - 1;
  const uint32_t k = (uint32_t)(4 - (pos & 3)) & 3u;
  uint32_t prev = 0u;
#pragma unroll
  for (int i = 0; i <= N; ++i) {
    const uint32_t cur = i < N ? w[i] : 0u;
    const uint32_t t = __builtin_amdgcn_alignbyte(cur, prev, k);
    __hip_atomic_fetch_or(dst + i, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    prev = cur;
  }
}

// What the fast formatter keeps of a row between the length scan and the write.
struct EmitFastRow { EmitDigits units, id; uint32_t frac; };

__device__ __forceinline__ bool emit_fast_ok(long long cents, long long id) {
  return (unsigned long long)cents <= 0xffffffffull && (unsigned long long)id < 100000000ull;
}
__device__ __forceinline__ EmitFastRow emit_fast_prepare(uint32_t cents, uint32_t id) {
  const uint32_t units = __umulhi(cents, 1374389535u) >> 5;   // cents / 100 (exact below 2^32)
  EmitFastRow f;
  f.frac = cents - units * 100u;
  f.units = emit_left_digits(units);
  f.id = emit_left_digits(id);
  return f;
}
// ".cc" as three bytes in the low end of a word
__device__ __forceinline__ uint32_t emit_frac3(uint32_t frac) {
  const uint32_t t = (frac * 205u) >> 11;
  return 0x0030302Eu + (t << 8) + ((frac - t * 10u) << 16);
}

template <int Style>
struct EmitFast;

template <>
struct EmitFast<kEmitSql> {
  static constexpr char kLit1[] = "update venta set TotalPagar=";       // 28 bytes; the quote goes with the digits
  static constexpr char kLit2[] = " where NumPedido";                   // 16 bytes between "'" and "='"
  static __device__ __forceinline__ int len(const EmitFastRow& f) { return 29 + f.units.nd + 3 + 19 + f.id.nd + 3; }
  static __device__ __forceinline__ void put(const EmitFastRow& f, unsigned char* s, int p) {
    uint32_t a[10];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = emit_w4(kLit1, 4 * i);
    a[7] = 0x27u | (f.units.w0 << 8);
    a[8] = __builtin_amdgcn_alignbyte(f.units.w1, f.units.w0, 3);
    a[9] = f.units.w1 >> 24;
    emit_or_words<10>(s, p, a);
    uint32_t b[8];
    b[0] = emit_frac3(f.frac) | 0x27000000u;                              // .cc'
#pragma unroll
    for (int i = 0; i < 4; ++i) b[1 + i] = emit_w4(kLit2, 4 * i);
    b[5] = 0x0000273Du | (f.id.w0 << 16);                                 // ='
    b[6] = (f.id.w0 >> 16) | (f.id.w1 << 16);
    b[7] = f.id.w1 >> 16;
    const int pb = p + 29 + f.units.nd;
    emit_or_words<8>(s, pb, b);
    const uint32_t c[1] = {0x000A3B27u};                                  // ';\n
    emit_or_words<1>(s, pb + 22 + f.id.nd, c);
  }
};

template <>
struct EmitFast<kEmitCsv> {
  static __device__ __forceinline__ int len(const EmitFastRow& f) { return f.id.nd + 1 + f.units.nd + 4; }
  static __device__ __forceinline__ void put(const EmitFastRow& f, unsigned char* s, int p) {
    const uint32_t a[2] = {f.id.w0, f.id.w1};
    emit_or_words<2>(s, p, a);
    const uint32_t b[3] = {0x2Cu | (f.units.w0 << 8), __builtin_amdgcn_alignbyte(f.units.w1, f.units.w0, 3), f.units.w1 >> 24};
    const int pb = p + f.id.nd;
    emit_or_words<3>(s, pb, b);
    const uint32_t c[1] = {emit_frac3(f.frac) | 0x0A000000u};            // .cc\n
    emit_or_words<1>(s, pb + 1 + f.units.nd, c);
  }
};

// Inclusive scan of one int per thread over the workgroup; returns the exclusive prefix and the total.
__device__ __forceinline__ int emit_block_scan(int v, int* s_w, int& total) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int off = 0;
  total = 0;
#pragma unroll
  for (int k = 0; k < kEmitThreads / 64; ++k) { if (k < wave) off += s_w[k]; total += s_w[k]; }
  __syncthreads();
  return off + incl - v;
}

template <int Kind, int Style>
__global__ void __launch_bounds__(kEmitThreads) k_emit_count(EmitArgs a) {
  __shared__ long long s_w[kEmitThreads / 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    const int64_t r0 = c * kEmitChunkRows;
    long long sum = 0;
    if (r0 + kEmitChunkRows <= a.n_rows) {                   // uniform: every chunk but the last
#pragma unroll
      for (int b = 0; b < kEmitChunkRows / kEmitThreads; ++b) sum += EmitRow<Kind, Style>::len(a, r0 + b * kEmitThreads + tid);
    } else {
      for (int b = 0; b < kEmitChunkRows / kEmitThreads; ++b) {
        const int64_t r = r0 + b * kEmitThreads + tid;
        if (r < a.n_rows) sum += EmitRow<Kind, Style>::len(a, r);
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) s_w[wave] = sum;
    __syncthreads();
    if (tid == 0) a.chunk_bytes[c] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
}

// Exclusive prefix sum of v[0..n) in place; v[n] and *total receive the sum.  One workgroup.
struct ScanArgs {
  long long* v;
  int64_t n;
  long long* total;
  int64_t capacity;
  uint32_t* status;
  uint32_t over_flag;        // raised when the sum exceeds capacity
};

__global__ void __launch_bounds__(1024) k_chunk_scan(ScanArgs a) {
  constexpr int kStrip = 8;                     // consecutive values per thread and round
  __shared__ long long s_w[16];
  __shared__ long long s_carry;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < a.n; i0 += 1024 * kStrip) {
    const int64_t i = i0 + (int64_t)tid * kStrip;
    long long v[kStrip];
    long long sum = 0;
#pragma unroll
    for (int k = 0; k < kStrip; ++k) { v[k] = i + k < a.n ? a.v[i + k] : 0; sum += v[k]; }
    long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    long long off = s_carry;
    for (int k = 0; k < wave; ++k) off += s_w[k];
    long long run = off + incl - sum;
#pragma unroll
    for (int k = 0; k < kStrip; ++k) { if (i + k < a.n) a.v[i + k] = run; run += v[k]; }
    __syncthreads();
    if (tid == 1023) s_carry = off + incl;
    __syncthreads();
  }
  if (tid == 0) {
    a.v[a.n] = s_carry;
    *a.total = s_carry;
    if (s_carry > a.capacity) atomicOr(a.status, a.over_flag);
  }
}

// Rows a thread formats per batch (consecutive rows, so that its text is one piece), and the longest row: short
// rows are taken several at a time, or the four barriers of a batch would cost more than its bytes.
template <int Kind, int Style> struct EmitBatch { static constexpr int kRows = 1, kMaxRow = kEmitMaxRow; };
template <> struct EmitBatch<kEmitTotals, kEmitCsv> { static constexpr int kRows = 4, kMaxRow = 36; };   // "-2147483648,-92233720368547758.08\n"

template <int Kind, int Style>
__global__ void __launch_bounds__(kEmitThreads) k_emit_write(EmitArgs a) {
  // [16 bytes in front][a batch of rows, from byte gofs % 16 on][slack]: words OR-ed in may reach a few bytes
  // past either end of a row.  Zero between batches: the copy-out clears what it has read.
  constexpr int R = EmitBatch<Kind, Style>::kRows;
  constexpr int kBatchRows = kEmitThreads * R;
  static_assert(kEmitChunkRows % kBatchRows == 0, "whole batches per chunk");
  constexpr int kBuf = 16 + 16 + kBatchRows * EmitBatch<Kind, Style>::kMaxRow + 48;
  __shared__ __align__(16) unsigned char s_buf[kBuf];
  __shared__ int s_w[kEmitThreads / 64];
  unsigned char* s_out = s_buf + 16;
  constexpr bool kFast = Kind == kEmitTotals && (Style == kEmitSql || Style == kEmitCsv);
  using Fast = EmitFast<(Kind == kEmitTotals && Style == kEmitCsv) ? kEmitCsv : kEmitSql>;
  const int tid = (int)threadIdx.x;
  for (int i = tid * 16; i < kBuf; i += kEmitThreads * 16) *reinterpret_cast<uint4*>(s_buf + i) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  for (int64_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    long long gofs = a.chunk_bytes[c];
    for (int b = 0; b < kEmitChunkRows / kBatchRows; ++b) {
      const int64_t r0 = c * kEmitChunkRows + (int64_t)b * kBatchRows + (int64_t)tid * R;     // this thread's first row
      int len[R];
      int mine = 0;
      bool fast = false;                                  // uniform in the wave
      EmitFastRow f[R];
      if (kFast) {
        long long v[R];
        bool ok = true;
        if (R == 4 && r0 + R <= a.n_rows && (reinterpret_cast<uintptr_t>(a.value) & 15u) == 0) {
          // the thread's four values as two 16-byte loads (r0 is a multiple of four; a column that is not 16-byte
          // aligned takes the 8-byte loads below):
          // half as many requests as four 8-byte loads that each touch every 64-byte piece of the wave's 2 KB
          const longlong2 x = *reinterpret_cast<const longlong2*>(a.value + r0);
          const longlong2 y = *reinterpret_cast<const longlong2*>(a.value + r0 + 2);
          v[0] = x.x; v[1 % R] = x.y; v[2 % R] = y.x; v[3 % R] = y.y;
#pragma unroll
          for (int j = 0; j < R; ++j) ok = ok && emit_fast_ok(v[j], a.order_id_base + r0 + j);
        } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const bool in = r0 + j < a.n_rows;
          v[j] = in ? a.value[r0 + j] : 0;
          ok = ok && emit_fast_ok(v[j], in ? a.order_id_base + r0 + j : 0);
        }
        }
        fast = __ballot(!ok) == 0ull;
        if (fast) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            f[j] = emit_fast_prepare((uint32_t)v[j], (uint32_t)(a.order_id_base + r0 + j));
            len[j] = r0 + j < a.n_rows ? Fast::len(f[j]) : 0;
          }
        }
      }
      if (!fast) {
#pragma unroll
        for (int j = 0; j < R; ++j) len[j] = r0 + j < a.n_rows ? EmitRow<Kind, Style>::len(a, r0 + j) : 0;
      }
#pragma unroll
      for (int j = 0; j < R; ++j) mine += len[j];
      int total;
      const int off = emit_block_scan(mine, s_w, total);
      if (total == 0) continue;                       // uniform: every thread sees the same total
      const int shift = (int)(gofs & 15);
      int p = shift + off;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (len[j]) {
          if (kFast && fast) Fast::put(f[j], s_out, p);
          else EmitRow<Kind, Style>::put(a, r0 + j, s_out, p);
        }
        p += len[j];
      }
      __syncthreads();
      // LDS byte i sits at text[g0 + i]; g0 is a multiple of 16
      const long long g0 = gofs - shift;
      const int end = shift + total;
      for (int w = tid; w * 16 < end; w += kEmitThreads) {
        const int lo = w * 16, hi = lo + 16;
        if (lo >= shift && hi <= end && g0 + hi <= a.capacity) {
          *reinterpret_cast<uint4*>(a.text + g0 + lo) = *reinterpret_cast<const uint4*>(s_out + lo);
        } else {
          const int from = lo < shift ? shift : lo, to = hi < end ? hi : end;
          for (int i = from; i < to; ++i) if (g0 + i < a.capacity) a.text[g0 + i] = s_out[i];
        }
        *reinterpret_cast<uint4*>(s_out + lo) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
      gofs += total;
    }
  }
}

}  // namespace acmpc
