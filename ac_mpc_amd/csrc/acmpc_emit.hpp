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
    for (int b = 0; b < kEmitChunkRows / kEmitThreads; ++b) {
      const int64_t r = r0 + b * kEmitThreads + tid;
      if (r < a.n_rows) sum += EmitRow<Kind, Style>::len(a, r);
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

template <int Kind, int Style>
__global__ void __launch_bounds__(kEmitThreads) k_emit_write(EmitArgs a) {
  __shared__ __align__(16) unsigned char s_out[16 + kEmitThreads * kEmitMaxRow + 16];
  __shared__ int s_w[kEmitThreads / 64];
  const int tid = (int)threadIdx.x;
  for (int64_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    long long gofs = a.chunk_bytes[c];
    for (int b = 0; b < kEmitChunkRows / kEmitThreads; ++b) {
      const int64_t r = c * kEmitChunkRows + b * kEmitThreads + tid;
      const int len = r < a.n_rows ? EmitRow<Kind, Style>::len(a, r) : 0;
      int total;
      const int off = emit_block_scan(len, s_w, total);
      if (total == 0) continue;                       // uniform: every thread sees the same total
      const int shift = (int)(gofs & 15);
      if (len) EmitRow<Kind, Style>::put(a, r, s_out, shift + off);
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
      }
      __syncthreads();
      gofs += total;
    }
  }
}

}  // namespace acmpc
