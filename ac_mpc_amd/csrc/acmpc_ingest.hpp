// acmpc_ingest.hpp — gfx950 device code of the `detalle` text ingest: SQL-dump / CSV bytes in HBM ->
// the three int32 columns (order_id, product_idx, qty) the aggregation kernels read.
//
// The text shapes are the ones the reference writes: the INSERT of process/confirmcompra.php:36
// (through library/consulSQL.php:29-30, all three values quoted), the phpMyAdmin dump style of
// store.sql:88-89,125-128 (header line, one "(v, 'v', v)," row per line) and plain comma / tab
// separated rows.  The grammar is spelled out in include/acmpc.h.
//
// Two passes over the text (byte work, HBM-bandwidth bound; no MFMA):
//   k_ingest_count   every 16 KiB chunk counts the data rows that START in it
//   k_ingest_scan    exclusive prefix of the chunk counts (one workgroup)
//   k_ingest_parse   every chunk is staged in LDS (plus 128 bytes of lookahead); each thread finds
//                    the row starts in its 64 bytes, ranks them with a block scan, parses them from
//                    LDS, looks the product code up in a hash dictionary and writes the columns
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acmpc {

constexpr int kIngThreads = 256;
constexpr int kIngPerThread = 64;                       // bytes of text per thread
constexpr int kIngChunk = kIngThreads * kIngPerThread;  // 16 KiB per workgroup iteration
constexpr int kIngLook = 128;                           // lookahead kept in LDS; longest row is 120 bytes
constexpr int kIngMaxRow = 120;
constexpr uint32_t kStIngestMalformed = 8u, kStIngestCapacity = 16u;

struct IngestArgs {
  const unsigned char* text;
  int64_t n_bytes;
  int64_t n_chunks;
  long long* chunk_rows;              // [n_chunks + 1]: counts, then exclusive offsets; last = total
  const unsigned long long* dict_hash;   // open-addressing dictionary of the product codes
  const int32_t* dict_idx;               // -1 = empty slot
  uint32_t dict_mask;
  const unsigned long long* codes;    // [P][4]: the codes, 32 bytes each, NUL padded
  int32_t* order_id;
  int32_t* product_idx;
  int32_t* qty;
  int64_t capacity;
  long long* n_rows;                  // device scalar: data rows found
  unsigned long long* n_malformed;    // device scalar
  uint32_t* status;
};

__device__ __forceinline__ bool ing_digit(uint32_t c) { return c - '0' < 10u; }

// Is the row that starts with bytes (c0, c1) a data row?
__device__ __forceinline__ bool ing_data_row(uint32_t c0, uint32_t c1) {
  return ing_digit(c0) || c0 == '(' || c0 == '\'' || c0 == '"' || (c0 == '-' && ing_digit(c1));
}

// Bit 7 of every byte of w that equals '\n'.
__device__ __forceinline__ uint32_t ing_newlines(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  return (x - 0x01010101u) & ~x & 0x80808080u;
}

// Calls f(offset) for every data-row start among the 64 text bytes held in w[16] (offset 0..63 is
// relative to the thread's first byte).  prev is the byte before the thread's range ('\n' at the
// start of the text), next2 holds the two bytes after it (0 past the end of the text).
template <typename F>
__device__ __forceinline__ void ing_row_starts(const uint32_t (&w)[16], uint32_t prev, uint32_t next2, int64_t base,
                                               int64_t n_bytes, F f) {
  // a row starts at byte k iff byte k-1 is '\n'; shift the newline mask by one byte
  uint32_t carry = prev == '\n' ? 0x80u : 0u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t nl = ing_newlines(w[j]);
    uint32_t starts = (nl << 8) | carry;             // bit 7 of byte b set: a row starts at byte 4j+b
    carry = nl >> 24;
    while (starts) {
      const int b = (__ffs((int)starts) - 1) >> 3;
      starts &= starts - 1u;
      const int k = 4 * j + b;
      if (base + k >= n_bytes) continue;
      // the row's first two bytes: inside w, or next2 at the very end of the range
      const uint32_t lo = j < 15 ? w[j + 1 < 16 ? j + 1 : 15] : next2;
      const unsigned long long pair = ((unsigned long long)lo << 32) | w[j];
      const uint32_t c0 = (uint32_t)(pair >> (8 * b)) & 0xffu;
      const uint32_t c1 = (uint32_t)(pair >> (8 * b + 8)) & 0xffu;
      if (ing_data_row(c0, base + k + 1 < n_bytes ? c1 : 0u)) f(k);
    }
  }
}

// Bit k set: a line starts at byte k of the thread's 64 bytes (byte k-1 is '\n', or the text starts
// there).  Lines are not classified here.
__device__ __forceinline__ unsigned long long ing_line_start_mask(const uint32_t (&w)[16], uint32_t prev) {
  unsigned long long mask = 0;
  uint32_t carry = prev == '\n' ? 0x80u : 0u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t nl = ing_newlines(w[j]);
    const uint32_t st = (nl << 8) | carry;
    carry = nl >> 24;
    const uint32_t nib = ((st >> 7) & 1u) | ((st >> 14) & 2u) | ((st >> 21) & 4u) | ((st >> 28) & 8u);
    mask |= (unsigned long long)nib << (4 * j);
  }
  return mask;
}

// Load the thread's 64 bytes (zero past the end of the text) and its neighbours.
__device__ __forceinline__ void ing_load64(const unsigned char* text, int64_t n_bytes, int64_t base, uint32_t (&w)[16],
                                           uint32_t& prev, uint32_t& next2) {
  if (base + kIngPerThread <= n_bytes) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 v = *reinterpret_cast<const uint4*>(text + base + 16 * j);
      w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int64_t p = base + 4 * j + b;
        if (p < n_bytes) v |= (uint32_t)text[p] << (8 * b);
      }
      w[j] = v;
    }
  }
  prev = base == 0 ? (uint32_t)'\n' : (base - 1 < n_bytes ? (uint32_t)text[base - 1] : 0u);
  uint32_t n0 = base + kIngPerThread < n_bytes ? text[base + kIngPerThread] : 0u;
  uint32_t n1 = base + kIngPerThread + 1 < n_bytes ? text[base + kIngPerThread + 1] : 0u;
  next2 = n0 | (n1 << 8);
}

__global__ void __launch_bounds__(kIngThreads) k_ingest_count(IngestArgs a) {
  __shared__ int s_w[kIngThreads / 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    const int64_t base = c * kIngChunk + (int64_t)tid * kIngPerThread;
    int cnt = 0;
    if (base < a.n_bytes) {
      uint32_t w[16], prev, next2;
      ing_load64(a.text, a.n_bytes, base, w, prev, next2);
      ing_row_starts(w, prev, next2, base, a.n_bytes, [&](int) { ++cnt; });
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) s_w[wave] = cnt;
    __syncthreads();
    if (tid == 0) a.chunk_rows[c] = (long long)s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
}

// Hash of a product code given as its 32 NUL-padded bytes in four little-endian 64-bit words.
__host__ __device__ __forceinline__ unsigned long long ing_code_hash(const unsigned long long (&k)[4]) {
  unsigned long long h = k[0] * 0x9E3779B97F4A7C15ull;
  h = (h ^ (h >> 29) ^ k[1]) * 0xBF58476D1CE4E5B9ull;
  h = (h ^ (h >> 32) ^ k[2]) * 0x94D049BB133111EBull;
  h = (h ^ (h >> 29) ^ k[3]) * 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 31);
}

// Parser state over the LDS copy of the text.  The staging code writes a '\n' right after the last
// staged byte, and the parser never moves past a '\n', so no bounds check is needed per byte.
struct IngCursor {
  const unsigned char* s;
  int p;
  __device__ __forceinline__ uint32_t peek() const { return (uint32_t)s[p]; }
};

__device__ __forceinline__ void ing_skip_sp(IngCursor& c) { while (c.peek() == ' ') ++c.p; }

__device__ __forceinline__ bool ing_int(IngCursor& c, int32_t& out) {
  uint32_t q = 0;
  uint32_t ch = c.peek();
  if (ch == '\'' || ch == '"') { q = ch; ++c.p; ch = c.peek(); }
  bool neg = false;
  if (ch == '-') { neg = true; ++c.p; ch = c.peek(); }
  int nd = 0;
  long long v = 0;
  while (ing_digit(ch)) {
    v = v * 10 + (long long)(ch - '0');
    ++c.p;
    if (++nd > 10) return false;
    ch = c.peek();
  }
  if (nd == 0) return false;
  if (neg) v = -v;
  if (v < -2147483648ll || v > 2147483647ll) return false;
  if (q) { if (ch != q) return false; ++c.p; }
  out = (int32_t)v;
  return true;
}

__device__ __forceinline__ bool ing_sep(IngCursor& c) {
  const uint32_t ch = c.peek();
  if (ch == ',' || ch == '\t') { ++c.p; return true; }
  return false;
}

// Parse one data row starting at c.p.  Returns false when malformed.
__device__ __forceinline__ bool ing_row(IngCursor& c, const IngestArgs& a, int32_t& oid, int32_t& pidx, int32_t& qty) {
  const int start = c.p;
  if (c.peek() == '(') ++c.p;
  ing_skip_sp(c);
  if (!ing_int(c, oid)) return false;
  ing_skip_sp(c);
  if (!ing_sep(c)) return false;
  ing_skip_sp(c);
  // product code: up to 30 bytes, packed into four 64-bit words (NUL padded) and hashed (FNV-1a)
  unsigned long long k[4] = {0, 0, 0, 0};
  int len = 0;
  uint32_t ch = c.peek();
  auto take = [&](uint32_t b) {
    if (len < 32) k[len >> 3] |= (unsigned long long)b << (8 * (len & 7));
    ++len;
  };
  if (ch == '\'' || ch == '"') {
    const uint32_t qt = ch;
    ++c.p;
    for (ch = c.peek(); ch != qt && ch != '\n'; ch = c.peek()) { take(ch); ++c.p; }
    if (ch != qt) return false;
    ++c.p;
    if (c.peek() == qt) return false;               // doubled quote: an escape this grammar does not take
  } else {
    for (; ch != ',' && ch != '\t' && ch != ' ' && ch != '\r' && ch != ')' && ch != '\n'; ch = c.peek()) { take(ch); ++c.p; }
  }
  if (len > 30) return false;
  ing_skip_sp(c);
  if (!ing_sep(c)) return false;
  ing_skip_sp(c);
  if (!ing_int(c, qty)) return false;
  ing_skip_sp(c);
  if (c.peek() == ')') ++c.p;
  ch = c.peek();
  if (ch == ',' || ch == ';') ++c.p;
  if (c.peek() == '\r') ++c.p;
  if (c.peek() != '\n') return false;
  if (c.p - start > kIngMaxRow) return false;
  // dictionary lookup: hash of the four packed words, then the 32 code bytes must be identical
  pidx = -1;
  const unsigned long long h = ing_code_hash(k);
  uint32_t slot = (uint32_t)h & a.dict_mask;
  for (;;) {
    const int32_t cand = a.dict_idx[slot];
    if (cand < 0) break;
    if (a.dict_hash[slot] == h) {
      const unsigned long long* cc = a.codes + (size_t)cand * 4;
      if (cc[0] == k[0] && cc[1] == k[1] && cc[2] == k[2] && cc[3] == k[3]) { pidx = cand; break; }
    }
    slot = (slot + 1u) & a.dict_mask;
  }
  return true;
}

__global__ void __launch_bounds__(kIngThreads) k_ingest_parse(IngestArgs a) {
  __shared__ __align__(16) unsigned char s_text[kIngChunk + kIngLook + 16];
  // start offsets of the chunk's data rows, in text order (a data row takes at least two bytes)
  __shared__ unsigned short s_row[kIngChunk / 2];
  __shared__ int s_w[kIngThreads / 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long bad = 0;
  for (int64_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    const int64_t cbase = c * kIngChunk;
    // stage chunk + lookahead (16-byte pieces; the text buffer is padded by the host side)
    const int64_t avail = a.n_bytes - cbase;
    const int want = (int)(avail < kIngChunk + kIngLook ? avail : kIngChunk + kIngLook);
    for (int o = tid * 16; o < want; o += kIngThreads * 16) {
      if (o + 16 <= want) {
        *reinterpret_cast<uint4*>(s_text + o) = *reinterpret_cast<const uint4*>(a.text + cbase + o);
      } else {
        for (int b = o; b < want; ++b) s_text[b] = a.text[cbase + b];
      }
    }
    if (tid == 0) s_text[want] = (unsigned char)'\n';     // sentinel: the parser stops at a newline
    __syncthreads();
    const int64_t base = cbase + (int64_t)tid * kIngPerThread;
    // line starts in this thread's 64 bytes, then the data rows among them (first one or two bytes)
    const int o = tid * kIngPerThread;
    unsigned long long dmask = 0;
    if (base < a.n_bytes) {
      uint32_t w[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (o + 16 * j + 16 <= want) v = *reinterpret_cast<const uint4*>(s_text + o + 16 * j);
        else {
          uint32_t t4[4] = {0, 0, 0, 0};
          for (int b = 0; b < 16; ++b) if (o + 16 * j + b < want) t4[b >> 2] |= (uint32_t)s_text[o + 16 * j + b] << (8 * (b & 3));
          v = make_uint4(t4[0], t4[1], t4[2], t4[3]);
        }
        w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
      }
      const uint32_t prev = base == 0 ? (uint32_t)'\n' : (tid == 0 ? (uint32_t)a.text[base - 1] : (uint32_t)s_text[o - 1]);
      unsigned long long smask = ing_line_start_mask(w, prev);
      const int64_t left = a.n_bytes - base;                 // bytes of text from this thread's first byte on
      if (left < 64) smask &= (1ull << left) - 1ull;
      for (unsigned long long m = smask; m; m &= m - 1ull) {
        const int k = __ffsll((long long)m) - 1;
        const uint32_t c0 = s_text[o + k];
        const uint32_t c1 = k + 1 < left ? (uint32_t)s_text[o + k + 1] : 0u;
        if (ing_data_row(c0, c1)) dmask |= 1ull << k;
      }
    }
    const int cnt = __popcll(dmask);
    // rank of this thread's first row inside the chunk
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int woff = 0, n_r = 0;
#pragma unroll
    for (int k = 0; k < kIngThreads / 64; ++k) { if (k < wave) woff += s_w[k]; n_r += s_w[k]; }
    // hand the rows out one per thread: every lane of a wave then parses a row of its own, and
    // consecutive lanes write consecutive output rows
    int r = woff + incl - cnt;
    for (unsigned long long m = dmask; m; m &= m - 1ull, ++r) s_row[r] = (unsigned short)(o + __ffsll((long long)m) - 1);
    __syncthreads();
    const long long rbase = a.chunk_rows[c];
    for (r = tid; r < n_r; r += kIngThreads) {
      IngCursor cur;
      cur.s = s_text;
      cur.p = (int)s_row[r];
      int32_t oid = 0, pidx = -1, qty = 0;
      const bool ok = ing_row(cur, a, oid, pidx, qty);
      if (!ok) { ++bad; oid = 0; pidx = -1; qty = 0; }
      const long long row = rbase + r;
      if (row < a.capacity) { a.order_id[row] = oid; a.product_idx[row] = pidx; a.qty[row] = qty; }
    }
    __syncthreads();
  }
  if (bad) {
    atomicAdd(a.n_malformed, bad);
    atomicOr(a.status, kStIngestMalformed);
  }
}

}  // namespace acmpc
