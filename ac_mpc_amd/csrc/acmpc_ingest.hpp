// acmpc_ingest.hpp — gfx950 device code of the `detalle` text ingest: SQL-dump / CSV bytes in HBM ->
// the three int32 columns (order_id, product_idx, qty) the aggregation kernels read.
//
// The text shapes are the ones the reference writes: the INSERT of process/confirmcompra.php:36
// (through library/consulSQL.php:29-30, all three values quoted), the phpMyAdmin dump style of
// store.sql:88-89,125-128 (header line, one "(v, 'v', v)," row per line) and plain comma / tab
// separated rows.  The grammar is spelled out in include/acmpc.h.
//
// One pass over the text (byte work; no MFMA):
//   k_ingest_parse   12 KiB chunks are handed out in order; every chunk is staged in LDS (plus 128 bytes of lookahead); each thread finds
//                    the row starts in its 48 bytes and ranks them with a block scan; the rows are
//                    then handed out one per thread, run through a table-driven automaton, converted,
//                    looked up in a hash dictionary of the product codes and written as columns
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace acmpc {

constexpr int kIngThreads = 256;
constexpr int kIngPerThread = 48;                       // bytes of text per thread (a multiple of 16)
constexpr int kIngChunk = kIngThreads * kIngPerThread;  // 12 KiB per workgroup iteration
constexpr int kIngWords = kIngPerThread / 4;
constexpr int kIngPerTicket = 1;                        // consecutive chunks a workgroup takes at once
constexpr int kIngBlocksPerCu = 5;                      // resident workgroups per CU (LDS and registers allow five)
constexpr int kIngLook = 256;                           // lookahead kept in LDS: a statement's row may start 100 bytes
                                                        // into its line, and the automaton looks at 128 bytes from there
constexpr int kIngStmtReach = 100;                      // farthest a statement's row may start from its line start
constexpr int kIngMaxRow = 120;
constexpr uint32_t kStIngestMalformed = 8u, kStIngestCapacity = 16u;

struct IngSlot;

struct IngestArgs {
  const unsigned char* text;
  int64_t n_bytes;
  int64_t n_chunks;
  unsigned long long* chunk_state;    // [n_chunks] zero on entry: see ing_row_base
  unsigned int* ticket;               // zero on entry: next group of kIngPerTicket chunks to hand out
  const IngSlot* dict;            // perfect-hash dictionary of the product codes: one slot per probe
  const unsigned short* dict_disp;       // displacement of every bucket
  uint32_t dict_mask, dict_seed, dict_bucket_shift;
  const unsigned long long* codes;    // [P][4]: the codes, 32 bytes each, NUL padded (emit kernels)
  const unsigned long long* tables;   // automaton: kIngTableWords 64-bit entries, then the 256 byte classes
  int32_t* order_id;
  int32_t* product_idx;
  int32_t* qty;
  int64_t capacity;
  long long* n_rows;                  // device scalar: data rows found
  unsigned long long* n_malformed;    // device scalar
  uint32_t* status;
  int32_t ablate;                     // developer switch for timing experiments (0 in production)
};

__device__ __forceinline__ bool ing_digit(uint32_t c) { return c - '0' < 10u; }

// Is the row that starts with bytes (c0, c1) a data row?
__device__ __forceinline__ bool ing_data_row(uint32_t c0, uint32_t c1) {
  return ing_digit(c0) || c0 == '(' || c0 == '\'' || c0 == '"' || (c0 == '-' && ing_digit(c1));
}

// The one-line statement of library/consulSQL.php:30 ("insert into detalle (...) VALUES('5', '1', '1')"):
// a line whose first six bytes are "insert" in any letter case and that has, at most kIngStmtReach bytes
// from its start and before its newline, a '(' directly preceded (spaces allowed) by "values" in any
// case.  Returns the offset of the first such '(' from the line start, or -1 (the header line of the dump
// style).  `s` points at the line start in the LDS text, which ends in a sentinel newline.
__device__ __forceinline__ bool ing_lower_is(const unsigned char* s, const char (&word)[7]) {
  bool same = true;
#pragma unroll
  for (int i = 0; i < 6; ++i) same &= (s[i] | 0x20u) == (uint32_t)word[i];
  return same;
}
// Bit 7 of every byte of w that equals the byte replicated in c4.
__device__ __forceinline__ uint32_t ing_bytes_eq(uint32_t w, uint32_t c4) {
  const uint32_t x = w ^ c4;
  return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
// `text` is the (4-byte aligned) LDS text, `pos` the offset of the line start in it.  The line is searched
// a word at a time for '(' and newline; four words are requested together (the reads of one step do not
// depend on each other).
__device__ __forceinline__ int ing_statement_row(const unsigned char* text, int pos) {
  const unsigned char* s = text + pos;
  if (!ing_lower_is(s, "insert")) return -1;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(text);
  const int first = pos + 6, last = pos + kIngStmtReach;            // bytes looked at: [first, last]
  uint32_t ignore = (first & 3) ? ((1u << (8 * (first & 3))) - 1u) : 0u;   // bytes of the first word in front of `first`
  for (int wi = first >> 2; (wi << 2) <= last; wi += 4) {
    uint32_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = w[wi + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t hit = (ing_bytes_eq(x[k], 0x28282828u) | ing_bytes_eq(x[k], 0x0A0A0A0Au)) & ~ignore;
      ignore = 0u;
      while (hit) {
        const int j = ((wi + k) << 2) + ((__ffs((int)hit) - 1) >> 3);     // offset in the text
        hit &= hit - 1u;
        if (j > last || text[j] == '\n') return -1;
        int q = j - 1;
        while (q > pos && text[q] == ' ') --q;
        if (q - 5 >= first && ing_lower_is(text + q - 5, "values")) return j - pos;
      }
    }
  }
  return -1;
}

// Bit 7 of every byte of w that equals '\n'.
__device__ __forceinline__ uint32_t ing_newlines(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;    // exact per byte: no borrow between bytes
}

// Bit k set: a line starts at byte k of the thread's 64 bytes (byte k-1 is '\n', or the text starts
// there).  Lines are not classified here.
__device__ __forceinline__ unsigned long long ing_line_start_mask(const uint32_t (&w)[kIngWords], uint32_t prev) {
  unsigned long long mask = 0;
  uint32_t carry = prev == '\n' ? 0x80u : 0u;
#pragma unroll
  for (int j = 0; j < kIngWords; ++j) {
    const uint32_t nl = ing_newlines(w[j]);
    const uint32_t st = (nl << 8) | carry;
    carry = nl >> 24;
    const uint32_t nib = ((st >> 7) & 1u) | ((st >> 14) & 2u) | ((st >> 21) & 4u) | ((st >> 28) & 8u);
    mask |= (unsigned long long)nib << (4 * j);
  }
  return mask;
}

// ---------------------------------------------------------------------------------------------
// Row base of a chunk without a counting pass: chunks are taken in order from a ticket counter, every
// chunk publishes its row count as soon as it knows it, and looks back over its predecessors' entries
// (512 at a time) until it meets one that already holds an inclusive prefix.  An entry is one 64-bit
// word: the top two bits say empty / own count / inclusive prefix, the rest is the value, so one
// relaxed atomic store publishes both.  A chunk only ever waits for chunks with smaller tickets, which
// were handed to workgroups that are already running.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kIngOwn = 1ull << 62, kIngInclusive = 2ull << 62, kIngValue = (1ull << 62) - 1ull;
constexpr uint32_t kStInternal = 64u;

// Publish the chunk's own row count (one thread).  Done as early as possible: successors can then add
// it up long before this chunk knows its own base.
__device__ __forceinline__ void ing_publish_own(unsigned long long* state, long long c, long long n_r) {
  __hip_atomic_store(&state[c], (c == 0 ? kIngInclusive : kIngOwn) | (unsigned long long)n_r, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// Called by one whole wave, after ing_publish_own.  Returns the number of data rows in chunks [0, c)
// and publishes the chunk's inclusive prefix.
__device__ __forceinline__ long long ing_row_base(unsigned long long* state, long long c, long long n_r, int lane,
                                                  uint32_t* status) {
  if (c == 0) return 0;
  long long excl = 0;
  long long p = c - 1;                                   // nearest predecessor not yet added
  constexpr int kPer = 1;                                // entries per lane and round
  for (uint32_t polls = 0;;) {
    // lane l looks at the entries at distance 8l .. 8l+7 behind p (nearest first)
    unsigned long long v[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const long long idx = p - ((long long)lane * kPer + j);
      v[j] = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : kIngInclusive;                   // in front of chunk 0: nothing
    }
    int fj = kPer;                                       // this lane's nearest inclusive entry
#pragma unroll
    for (int j = kPer - 1; j >= 0; --j) fj = (v[j] >> 62) == 2ull ? j : fj;
    const unsigned long long incl_mask = __ballot(fj < kPer);
    const int first = incl_mask ? __ffsll((long long)incl_mask) - 1 : 64;     // lane holding the nearest one
    // entries nearer than (and including) that one must all be published
    const int upto = lane < first ? kPer - 1 : (lane == first ? fj : -1);
    bool waiting = false;
    long long val = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (j <= upto) {
        waiting |= (v[j] >> 62) == 0ull;
        val += (long long)(v[j] & kIngValue);
      }
    }
    if (__ballot(waiting)) {                             // a predecessor has not published yet
      if (++polls > (1u << 24)) { if (lane == 0) atomicOr(status, kStInternal); break; }   // never expected
      // back off: every poll reads lines that the publishers are about to write, and the memory side
      // serves one address only so often
      if (polls < 4) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
      continue;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
    excl += val;
    if (incl_mask) break;
    p -= 64 * kPer;
  }
  if (lane == 0)
    __hip_atomic_store(&state[c], kIngInclusive | (unsigned long long)(excl + n_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// Two hashes of a product code given as its 32 NUL-padded bytes in eight little-endian 32-bit words.
// `a` is a seeded multiplicative chain over the first four words (codes that differ only past byte 16
// are told apart by b), `b` an independent rotation fold over all eight; the dictionary
// is a perfect hash built on them (hash-and-displace): bucket = top bits of a, and every bucket has a
// displacement d such that slot = (a + d * (b | 1)) & mask is a slot of its own for each of its codes.
struct IngHash { uint32_t a, b; };
__host__ __device__ __forceinline__ uint32_t ing_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ __forceinline__ IngHash ing_code_hash(const uint32_t (&w)[8], uint32_t seed) {
  uint32_t a = 0x811C9DC5u ^ seed;
#pragma unroll
  for (int i = 0; i < 4; ++i) a = (a ^ w[i]) * 0x01000193u;
  a ^= a >> 15;
  a *= 0x85EBCA77u;
  a ^= a >> 13;
  uint32_t b = w[0] ^ ing_rotl(w[1], 5) ^ ing_rotl(w[2], 11) ^ ing_rotl(w[3], 17) ^ ing_rotl(w[4], 23) ^
               ing_rotl(w[5], 29) ^ ing_rotl(w[6], 3) ^ ing_rotl(w[7], 9) ^ (seed * 0x9E3779B1u);
  b *= 0x9E3779B1u;
  b ^= b >> 16;
  IngHash h;
  h.a = a;
  h.b = b;
  return h;
}
__host__ __device__ __forceinline__ uint32_t ing_slot_of(IngHash h, uint32_t disp, uint32_t mask) {
  return (h.a + disp * (h.b | 1u)) & mask;
}

// One dictionary slot, 64 bytes (one piece of a cache line).  The first 16 bytes decide a lookup of a
// code of at most 12 bytes on their own: its three words and `short_idx`, which is the product index
// when the slot's code has no byte past the twelfth and -1 otherwise (a longer code cannot equal a
// shorter one).  Longer codes compare all eight words and take `idx`.  An empty slot holds the
// all-zero code and -1 twice.
struct alignas(64) IngSlot {
  uint32_t head[3];        // code words 0..2
  int32_t short_idx;
  uint32_t tail[5];        // code words 3..7
  int32_t idx;
  int32_t pad[6];
};

// ---------------------------------------------------------------------------------------------
// The row grammar as a table-driven automaton.
//
// A row is consumed one byte per step: class of the byte (13 classes) x state (33 states) -> one 64-bit
// table entry holding the next state and two words of packed increments.  Summed over the row, the
// increments give everything the parser needs to pick the fields out of the text afterwards —
//   word A: digits of the first int, digits of the second int, code bytes, minus signs, row length
//   word B: bytes up to the end of the first int's digits, bytes before the first code byte,
//           bytes up to the end of the second int's digits
// (7 bits each: a row is cut off after 128 bytes, and a field can only reach 128 in a row that never
// arrives at the final state).  The numbers are then converted from their digits with a few
// word-wide operations, the code is cut out with byte-aligned word reads.  No per-byte branches.
// ---------------------------------------------------------------------------------------------
enum : int {
  kcDig, kcQ1, kcQ2, kcMinus, kcSp, kcComma, kcTab, kcSemi, kcRpar, kcLpar, kcCr, kcNl, kcOther, kNumClasses = 16
};
enum : int {
  sStart, sASp, sAQ1, sAQ2, sANeg, sANegQ1, sANegQ2, sADig, sADigQ1, sADigQ2, sAEnd, sBSp, sCQ1, sCQ2, sCQEnd1,
  sCQEnd2, sCBare, sCEnd, sDSp, sDQ1, sDQ2, sDNeg, sDNegQ1, sDNegQ2, sDDig, sDDigQ1, sDDigQ2, sESp, sEPar, sETerm,
  sECr, sDone, sErr, kNumStates
};
constexpr int kIngTableWords = kNumStates * kNumClasses;     // 64-bit entries
constexpr int kIngSteps = 128;                               // bytes of a row the automaton looks at
constexpr int kIngFront = 16;                                // LDS bytes kept in front of the chunk

// word A fields
constexpr int kANd1 = 0, kANd2 = 7, kACl = 14, kANeg1 = 21, kANeg2 = 22, kALen = 23;
// word B fields (bits 21..31 of the entry's low word hold 16 * next state)
constexpr int kBE1 = 0, kBCs = 7, kBE2 = 14, kBState = 21;

inline int ing_class_of(int c) {
  if (c >= '0' && c <= '9') return kcDig;
  switch (c) {
    case '\'': return kcQ1;
    case '"': return kcQ2;
    case '-': return kcMinus;
    case ' ': return kcSp;
    case ',': return kcComma;
    case '\t': return kcTab;
    case ';': return kcSemi;
    case ')': return kcRpar;
    case '(': return kcLpar;
    case '\r': return kcCr;
    case '\n': return kcNl;
    default: return kcOther;
  }
}

// Host: fill the transition table T[kNumStates][kNumClasses] and the byte-class table CT[256].
inline void ing_build_tables(unsigned long long* T, unsigned char* CT) {
  for (int c = 0; c < 256; ++c) CT[c] = (unsigned char)ing_class_of(c);
  int next[kNumStates][kNumClasses];
  for (int s = 0; s < kNumStates; ++s)
    for (int c = 0; c < kNumClasses; ++c) next[s][c] = s == sDone ? sDone : sErr;
  auto int_start = [&](int s, int q1, int q2, int neg, int dig) {        // before an int: [q] ['-'] digits
    next[s][kcSp] = s == sStart ? sASp : s;
    next[s][kcQ1] = q1; next[s][kcQ2] = q2; next[s][kcMinus] = neg; next[s][kcDig] = dig;
  };
  int_start(sStart, sAQ1, sAQ2, sANeg, sADig);
  next[sStart][kcLpar] = sASp;
  int_start(sASp, sAQ1, sAQ2, sANeg, sADig);
  int_start(sDSp, sDQ1, sDQ2, sDNeg, sDDig);
  next[sAQ1][kcMinus] = sANegQ1; next[sAQ1][kcDig] = sADigQ1;
  next[sAQ2][kcMinus] = sANegQ2; next[sAQ2][kcDig] = sADigQ2;
  next[sANeg][kcDig] = sADig; next[sANegQ1][kcDig] = sADigQ1; next[sANegQ2][kcDig] = sADigQ2;
  next[sDQ1][kcMinus] = sDNegQ1; next[sDQ1][kcDig] = sDDigQ1;
  next[sDQ2][kcMinus] = sDNegQ2; next[sDQ2][kcDig] = sDDigQ2;
  next[sDNeg][kcDig] = sDDig; next[sDNegQ1][kcDig] = sDDigQ1; next[sDNegQ2][kcDig] = sDDigQ2;
  // first int: digits, then (closing quote) spaces and the separator
  next[sADig][kcDig] = sADig; next[sADig][kcSp] = sAEnd; next[sADig][kcComma] = sBSp; next[sADig][kcTab] = sBSp;
  next[sADigQ1][kcDig] = sADigQ1; next[sADigQ1][kcQ1] = sAEnd;
  next[sADigQ2][kcDig] = sADigQ2; next[sADigQ2][kcQ2] = sAEnd;
  next[sAEnd][kcSp] = sAEnd; next[sAEnd][kcComma] = sBSp; next[sAEnd][kcTab] = sBSp;
  // code: quoted, or bare up to one of , \t space \r \n )
  for (int c = 0; c < kNumClasses; ++c) {
    const bool stop = c == kcComma || c == kcTab || c == kcSp || c == kcCr || c == kcNl || c == kcRpar;
    if (!stop) { next[sBSp][c] = sCBare; next[sCBare][c] = sCBare; }
    if (c != kcNl) { next[sCQ1][c] = sCQ1; next[sCQ2][c] = sCQ2; }
  }
  next[sBSp][kcSp] = sBSp; next[sBSp][kcQ1] = sCQ1; next[sBSp][kcQ2] = sCQ2;
  next[sBSp][kcComma] = sDSp; next[sBSp][kcTab] = sDSp;                  // empty bare code
  next[sCQ1][kcQ1] = sCQEnd1; next[sCQ2][kcQ2] = sCQEnd2;
  next[sCQEnd1][kcSp] = sCEnd; next[sCQEnd1][kcComma] = sDSp; next[sCQEnd1][kcTab] = sDSp;   // a doubled quote: error
  next[sCQEnd2][kcSp] = sCEnd; next[sCQEnd2][kcComma] = sDSp; next[sCQEnd2][kcTab] = sDSp;
  next[sCBare][kcSp] = sCEnd; next[sCBare][kcComma] = sDSp; next[sCBare][kcTab] = sDSp;
  next[sCEnd][kcSp] = sCEnd; next[sCEnd][kcComma] = sDSp; next[sCEnd][kcTab] = sDSp;
  // second int and the row's tail: spaces, ')', ',' or ';', '\r', '\n'
  auto tail = [&](int s, bool sp, bool par, bool term, bool cr) {
    if (sp) next[s][kcSp] = sESp;
    if (par) next[s][kcRpar] = sEPar;
    if (term) { next[s][kcComma] = sETerm; next[s][kcSemi] = sETerm; }
    if (cr) next[s][kcCr] = sECr;
    next[s][kcNl] = sDone;
  };
  next[sDDig][kcDig] = sDDig; tail(sDDig, true, true, true, true);
  next[sDDigQ1][kcDig] = sDDigQ1; next[sDDigQ1][kcQ1] = sESp;
  next[sDDigQ2][kcDig] = sDDigQ2; next[sDDigQ2][kcQ2] = sESp;
  tail(sESp, true, true, true, true);
  tail(sEPar, false, false, true, true);
  tail(sETerm, false, false, false, true);
  tail(sECr, false, false, false, false);

  auto in_a = [](int s) { return s >= sASp && s <= sADigQ2; };          // inside the first int (prefix or digits)
  auto pre_code = [&](int s) { return s == sStart || in_a(s) || s == sAEnd || s == sBSp; };
  for (int s = 0; s < kNumStates; ++s) {
    for (int c = 0; c < kNumClasses; ++c) {
      const int n = next[s][c];
      unsigned long long a = 0, b = 0;
      if (s != sDone && s != sErr) {
        if (n != sDone) a += 1ull << kALen;                              // every byte of the row but its '\n'
        if (c == kcDig && (n == sADig || n == sADigQ1 || n == sADigQ2)) a += 1ull << kANd1;
        if (c == kcDig && (n == sDDig || n == sDDigQ1 || n == sDDigQ2)) a += 1ull << kANd2;
        if (n == sANeg || n == sANegQ1 || n == sANegQ2) a += 1ull << kANeg1;
        if (n == sDNeg || n == sDNegQ1 || n == sDNegQ2) a += 1ull << kANeg2;
        if (n == sCBare || ((s == sCQ1 || s == sCQ2) && n == s)) a += 1ull << kACl;
        if (in_a(n)) b += 1ull << kBE1;
        if (pre_code(s) && (in_a(n) || n == sAEnd || n == sBSp || n == sCQ1 || n == sCQ2)) b += 1ull << kBCs;
        if (n < sESp) b += 1ull << kBE2;
      }
      b |= (unsigned long long)(n * kNumClasses) << kBState;
      T[s * kNumClasses + c] = (a << 32) | b;
    }
  }
}

// hi:lo shifted right by sh bytes (sh = 0..3), low word.
__device__ __forceinline__ uint32_t ing_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// Four ASCII digits, most significant in the lowest byte (bytes already reduced to 0..9) -> 0..9999.
__device__ __forceinline__ uint32_t ing_four_digits(uint32_t y) {
  uint32_t t = (y << 3) + (y >> 8);
  t += y << 1;                                    // byte 0: 10*d0 + d1, byte 2: 10*d2 + d3 (each < 100)
  return (t & 0xffu) * 100u + ((t >> 16) & 0xffu);
}

// Value of the nd decimal digits (nd <= 4 * W, W = 1..3 words) that end just before byte `pe` of the LDS
// text (pe >= 4 * W).
template <int W>
__device__ __forceinline__ unsigned long long ing_digits_value(const unsigned char* s_text, int pe, int nd) {
  const int p0 = pe - 4 * W;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(s_text) + (p0 >> 2);
  const uint32_t sh = (uint32_t)p0 & 3u;
  uint32_t d[W + 1];
#pragma unroll
  for (int i = 0; i <= W; ++i) d[i] = w[i];
  const int drop = 4 * W - nd;                    // bytes in front of the number (little endian: low bytes first)
  uint32_t v[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const int di = min(max(drop - 4 * i, 0), 4);
    const uint32_t mask = di >= 4 ? 0u : (0xffffffffu << (8 * di));
    v[i] = ing_four_digits(ing_alignbyte(d[i + 1], d[i], sh) & mask & 0x0f0f0f0fu);
  }
  if (W == 1) return v[0];
  if (W == 2) return v[0] * 10000u + v[W - 1];
  return (unsigned long long)v[0] * 100000000ull + (unsigned long long)(v[W > 1 ? 1 : 0] * 10000u + v[W - 1]);
}

// Run the automaton over N rows of this lane at once (they start at bytes p[0..N) of the LDS text).  The
// table look-ups of one row depend on each other; stepping several rows together fills the waits.
template <int N>
__device__ __forceinline__ void ing_scan_rows(const unsigned char* s_text, const unsigned char* s_ct, const uint2* s_T,
                                              const int (&p)[N], uint32_t (&st16)[N], uint32_t (&A)[N], uint32_t (&B)[N]) {
#pragma unroll
  for (int q = 0; q < N; ++q) { st16[q] = sStart * kNumClasses; A[q] = 0; B[q] = 0; }
  for (int j0 = 0; j0 < kIngSteps; j0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const uint32_t cls = s_ct[s_text[p[q] + j0 + j]];
        const uint2 e = s_T[st16[q] + cls];
        B[q] += e.x;
        A[q] += e.y;
        st16[q] = e.x >> kBState;
      }
    }
    bool running = false;
#pragma unroll
    for (int q = 0; q < N; ++q) running |= st16[q] < (uint32_t)(sDone * kNumClasses);
    if (__ballot(running) == 0ull) break;                  // every row of every active lane is done
  }
}

// Turn the automaton's final state and counters for the row at byte `p` into the three columns.
// Returns false when the row is malformed.  W1, W2: words of digits read for the two numbers; NC: words
// of the code that are cut out (the rest are zero).  The caller picks <3, 3, 8>, which covers every row,
// or a cheaper combination when no row of the wave needs more.
template <int W1, int W2, int NC>
__device__ __forceinline__ bool ing_finish_row(const unsigned char* s_text, int p, uint32_t st16, uint32_t A, uint32_t B,
                                               const IngestArgs& a, int32_t& oid, int32_t& pidx, int32_t& qty) {
  const int nd1 = (A >> kANd1) & 127, nd2 = (A >> kANd2) & 127, cl = (A >> kACl) & 127, len = (A >> kALen) & 127;
  const bool neg1 = (A >> kANeg1) & 1u, neg2 = (A >> kANeg2) & 1u;
  const int e1 = (B >> kBE1) & 127, cs = (B >> kBCs) & 127, e2 = (B >> kBE2) & 127;
  bool ok = st16 == (uint32_t)(sDone * kNumClasses) && len <= kIngMaxRow && nd1 <= 10 && nd2 <= 10 && cl <= 30;
  const unsigned long long v1 = ing_digits_value<W1>(s_text, p + e1, min(nd1, W1 == 3 ? 10 : 4 * W1));
  const unsigned long long v2 = ing_digits_value<W2>(s_text, p + e2, min(nd2, W2 == 3 ? 10 : 4 * W2));
  ok = ok && v1 <= (neg1 ? 2147483648ull : 2147483647ull) && v2 <= (neg2 ? 2147483648ull : 2147483647ull);
  oid = (int32_t)(neg1 ? 0u - (uint32_t)v1 : (uint32_t)v1);
  qty = (int32_t)(neg2 ? 0u - (uint32_t)v2 : (uint32_t)v2);
  // the code: cl bytes from p + cs, as four NUL-padded little-endian 64-bit words
  const int pc = p + cs;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(s_text) + (pc >> 2);
  const uint32_t sh = (uint32_t)pc & 3u;
  uint32_t cw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t lo = w[0];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const uint32_t hi = w[i + 1];
    const int keep = min(max(cl - 4 * i, 0), 4);
    const uint32_t mask = keep >= 4 ? 0xffffffffu : ((1u << (8 * keep)) - 1u);
    cw[i] = ing_alignbyte(hi, lo, sh) & mask;
    lo = hi;
  }
  pidx = -1;
  if (ok && !(a.ablate & 2)) {
    // dictionary lookup: bucket displacement (a small, cache-resident table), then the one slot this code
    // can live in; the slot holds the code itself, so equality is decided without another round trip
    const IngHash h = ing_code_hash(cw, a.dict_seed);
    const uint32_t slot = ing_slot_of(h, a.dict_disp[h.a >> a.dict_bucket_shift], a.dict_mask);
    const uint4* sp = reinterpret_cast<const uint4*>(a.dict + slot);
    if (NC <= 3) {
      const uint4 s0 = sp[0];
      if (((s0.x ^ cw[0]) | (s0.y ^ cw[1]) | (s0.z ^ cw[2])) == 0u) pidx = (int32_t)s0.w;
    } else {
      const uint4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
      const uint32_t diff = (s0.x ^ cw[0]) | (s0.y ^ cw[1]) | (s0.z ^ cw[2]) | (s1.x ^ cw[3]) | (s1.y ^ cw[4]) |
                            (s1.z ^ cw[5]) | (s1.w ^ cw[6]) | (s2.x ^ cw[7]);
      if (diff == 0u) pidx = (int32_t)s2.y;
    }
  }
  return ok;
}

__global__ void __launch_bounds__(kIngThreads) __attribute__((amdgpu_waves_per_eu(kIngBlocksPerCu * kIngThreads / 256, kIngBlocksPerCu * kIngThreads / 256))) k_ingest_parse(IngestArgs a) {
  // [16 bytes of slack][chunk][look-ahead][slack]: number and code windows may reach a few bytes outside a row
  __shared__ __align__(16) unsigned char s_text[kIngFront + kIngChunk + kIngLook + 64];
  // start offsets of the chunk's data rows, in text order (a data row takes at least two bytes)
  __shared__ __align__(16) unsigned short s_row[kIngChunk / 2];
  constexpr int kKeep = kIngChunk / 16;                    // parsed rows a chunk can park in LDS
  static_assert(kIngChunk / 2 * 2 >= kKeep * 2 + 3 * kKeep * 4, "parsed columns are parked behind the row list");
  __shared__ __align__(8) uint2 s_T[kIngTableWords];
  __shared__ unsigned char s_ct[256];
  __shared__ int s_w[kIngThreads / 64];
  __shared__ long long s_base;
  __shared__ long long s_ticket;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kIngTableWords; i += kIngThreads) {
    const unsigned long long e = a.tables[i];
    s_T[i] = make_uint2((uint32_t)e, (uint32_t)(e >> 32));
  }
  if (tid < 256) s_ct[tid] = reinterpret_cast<const unsigned char*>(a.tables + kIngTableWords)[tid];
  if (tid < kIngFront) s_text[tid] = 0;
  unsigned char* s_chunk = s_text + kIngFront;
  unsigned long long bad = 0;
  unsigned round = 0;                                      // (timing experiments only)
  int32_t* s_res = reinterpret_cast<int32_t*>(s_row + kKeep);   // parked columns: [oid | pidx | qty], kKeep each

  // The first output row of chunk c (n_r data rows) comes from its predecessors' counts.  One wave looks
  // back; the last chunk also knows the total.
  auto resolve_base = [&](long long c, int n_r) {
    if (wave == 0) {
      const long long excl = (a.ablate & 64) ? c * 400 : ing_row_base(a.chunk_state, c, n_r, lane, a.status);
      if (lane == 0) {
        s_base = excl;
        if (c == a.n_chunks - 1) {
          *a.n_rows = excl + n_r;
          if (excl + n_r > a.capacity) atomicOr(a.status, kStIngestCapacity);
        }
      }
    }
    __syncthreads();
  };
  // Looking back is cheapest when it is done late (the predecessors have had time to publish), so a
  // chunk's parsed columns are parked in LDS and written out only after the NEXT chunk of this workgroup
  // has been staged, ranked and published.  (Every thread parks and reads back the same rows.)
  long long c_parked = -1;
  int n_parked = 0;
  auto flush_parked = [&]() {                              // uniform
    if (c_parked < 0) return;
    resolve_base(c_parked, n_parked);
    const long long rbase = s_base;
    for (int r = tid; r < n_parked; r += kIngThreads) {
      const long long row = rbase + r;
      if (row < a.capacity && !(a.ablate & 8)) { a.order_id[row] = s_res[r]; a.product_idx[row] = s_res[kKeep + r]; a.qty[row] = s_res[2 * kKeep + r]; }
    }
    c_parked = -1;
  };

  for (;;) {
    // A ticket is kIngPerTicket consecutive chunks.  Tickets go out in text order and a workgroup works
    // through its chunks in order, so a chunk only ever waits for smaller chunks held by workgroups that
    // are running.  (Striping the chunks over eight ticket counters, a workgroup serving the stripe
    // blockIdx % 8, was measured and changes nothing: 2.51 ms either way for 100 M dump-style rows, so the
    // one counter's update rate is not what bounds this kernel.  Two chunks per ticket lengthen the
    // look-back by more than they gain.)
    if (tid == 0) s_ticket = (a.ablate & 128) ? (long long)(blockIdx.x + gridDim.x * round++) : (long long)atomicAdd(a.ticket, 1u);
    __syncthreads();
    const int64_t c_first = (int64_t)s_ticket * kIngPerTicket;
    if (c_first >= a.n_chunks) break;                      // uniform
    for (int part = 0; part < kIngPerTicket; ++part) {
      const int64_t c = c_first + part;
      if (c >= a.n_chunks) break;                            // uniform
      const int64_t cbase = c * kIngChunk;
      // stage chunk + lookahead (16-byte pieces; the text buffer is padded by the host side)
      const int64_t avail = a.n_bytes - cbase;
      const int want = (int)(avail < kIngChunk + kIngLook ? avail : kIngChunk + kIngLook);
      for (int o = tid * 16; o < want && !((a.ablate & 16) && c >= (int64_t)gridDim.x); o += kIngThreads * 16) {
        if (o + 16 <= want) {
          *reinterpret_cast<uint4*>(s_chunk + o) = *reinterpret_cast<const uint4*>(a.text + cbase + o);
        } else {
          for (int b = o; b < want; ++b) s_chunk[b] = a.text[cbase + b];
        }
      }
      if (tid == 0) s_chunk[want] = (unsigned char)'\n';     // sentinel: every row ends in a newline
      // the 16 bytes in front of the chunk come along (the first thread needs the byte before its slice)
      if (tid == kIngThreads - 1 && c > 0) *reinterpret_cast<uint4*>(s_text) = *reinterpret_cast<const uint4*>(a.text + cbase - kIngFront);
      __syncthreads();
      const int64_t base = cbase + (int64_t)tid * kIngPerThread;
      // line starts in this thread's slice, then the data rows among them (first one or two bytes)
      const int o = tid * kIngPerThread;
      unsigned long long dmask = 0;
      // statement lines of this slice (a statement line takes at least 20 bytes: at most three start here)
      // hand out the position of their '(' instead of the line start
      unsigned long long smask_stmt = 0;
      unsigned short stmt_at[3] = {0, 0, 0};
      int n_stmt = 0;
      if (base < a.n_bytes) {
        uint32_t w[kIngWords];
#pragma unroll
        for (int j = 0; j < kIngWords / 4; ++j) {
          // whole 16-byte pieces even at the end of the text: what lies past the last byte (the sentinel,
          // leftovers of an earlier chunk) only sets mask bits that are cleared below
          const uint4 v = *reinterpret_cast<const uint4*>(s_chunk + o + 16 * j);
          w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        }
        const uint32_t prev = base == 0 ? (uint32_t)'\n' : (uint32_t)s_chunk[o - 1];
        unsigned long long smask = ing_line_start_mask(w, prev);
        const int64_t left = a.n_bytes - base;                 // bytes of text from this thread's first byte on
        if (left < kIngPerThread) smask &= (1ull << left) - 1ull;
        for (unsigned long long m = smask; m; m &= m - 1ull) {
          const int k = __ffsll((long long)m) - 1;
          const uint32_t c0 = s_chunk[o + k];
          const uint32_t c1 = k + 1 < left ? (uint32_t)s_chunk[o + k + 1] : 0u;
          if (ing_data_row(c0, c1)) dmask |= 1ull << k;
          else if ((c0 | 0x20u) == 'i') {
            // (bytes past the end of the text: the sentinel newline stops the search)
            const int at = ing_statement_row(s_text, kIngFront + o + k);
            if (at >= 0) { dmask |= 1ull << k; stmt_at[n_stmt < 3 ? n_stmt : 2] = (unsigned short)(o + k + at); ++n_stmt; smask_stmt |= 1ull << k; }
          }
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
      // publish as early as possible: successors add this count up long before this chunk knows its base
      if (tid == 0 && !(a.ablate & 64)) ing_publish_own(a.chunk_state, c, n_r);
      // a long row list would run into the parked columns of the chunk before: write those out first
      if (n_r > kKeep) { flush_parked(); __syncthreads(); }  // (uniform) every thread has read its parked rows back
      // hand the rows out one per thread: every lane of a wave then parses a row of its own, and
      // consecutive lanes write consecutive output rows
      int r = woff + incl - cnt;
      {
        int si = 0;
        for (unsigned long long m = dmask; m; m &= m - 1ull, ++r) {
          const int k = __ffsll((long long)m) - 1;
          const bool stmt = (smask_stmt >> k) & 1ull;
          s_row[r] = stmt ? stmt_at[si < 3 ? si : 2] : (unsigned short)(o + k);
          si += stmt ? 1 : 0;
        }
      }
      __syncthreads();
      if (n_r <= kKeep) {
        flush_parked();                                      // the chunk before: its look-back comes late, after this publish
        // A thread steps two rows through the automaton at once while more than one batch of rows is left
        // (the table look-ups of a row depend on each other; a second row fills the waits), one otherwise.
        auto parse_batch = [&](auto together, int r0) {
          constexpr int kTogether = decltype(together)::value;
          int p[kTogether];
          uint32_t st16[kTogether], A[kTogether], B[kTogether];
#pragma unroll
          for (int q = 0; q < kTogether; ++q) {                // a missing row repeats the thread's first
            const int rq = r0 + q * kIngThreads;
            p[q] = kIngFront + (int)s_row[rq < n_r ? rq : r0];
          }
          if (a.ablate & 4) {
#pragma unroll
            for (int q = 0; q < kTogether; ++q) { st16[q] = sDone * kNumClasses; A[q] = 0x01000000u | (uint32_t)p[q]; B[q] = 0; }
          } else {
            ing_scan_rows<kTogether>(s_text, s_ct, s_T, p, st16, A, B);
          }
#pragma unroll
          for (int q = 0; q < kTogether; ++q) {
            const int rq = r0 + q * kIngThreads;
            if (rq < n_r) {
              int32_t oid = 0, pidx = -1, qty = 0;
              // the cheaper conversion when no row of the wave has a long number or a long code
              const uint32_t Aq = A[q];
              const bool plain = ((Aq >> kANd1) & 127u) <= 8u && ((Aq >> kANd2) & 127u) <= 4u && ((Aq >> kACl) & 127u) <= 12u;
              bool ok = true;
              if (!(a.ablate & 1)) {
                if (__ballot(!plain) == 0ull) ok = ing_finish_row<2, 1, 3>(s_text, p[q], st16[q], Aq, B[q], a, oid, pidx, qty);
                else ok = ing_finish_row<3, 3, 8>(s_text, p[q], st16[q], Aq, B[q], a, oid, pidx, qty);
              }
              if (!ok) { ++bad; oid = 0; pidx = -1; qty = 0; }
              s_res[rq] = oid; s_res[kKeep + rq] = pidx; s_res[2 * kKeep + rq] = qty;
            }
          }
        };
        int b0 = 0;                                            // first row of the batch (uniform)
        for (; n_r - b0 > kIngThreads; b0 += 2 * kIngThreads)
          if (b0 + tid < n_r) parse_batch(std::integral_constant<int, 2>{}, b0 + tid);
        if (b0 + tid < n_r) parse_batch(std::integral_constant<int, 1>{}, b0 + tid);
        c_parked = c;
        n_parked = n_r;
      } else {
        // more rows than can be parked (rows of a few bytes): look back first, write as they are parsed
        resolve_base(c, n_r);
        const long long rbase = s_base;
        for (r = tid; r < n_r; r += kIngThreads) {
          int32_t oid = 0, pidx = -1, qty = 0;
          const int p[1] = {kIngFront + (int)s_row[r]};
          uint32_t st16[1], A[1], B[1];
          ing_scan_rows<1>(s_text, s_ct, s_T, p, st16, A, B);
          const bool ok = ing_finish_row<3, 3, 8>(s_text, p[0], st16[0], A[0], B[0], a, oid, pidx, qty);
          if (!ok) { ++bad; oid = 0; pidx = -1; qty = 0; }
          const long long row = rbase + r;
          if (row < a.capacity) { a.order_id[row] = oid; a.product_idx[row] = pidx; a.qty[row] = qty; }
        }
      }
      __syncthreads();                                       // the text and the row list are free again
    }
  }
  flush_parked();
  if (bad) {
    atomicAdd(a.n_malformed, bad);
    atomicOr(a.status, kStIngestMalformed);
  }
}

}  // namespace acmpc
