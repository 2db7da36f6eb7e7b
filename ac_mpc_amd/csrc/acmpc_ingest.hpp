// acmpc_ingest.hpp — gfx950 device code of the `detalle` text ingest: SQL-dump / CSV bytes in HBM ->
// the three int32 columns (order_id, product_idx, qty) the aggregation kernels read.
//
// The text shapes are the ones the reference writes: the INSERT of process/confirmcompra.php:36
// (through library/consulSQL.php:29-30, all three values quoted), the phpMyAdmin dump style of
// store.sql:88-89,125-128 (header line, one "(v, 'v', v)," row per line) and plain comma / tab
// separated rows.  The grammar is spelled out in include/acmpc.h.
//
// One pass over the text (byte work; no MFMA):
//   k_ingest_parse   12 KiB chunks are handed out in text order (a ticket counter per stripe of chunks); a chunk
//                    is staged in LDS (plus 256 bytes of look-ahead), every thread finds the rows that start
//                    in its own 48 bytes and parses them itself: a table-driven automaton over the row,
//                    word-wide number conversion, a perfect-hash dictionary of the product codes, and the
//                    three columns written at the row's rank in the text (chunk counts are published and
//                    summed by decoupled look-back; there is no counting pass)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace acmpc {

constexpr int kIngThreads = 256;
constexpr int kIngPerThread = 48;                       // bytes of text per thread (a multiple of 16)
constexpr int kIngChunk = kIngThreads * kIngPerThread;  // 12 KiB per workgroup iteration
constexpr int kIngWords = kIngPerThread / 4;
constexpr int kIngStripes = 8;                          // chunk c belongs to stripe c % n_stripes; a counter per stripe
constexpr int kIngTicketPitch = 64;                     // 32-bit words between two counters: a line of their own each
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
  unsigned int* ticket;               // zero on entry: kIngStripes counters, kIngTicketPitch words apart
  int32_t n_stripes;                  // stripes in use (fewer than kIngStripes when there are fewer workgroups)
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
  int32_t ablate;                     // developer switch for timing experiments (0 in production): 1 no field
                                      // conversion, 2 no dictionary, 4 no automaton, 8 no stores, 16 no line starts
                                      // looked at, 32 no newline search, 64 no look-back
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
// The statement exactly as library/consulSQL.php:30 builds it from the arguments of process/confirmcompra.php:36:
// "insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES(" — 70 bytes, the row's '(' at offset 69 —
// as little-endian words.  A line that begins with these bytes needs no search: its only earlier '(' follows
// "detalle", not "values", and there is no newline in between, so the search below would return 69 as well.
constexpr int kIngCanonAt = 69;
constexpr int kIngCanonWords = 18;
__device__ __forceinline__ bool ing_canonical_statement(const unsigned char* text, int pos) {
  constexpr uint32_t canon[kIngCanonWords] = {
      0x65736e69u, 0x69207472u, 0x206f746eu, 0x61746564u, 0x20656c6cu, 0x6d754e28u, 0x69646550u, 0x202c6f64u, 0x69646f43u,
      0x72506f67u, 0x202c646fu, 0x746e6143u, 0x64616469u, 0x646f7250u, 0x6f746375u, 0x56202973u, 0x45554c41u, 0x00002853u};
  const uint32_t* w = reinterpret_cast<const uint32_t*>(text) + (pos >> 2);
  const uint32_t sh = (uint32_t)pos & 3u;
  uint32_t x[kIngCanonWords + 1];
#pragma unroll
  for (int i = 0; i <= kIngCanonWords; ++i) x[i] = w[i];             // (none of these reads depends on another)
  uint32_t diff = 0u;
#pragma unroll
  for (int i = 0; i < kIngCanonWords; ++i) {
    const uint32_t d = __builtin_amdgcn_alignbyte(x[i + 1], x[i], sh) ^ canon[i];
    diff |= i == kIngCanonWords - 1 ? (d & 0xffffu) : d;             // the last word holds the prefix's last two bytes
  }
  return diff == 0u;
}
// `text` is the (4-byte aligned) LDS text, `pos` the offset of the line start in it.  The line is searched
// a word at a time for '(' and newline; four words are requested together (the reads of one step do not
// depend on each other).
__device__ __forceinline__ int ing_statement_row(const unsigned char* text, int pos) {
  if (ing_canonical_statement(text, pos)) return kIngCanonAt;
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

// ---------------------------------------------------------------------------------------------
// Row base of a chunk without a counting pass: chunks are taken in text order WITHIN A STRIPE (chunk c
// belongs to stripe c % n_stripes, every stripe has a ticket counter of its own), every chunk publishes
// its row count as soon as it knows it (before its workgroup waits for anything), and looks back over its
// predecessors' entries (64 at a time) until it meets one that already holds an inclusive prefix.  An
// entry is one 64-bit word: the top two bits say empty / own count / inclusive prefix, the rest is the
// value, so one relaxed atomic store publishes both.
// Forward progress.  A chunk waits only for chunks in front of it.  Take the first chunk c* of the text
// that is not published.  If a workgroup has drawn it, that workgroup publishes it without waiting for
// anyone.  If nobody has, c* is the next ticket of its stripe, so it lies past every chunk the stripe's
// workgroups hold; those workgroups wait only for chunks in front of the ones they hold, all of which are
// published (they lie in front of c*), so they run on and one of them draws c*.  This needs every stripe to
// have a workgroup that is running: the grid has at least n_stripes workgroups, workgroup b serves stripe
// b % n_stripes, and the hardware starts workgroups in index order, so workgroups 0 .. n_stripes - 1 are
// resident from the start and stay until their stripe has no ticket left.  Any grid size is therefore safe,
// also fewer workgroups than stripes would be (n_stripes is then the grid size) and more workgroups than the
// chip holds at once (the late ones start when a stripe is exhausted, draw a ticket past the end and leave).
// Should a dispatcher ever start workgroups out of order, the wait is bounded: it gives up after 2^24 polls
// and reports kStInternal instead of hanging.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kIngOwn = 1ull << 62, kIngInclusive = 2ull << 62, kIngValue = (1ull << 62) - 1ull;
constexpr uint32_t kStInternal = 64u;

// Publish the chunk's own row count (one thread).  Done as early as possible: successors can then add
// it up long before this chunk knows its own base.
__device__ __forceinline__ void ing_publish_own(unsigned long long* state, long long c, long long n_r) {
  __hip_atomic_store(&state[c], (c == 0 ? kIngInclusive : kIngOwn) | (unsigned long long)n_r, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// The entry that lane `lane` looks at in the first round of chunk c's look-back: the one at distance `lane`
// in front of c - 1.  Issued early, the load is on its way while the wave does something else.
__device__ __forceinline__ unsigned long long ing_look_first(const unsigned long long* state, long long c, int lane) {
  const long long idx = c - 1 - lane;
  return idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kIngInclusive;
}

// Called by a whole wave, after ing_publish_own.  Returns the number of data rows in chunks [0, c) and
// publishes the chunk's inclusive prefix.  `first` is ing_look_first(state, c, lane).  Every round looks at
// the 64 entries behind p (nearest first), one per lane: wider rounds were measured and cost more than
// they save, every poll competes with the publishers for the same lines.
__device__ __forceinline__ long long ing_row_base(unsigned long long* state, long long c, long long n_r, int lane,
                                                  uint32_t* status, unsigned long long first) {
  if (c == 0) return 0;
  long long excl = 0;
  long long p = c - 1;                                   // nearest predecessor not yet added
  bool preloaded = true;
  for (uint32_t polls = 0;;) {
    const long long idx = p - lane;
    const unsigned long long v = preloaded ? first
                                 : (idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : kIngInclusive);                   // in front of chunk 0: nothing
    preloaded = false;
    const unsigned long long incl_mask = __ballot((v >> 62) == 2ull);
    const int nearest = incl_mask ? __ffsll((long long)incl_mask) - 1 : 64;      // lane holding the nearest prefix
    // entries nearer than (and including) that one must all be published
    const bool mine = lane <= nearest;
    if (__ballot(mine && (v >> 62) == 0ull)) {             // a predecessor has not published yet
      if (++polls > (1u << 24)) { if (lane == 0) atomicOr(status, kStInternal); break; }   // never expected
      // back off: every poll reads lines that the publishers are about to write, and the memory side
      // serves one address only so often
      if (polls < 4) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
      continue;
    }
    long long val = mine ? (long long)(v & kIngValue) : 0ll;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) val += __shfl_xor(val, d);
    excl += val;
    if (incl_mask) break;
    p -= 64;
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
// word B fields (bits 21..31 of the entry's low word hold kStUnit * next state)
constexpr int kBE1 = 0, kBCs = 7, kBE2 = 14, kBState = 21;
// A state travels as its number times kStUnit: shifted left by four that is the byte offset of the state's
// row of 16 eight-byte entries, to which the byte class, kept as its number times eight, is added.
constexpr int kStUnit = 8;

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
      b |= (unsigned long long)(n * kStUnit) << kBState;
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
__device__ __forceinline__ void ing_scan_rows(const unsigned char* s_text, const unsigned char* s_ct8, const uint2* s_T,
                                              const int (&p)[N], const bool (&have)[N], uint32_t (&st)[N],
                                              uint32_t (&A)[N], uint32_t (&B)[N]) {
  // a lane without a row starts in the final state, which stays where it is and counts nothing
#pragma unroll
  for (int q = 0; q < N; ++q) { st[q] = (have[q] ? sStart : sDone) * kStUnit; A[q] = 0; B[q] = 0; }
  const unsigned char* tab = reinterpret_cast<const unsigned char*>(s_T);
  for (int j0 = 0; j0 < kIngSteps; j0 += 4) {
    // four bytes of every row and their classes first (none of these reads waits for a state), then the
    // chain of table look-ups, which does
    uint32_t cls8[N][4];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      uint32_t c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = s_text[p[q] + j0 + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) cls8[q][j] = s_ct8[c[j]];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const uint2 e = *reinterpret_cast<const uint2*>(tab + ((st[q] << 4) + cls8[q][j]));
        B[q] += e.x;
        A[q] += e.y;
        st[q] = e.x >> kBState;
        // (kept apart from the shift of the next look-up: folded together the two shifts need a mask and an
        // add, three instructions where a shift and a shift-add do)
        asm volatile("" : "+v"(st[q]));
      }
    }
    bool running = false;
#pragma unroll
    for (int q = 0; q < N; ++q) running |= st[q] < (uint32_t)(sDone * kStUnit);
    if (__ballot(running) == 0ull) break;                  // every row of every active lane is done
  }
}

// Turn the automaton's final state and counters for the row at byte `p` into the three columns.
// Returns false when the row is malformed.  W1, W2: words of digits read for the two numbers; NC: words
// of the code that are cut out (the rest are zero).  The caller picks <3, 3, 8>, which covers every row,
// or a cheaper combination when no row of the wave needs more.
template <int W1, int W2, int NC, bool kDev = false>
__device__ __forceinline__ bool ing_finish_row(const unsigned char* s_text, int p, uint32_t st16, uint32_t A, uint32_t B,
                                               const IngestArgs& a, int32_t& oid, int32_t& pidx, int32_t& qty) {
  const int nd1 = (A >> kANd1) & 127, nd2 = (A >> kANd2) & 127, cl = (A >> kACl) & 127, len = (A >> kALen) & 127;
  const bool neg1 = (A >> kANeg1) & 1u, neg2 = (A >> kANeg2) & 1u;
  const int e1 = (B >> kBE1) & 127, cs = (B >> kBCs) & 127, e2 = (B >> kBE2) & 127;
  bool ok = st16 == (uint32_t)(sDone * kStUnit) && len <= kIngMaxRow && nd1 <= 10 && nd2 <= 10 && cl <= 30;
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
  if (ok && !(kDev && (a.ablate & 2))) {
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

// Inclusive prefix sum of a small integer over the 64 lanes in DPP adds (no LDS traffic).
__device__ __forceinline__ int ing_wave_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// Sixteen bytes of text at offset `off`.  A piece that reaches past the end of the text is put together
// byte by byte: the byte at n_bytes is the sentinel newline (every row ends in one), bytes past it are 0.
__device__ __forceinline__ uint4 ing_load_piece(const unsigned char* text, int64_t off, int64_t n_bytes) {
  if (off + 16 <= n_bytes) return *reinterpret_cast<const uint4*>(text + off);
  uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    uint32_t c = 0u;
    if (off + b < n_bytes) c = text[off + b];
    else if (off + b == n_bytes) c = (uint32_t)'\n';
    w[b >> 2] |= c << (8 * (b & 3));
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// Bit k set: byte k of the thread's 48 bytes (twelve little-endian words) is not a newline.  Per word the
// exact zero-byte test leaves one flag in bit 7 of every byte; three shift-or steps bring the four flags
// together in the top nibble, and a funnel shift appends the nibble to the mask (words are taken last to
// first, so that word 0 ends in the lowest bits).
__device__ __forceinline__ uint32_t ing_not_newline_nibble(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  const uint32_t u = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bit 7 of a byte: it is not '\n'
  uint32_t acc = (u << 21) | u;
  acc = (u << 14) | acc;
  acc = (u << 7) | acc;
  return acc;                                                                // flags of bytes 0..3 in bits 28..31
}
__device__ __forceinline__ unsigned long long ing_newline_mask(const uint32_t (&w)[kIngWords]) {
  static_assert(kIngWords == 12, "two mask words: bytes 0..31 and 32..47");
  uint32_t lo = 0u, hi = 0u;
#pragma unroll
  for (int j = kIngWords - 1; j >= 8; --j) hi = __builtin_amdgcn_alignbit(hi, ing_not_newline_nibble(w[j]), 28);
#pragma unroll
  for (int j = 7; j >= 0; --j) lo = __builtin_amdgcn_alignbit(lo, ing_not_newline_nibble(w[j]), 28);
  return ((unsigned long long)(~hi & 0xffffu) << 32) | (unsigned long long)~lo;
}

// What a thread knows about the data rows that start in its 48 bytes of a chunk.
struct IngOwn {
  unsigned long long rows;    // bit k: a data row starts at byte k of the slice (for a statement line: the line)
  uint32_t stmt_at;           // statement lines (the ones that start with a letter): offsets of their '(' from the
                              // line start, where the row starts; one byte each, in text order
  int first;                  // rank of the thread's first row among the rows of the chunk
};

constexpr int kIngBuf = kIngFront + kIngChunk + kIngLook + 64;
// A chunk with at most this many data rows (long lines: one-statement-per-line text has a row every 98 bytes, 125 a
// chunk) hands its rows out by rank in the chunk, one to a lane (wave w takes rows 64 w + lane), instead of leaving
// every row to the thread in whose 48 bytes it starts: the lanes of the first waves are then all busy and the waves
// past the last row skip the parse, where otherwise every wave runs the automaton for the few lanes that own a row.
// (Two rows to a lane, which would also serve the dump style's 480 rows a chunk, was measured and lost: a chunk's
// time is the time of its busiest wave, and one wave with two rows per lane takes longer than two waves with one.)
constexpr int kIngSparse = kIngThreads;
constexpr int kIngLookWave = 0;                            // the wave that looks back over the chunks in front for the row base
constexpr int kWaveRows = 64;                              // rows of one wave-instruction in the rank-based parse

// Every thread owns 48 bytes of a 12 KiB chunk and parses the rows that START there (a row of the shapes the
// reference writes takes 19..26 bytes, so a thread has one to three of them; any other mix only changes how
// evenly the lanes are loaded).  A workgroup works on two chunks at a time: chunk i + 1 is staged in LDS,
// its rows are found, counted and the count published, and only then is chunk i — staged one round earlier
// in the other text buffer — parsed and written, because its first output row is the sum of the counts
// of all chunks before it, and looking back over them is cheapest when it is done late.  The text of chunk
// i + 2 is requested before the parse and waits in registers; so does the ticket after it.
// kDev: the developer switches of IngestArgs::ablate are compiled in (timing experiments only).
template <bool kDev>
__global__ void __launch_bounds__(kIngThreads) __attribute__((amdgpu_waves_per_eu(kIngBlocksPerCu * kIngThreads / 256, kIngBlocksPerCu * kIngThreads / 256))) k_ingest_parse(IngestArgs a) {
  const int ablate = kDev ? a.ablate : 0;
  // [16 bytes of slack][chunk][look-ahead][slack]: number and code windows may reach a few bytes outside a row
  __shared__ __align__(16) unsigned char s_text[2][kIngBuf];
  __shared__ __align__(8) uint2 s_T[kIngTableWords];
  __shared__ unsigned char s_ct[256];
  __shared__ int s_w[kIngThreads / 64];
  __shared__ long long s_tk[2];
  __shared__ long long s_base[2], s_base_of[2];            // row base of a chunk, and which chunk (+ 1) it belongs to
  __shared__ unsigned short s_list[2][kIngSparse];         // chunks of few rows: where the row of every rank starts in s_text
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kIngTableWords; i += kIngThreads) {
    const unsigned long long e = a.tables[i];
    s_T[i] = make_uint2((uint32_t)e, (uint32_t)(e >> 32));
  }
  if (tid < 256) s_ct[tid] = (unsigned char)(8u * reinterpret_cast<const unsigned char*>(a.tables + kIngTableWords)[tid]);   // class * 8
  if (tid < kIngFront) { s_text[0][tid] = 0; s_text[1][tid] = 0; }
  if (tid < 2) s_base_of[tid] = 0;
  uint32_t bad = 0;                                        // malformed rows met by this thread (a chunk holds < 2^13 rows)
  const int o = tid * kIngPerThread;                       // this thread's slice inside a chunk

  // text of a chunk on its way to LDS: three 16-byte pieces per thread, each load contiguous over the workgroup
  // (piece j of thread t is bytes [j * 4096 + 16 t, + 16) of the chunk; the thread's OWN 48 bytes are read back
  // from LDS once the chunk is staged: loaded directly they are 48 bytes apart from lane to lane, and every one
  // of the three loads then touches every 64-byte piece of the wave's 3 KB), a piece of the look-ahead (the
  // first kIngLook / 16 threads) and the byte in front of the chunk (thread 0)
  constexpr int kIngPieces = kIngPerThread / 16, kIngPieceStride = kIngThreads * 16;
  uint4 rt[kIngPieces], rl;
  uint32_t rprev = (uint32_t)'\n';
  auto request_text = [&](int64_t c) {
    const int64_t cbase = c * kIngChunk;
    if (cbase + kIngChunk + kIngLook <= a.n_bytes) {       // uniform: every chunk but the last one or two
#pragma unroll
      for (int j = 0; j < kIngPieces; ++j)
        rt[j] = *reinterpret_cast<const uint4*>(a.text + cbase + j * kIngPieceStride + 16 * tid);
      if (tid < kIngLook / 16) rl = *reinterpret_cast<const uint4*>(a.text + cbase + kIngChunk + 16 * tid);
    } else {
#pragma unroll
      for (int j = 0; j < kIngPieces; ++j) rt[j] = ing_load_piece(a.text, cbase + j * kIngPieceStride + 16 * tid, a.n_bytes);
      if (tid < kIngLook / 16) rl = ing_load_piece(a.text, cbase + kIngChunk + 16 * tid, a.n_bytes);
    }
    if (tid == 0) rprev = c > 0 ? (uint32_t)a.text[cbase - 1] : (uint32_t)'\n';
  };
  // Chunks are handed out in text order within a stripe: stripe s = blockIdx % n_stripes owns the chunks
  // c with c % n_stripes == s, and its workgroups draw them from a counter of their own.  (All workgroups
  // drawing from one counter are served one after the other by the memory side, 11 ns apiece: 2.4 ms for
  // the 213 k chunks of 100 M rows, whatever else the kernel does.)  A chunk waits only for chunks in front
  // of it; the first of those that is unfinished is either being worked on or is the next ticket of its
  // stripe, and workgroups are started in index order, so every stripe has a running workgroup to draw it.
  const int stripe = (int)(blockIdx.x % (unsigned)a.n_stripes);
  auto take_ticket = [&]() -> long long {
    return (long long)atomicAdd(a.ticket + kIngTicketPitch * stripe, 1u) * a.n_stripes + stripe;
  };

  if (tid == 0) s_tk[0] = take_ticket();
  __syncthreads();
  int64_t c_next = s_tk[0];
  if (c_next >= a.n_chunks) return;                        // uniform
  request_text(c_next);
  long long tk_pending = 0;
  if (tid == 0) tk_pending = take_ticket();
  int64_t c_cur = -1;
  int buf = 0, par = 1;                                    // text buffer and ticket slot of the chunk being staged
  IngOwn cur = {0ull, 0u, 0}, nxt = {0ull, 0u, 0};
  int n_cur = 0;

  for (;;) {
    __syncthreads();                                       // the rows parsed last round have been read out of s_text[buf]
    if (c_next >= 0) {                                     // uniform
      unsigned char* t = s_text[buf] + kIngFront;
#pragma unroll
      for (int j = 0; j < kIngPieces; ++j) *reinterpret_cast<uint4*>(t + j * kIngPieceStride + 16 * tid) = rt[j];
      if (tid < kIngLook / 16) *reinterpret_cast<uint4*>(t + kIngChunk + 16 * tid) = rl;
      if (tid == 0) { t[-1] = (unsigned char)rprev; s_tk[par] = tk_pending; }
    }
    __syncthreads();
    if (c_next >= 0) {
      unsigned long long nl;
      {
        uint32_t w[kIngWords];
#pragma unroll
        for (int j = 0; j < kIngPieces; ++j) {
          const uint4 x = *reinterpret_cast<const uint4*>(s_text[buf] + kIngFront + o + 16 * j);
          w[4 * j] = x.x; w[4 * j + 1] = x.y; w[4 * j + 2] = x.z; w[4 * j + 3] = x.w;
        }
        nl = ing_newline_mask(w);
        if (ablate & 32) nl = ~0ull;
      }
      // line starts of the slice (the byte in front is a newline, or the text starts here), then the data
      // rows among them by their first one or two bytes; a statement line hands out the position of its '('
      const unsigned char* t = s_text[buf] + kIngFront;
      const int64_t left = a.n_bytes - (c_next * kIngChunk + o);      // bytes of text from this thread's first byte on
      unsigned long long smask = ((nl << 1) | (t[o - 1] == (unsigned char)'\n' ? 1ull : 0ull)) & ((1ull << kIngPerThread) - 1ull);
      if (left < kIngPerThread) smask = left > 0 ? (smask & ((1ull << left) - 1ull)) : 0ull;
      if (ablate & 16) smask = 0ull;
      nxt.rows = 0; nxt.stmt_at = 0;
      int n_stmt = 0;
      for (unsigned long long m = smask; m; m &= m - 1ull) {
        const int k = __ffsll((long long)m) - 1;
        const uint32_t c0 = t[o + k];
        const uint32_t c1 = t[o + k + 1];                    // (past the end of the text: the sentinel or 0, no digit)
        if (ing_data_row(c0, c1)) nxt.rows |= 1ull << k;
        else if ((c0 | 0x20u) == 'i') {
          // (bytes past the end of the text: the sentinel newline stops the search; a statement line takes
          // at least 20 bytes, so at most three start in a slice)
          const int at = ing_statement_row(s_text[buf], kIngFront + o + k);
          if (at >= 0) {
            nxt.rows |= 1ull << k;
            nxt.stmt_at |= (uint32_t)at << (8 * (n_stmt < 3 ? n_stmt : 3));
            ++n_stmt;
          }
        }
      }
      const int cnt = __popcll(nxt.rows);
      const int incl = ing_wave_scan(cnt);
      if (lane == 63) s_w[wave] = incl;
      nxt.first = incl - cnt;
    }
    __syncthreads();
    int n_next = 0;
    int64_t c_after = -1;
    if (c_next >= 0) {
      int woff = 0;
#pragma unroll
      for (int k = 0; k < kIngThreads / 64; ++k) { if (k < wave) woff += s_w[k]; n_next += s_w[k]; }
      nxt.first += woff;
      // publish as early as possible: successors add this count up long before this chunk knows its base
      if (tid == 0 && !(ablate & 64)) ing_publish_own(a.chunk_state, c_next, n_next);
      if (n_next <= kIngSparse) {                          // uniform: few rows, handed out by rank when the chunk is parsed
        const unsigned char* t = s_text[buf] + kIngFront;
        uint32_t at = nxt.stmt_at;
        int r = nxt.first;                                 // ranks of this thread's rows: [first, first + its rows) < n_next
        for (unsigned long long mm = nxt.rows; mm; mm &= mm - 1ull) {
          const int k = __ffsll((long long)mm) - 1;
          int p = kIngFront + o + k;
          if (at != 0u && (t[o + k] | 0x20u) == (uint32_t)'i') { p += (int)(at & 0xffu); at >>= 8; }   // (as take() below)
          s_list[buf][r++] = (unsigned short)p;
        }
      }
      c_after = s_tk[par];
      if (c_after < a.n_chunks) {
        request_text(c_after);
        if (tid == 0) tk_pending = take_ticket();
      } else {
        c_after = -1;
      }
    }
    if (c_cur >= 0) {
      // One wave (kIngLookWave) looks back over the counts of the chunks in front and hands the row base to the
      // other waves through LDS, where they wait for it (no barrier: that wave never waits for them, and polling
      // LDS costs the memory side nothing).  The first round of its loads is issued here and used after its first
      // rows have been through the automaton, which needs no row base.  (Issued a whole staging and discovery phase
      // earlier, at the top of the iteration, the answer is older and as often incomplete: measured, no gain.)
      unsigned long long look = kIngInclusive;
      if (wave == kIngLookWave && !(ablate & 64)) look = ing_look_first(a.chunk_state, c_cur, lane);
      auto row_base = [&]() -> long long {
        long long rbase;
        if (wave == kIngLookWave) {
          rbase = (ablate & 64) ? c_cur * 400 : ing_row_base(a.chunk_state, c_cur, n_cur, lane, a.status, look);
          if (lane == 0) {
            s_base[par] = rbase;
            __hip_atomic_store(&s_base_of[par], c_cur + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (c_cur == a.n_chunks - 1) {
              *a.n_rows = rbase + n_cur;
              if (rbase + n_cur > a.capacity) atomicOr(a.status, kStIngestCapacity);
            }
          }
        } else {
          while (__hip_atomic_load(&s_base_of[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != c_cur + 1)
            __builtin_amdgcn_s_sleep(2);
          rbase = s_base[par];
        }
        return rbase;
      };
      const unsigned char* tx = s_text[buf ^ 1];
      // A chunk of few rows hands them out by rank, one to a thread (thread r takes row r, whose place the list
      // holds); otherwise every thread parses the rows that start in its own 48 bytes.
      const bool sparse = n_cur <= kIngSparse;             // uniform
      unsigned long long m = sparse ? (tid < n_cur ? 1ull : 0ull) : cur.rows;
      uint32_t at = sparse ? 0u : cur.stmt_at;
      const int first_row = sparse ? tid : cur.first;
      long long row = -1;                                  // output row of the lane's next data row, once the base is known
      auto take = [&](bool& have) -> int {
        have = m != 0ull;
        const int k = have ? __ffsll((long long)m) - 1 : 0;
        int p = kIngFront + o + k;
        if (have && at != 0u && (tx[p] | 0x20u) == (uint32_t)'i') { p += (int)(at & 0xffu); at >>= 8; }
        m &= m - 1ull;
        if (sparse) p = have ? (int)s_list[buf ^ 1][tid] : kIngFront;
        return p;
      };
      auto emit = [&](bool have, int p, uint32_t st16, uint32_t A, uint32_t B, bool wide) {
        if (!have) return;
        int32_t oid = 0, pidx = -1, qty = 0;
        bool ok = true;
        if (!(ablate & 1)) {
          if (!wide) ok = ing_finish_row<2, 1, 3, kDev>(tx, p, st16, A, B, a, oid, pidx, qty);
          else ok = ing_finish_row<3, 3, 8, kDev>(tx, p, st16, A, B, a, oid, pidx, qty);
        }
        if (!ok) { ++bad; oid = 0; pidx = -1; qty = 0; }
        if (row < a.capacity && !(ablate & 8)) { a.order_id[row] = oid; a.product_idx[row] = pidx; a.qty[row] = qty; }
        ++row;
      };
      auto plain = [](uint32_t A) {
        return ((A >> kANd1) & 127u) <= 8u && ((A >> kANd2) & 127u) <= 4u && ((A >> kACl) & 127u) <= 12u;
      };
      // two rows of the lane go through the automaton together while any lane has two left (the table
      // look-ups of one row depend on each other; a second row fills the waits).  With rows handed out by rank a
      // wave past the last row has nothing to do (the wave that looks back always goes through: it publishes the
      // chunk's prefix whether or not the chunk has rows).
      if (!sparse || wave == kIngLookWave || wave * kWaveRows < n_cur)      // wave-uniform
      do {
        bool have[2];
        int p[2];
        p[0] = take(have[0]);
        p[1] = take(have[1]);
        uint32_t st16[2], A[2], B[2];
        const bool two = __ballot(have[1]) != 0ull;
        if (ablate & 4) { st16[0] = st16[1] = sDone * kStUnit; A[0] = A[1] = 0x01000000u; B[0] = B[1] = 0; }
        else if (two) ing_scan_rows<2>(tx, s_ct, s_T, p, have, st16, A, B);
        else {
          const int p1[1] = {p[0]};
          const bool h1[1] = {have[0]};
          uint32_t s1[1], A1[1], B1[1];
          ing_scan_rows<1>(tx, s_ct, s_T, p1, h1, s1, A1, B1);
          st16[0] = s1[0]; A[0] = A1[0]; B[0] = B1[0];
          st16[1] = sDone * kStUnit; A[1] = 0; B[1] = 0;
        }
        if (row < 0) row = row_base() + first_row;          // uniform: the first pass of this chunk
        // the cheaper conversion when no row of the wave has a long number or a long code
        const bool wide = __ballot((have[0] && !plain(A[0])) || (have[1] && !plain(A[1]))) != 0ull;
        emit(have[0], p[0], st16[0], A[0], B[0], wide);
        if (two) emit(have[1], p[1], st16[1], A[1], B[1], wide);
      } while (__ballot(m != 0ull));
    }
    if (c_next < 0) break;                                 // uniform: nothing was staged, the last chunk is written
    c_cur = c_next; cur = nxt; n_cur = n_next;
    c_next = c_after;
    buf ^= 1; par ^= 1;
  }
  if (bad) {
    atomicAdd(a.n_malformed, (unsigned long long)bad);
    atomicOr(a.status, kStIngestMalformed);
  }
}

}  // namespace acmpc
