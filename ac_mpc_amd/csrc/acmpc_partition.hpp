// acmpc_partition.hpp — gfx950 device code of the order-restoring path for `detalle` rows in any order
// (concurrent checkouts interleave their inserts, process/confirmcompra.php:27,35-37; the table has no key
// that would order them, store.sql:192-194).
//
// One 64-bit global atomic per line is bound by the part's scattered-atomic rate (about 23 G/s).  Instead
// the lines are moved next to the other lines of their order's neighbourhood, and summed there in LDS:
//
//   k_part_count      lines per bucket of 16384 consecutive orders (32768 or 65536 beyond 2^25 / 2^26 orders: there
//                     are at most 2048 buckets), one pass over order_id, and how often the order id changes from
//                     one line to the next
//   k_part_scan       bucket offsets, cursors of the two scatter passes; decides whether the rows are far
//                     enough from grouped for this path (else k_lines_any, one atomic per run, is cheaper)
//   k_part_scatter_a  every line becomes one 64-bit record, order index (25 to 27 bits) above price * qty (39 to
//                     37 bits), written into one of 32 coarse bins (64 buckets each); the stock deltas are taken
//                     here as well.  A value that does not fit its bits does not fit a record: the pass then hands
//                     the totals (not the stock deltas, which it has) to the atomic kernel
//   k_part_scatter_b  every coarse bin split into its buckets
//   k_part_reduce     one workgroup per 16384 orders: 64-bit adds into a 128 KB LDS table, totals written once (a
//                     bucket of 32768 or 65536 orders is read by two or four workgroups, each keeping its part)
//
// Both scatter passes bin a round of records (4096 / 8192) in LDS and write whole runs, and reserve the place
// of a run with one returning atomic per (round, bin): the records of a bucket end up in no particular order,
// which integer sums do not mind.  HBM-bound byte work; no MFMA.
#pragma once
#include "acmpc_kernels.hpp"

namespace acmpc {

constexpr int kPartShift = 14;                         // 16384 orders per reduce workgroup: int64 cells fill 128 KB of LDS
constexpr int kPartBucket = 1 << kPartShift;
constexpr int kPartMaxBuckets = 2048;                  // buckets of 2^14 orders reach 2^25 orders, of 2^16 orders 2^27
constexpr int kPartMaxShift = 16;
constexpr int kPartCoarseShift = 6;                    // 64 buckets per coarse bin, at most 32 coarse bins
constexpr int kPartCoarse = kPartMaxBuckets >> kPartCoarseShift;
constexpr int kPartThreads = 1024;
constexpr int kPartWaves = kPartThreads / kWave;
constexpr int kPartRound = kPartThreads * 4;           // records a workgroup bins per round
constexpr int kPartCntPitch = 36;                      // counters per wave in pass A: 32 bins, the drop bin, padding

struct PartArgs {
  const int32_t* order_id;
  const int32_t* product_idx;
  const int32_t* qty;
  int64_t n_lines;
  int32_t base;
  int32_t n_orders;
  int32_t n_buckets;                  // ceil(n_orders / 2^shift), at most kPartMaxBuckets
  int32_t shift;                      // log2 of the orders per bucket: 14, or 15 / 16 for more than 2^25 / 2^26 orders
  int32_t val_bits;                   // bits of a record below the order index: 64 - 11 - shift
  const int64_t* price;               // [P + 1], entry P is the null product
  int32_t n_products, p_pad;
  uint32_t q_small;                   // quantities in [0, q_small) go to the 32-bit LDS histogram
  uint32_t* prod_slab;                // [gridDim.x][p_pad] (pass A)
  unsigned long long* prod_carry;     // [P] 64-bit cells for every other quantity
  int32_t totals_only;                // 1: the stock deltas are already known (fallback of the grouped pass)
  int32_t only_if_unsorted;           // 1: leave at once unless the grouped pass raised kStUnsorted
  unsigned long long* counts;         // [kPartMaxBuckets] lines per bucket (zero on entry of k_part_count)
  unsigned long long* bucket_off;     // [kPartMaxBuckets + 1] first record of every bucket
  unsigned long long* cursor_a;       // [kPartCoarse] next free record of every coarse bin
  unsigned long long* cursor_b;       // [kPartMaxBuckets] next free record of every bucket
  unsigned long long* rec_a;          // records after pass A   [n_lines]
  unsigned long long* rec_b;          // records after pass B   [n_lines]
  int64_t* order_total;               // [n_orders]
  uint32_t* status;
  // How far from grouped the rows are decides which path sums them: k_part_count counts the places where the
  // order id changes from one line to the next (inside every lane's four lines: three of four places), and
  // k_part_scan sets *mode to 1 (this path) when the estimate exceeds 0.4 changes per line, else to 0 (one
  // 64-bit atomic per run of equal ids, k_lines_any, which costs about 0.05 ms per million runs).  Pass A sets
  // it to 2 when it meets a value that does not fit a record: the atomic kernel then redoes the order totals.
  // Rows whose ids never go down are grouped after all: mode 3, and the grouped line pass takes them.
  // order lifecycle (each may be null): the price a line was sold at instead of the table price, whether an order
  // counts towards the stock deltas (bit 0 of its state), and venta.Descuento, taken off where the reduce writes
  const long long* line_price;        // [n_lines]
  const unsigned char* order_state;   // [n_orders]
  uint32_t* state_bits;               // [ceil(n_orders / 32)] bit 0 of every order's state, packed (filled by k_part_state_bits)
  const int32_t* order_descuento;     // [n_orders]
  int32_t descuento_mode;
  unsigned long long* changes;        // zero on entry of k_part_count
  unsigned long long* descents;       // zero on entry of k_part_count; nonzero: the order id goes down somewhere
  unsigned long long* mode;
};

__device__ __forceinline__ bool part_skip(const PartArgs& a) {
  return a.only_if_unsorted && !(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStUnsorted);
}
// the scatter passes and the reduce: also when the rows are near enough to grouped for the atomic kernel
__device__ __forceinline__ bool part_skip_moving(const PartArgs& a) { return part_skip(a) || *a.mode != 1ull; }
// pass A itself may be the one that sets the mode to 2 while some of its workgroups have yet to start
__device__ __forceinline__ bool part_skip_scatter_a(const PartArgs& a) {
  const unsigned long long m = *a.mode;
  return part_skip(a) || m == 0ull || m == 3ull;
}

// A record: the order index (11 bits of bucket number above `shift` bits inside the bucket) above the val_bits
// bits of the line's value: 25 + 39 bits up to 2^25 orders, 26 + 38 up to 2^26, 27 + 37 up to 2^27.
__device__ __forceinline__ unsigned long long part_val_mask(const PartArgs& a) { return (1ull << a.val_bits) - 1ull; }

// Position of a lane's records among the records of their bins in this round.  The lanes of a wave that share
// a bin take one returning add on the bin's counter between them (made by the lowest of them), so adds to one
// LDS word from inside a wave never collide, however the keys are distributed.  The lanes whose key equals
// this lane's are found with one ballot per key bit — only the bits in which two keys of the wave differ at
// all (one OR-reduction over the wave finds those): a round of pass B holds the buckets of one or two coarse
// bins, six or seven bits of twelve, and rows that are nearly grouped differ in fewer still.  K records of
// every lane are ranked at once.
__device__ __forceinline__ uint32_t part_wave_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int K>
__device__ __forceinline__ void part_rank_many(uint32_t* s_cnt, const uint32_t (&key)[K], int lane, uint32_t (&pos)[K]) {
  const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key[0]);
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) d |= key[i] ^ k0;
  unsigned long long m[K];
#pragma unroll
  for (int i = 0; i < K; ++i) m[i] = ~0ull;
  for (uint32_t bits = part_wave_or(d); bits; bits &= bits - 1u) {      // uniform
    const int b = __ffs((int)bits) - 1;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const bool bit = (key[i] >> b) & 1u;
      const unsigned long long s = __ballot(bit);
      m[i] &= bit ? s : ~s;
    }
  }
  // records of different slots i that share a bin: slot by slot, so that the counter orders them
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const unsigned long long below = lane == 0 ? 0ull : (m[i] & (~0ull >> (64 - lane)));
    const int leader = __ffsll((long long)m[i]) - 1;
    uint32_t base = 0;
    if (lane == leader) base = __hip_atomic_fetch_add(&s_cnt[key[i]], (uint32_t)__popcll(m[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = (uint32_t)__shfl((int)base, leader);
    pos[i] = base + (uint32_t)__popcll(below);
  }
}

// ---- order states, one bit per order ---------------------------------------------------------------
// Pass A needs the state of every line's order, and rows in any order ask for them at random: from the byte column
// (25 MB for 25 M orders) every look is a miss of the XCD's L2; one bit per order is 3 MB and stays there.
__global__ void __launch_bounds__(256) k_part_state_bits(PartArgs a) {
  if (part_skip(a)) return;
  const long long n_words = ((long long)a.n_orders + 31) / 32;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (long long)gridDim.x * blockDim.x) {
    const long long o0 = w * 32;
    uint32_t bits = 0u;
    if (o0 + 32 <= (long long)a.n_orders) {
      const uint4 x = *reinterpret_cast<const uint4*>(a.order_state + o0);        // (the column is 16-byte aligned)
      const uint4 y = *reinterpret_cast<const uint4*>(a.order_state + o0 + 16);
      const uint32_t v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t t = v[k] & 0x01010101u;                                    // bit 0 of four states
        bits |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xfu) << (4 * k);
      }
    } else {
      for (int k = 0; k < 32; ++k)
        if (o0 + k < (long long)a.n_orders && (a.order_state[o0 + k] & 1u)) bits |= 1u << k;
    }
    a.state_bits[w] = bits;
  }
}

// ---- lines per bucket ----------------------------------------------------------------------------
__global__ void __launch_bounds__(kPartThreads) k_part_count(PartArgs a) {
  __shared__ uint32_t s_cnt[kPartMaxBuckets];
  __shared__ uint32_t s_changes, s_descents;
  if (part_skip(a)) return;                               // uniform: the whole grid leaves
  for (int i = (int)threadIdx.x; i < kPartMaxBuckets; i += kPartThreads) s_cnt[i] = 0u;
  if (threadIdx.x == 0) { s_changes = 0u; s_descents = 0u; }
  __syncthreads();
  uint32_t st = 0;
  uint32_t changes = 0, descents = 0;
  const long long n4 = a.n_lines / 4;
  // Whether the ids ever go down is a yes or no: a wave that has met one place where they do stops looking
  // (rows in no order at all pay for the comparisons of a wave's first lines only).
  bool looking = true;                                      // uniform in the wave
  auto take4 = [&](const int4& v, int before) {
    const int ids[4] = {v.x, v.y, v.z, v.w};
    changes += (v.x != v.y) + (v.y != v.z) + (v.z != v.w);
    if (looking) descents += (before > v.x) + (v.x > v.y) + (v.y > v.z) + (v.z > v.w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t idx = (uint32_t)ids[i] - (uint32_t)a.base;
      const bool ok = idx < (uint32_t)a.n_orders;
      if (!ok) st |= kStOrderRange;
      // rows that are grouped after all put every lane of a wave into one bucket: those lanes share one add
      const uint32_t bucket = ok ? idx >> a.shift : 0xffffffffu;
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bucket);
      const unsigned long long same = __ballot(bucket == b0);
      const bool shared = __popcll(same) >= 8 && bucket == b0;
      if (shared) {
        if ((int)(threadIdx.x & 63) == __ffsll((long long)same) - 1 && ok)
          __hip_atomic_fetch_add(&s_cnt[bucket], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (ok) {
        __hip_atomic_fetch_add(&s_cnt[bucket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  // four 16-byte loads of a thread in flight
  const long long stride = (long long)gridDim.x * kPartThreads;
  const long long q0 = (long long)blockIdx.x * kPartThreads + (threadIdx.x & ~63u);      // the wave's first group
  const int lane = (int)(threadIdx.x & 63);
  int vzero = 0;
  asm volatile("" : "+v"(vzero));                       // 0, in a vector register
  for (long long qw = q0; qw < n4; qw += 4 * stride) {
    int4 v[4];
    bool in[4];
    int front[4];                                            // the id in front of the wave's first line of a group
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long qg = qw + (long long)k * stride;       // (uniform in the wave)
      const long long q = qg + lane;
      in[k] = q < n4;
      v[k] = in[k] ? load_stream16(a.order_id + 4 * q) : make_int4(0, 0, 0, 0);
      // (read as a vector load like the others: as a scalar load its wait would also wait for the LDS adds)
      front[k] = (looking && qg > 0 && qg < n4) ? a.order_id[4 * qg - 1 + vzero] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // the id in front of a lane's four: the last of the lane below, or (lane 0) the one in front of the group;
      // the very first line of the data has none
      int before = v[k].x;
      if (looking) {
        const int below = wave_shift_up1(v[k].w, front[k]);
        const bool first_line = lane == 0 && qw + (long long)k * stride == 0;
        if (!first_line) before = below;
      }
      if (in[k]) take4(v[k], before);                        // (the ballots in there count the lanes that are here)
    }
    if (looking) looking = __ballot(descents != 0u) == 0ull;
  }
  if (blockIdx.x == 0 && (long long)threadIdx.x < a.n_lines - 4 * n4) {
    const long long l = 4 * n4 + threadIdx.x;
    if (l > 0 && a.order_id[l - 1] > a.order_id[l]) ++descents;
    const uint32_t idx = (uint32_t)a.order_id[4 * n4 + threadIdx.x] - (uint32_t)a.base;
    if (idx < (uint32_t)a.n_orders) __hip_atomic_fetch_add(&s_cnt[idx >> a.shift], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else st |= kStOrderRange;
  }
  if (st) atomicOr(a.status, st);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { changes += __shfl_xor(changes, d); descents += __shfl_xor(descents, d); }
  // one add per workgroup: thousands of waves adding to one address wait for each other
  if ((threadIdx.x & 63) == 0 && changes) __hip_atomic_fetch_add(&s_changes, changes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if ((threadIdx.x & 63) == 0 && descents) s_descents = 1u;
  __syncthreads();
  if (threadIdx.x == 0 && s_changes) atomicAdd(a.changes, (unsigned long long)s_changes);
  if (threadIdx.x == 0 && s_descents) *a.descents = 1ull;   // (a yes or no: every writer stores the same)
  for (int i = (int)threadIdx.x; i < a.n_buckets; i += kPartThreads)
    if (s_cnt[i]) atomicAdd(&a.counts[i], (unsigned long long)s_cnt[i]);
}

// ---- offsets and cursors (one workgroup; two buckets per thread) -----------------------------------
__global__ void __launch_bounds__(kPartThreads) k_part_scan(PartArgs a) {
  __shared__ unsigned long long s_w[kPartWaves];
  if (part_skip(a)) {                                       // this path does nothing in this call: say so
    if (threadIdx.x == 0) *a.mode = 0ull;
    return;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long c0 = 2 * tid < a.n_buckets ? a.counts[2 * tid] : 0ull;
  const unsigned long long c1 = 2 * tid + 1 < a.n_buckets ? a.counts[2 * tid + 1] : 0ull;
  const long long incl = wave_inclusive_scan((long long)(c0 + c1));
  if (lane == 63) s_w[wave] = (unsigned long long)incl;
  __syncthreads();
  unsigned long long off = (unsigned long long)incl - (c0 + c1);
  for (int w = 0; w < wave; ++w) off += s_w[w];
  a.bucket_off[2 * tid] = off;
  a.bucket_off[2 * tid + 1] = off + c0;
  a.cursor_b[2 * tid] = off;
  a.cursor_b[2 * tid + 1] = off + c0;
  if (((2 * tid) & ((1 << kPartCoarseShift) - 1)) == 0) a.cursor_a[(2 * tid) >> kPartCoarseShift] = off;
  if (tid == kPartThreads - 1) a.bucket_off[kPartMaxBuckets] = off + c0 + c1;
  a.counts[2 * tid] = 0ull;                                 // clean for the next call
  a.counts[2 * tid + 1] = 0ull;
  if (tid == 0) {
    // three of every four places were looked at: changes per line is about (4/3) * counted / n_lines
    const unsigned long long c = *a.changes;
    *a.mode = *a.descents == 0ull ? 3ull : (c * 10ull > (unsigned long long)a.n_lines * 3ull) ? 1ull : 0ull;
    *a.changes = 0ull;
    *a.descents = 0ull;
  }
}

// The binning both scatter passes share, in two steps.  On entry every thread holds up to K records (bin, rec;
// bin == n_bins: not a record) and has taken its rank inside its bin from the round's counters; s_start holds
// the first staging slot of every bin.  part_stage_round puts the records bin by bin into LDS; after a barrier
// part_write_round writes the runs out, to s_gbase[bin] onwards.  The returning atomics that reserve the runs
// are issued before the staging and their results stored into s_gbase after it, so that they are on their way
// while the round is staged.
//   s_start [n_bins + 1]   first staging slot of every bin      s_gbase [n_bins]   first record of its run
template <int N_BINS, int K>
__device__ __forceinline__ void part_stage_round(const uint32_t (&bin)[K], const unsigned long long (&rec)[K],
                                                 const uint32_t (&pos)[K], const uint32_t* s_start,
                                                 unsigned long long* s_rec, unsigned short* s_bin) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    if (bin[i] < (uint32_t)N_BINS) {
      const uint32_t p = s_start[bin[i]] + pos[i];
      s_rec[p] = rec[i];
      s_bin[p] = (unsigned short)bin[i];
    }
  }
}
template <int N_BINS>
__device__ __forceinline__ void part_write_round(const uint32_t* s_start, const unsigned long long* s_gbase,
                                                 const unsigned long long* s_rec, const unsigned short* s_bin,
                                                 unsigned long long* rec_out) {
  const uint32_t total = s_start[N_BINS];
  for (uint32_t e = threadIdx.x; e < total; e += kPartThreads) {
    const uint32_t b = s_bin[e];
    rec_out[s_gbase[b] + (e - s_start[b])] = s_rec[e];
  }
}

// ---- pass A: lines -> (order index, value) records in 32 coarse bins; stock deltas --------------------
// LDS: [price int32 p_pad][histogram p_pad][counters 36][start 36][gbase 34 x 8][records 4096 x 8][bin 4096 x 2]
// kLife: the lifecycle columns are looked at (line prices, order states); a template parameter so that the plain
// kernel keeps its registers and its loop.
template <bool kLife>
__global__ void __launch_bounds__(kPartThreads) k_part_scatter_a(PartArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (part_skip_scatter_a(a)) return;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  int32_t* s_price = reinterpret_cast<int32_t*>(smem);
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem) + a.p_pad;
  uint32_t* s_cnt = s_hist + a.p_pad;                                     // [33] (+3): records per bin of this round
  uint32_t* s_start = s_cnt + kPartCntPitch;                              // [33] (+3)
  unsigned long long* s_gbase = reinterpret_cast<unsigned long long*>(s_start + kPartCntPitch);   // [32] (+2)
  unsigned long long* s_rec = s_gbase + 34;                               // [4096]
  unsigned short* s_bin = reinterpret_cast<unsigned short*>(s_rec + kPartRound);      // [4096]
  const bool hist = !a.totals_only;
  for (int i = tid; i < a.n_products + 1; i += kPartThreads) { s_price[i] = (int32_t)a.price[i]; s_hist[i] = 0u; }
  if (tid < kPartCntPitch) s_cnt[tid] = 0u;
  __syncthreads();
  const uint32_t P = (uint32_t)a.n_products;
  const long long n_rounds = (a.n_lines + kPartRound - 1) / kPartRound;
  const unsigned long long val_mask = part_val_mask(a);
  const int coarse_shift = a.shift + kPartCoarseShift;
  uint32_t st = 0;

  const bool snapped = kLife && a.line_price != nullptr;   // uniform
  const bool stated = kLife && a.state_bits != nullptr;
  const uint32_t top = (uint32_t)a.n_orders - 1u;           // (n_orders > 0 whenever there are lines)
  auto load_round = [&](long long r, int (&ids)[4], int (&prs)[4], int (&qs)[4], long long (&lp)[4]) {
    const long long l0 = r * kPartRound + (long long)tid * 4;
    if (l0 + 4 <= a.n_lines) {
      load_lines<4>(a.order_id + l0, ids);
      load_lines<4>(a.product_idx + l0, prs);
      load_lines<4>(a.qty + l0, qs);
      if (snapped) load_prices<4>(a.line_price + l0, lp);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool in = l0 + i < a.n_lines;
        ids[i] = in ? a.order_id[l0 + i] : a.base - 1;      // outside the order range: dropped below, not reported
        prs[i] = in ? a.product_idx[l0 + i] : -1;
        qs[i] = in ? a.qty[l0 + i] : 0;
        if (kLife) lp[i] = (snapped && in) ? a.line_price[l0 + i] : 0ll;
      }
    }
  };

  long long r = blockIdx.x;
  int nid[4], npr[4], nq[4];
  long long nlp[4] = {0, 0, 0, 0};
  if (r < n_rounds) load_round(r, nid, npr, nq, nlp);
  for (; r < n_rounds; r += gridDim.x) {
    int ids[4], prs[4], qs[4];
    long long lps[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ids[i] = nid[i]; prs[i] = npr[i]; qs[i] = nq[i]; lps[i] = nlp[i]; }
    const long long l0 = r * kPartRound + (long long)tid * 4;
    if (r + gridDim.x < n_rounds) load_round(r + gridDim.x, nid, npr, nq, nlp);     // in flight while this round is binned
    uint32_t bin[4], pos[4];
    unsigned long long rec[4];
    bool wide = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t idx = (uint32_t)ids[i] - (uint32_t)a.base;
      const bool ok = idx < (uint32_t)a.n_orders;
      if (!ok && l0 + i < a.n_lines) st |= kStOrderRange;
      bin[i] = ok ? (idx >> coarse_shift) : (uint32_t)kPartCoarse;
      const uint32_t pi = min((uint32_t)prs[i], P);
      // the line's own price when the column is there (an unknown product still adds nothing), else the table's
      const long long price = snapped ? (pi < P ? lps[i] : 0ll) : (long long)s_price[pi];
      const unsigned long long val = (unsigned long long)(price * (long long)qs[i]);
      wide |= ok && val > val_mask;
      rec[i] = ((unsigned long long)idx << a.val_bits) | (val & val_mask);
      if (hist) {
        // the lines of an order that does not count leave the stock deltas (its total is kept)
        int qh = qs[i];
        if (stated) { const uint32_t o = min(idx, top); qh = ((a.state_bits[o >> 5] >> (o & 31u)) & 1u) ? qs[i] : 0; }
        if ((uint32_t)qh < a.q_small)
          __hip_atomic_fetch_add(&s_hist[pi], (uint32_t)qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (pi < P)
          atomicAdd(&a.prod_carry[pi], (unsigned long long)(long long)qh);
      }
    }
    part_rank_many<4>(s_cnt, bin, lane, pos);
    __syncthreads();
    // one thread per bin: where the bin starts in the staging area; reserve its run
    unsigned long long g = 0;
    uint32_t mine = 0;
    if (tid <= kPartCoarse) {
      mine = tid < kPartCoarse ? s_cnt[tid] : 0u;                          // the drop bin takes no room
      s_cnt[tid] = 0u;                                                     // for the next round (three barriers from here)
      if (mine) g = atomicAdd(&a.cursor_a[tid], (unsigned long long)mine); // on its way during the staging
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
      s_start[tid] = incl - mine;                                          // s_start[32] == records of this round
    }
    __syncthreads();
    if (wide) __hip_atomic_store(a.mode, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the records are not used then)
    part_stage_round<kPartCoarse, 4>(bin, rec, pos, s_start, s_rec, s_bin);
    if (mine) s_gbase[tid] = g;
    __syncthreads();
    // (no barrier here: the next round touches the staging area only after its own three barriers)
    part_write_round<kPartCoarse>(s_start, s_gbase, s_rec, s_bin, a.rec_a);
  }
  if (st) atomicOr(a.status, st);
  if (hist) {
    uint32_t* slab = a.prod_slab + (size_t)blockIdx.x * a.p_pad;
    for (int i = tid; i < a.n_products; i += kPartThreads) slab[i] = s_hist[i];
  }
}

// ---- pass B: records of the coarse bins -> records of the buckets ----------------------------------
// Rounds of 8192 records (two groups of four per thread): the wait for the returning atomics that reserve
// the runs comes once per round, and a round's records fall into the 64 to 128 buckets of one or two coarse
// bins, so the longer the round the longer the runs.
// LDS: [counters 2052][start 2052][gbase 2048 x 8][records 8192 x 8][bin 8192 x 2]
constexpr int kPartRoundB = 2 * kPartRound;
__global__ void __launch_bounds__(kPartThreads) k_part_scatter_b(PartArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (part_skip_moving(a)) return;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem);                    // [2049] (+3): records per bucket of this round
  uint32_t* s_start = s_cnt + kPartMaxBuckets + 4;                        // [2049] (+3)
  unsigned long long* s_gbase = reinterpret_cast<unsigned long long*>(s_start + kPartMaxBuckets + 4);   // [2048]
  unsigned long long* s_rec = s_gbase + kPartMaxBuckets;
  unsigned short* s_bin = reinterpret_cast<unsigned short*>(s_rec + kPartRoundB);
  __shared__ uint32_t s_w[kPartWaves];
  for (int i = tid; i < kPartMaxBuckets + 4; i += kPartThreads) s_cnt[i] = 0u;
  __syncthreads();
  const long long n_rec = (long long)a.bucket_off[kPartMaxBuckets];
  const long long n_rounds = (n_rec + kPartRoundB - 1) / kPartRoundB;
  const int bucket_bits = a.val_bits + a.shift;             // a record's bucket number stands above these

  // A round is four quarters of 2048 records; thread t holds records 2t, 2t + 1 of each quarter: one 16-byte load
  // per quarter, contiguous over the workgroup.  (Four consecutive records per thread would make every load cover
  // half of each 64-byte piece it touches, and the vector cache would ask for every piece twice.)
  constexpr int kQuarter = kPartRoundB / 4;
  auto load_round = [&](long long r, unsigned long long (&rec)[8]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const long long e0 = r * kPartRoundB + (long long)h * kQuarter + (long long)tid * 2;
      if (e0 + 2 <= n_rec) {
        long long q[2];
        load_prices<2>(reinterpret_cast<const long long*>(a.rec_a + e0), q);
        rec[2 * h] = (unsigned long long)q[0];
        rec[2 * h + 1] = (unsigned long long)q[1];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) rec[2 * h + i] = e0 + i < n_rec ? a.rec_a[e0 + i] : 0ull;    // (past the end: dropped below)
      }
    }
  };

  long long r = blockIdx.x;
  unsigned long long nrec[8];
  if (r < n_rounds) load_round(r, nrec);
  for (; r < n_rounds; r += gridDim.x) {
    uint32_t bin[8], pos[8];
    unsigned long long rec[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rec[i] = nrec[i];
    if (r + gridDim.x < n_rounds) load_round(r + gridDim.x, nrec);
    // every bucket number is a record's; places past the last record go to the drop bin
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long e = r * kPartRoundB + (long long)(i >> 1) * kQuarter + (long long)tid * 2 + (i & 1);
      bin[i] = e < n_rec ? (uint32_t)(rec[i] >> bucket_bits) : (uint32_t)kPartMaxBuckets;
    }
    part_rank_many<8>(s_cnt, bin, lane, pos);
    __syncthreads();
    // one thread per pair of buckets: where they start in the staging area; reserve their runs
    const uint32_t t0 = s_cnt[2 * tid], t1 = s_cnt[2 * tid + 1];
    s_cnt[2 * tid] = 0u;                               // for the next round (three barriers from here)
    s_cnt[2 * tid + 1] = 0u;
    if (tid == 0) s_cnt[kPartMaxBuckets] = 0u;
    unsigned long long g0 = 0, g1 = 0;
    if (t0) g0 = atomicAdd(&a.cursor_b[2 * tid], (unsigned long long)t0);       // in flight during the scan
    if (t1) g1 = atomicAdd(&a.cursor_b[2 * tid + 1], (unsigned long long)t1);
    uint32_t incl = t0 + t1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t off = incl - (t0 + t1);
    for (int w = 0; w < wave; ++w) off += s_w[w];
    s_start[2 * tid] = off;
    s_start[2 * tid + 1] = off + t0;
    if (tid == kPartThreads - 1) s_start[kPartMaxBuckets] = off + t0 + t1;
    __syncthreads();
    part_stage_round<kPartMaxBuckets, 8>(bin, rec, pos, s_start, s_rec, s_bin);
    if (t0) s_gbase[2 * tid] = g0;                     // (the atomics have been on their way since before the scan)
    if (t1) s_gbase[2 * tid + 1] = g1;
    __syncthreads();
    // (no barrier here: the next round touches the staging area only after its own three barriers)
    part_write_round<kPartMaxBuckets>(s_start, s_gbase, s_rec, s_bin, a.rec_b);
  }
}

// ---- one workgroup per 16384 orders: their totals ----------------------------------------------------
// A bucket holds 2^shift orders; its 2^(shift - 14) workgroups (consecutive in the grid) all read the bucket's
// records and each keeps the ones of its own 16384 orders.
__global__ void __launch_bounds__(kPartThreads) k_part_reduce(PartArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (part_skip_moving(a)) return;
  unsigned long long* s_tab = reinterpret_cast<unsigned long long*>(smem);   // [16384]
  const int tid = (int)threadIdx.x;
  const int part_bits = a.shift - kPartShift;                // 0, 1 or 2
  const int b = (int)(blockIdx.x >> part_bits);
  const uint32_t part = blockIdx.x & ((1u << part_bits) - 1u);
  const unsigned long long val_mask = part_val_mask(a);
  const uint32_t in_bucket = (1u << a.shift) - 1u;
  for (int i = tid * 2; i < kPartBucket; i += kPartThreads * 2) *reinterpret_cast<uint4*>(s_tab + i) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const long long lo = (long long)a.bucket_off[b], hi = (long long)a.bucket_off[b + 1];
  // Sixteen records a thread in flight, as eight streaming 16-byte loads of two records each (a bucket starts at
  // any record: the pairs are taken from the even record at or below it, and records outside [lo, hi) are skipped).
  constexpr int kDepth = 8;
  const long long lo2 = lo & ~1ll;
  for (long long e = lo2 + 2 * tid; e < hi; e += 2 * kDepth * kPartThreads) {
    long long rec[kDepth][2];
#pragma unroll
    for (int k = 0; k < kDepth; ++k) {
      const long long ek = e + 2ll * k * kPartThreads;
      if (ek < hi) load_prices<2>(reinterpret_cast<const long long*>(a.rec_b + ek), rec[k]);   // (the workspace ends on an even record)
      else rec[k][0] = rec[k][1] = 0;
    }
#pragma unroll
    for (int k = 0; k < kDepth; ++k) {
      const long long ek = e + 2ll * k * kPartThreads;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned long long r = (unsigned long long)rec[k][j];
        const uint32_t cell = ((uint32_t)(r >> a.val_bits) & in_bucket) - part * (uint32_t)kPartBucket;   // < 16384: this part's
        if (ek + j >= lo && ek + j < hi && cell < (uint32_t)kPartBucket)
          __hip_atomic_fetch_add(&s_tab[cell], r & val_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  const long long o0 = ((long long)b << a.shift) + (long long)part * kPartBucket;
  long long n = (long long)a.n_orders - o0;                  // (may be <= 0 for the last bucket's later parts)
  if (n > kPartBucket) n = kPartBucket;
  // o0 is a multiple of 16384 and order_total is 16-byte aligned: two totals per store
  if (a.order_descuento) {                                   // uniform: the totals are stored net of venta.Descuento
    for (long long i = (long long)tid * 2; i + 1 < n; i += kPartThreads * 2) {
      longlong2 t;
      t.x = net_total((long long)s_tab[i], (uint32_t)(o0 + i), a.order_descuento, a.descuento_mode);
      t.y = net_total((long long)s_tab[i + 1], (uint32_t)(o0 + i + 1), a.order_descuento, a.descuento_mode);
      *reinterpret_cast<longlong2*>(a.order_total + o0 + i) = t;
    }
    if (n > 0 && (n & 1) && tid == 0)
      a.order_total[o0 + n - 1] = net_total((long long)s_tab[n - 1], (uint32_t)(o0 + n - 1), a.order_descuento, a.descuento_mode);
    return;
  }
  for (long long i = (long long)tid * 2; i + 1 < n; i += kPartThreads * 2)
    *reinterpret_cast<uint4*>(a.order_total + o0 + i) = *reinterpret_cast<const uint4*>(s_tab + i);
  if (n > 0 && (n & 1) && tid == 0) a.order_total[o0 + n - 1] = (int64_t)s_tab[n - 1];
}

}  // namespace acmpc
