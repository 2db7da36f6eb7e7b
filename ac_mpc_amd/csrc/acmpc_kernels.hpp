// acmpc_kernels.hpp — gfx950 (MI355X, CDNA4) device code of the checkout-aggregation engine.
//
// Data layout in HBM (all columns are plain dense arrays):
//   order_id[N], product_idx[N], qty[N]   int32   the `detalle` table (store.sql:97-101), rows of
//                                                  one order contiguous, order ids non-decreasing
//   order_customer[M]                      int32   venta.NIT as a dictionary index (store.sql:161)
//   price_cents[P]                         int64   producto.Precio in cents (store.sql:113)
//   order_total_cents[M], product_qty[P], customer_total_cents[C]   int64 outputs
//
// Kernels (all HBM-bandwidth bound; there is no dense contraction anywhere, so no MFMA):
//   k_lines      one pass over the three line columns (12 B/line read, 8 B/order written).
//                One workgroup per CU keeps the price table and a private stock-delta
//                histogram in LDS; every WAVE owns a contiguous run of 256-line tiles and
//                carries the open order from tile to tile in registers, so the main loop has
//                no workgroup barrier.
//   k_fixup      block 0 resolves the orders cut by a wave boundary (two records per wave);
//                the other blocks merge the per-workgroup histograms.
//   k_customers  per-order pass (4 B + 8 B read per order): customer ranges are split over
//                sibling workgroups that read the same orders (from L2) and each keep their
//                range of partial sums in LDS.
//   k_custmerge  merges the per-stream customer partials.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acmpc {

constexpr int kWave = 64;
constexpr int kLinesPerLane = 4;                  // one 16-byte load per column per lane
constexpr int kTile = kWave * kLinesPerLane;      // 256 lines (or orders) per wave-tile
constexpr int kMaxWavesTotal = 4096;              // k_lines waves; two records each for k_fixup
constexpr int kFixThreads = 1024;
constexpr int kFixStrip = 8;                      // records per thread in k_fixup block 0
constexpr int kMergeProducts = 64;                // products per merge block in k_fixup
constexpr int kLinesDepth = 1;                    // line tiles in flight per wave in k_lines
constexpr int kCustDepth = 4;                     // order tiles in flight per wave in k_customers

constexpr uint32_t kStOrderRange = 1u, kStUnsorted = 2u, kStCustomerRange = 4u;

struct LinesArgs {
  const int32_t* order_id;
  const int32_t* product_idx;
  const int32_t* qty;
  int64_t n_lines;
  int64_t n_tiles;          // ceil(n_lines / tile_lines)
  int64_t tiles_per_wave;
  int32_t base;             // order_id_base
  int32_t n_orders;         // M
  const int64_t* price;     // [P + 1] device; entry P is 0 (the null product)
  int32_t n_products;
  int32_t p_pad;            // P + 1 rounded up to a multiple of 4
  uint32_t q_small;         // quantities in [0, q_small) use the 32-bit LDS histogram
  uint32_t stage_cap;       // staging slots per wave (orders buffered in LDS before they are written)
  uint32_t flush_reserve;   // flush when fewer than this many slots are left (lines per tile)
  int64_t tile_lines;       // 64 * lines per lane
  int64_t* order_total;     // [M]
  uint32_t* prod_slab;      // [gridDim.x][p_pad] low words of the per-workgroup stock deltas
  unsigned long long* prod_carry;   // [P] high-word corrections (LDS mode) / product_qty itself (global mode)
  int32_t* rec_id;          // [2 * waves]  head/tail order id of every wave
  int64_t* rec_sum;         // [2 * waves]  head/tail partial sums
  uint32_t* status;
  // order lifecycle (kLife kernels only; each may be null)
  const long long* line_price;        // [n_lines] price at the time of sale, instead of the table
  int32_t price_in_lds;               // 0: no price table in LDS (line prices make it unnecessary, or it does not fit)
  // k_lines_any as the fallback of the grouped kernel: run only if the grouped pass found the ids
  // unsorted, and redo the order totals only (stock deltas do not depend on the row order)
  int32_t only_if_unsorted;
  int32_t totals_only;
  // k_lines: 1 = every wave also zeroes the totals of the orders without lines that lie between the
  // previous wave's last order and its own first one (the first wave: before its first order; the last
  // wave: after its last order).  Then no later kernel has to write zeros before the totals are read.
  int32_t fill_edges;
  // k_lines: > 0 = the waves of a workgroup write their closed orders out together every `burst` tiles
  // (a workgroup barrier, then every wave flushes): the memory system takes the order totals better in
  // bursts than trickling in between the reads.  Only in workgroups whose waves all have full runs.
  int32_t burst;
  const unsigned char* order_state;   // [M] bit 0: the order counts towards stock and customer sums
  int32_t no_hist;                    // the stock deltas are left to k_stock (global-table kernels, k_lines_any)
  const int32_t* order_descuento;     // [M] or null (grouped kLife kernels): totals are stored net of it
  int32_t descuento_mode;
  // k_lines_any: when not null, the word in which the order-restoring path (acmpc_partition.hpp) says what it
  // did: 0 nothing (this kernel runs in full), 1 everything (this kernel leaves), 2 the stock deltas only,
  // 3 nothing, because the rows turned out to be grouped (this kernel leaves; the grouped kernel runs)
  const unsigned long long* run_if_zero;
  // k_lines: when not null, run only if that word is 3
  const unsigned long long* run_if_sorted;
};

// Lifecycle: quantities that reach the stock delta (0 for lines of orders that do not count).
template <int LPL>
__device__ __forceinline__ void load_states(const LinesArgs& a, const int (&ids)[LPL], uint32_t (&s)[LPL]) {
  if (a.order_state) {
    const uint32_t top = (uint32_t)a.n_orders - 1u;         // n_orders > 0 whenever there are lines
#pragma unroll
    for (int i = 0; i < LPL; ++i) s[i] = a.order_state[min((uint32_t)ids[i] - (uint32_t)a.base, top)];
  } else {
#pragma unroll
    for (int i = 0; i < LPL; ++i) s[i] = 1u;
  }
}

// The quantities that reach the stock delta: those of lines whose order counts (bit 0 of its state).
// Loading the states (load_states) is kept apart so that a tile loop can issue those loads before the
// next tile's: vector loads return in order, and a wait for the states would otherwise be a wait for the
// tile that was just requested.
template <int LPL>
__device__ __forceinline__ void live_quantities(const int (&qs)[LPL], const uint32_t (&s)[LPL], int (&qh)[LPL]) {
#pragma unroll
  for (int i = 0; i < LPL; ++i) qh[i] = (s[i] & 1u) ? qs[i] : 0;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// 16-byte streaming load: the line columns are read exactly once, so they are loaded with the
// non-temporal hint and do not displace the output lines and tables that the caches should keep.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 load_stream16(const int32_t* p) {
  const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
  return make_int4(v.x, v.y, v.z, v.w);
}

// Inclusive prefix sum of a 64-bit value over the 64 lanes, entirely in DPP adds with carry (no LDS
// traffic, no waits): row_shr 1,2,4,8 inside each 16-lane row, then the row totals are broadcast
// into the following rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
// Each step is v_add_co_u32 / v_addc_co_u32 with the DPP operand on the running value itself.
// hipcc does not see inside an asm block, so the two wait states a DPP read needs after a VALU
// write of the same register are spelled out as s_nop.
__device__ __forceinline__ long long wave_inclusive_scan(long long v) {
  unsigned int lo = (unsigned int)v, hi = (unsigned int)((unsigned long long)v >> 32);
  asm volatile(
      "s_nop 1\n\t"
      "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 0\n\t"
      "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 0\n\t"
      "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 0\n\t"
      "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 0\n\t"
      "v_add_co_u32_dpp %0, vcc, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_co_u32_dpp %0, vcc, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "+v"(lo), "+v"(hi)
      :
      : "vcc");
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// Value of lane-1 (lane 0 receives `first`).
__device__ __forceinline__ int wave_shift_up1(int v, int first) {
  return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

__device__ __forceinline__ long long readlane64(long long v, int l) {
  const int lo = __builtin_amdgcn_readlane((int)v, l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}

// Add a 64-bit value to an accumulator that is split into a 32-bit low word in LDS and a (rare)
// high-word correction in global memory.  Exact modulo 2^64.
__device__ __forceinline__ void split_add64(uint32_t* lds_lo, unsigned long long* g_hi, long long v) {
  const uint32_t lo = (uint32_t)v;
  const uint32_t old = atomicAdd(lds_lo, lo);
  const long long carry = ((uint32_t)(old + lo) < old) ? 1 : 0;
  const long long hi = (v >> 32) + carry;
  if (hi != 0) atomicAdd(g_hi, ((unsigned long long)hi) << 32);
}

// venta.Descuento (INT, store.sql:162; always 0 in the reference, process/confirmcompra.php:24):
// order_total[o] -= discount.  mode 0: the column is an amount in whole currency units; mode 1: a
// percentage of the gross total, rounded half away from zero to a cent.  Wrapping 64-bit arithmetic.
__device__ __forceinline__ long long descuento_cents(long long gross, int d, int mode) {
  if (mode == 0) return 100ll * (long long)d;
  if ((unsigned long long)gross < (1ull << 32) && (uint32_t)d < (1u << 25)) {
    // the usual sizes: everything fits 32-bit unsigned arithmetic, where a division by 100 is a multiply and a
    // shift (the 64-bit divisions below are some hundred instructions each); the same value as the general form
    const uint32_t g = (uint32_t)gross, q = g / 100u, r = g - q * 100u;
    const uint32_t part = (r * (uint32_t)d + 50u) / 100u;         // r * d < 100 * 2^25
    return (long long)((unsigned long long)q * (unsigned long long)(uint32_t)d + part);
  }
  const long long q = gross / 100ll, r = gross - q * 100ll;       // truncating: r has the sign of gross
  const long long x = r * (long long)d;                           // |x| < 100 * 2^31
  const long long part = x >= 0 ? (x + 50ll) / 100ll : -((-x + 50ll) / 100ll);
  return (long long)((unsigned long long)q * (unsigned long long)(long long)d) + part;
}

// The total that is stored for order index idx: the gross total less its discount when the kernel has
// been given the Descuento column (grouped lifecycle kernels; otherwise k_order_adjust does it afterwards).
__device__ __forceinline__ long long net_total(long long gross, uint32_t idx, const int32_t* descuento, int mode) {
  if (!descuento) return gross;
  return (long long)((unsigned long long)gross - (unsigned long long)descuento_cents(gross, descuento[idx], mode));
}

// Zero order_total for the order ids strictly between a and b (exclusive), all 64 lanes helping.
// (64-bit bounds: base - 1 and base + n_orders may lie outside int32.)
template <bool kLife = false>
__device__ __forceinline__ void wave_fill_gap(const LinesArgs& a, long long ida, long long idb, int lane) {
  for (long long k = ida + 1 + lane; k < idb; k += kWave) {
    const uint32_t idx = (uint32_t)((int)k) - (uint32_t)a.base;
    if (idx < (uint32_t)a.n_orders) a.order_total[idx] = kLife ? net_total(0, idx, a.order_descuento, a.descuento_mode) : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// k_lines
// ---------------------------------------------------------------------------------------------
// Per-wave state carried from tile to tile.
struct WaveRun {
  int id;               // order id of the last line seen (the order still open)
  long long sum;        // its partial total so far
  bool first_open;      // no order boundary seen yet in this wave's run
  long long head_sum;   // partial total of the run's first order (valid once first_open is false)
  uint32_t st;          // data-contract flags raised by this lane
  uint32_t buf_base;    // order index (id - base) that staging slot 0 stands for
};

// A staging slot holds a finished order total as an unsigned 32-bit number: half the LDS of a 64-bit
// slot, so twice the orders fit between two flushes.  A tile in which a total might not fit (any line
// value or running sum outside [0, 2^32)) takes the rare path, which writes 64-bit totals directly.
using StageT = uint32_t;

// Write the finished part of the wave's staging buffer — order indices [buf_base, buf_base + n) —
// to order_total as contiguous 8-byte stores, and leave the buffer zeroed.  Slots never written
// belong to orders without lines and flush as 0.
template <bool kLife = false>
__device__ __forceinline__ void stage_flush(const LinesArgs& a, StageT* s_stage, WaveRun& o, uint32_t n, int lane) {
  if (kLife && a.order_descuento) {
    // with discounts: the discounts of six rounds of 64 orders are requested together, then taken off (one
    // round at a time every store would wait a full trip to memory for its discount)
    constexpr int kBatch = 6;
    for (uint32_t k0 = (uint32_t)lane; k0 < n; k0 += kBatch * kWave) {
      int dsc[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const uint32_t k = k0 + (uint32_t)j * kWave, idx = o.buf_base + k;
        dsc[j] = (k < n && idx < (uint32_t)a.n_orders) ? a.order_descuento[idx] : 0;
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const uint32_t k = k0 + (uint32_t)j * kWave, idx = o.buf_base + k;
        if (k < n) {
          const long long val = (long long)(unsigned long long)s_stage[k];
          s_stage[k] = 0;
          if (idx < (uint32_t)a.n_orders)
            a.order_total[idx] = (long long)((unsigned long long)val - (unsigned long long)descuento_cents(val, dsc[j], a.descuento_mode));
          else o.st |= kStOrderRange;
        }
      }
    }
    o.buf_base += n;
    return;
  }
  for (uint32_t k = (uint32_t)lane; k < n; k += kWave) {
    const long long val = (long long)(unsigned long long)s_stage[k];
    s_stage[k] = 0;
    const uint32_t idx = o.buf_base + k;
    if (idx < (uint32_t)a.n_orders) a.order_total[idx] = kLife ? net_total(val, idx, a.order_descuento, a.descuento_mode) : val;
    else o.st |= kStOrderRange;
  }
  o.buf_base += n;
}

// One tile of 64 * LPL lines: lane l holds lines LPL*l .. LPL*l+LPL-1 of the tile in (ids, prs, qs).
//
// Tables carry one extra entry, index P, the "null product": price 0 and a scratch histogram
// cell.  Any product_idx outside [0, P) is clamped to it with one unsigned min, so unknown codes
// need no branch and add nothing.
//
// kLife: `qh` holds the quantities that reach the stock delta (see live_quantities) and, when `snapped`,
// snap[i] is the line's own price; otherwise qh is qs and the table price is used.  (snap is an array
// reference, not a pointer: indexed through a pointer that may be null the prices end up in scratch
// memory, and waiting for a scratch load also waits for the next tile's loads, which were issued first.)
template <typename PriceT, bool kLdsTables, int LPL, bool kLife = false, bool kPriceGlobal = false, bool kNoHist = false>
__device__ __forceinline__ void lines_tile(const LinesArgs& a, const PriceT* s_price, uint32_t* s_hist,
                                           StageT* s_stage, const int (&ids)[LPL], const int (&prs)[LPL],
                                           const int (&qs)[LPL], int lane, uint32_t lt_lo, uint32_t lt_hi,
                                           WaveRun& o, const int (&qh)[LPL], const long long (&snap)[LPL],
                                           bool snapped) {
  const uint32_t P = (uint32_t)a.n_products;
  uint32_t pi[LPL];
  long long v[LPL];
  uint32_t qmax = 0;               // largest quantity (as unsigned) seen by this lane
#pragma unroll
  for (int i = 0; i < LPL; ++i) pi[i] = min((uint32_t)prs[i], P);
  if (kLdsTables) {
    // Stock delta: every quantity is added to the 32-bit LDS histogram with a non-returning add.
    // The host sizes q_small so that a workgroup's sum of quantities below q_small cannot wrap; a
    // quantity outside [0, q_small) (never produced by the reference, which always writes 1) is
    // detected through qmax, taken back out of the histogram and sent to a 64-bit global atomic.
    if constexpr (!kNoHist) {
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        __hip_atomic_fetch_add(&s_hist[pi[i]], (uint32_t)qh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        qmax = max(qmax, (uint32_t)qh[i]);
      }
    }
    if (kLife && snapped) {
#pragma unroll
      for (int i = 0; i < LPL; ++i) v[i] = (pi[i] < P ? snap[i] : 0ll) * (long long)qs[i];
    } else if (!kPriceGlobal) {
      PriceT price[LPL];
#pragma unroll
      for (int i = 0; i < LPL; ++i) price[i] = s_price[pi[i]];
#pragma unroll
      for (int i = 0; i < LPL; ++i) v[i] = (long long)price[i] * (long long)qs[i];
    } else {
      // the price table did not fit beside the histogram: gathered from global memory (it stays in L2)
      long long price[LPL];
#pragma unroll
      for (int i = 0; i < LPL; ++i) price[i] = a.price[pi[i]];
#pragma unroll
      for (int i = 0; i < LPL; ++i) v[i] = price[i] * (long long)qs[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      const long long pr = (kLife && snapped) ? (pi[i] < P ? snap[i] : 0ll) : a.price[pi[i]];
      v[i] = pr * (long long)qs[i];
      if (pi[i] < P && !a.no_hist) atomicAdd(&a.prod_carry[pi[i]], (unsigned long long)(long long)qh[i]);
    }
  }

  // order boundaries inside the lane's lines: d[i] != 0 where line i starts a new order
  const int prev = wave_shift_up1(ids[LPL - 1], o.id);
  int d[LPL];
  d[0] = ids[0] - prev;
#pragma unroll
  for (int i = 1; i < LPL; ++i) d[i] = ids[i] - ids[i - 1];
  int dmin = d[0], dor = d[0];
#pragma unroll
  for (int i = 1; i < LPL; ++i) { dmin = min(dmin, d[i]); dor |= d[i]; }
  // r: sum of the lane's lines since its last boundary (all of them if it has none); tot: all of them
  long long r = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < LPL; ++i) {
    r = d[i] ? 0ll : r;
    r += v[i];
    tot += v[i];
  }
  // incoming[l] = partial total, at the start of lane l, of the order that is open there:
  // everything since the nearest boundary in a lower lane, or the carry plus everything so far.
  const long long incl = wave_inclusive_scan(tot);
  const long long excl = incl - tot;
  const unsigned long long headmask = __ballot(dor != 0);
  const uint32_t below_lo = (uint32_t)headmask & lt_lo, below_hi = (uint32_t)(headmask >> 32) & lt_hi;
  const bool any_below = (below_lo | below_hi) != 0u;
  const int j = below_hi ? (63 - __clz(below_hi)) : (31 - __clz(below_lo | 1u));
  const long long gj = __shfl(r - incl, j);                 // (sum since last boundary) - incl, of lane j
  const long long incoming = excl + (any_below ? gj : o.sum);

  if (o.first_open && headmask != 0ull) {                     // wave-uniform, once per run
    // pre: lines of the first boundary lane that still belong to the run's first order
    long long pre = 0;
    bool open = true;
#pragma unroll
    for (int i = 0; i < LPL - 1; ++i) { open &= d[i] == 0; pre += open ? v[i] : 0ll; }
    const int L = __ffsll((long long)headmask) - 1;
    o.head_sum = readlane64(incoming + pre, L);
    o.first_open = false;
  }

  const int new_id = __builtin_amdgcn_readlane(ids[LPL - 1], kWave - 1);
  const uint32_t span = ((uint32_t)new_id - (uint32_t)a.base) - o.buf_base;   // slots this tile can touch
  // Every running sum of this lane lies between 0 and incoming + tot when incoming and the line values
  // are not negative; all of them below 2^32 is what a staging slot needs.
  unsigned long long wide = (unsigned long long)incoming | (unsigned long long)(incoming + tot);
#pragma unroll
  for (int i = 0; i < LPL; ++i) wide |= (unsigned long long)v[i];
  const bool lane_rare = (kLdsTables && qmax >= a.q_small) || dmin < 0 || (wide >> 32) != 0ull;
  const unsigned long long rare_mask = __ballot(lane_rare);
  long long run = incoming;
  int pid = prev;

  if (rare_mask == 0ull && span <= a.stage_cap) {
    // ---- common path: every order that closes in this tile goes to its staging slot; a lane whose
    // line i is not a boundary writes its scratch slot (stage_cap + lane) instead of branching, and
    // the same min() keeps any slot inside the wave's slice.
    const uint32_t off = (uint32_t)a.base + o.buf_base;
    const uint32_t scratch = a.stage_cap + (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      const uint32_t slot = (uint32_t)pid - off;
      const uint32_t w = min(d[i] ? slot : 0xffffffffu, scratch);
      s_stage[w] = (StageT)run;
      run = d[i] ? 0ll : run;
      run += v[i];
      pid = ids[i];
    }
    o.id = new_id;
    o.sum = readlane64(run, kWave - 1);
    if (span > a.stage_cap - a.flush_reserve) stage_flush<kLife>(a, s_stage, o, span, lane);
    return;
  }

  // ---- rare path: quantities outside [0, q_small), unsorted ids, a total that may need more than 32
  // bits, or a jump in order ids wider than the staging buffer.  Empty the buffer, then write this tile's orders (and the ids they skip)
  // directly.
  stage_flush<kLife>(a, s_stage, o, min(((uint32_t)o.id - (uint32_t)a.base) - o.buf_base, a.stage_cap), lane);
  if (kLdsTables && !kNoHist) {
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      if ((uint32_t)qh[i] >= a.q_small) {
        __hip_atomic_fetch_add(&s_hist[pi[i]], 0u - (uint32_t)qh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pi[i] < P) atomicAdd(&a.prod_carry[pi[i]], (unsigned long long)(long long)qh[i]);
      }
    }
  }
  bool gap = false;
#pragma unroll
  for (int i = 0; i < LPL; ++i) {
    if (ids[i] != pid) {
      const uint32_t idx = (uint32_t)pid - (uint32_t)a.base;
      if (idx < (uint32_t)a.n_orders) a.order_total[idx] = kLife ? net_total(run, idx, a.order_descuento, a.descuento_mode) : run;
      else o.st |= kStOrderRange;
      if (ids[i] < pid) o.st |= kStUnsorted;
      else if ((uint32_t)ids[i] - (uint32_t)pid > 1u) gap = true;
      run = 0;
    }
    run += v[i];
    pid = ids[i];
  }
  unsigned long long gm = __ballot(gap);
  while (gm) {                                                 // orders with no lines
    const int L = __ffsll((long long)gm) - 1;
    gm &= gm - 1ull;
    int before = __builtin_amdgcn_readlane(prev, L);
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      const int here = __builtin_amdgcn_readlane(ids[i], L);
      wave_fill_gap<kLife>(a, before, here, lane);
      before = here;
    }
  }
  o.id = new_id;
  o.sum = readlane64(run, kWave - 1);
  o.buf_base = (uint32_t)new_id - (uint32_t)a.base;            // staging restarts at the open order
}

// LPL consecutive int32 of one column for this lane, as 16-byte streaming loads.
template <int LPL>
__device__ __forceinline__ void load_lines(const int32_t* p, int (&out)[LPL]) {
#pragma unroll
  for (int k = 0; k < LPL / 4; ++k) {
    const int4 x = load_stream16(p + 4 * k);
    out[4 * k] = x.x; out[4 * k + 1] = x.y; out[4 * k + 2] = x.z; out[4 * k + 3] = x.w;
  }
}

// The 256 int64 line prices of a tile (2 KB, `tile` points at the tile's first line) as two fully
// coalesced 16-byte streaming loads per lane: load k brings prices 128k + 2*lane and 128k + 2*lane + 1.
// (Letting every lane load its own four prices directly makes each load touch half of every cache
// line, and the lines are then fetched twice.)
__device__ __forceinline__ void load_prices_raw(const long long* tile, int lane, int4 (&raw)[2]) {
  const int32_t* p = reinterpret_cast<const int32_t*>(tile);
  raw[0] = load_stream16(p + lane * 4);
  raw[1] = load_stream16(p + 256 + lane * 4);
}

// Hands every lane the prices of its own four lines (4*lane .. 4*lane+3) through the wave's 2 KB LDS
// slice.  LDS operations of one wave execute in order, so no barrier is needed.
__device__ __forceinline__ void own_prices(int4* xp, int lane, const int4 (&raw)[2], long long (&out)[4]) {
  xp[lane] = raw[0];
  xp[64 + lane] = raw[1];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int4 a = xp[2 * lane], b = xp[2 * lane + 1];
  out[0] = (long long)(((unsigned long long)(uint32_t)a.y << 32) | (uint32_t)a.x);
  out[1] = (long long)(((unsigned long long)(uint32_t)a.w << 32) | (uint32_t)a.z);
  out[2] = (long long)(((unsigned long long)(uint32_t)b.y << 32) | (uint32_t)b.x);
  out[3] = (long long)(((unsigned long long)(uint32_t)b.w << 32) | (uint32_t)b.z);
  __builtin_amdgcn_wave_barrier();                   // the slice is rewritten for the next tile
}

// LPL consecutive int64 prices of the line-price column for this lane (strided; any-order kernel).
template <int LPL>
__device__ __forceinline__ void load_prices(const long long* p, long long (&out)[LPL]) {
#pragma unroll
  for (int k = 0; k < LPL / 2; ++k) {
    const int4 x = load_stream16(reinterpret_cast<const int32_t*>(p + 2 * k));
    out[2 * k] = (long long)(((unsigned long long)(uint32_t)x.y << 32) | (uint32_t)x.x);
    out[2 * k + 1] = (long long)(((unsigned long long)(uint32_t)x.w << 32) | (uint32_t)x.z);
  }
}

// kSnap (lifecycle only): the line-price column is present.  A template parameter rather than a test of
// the pointer, so that the loads of that column and their uses are on one path for the compiler: with a
// run-time test it has to assume that a load into the same registers may still be pending from a path
// that never happens, and waits for it (and for everything requested before it) in the middle of the loop.
// kPriceGlobal: only the stock-delta histogram is in LDS, the prices are gathered from global memory (a
// product table too large for both; a template parameter because a run-time test in the tile routine
// costs the common case 8 %).
// kNoHist: only the price table is in LDS (both do not fit, and reading prices from LDS matters more):
// the stock deltas are left to k_stock.
template <typename PriceT, bool kLdsTables, int LPL, bool kLife = false, bool kSnap = false, bool kPriceGlobal = false,
          bool kNoHist = false>
__global__ void __launch_bounds__(1024) k_lines(LinesArgs a) {
  static_assert(!kLife || LPL == 4, "lifecycle columns are read four lines per lane");
  static_assert(kLife || !kSnap, "line prices are a lifecycle column");
  constexpr int kTileL = kWave * LPL;                 // lines per wave-tile
  extern __shared__ __align__(16) unsigned char smem[];
  if (a.run_if_sorted && *a.run_if_sorted != 3ull) return;   // uniform (rows in any order that were not grouped after all)
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int waves_per_block = (int)(blockDim.x >> 6);
  const int P1 = a.n_products + 1;                    // tables include the null product
  // LDS: [price table][stock-delta histogram] (only with kLdsTables), then per wave a staging slice
  // of stage_cap slots followed by 64 scratch slots.
  // With line prices (lifecycle) the price table is left out and every wave gets a 2 KB slice, after
  // the staging slices, through which it re-orders the coalesced price loads.
  constexpr bool price_lds = !kSnap && !kPriceGlobal;
  const size_t price_bytes = (kLdsTables && price_lds) ? (size_t)a.p_pad * sizeof(PriceT) : 0;
  const size_t tables = kLdsTables ? price_bytes + (kNoHist ? 0 : (size_t)a.p_pad * sizeof(uint32_t)) : 0;
  PriceT* s_price = reinterpret_cast<PriceT*>(smem);
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + price_bytes);
  StageT* s_stage = reinterpret_cast<StageT*>(smem + tables) + (size_t)wave * (a.stage_cap + kWave);
  int4* s_xpose = reinterpret_cast<int4*>(smem + tables + (size_t)waves_per_block * (a.stage_cap + kWave) * sizeof(StageT)) + wave * 128;

  for (uint32_t k = (uint32_t)lane; k < a.stage_cap; k += kWave) s_stage[k] = 0;
  if (kLdsTables) {
    for (int i = (int)threadIdx.x; i < P1; i += (int)blockDim.x) {
      if (price_lds) s_price[i] = (PriceT)a.price[i];
      if (!kNoHist) s_hist[i] = 0u;
    }
    __syncthreads();
  }
  const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xffffffffu;      // lanes below this one
  const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);

  const long long g = (long long)blockIdx.x * waves_per_block + wave;   // global wave index
  const long long t0 = g * a.tiles_per_wave;
  long long t1 = t0 + a.tiles_per_wave;
  if (t1 > a.n_tiles) t1 = a.n_tiles;

  if (t0 < t1) {
    const long long n_full = a.n_lines / kTileL;     // tiles [0, n_full) are complete
    const long long tf = t1 < n_full ? t1 : n_full;  // this wave's complete tiles are [t0, tf)
    WaveRun o;
    o.id = a.order_id[t0 * kTileL];                  // the wave's run starts inside this order
    o.sum = 0;
    o.first_open = true;
    o.head_sum = 0;
    o.st = 0;
    o.buf_base = (uint32_t)o.id - (uint32_t)a.base;
    const int head_id = o.id;

    if (t0 < tf) {
      // software pipeline: the next tile's loads are in flight while this tile is processed.  The
      // last iteration re-loads its own tile instead of branching around the loads.
      const long long last = tf - 1;
      long long off = t0 * kTileL + lane * LPL;
      int nid[LPL], npr[LPL], nq[LPL];
      int4 nraw[2] = {};
      constexpr bool snapped = kSnap;
      load_lines<LPL>(a.order_id + off, nid);
      load_lines<LPL>(a.product_idx + off, npr);
      load_lines<LPL>(a.qty + off, nq);
      if constexpr (kLife) { if (snapped) load_prices_raw(a.line_price + (off - lane * LPL), lane, nraw); }
      const bool bursts = a.burst > 0 &&
                          (long long)(blockIdx.x + 1) * waves_per_block * a.tiles_per_wave <= n_full;   // workgroup-uniform
      int until_burst = a.burst;
      for (long long t = t0; t <= last; ++t) {
        int ids[LPL], prs[LPL], qs[LPL];
        long long sn[LPL] = {};
#pragma unroll
        for (int i = 0; i < LPL; ++i) { ids[i] = nid[i]; prs[i] = npr[i]; qs[i] = nq[i]; }
        if constexpr (kLife) { if (snapped) own_prices(s_xpose, lane, nraw, sn); }
        uint32_t stt[LPL];
        if constexpr (kLife) load_states<LPL>(a, ids, stt);      // before the next tile's loads: see live_quantities
        off += (t < last) ? kTileL : 0;
        load_lines<LPL>(a.order_id + off, nid);
        load_lines<LPL>(a.product_idx + off, npr);
        load_lines<LPL>(a.qty + off, nq);
        if constexpr (kLife) { if (snapped) load_prices_raw(a.line_price + (off - lane * LPL), lane, nraw); }
        if constexpr (kLife) {
          // keep the uses of the states behind the loads just issued (the scheduler would otherwise move
          // them, and the wait for the states, in front of the next tile's loads)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < LPL; ++i) asm volatile("" : "+v"(stt[i]));   // the states are first looked at here
          int qh[LPL];
          live_quantities<LPL>(qs, stt, qh);
          lines_tile<PriceT, kLdsTables, LPL, true, kPriceGlobal, kNoHist>(a, s_price, s_hist, s_stage, ids, prs, qs, lane, lt_lo, lt_hi, o, qh,
                                                    sn, snapped);
        } else {
          lines_tile<PriceT, kLdsTables, LPL, false, kPriceGlobal, kNoHist>(a, s_price, s_hist, s_stage, ids, prs, qs, lane, lt_lo, lt_hi, o, qs, sn, false);
        }
        if (bursts && --until_burst == 0) {
          until_burst = a.burst;
          __syncthreads();
          stage_flush<kLife>(a, s_stage, o, min(((uint32_t)o.id - (uint32_t)a.base) - o.buf_base, a.stage_cap), lane);
        }
      }
    }
    if (tf < t1) {
      // the data set's last, partial tile: lines past the end repeat the last order id, add nothing
      const long long l0 = tf * kTileL + (long long)lane * LPL;
      const int last_id = a.order_id[a.n_lines - 1];
      int ids[LPL], prs[LPL], qs[LPL];
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        const long long l = l0 + i;
        const bool in = l < a.n_lines;
        ids[i] = in ? a.order_id[l] : last_id;
        prs[i] = in ? a.product_idx[l] : -1;
        qs[i] = in ? a.qty[l] : 0;
      }
      if constexpr (kLife) {
        long long sn[LPL];
        int qh[LPL];
        constexpr bool snapped = kSnap;
#pragma unroll
        for (int i = 0; i < LPL; ++i) sn[i] = (snapped && l0 + i < a.n_lines) ? a.line_price[l0 + i] : 0ll;
        uint32_t stt[LPL];
        load_states<LPL>(a, ids, stt);
        live_quantities<LPL>(qs, stt, qh);
        lines_tile<PriceT, kLdsTables, LPL, true, kPriceGlobal, kNoHist>(a, s_price, s_hist, s_stage, ids, prs, qs, lane, lt_lo, lt_hi, o, qh,
                                                  sn, snapped);
      } else {
        const long long none[LPL] = {};
        lines_tile<PriceT, kLdsTables, LPL, false, kPriceGlobal, kNoHist>(a, s_price, s_hist, s_stage, ids, prs, qs, lane, lt_lo, lt_hi, o, qs, none, false);
      }
    }
    // orders closed since the last flush (everything below the order that is still open)
    stage_flush<kLife>(a, s_stage, o, min(((uint32_t)o.id - (uint32_t)a.base) - o.buf_base, a.stage_cap), lane);

    // the run's first and last order may continue in the neighbouring waves: hand both to k_fixup
    if (lane == 0) {
      const uint32_t idx = (uint32_t)o.id - (uint32_t)a.base;
      if (idx >= (uint32_t)a.n_orders) o.st |= kStOrderRange;
      a.rec_id[2 * g] = head_id;
      a.rec_id[2 * g + 1] = o.id;
      if (o.first_open) { a.rec_sum[2 * g] = o.sum; a.rec_sum[2 * g + 1] = 0; }
      else              { a.rec_sum[2 * g] = o.head_sum; a.rec_sum[2 * g + 1] = o.sum; }
    }
    if (a.fill_edges) {
      const long long before = t0 == 0 ? (long long)a.base - 1 : (long long)a.order_id[t0 * kTileL - 1];
      wave_fill_gap<kLife>(a, before, head_id, lane);                // behind the last order of the wave before
      if (t1 == a.n_tiles) wave_fill_gap<kLife>(a, o.id, (long long)a.base + a.n_orders, lane);
    }
    if (o.st) atomicOr(a.status, o.st);
  }

  if (kLdsTables && !kNoHist) {
    __syncthreads();
    uint32_t* slab = a.prod_slab + (size_t)blockIdx.x * a.p_pad;
    for (int i = (int)threadIdx.x; i < a.n_products; i += (int)blockDim.x) slab[i] = s_hist[i];
  }
}

// ---------------------------------------------------------------------------------------------
// k_lines_any — the same sums when the `detalle` rows are NOT grouped by order (concurrent checkouts
// interleave their inserts: process/confirmcompra.php:27,35-37 and the table has no key that would
// order them, store.sql:192-194).  order_total must be zero on entry.  Runs of equal order ids are
// still combined inside a tile (one 64-bit global atomic per run), so almost-grouped input costs about
// one atomic per order; fully shuffled input costs one per line.
// ---------------------------------------------------------------------------------------------
template <typename PriceT, bool kLdsTables, bool kLife = false>
__global__ void __launch_bounds__(1024) k_lines_any(LinesArgs a) {
  constexpr int LPL = kLinesPerLane;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int waves_per_block = (int)(blockDim.x >> 6);
  const uint32_t P = (uint32_t)a.n_products;
  if (a.only_if_unsorted && !(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStUnsorted))
    return;                                             // uniform: the whole grid leaves
  // run_if_zero names the word in which the order-restoring path says what it did: 0 = nothing (the rows are
  // near enough to grouped for this kernel), 1 = everything, 2 = the stock deltas but not the order totals
  const unsigned long long elsewhere = a.run_if_zero ? *a.run_if_zero : 0ull;
  if (elsewhere == 1ull || elsewhere == 3ull) return;    // uniform
  const bool hist = !a.totals_only && !a.no_hist && elsewhere == 0ull;   // no_hist: k_stock does the stock deltas
  const bool price_lds = a.price_in_lds != 0;
  PriceT* s_price = reinterpret_cast<PriceT*>(smem);
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + (price_lds ? (size_t)a.p_pad * sizeof(PriceT) : 0));
  if (kLdsTables) {
    for (int i = (int)threadIdx.x; i < (int)P + 1; i += (int)blockDim.x) {
      if (price_lds) s_price[i] = (PriceT)a.price[i];
      if (hist) s_hist[i] = 0u;                          // (without it the histogram has no LDS of its own)
    }
    __syncthreads();
  }
  const uint32_t lt_lo = lane < 32 ? ((1u << lane) - 1u) : 0xffffffffu;
  const uint32_t lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
  unsigned long long* tot_cells = reinterpret_cast<unsigned long long*>(a.order_total);

  const long long g = (long long)blockIdx.x * waves_per_block + wave;
  const long long t0 = g * a.tiles_per_wave;
  long long t1 = t0 + a.tiles_per_wave;
  if (t1 > a.n_tiles) t1 = a.n_tiles;
  uint32_t st = 0;
  for (long long t = t0; t < t1; ++t) {
    const long long l0 = t * kTile + (long long)lane * LPL;
    int ids[LPL], prs[LPL], qs[LPL], qh[LPL];
    long long sn[LPL] = {};
    const bool snapped = kLife && a.line_price != nullptr;
    if (l0 + LPL <= a.n_lines) {
      load_lines<LPL>(a.order_id + l0, ids);
      load_lines<LPL>(a.product_idx + l0, prs);
      load_lines<LPL>(a.qty + l0, qs);
      if (snapped) load_prices<LPL>(a.line_price + l0, sn);
    } else {
      // past the end: repeat the last real order id of the data set, unknown product, quantity 0
      const int last_id = a.order_id[a.n_lines - 1];
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        const bool in = l0 + i < a.n_lines;
        ids[i] = in ? a.order_id[l0 + i] : last_id;
        prs[i] = in ? a.product_idx[l0 + i] : -1;
        qs[i] = in ? a.qty[l0 + i] : 0;
        sn[i] = (snapped && in) ? a.line_price[l0 + i] : 0ll;
      }
    }
    if (kLife) {
      uint32_t stt[LPL];
      load_states<LPL>(a, ids, stt);
      live_quantities<LPL>(qs, stt, qh);
    } else {
#pragma unroll
      for (int i = 0; i < LPL; ++i) qh[i] = qs[i];
    }
    uint32_t pi[LPL];
    long long v[LPL];
    uint32_t qmax = 0;
#pragma unroll
    for (int i = 0; i < LPL; ++i) pi[i] = min((uint32_t)prs[i], P);
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      const long long own = pi[i] < P ? sn[i] : 0ll;        // the line's own price (lifecycle)
      if (kLdsTables) {
        v[i] = (snapped ? own : price_lds ? (long long)s_price[pi[i]] : a.price[pi[i]]) * (long long)qs[i];
        if (hist) {
          __hip_atomic_fetch_add(&s_hist[pi[i]], (uint32_t)qh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          qmax = max(qmax, (uint32_t)qh[i]);
        }
      } else {
        v[i] = (snapped ? own : a.price[pi[i]]) * (long long)qs[i];
        if (hist && pi[i] < P) atomicAdd(&a.prod_carry[pi[i]], (unsigned long long)(long long)qh[i]);
      }
    }
    if (kLdsTables && qmax >= a.q_small) {
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        if ((uint32_t)qh[i] >= a.q_small) {
          __hip_atomic_fetch_add(&s_hist[pi[i]], 0u - (uint32_t)qh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pi[i] < P) atomicAdd(&a.prod_carry[pi[i]], (unsigned long long)(long long)qh[i]);
        }
      }
    }
    // runs of equal ids inside the tile (no carry between tiles: every tile settles its own runs)
    const int first = __builtin_amdgcn_readfirstlane(ids[0]);
    const int prev = wave_shift_up1(ids[LPL - 1], first);
    int d[LPL];
    d[0] = ids[0] - prev;
#pragma unroll
    for (int i = 1; i < LPL; ++i) d[i] = ids[i] - ids[i - 1];
    int dor = d[0];
    long long r = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      if (i) dor |= d[i];
      r = d[i] ? 0ll : r;
      r += v[i];
      tot += v[i];
    }
    const long long incl = wave_inclusive_scan(tot);
    const long long excl = incl - tot;
    const unsigned long long headmask = __ballot(dor != 0);
    const uint32_t below_lo = (uint32_t)headmask & lt_lo, below_hi = (uint32_t)(headmask >> 32) & lt_hi;
    const bool any_below = (below_lo | below_hi) != 0u;
    const int j = below_hi ? (63 - __clz(below_hi)) : (31 - __clz(below_lo | 1u));
    const long long gj = __shfl(r - incl, j);
    long long run = excl + (any_below ? gj : 0ll);
    int pid = prev;
    auto settle = [&](int oid, long long sum) {
      const uint32_t idx = (uint32_t)oid - (uint32_t)a.base;
      if (idx >= (uint32_t)a.n_orders) st |= kStOrderRange;
      else if (sum != 0) atomicAdd(&tot_cells[idx], (unsigned long long)sum);
    };
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      if (ids[i] != pid) { settle(pid, run); run = 0; }
      run += v[i];
      pid = ids[i];
    }
    if (lane == kWave - 1) settle(pid, run);          // the tile's last run
  }
  if (st) atomicOr(a.status, st);
  if (kLdsTables && hist) {
    __syncthreads();
    uint32_t* slab = a.prod_slab + (size_t)blockIdx.x * a.p_pad;
    for (int i = (int)threadIdx.x; i < a.n_products; i += (int)blockDim.x) slab[i] = s_hist[i];
  }
}

// Fallback helpers: act only if the grouped pass raised kStUnsorted.
// (need_unsorted: only if the grouped pass raised kStUnsorted; run_if_zero: not if that word is 1)
__global__ void __launch_bounds__(256) k_zero_if_unsorted(int64_t* out, int64_t n, const uint32_t* status, int need_unsorted,
                                                          const unsigned long long* run_if_zero) {
  if (need_unsorted && !(__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStUnsorted)) return;
  if (run_if_zero && (*run_if_zero == 1ull || *run_if_zero == 3ull)) return;   // (another path writes every total)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 0;
}

__global__ void k_clear_unsorted(uint32_t* status) {
  if (threadIdx.x == 0 && blockIdx.x == 0) atomicAnd(status, ~kStUnsorted);
}

// ---------------------------------------------------------------------------------------------
// k_fixup
// ---------------------------------------------------------------------------------------------
struct FixArgs {
  const int32_t* rec_id;
  const int64_t* rec_sum;
  int32_t n_rec;            // 2 * (waves that had work), <= kFixThreads * kFixStrip
  int32_t base;
  int32_t n_orders;
  int64_t* order_total;
  uint32_t* status;
  const uint32_t* prod_slab;
  int32_t n_slabs;
  int32_t p_pad;
  int32_t n_products;
  int32_t merge_products;   // 1: blocks >= 1 merge the stock-delta slabs
  int32_t fix_orders;       // 1: block 0 finishes the orders cut by wave boundaries (grouped input)
  const unsigned long long* fix_if_sorted;   // when not null: only if that word is 3 (see LinesArgs::run_if_sorted)
  unsigned long long* prod_carry;
  int64_t* product_qty;
  int32_t ablate;           // developer switch for timing experiments: 1 skip merge blocks, 2 skip block 0
  // Deferred mode (the record block runs beside the customer pass instead of before it).  The customer
  // pass then reads, for an order cut by a wave boundary, whatever k_lines left in its slot; this block
  // reads the same word, adds (total - that word) to the customer's 64-bit cell — exact modulo 2^64,
  // whatever the word was — and leaves (index, total) in the edge list; the slot itself is written after
  // the customer pass, by k_tail.  Orders without lines were zeroed by k_lines (fill_edges).
  int32_t deferred;
  const int32_t* order_customer;
  unsigned long long* cust_carry;
  int32_t n_customers;
  uint32_t* edge_idx;       // [n_rec] order index, or 0xffffffff
  int64_t* edge_total;      // [n_rec]
  const int32_t* order_descuento;   // [M] or null: the totals (and the zeros of orders without lines) are stored net of it
  int32_t descuento_mode;
  const unsigned char* order_state; // [M] or null (deferred mode): an order with bit 0 clear is not in its customer's sum
};

// Orders without lines, indices [lo, hi): a total of zero (less the order's discount, if any).
__device__ __forceinline__ void block_fill_zero(int64_t* out, long long lo, long long hi, int n_orders,
                                                const int32_t* descuento = nullptr, int mode = 0) {
  if (lo < 0) lo = 0;
  if (hi > n_orders) hi = n_orders;
  for (long long k = lo + threadIdx.x; k < hi; k += blockDim.x) out[k] = net_total(0, (uint32_t)k, descuento, mode);
}

// Stock-delta merge of one group of 64 products by a workgroup of `waves` waves: wave w sums slabs
// w, w + waves, ...; s_part holds waves * 64 partial sums.
__device__ __forceinline__ void merge_product_group(const FixArgs& a, int group, int waves, unsigned long long* s_part) {
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int p = group * kMergeProducts + lane;
  unsigned long long acc = 0;
  if (p < a.n_products) {
    // sixteen independent loads per round: the loop is bound by latency, not by bytes
    for (int s0 = wave; s0 < a.n_slabs; s0 += 16 * waves) {
      uint32_t u[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int sj = s0 + j * waves;
        u[j] = sj < a.n_slabs ? a.prod_slab[(size_t)sj * a.p_pad + p] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += u[j];
    }
  }
  s_part[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && p < a.n_products) {
    unsigned long long sum = a.prod_carry[p];
    for (int w = 0; w < waves; ++w) sum += s_part[w * 64 + lane];
    a.product_qty[p] = (int64_t)sum;
    a.prod_carry[p] = 0ull;           // leave the workspace clean for the next call
  }
}

__device__ __forceinline__ void fixup_records(const FixArgs& a, unsigned char* smem);

__global__ void __launch_bounds__(kFixThreads) k_fixup(FixArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (blockIdx.x > 0) {
    // ---- stock-delta merge: 64 products per block
    if (!a.merge_products || (a.ablate & 1)) return;
    merge_product_group(a, (int)blockIdx.x - 1, kFixThreads / 64, reinterpret_cast<unsigned long long*>(smem));
    return;
  }
  if ((a.ablate & 2) || !a.fix_orders) return;
  if (a.fix_if_sorted && *a.fix_if_sorted != 3ull) return;
  fixup_records(a, smem);
}

// One workgroup of kFixThreads threads; smem: kFixThreads * kFixStrip * 12 bytes.
__device__ __forceinline__ void fixup_records(const FixArgs& a, unsigned char* smem) {
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // ---- block 0: orders cut by wave boundaries.  Records are (order id, partial sum), two per
  // wave, in line order, hence sorted by id.  Every maximal run of equal ids is one order.
  long long* s_excl = reinterpret_cast<long long*>(smem);                       // [n_rec]
  int* s_head = reinterpret_cast<int*>(smem + (size_t)kFixThreads * kFixStrip * 8);   // [n_rec]
  __shared__ long long s_wsum[kFixThreads / 64];
  __shared__ int s_wmax[kFixThreads / 64];
  __shared__ int s_gap_n;
  __shared__ int s_gap_lo[256], s_gap_hi[256];

  const int n = a.n_rec;
  const int i0 = tid * kFixStrip;
  int ids[kFixStrip];
  long long sums[kFixStrip];
  int prev_id = 0;
  uint32_t st = 0;
  if (tid == 0) s_gap_n = 0;
#pragma unroll
  for (int k = 0; k < kFixStrip; ++k) {
    const int i = i0 + k;
    ids[k] = i < n ? a.rec_id[i] : 0;
    sums[k] = i < n ? a.rec_sum[i] : 0;
  }
  if (i0 > 0 && i0 < n) prev_id = a.rec_id[i0 - 1];
  // everything this block reads from global memory is requested here, in one round trip
  const int after_id = i0 + kFixStrip < n ? a.rec_id[i0 + kFixStrip] : 0;     // the id after this thread's strip
  const int first_id = n > 0 ? a.rec_id[0] : 0, last_id = n > 0 ? a.rec_id[n - 1] : 0;

  long long tsum = 0;
  int tmax = -1;
#pragma unroll
  for (int k = 0; k < kFixStrip; ++k) {
    const int i = i0 + k;
    if (i < n) {
      const bool head = (i == 0) || (ids[k] != (k == 0 ? prev_id : ids[k - 1]));
      if (head) tmax = i;
      tsum += sums[k];
    }
  }
  // block scans: exclusive sum of tsum, exclusive running max of tmax
  long long incl = wave_inclusive_scan(tsum);
  int mx = tmax;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(mx, d); if (lane >= d) mx = max(mx, t); }
  if (lane == 63) { s_wsum[wave] = incl; s_wmax[wave] = mx; }
  __syncthreads();
  long long woff = 0;
  int wmx = -1;
  for (int w = 0; w < wave; ++w) { woff += s_wsum[w]; wmx = max(wmx, s_wmax[w]); }
  long long run = woff + incl - tsum;                       // exclusive prefix at record i0
  int mxe = __shfl_up(mx, 1);
  int cur_head = max(wmx, lane == 0 ? -1 : mxe);            // last head before record i0
#pragma unroll
  for (int k = 0; k < kFixStrip; ++k) {
    const int i = i0 + k;
    if (i < n) {
      const bool head = (i == 0) || (ids[k] != (k == 0 ? prev_id : ids[k - 1]));
      if (head) cur_head = i;
      s_excl[i] = run;
      s_head[i] = cur_head;
      run += sums[k];
    }
  }
  __syncthreads();

  // every run's last record writes the order's total and zero-fills the ids up to the next run
#pragma unroll
  for (int k = 0; k < kFixStrip; ++k) {
    const int i = i0 + k;
    if (i < n) {
      if (a.deferred) a.edge_idx[i] = 0xffffffffu;              // overwritten below by a run's last record
      const int next_id = (i + 1 < n) ? ((k + 1 < kFixStrip) ? ids[(k + 1) % kFixStrip] : after_id) : 0;
      const bool tail = (i + 1 == n) || (next_id != ids[k]);
      if (tail) {
        const long long total = s_excl[i] + sums[k] - s_excl[s_head[i]];
        const uint32_t idx = (uint32_t)ids[k] - (uint32_t)a.base;
        if (idx >= (uint32_t)a.n_orders) st |= kStOrderRange;
        else if (!a.deferred) a.order_total[idx] = net_total(total, idx, a.order_descuento, a.descuento_mode);
        else {
          const long long net = net_total(total, idx, a.order_descuento, a.descuento_mode);
          const long long seen = a.order_total[idx];           // what the customer pass reads for this order
          const int32_t c = a.order_customer[idx];
          const bool counts = !a.order_state || (a.order_state[idx] & 1);   // (a cancelled order: that pass adds nothing)
          if (counts && (uint32_t)c < (uint32_t)a.n_customers)   // out of range: flagged by the customer pass
            atomicAdd(&a.cust_carry[c], (unsigned long long)net - (unsigned long long)seen);
          a.edge_idx[i] = idx;
          a.edge_total[i] = net;
        }
        // Only the step from a wave's tail record (odd i) to the next wave's head record can hide
        // orders without lines; between a wave's own head and tail k_lines wrote everything.
        if (i + 1 < n && (i & 1)) {
          if (next_id < ids[k]) st |= kStUnsorted;
          else if (!a.deferred && (uint32_t)next_id - (uint32_t)ids[k] > 1u) {
            const int slot = atomicAdd(&s_gap_n, 1);
            if (slot < 256) { s_gap_lo[slot] = ids[k]; s_gap_hi[slot] = next_id; }
            else {
              for (long long q = (long long)ids[k] + 1; q < (long long)next_id; ++q) {
                const uint32_t qi = (uint32_t)((int)q) - (uint32_t)a.base;
                if (qi < (uint32_t)a.n_orders) a.order_total[qi] = net_total(0, qi, a.order_descuento, a.descuento_mode);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const int n_gaps = min(s_gap_n, 256);
  for (int gidx = 0; gidx < n_gaps; ++gidx)
    block_fill_zero(a.order_total, (long long)s_gap_lo[gidx] + 1 - a.base, (long long)s_gap_hi[gidx] - a.base,
                    a.n_orders, a.order_descuento, a.descuento_mode);
  // orders before the first and after the last line
  if (n > 0 && !a.deferred) {
    block_fill_zero(a.order_total, 0, (long long)first_id - a.base, a.n_orders, a.order_descuento, a.descuento_mode);
    block_fill_zero(a.order_total, (long long)last_id + 1 - a.base, a.n_orders, a.n_orders, a.order_descuento, a.descuento_mode);
  }
  if (st) atomicOr(a.status, st);
}

// ---------------------------------------------------------------------------------------------
// k_stock — the stock deltas on their own, for product tables whose 32-bit histogram does not fit one
// workgroup's LDS.  Built like the customer pass: the products are cut into R ranges, the lines into
// streams; workgroup (stream s, range r) reads the stream's product_idx / qty columns (the R siblings of
// a stream share them through their XCD's L2), adds the quantities of its range into an LDS histogram
// and writes it as slice r of slab s.  merge_product_group sums the slabs.  The alternative, one 64-bit
// global atomic per line, is bound by the 23 G/s scattered-atomic rate of the part.
// ---------------------------------------------------------------------------------------------
struct StockArgs {
  const int32_t* order_id;            // only read when order_state is not null
  const int32_t* product_idx;
  const int32_t* qty;
  int64_t n_lines;
  int64_t n_tiles;                    // ceil(N / 256)
  int64_t tiles_per_stream;
  int32_t n_products;
  int32_t p_pad;                      // slab row pitch
  int32_t span;                       // products per range (a multiple of 4)
  int32_t n_ranges;
  int32_t n_streams;
  uint32_t q_small;                   // quantities in [0, q_small) go to the LDS histogram
  uint32_t* prod_slab;                // [n_streams][p_pad]
  unsigned long long* prod_carry;     // [P] 64-bit cells for every other quantity
  int32_t base, n_orders;
  const unsigned char* order_state;   // [M] or null
};

__global__ void __launch_bounds__(1024) k_stock(StockArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(smem);
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int waves_per_block = (int)(blockDim.x >> 6);
  const int x = (int)blockIdx.x & 7, jslot = (int)blockIdx.x >> 3;
  const int r = jslot % a.n_ranges;
  const int s = x + 8 * (jslot / a.n_ranges);
  const int lo = r * a.span;
  int len = a.n_products - lo;
  if (len > a.span) len = a.span;
  if (len < 0) len = 0;
  for (int i = (int)threadIdx.x * 4; i < a.span; i += (int)blockDim.x * 4)
    *reinterpret_cast<uint4*>(s_tab + i) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const long long t0 = (long long)s * a.tiles_per_stream;
  long long t1 = t0 + a.tiles_per_stream;
  if (t1 > a.n_tiles) t1 = a.n_tiles;
  const uint32_t top = (uint32_t)a.n_orders - 1u;
  for (long long t = t0 + wave; t < t1; t += waves_per_block) {
    const long long l0 = t * kTile + (long long)lane * 4;
    int pr[4], q[4], id[4] = {0, 0, 0, 0};
    if (l0 + 4 <= a.n_lines) {
      load_lines<4>(a.product_idx + l0, pr);
      load_lines<4>(a.qty + l0, q);
      if (a.order_state) load_lines<4>(a.order_id + l0, id);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool in = l0 + i < a.n_lines;
        pr[i] = in ? a.product_idx[l0 + i] : -1;
        q[i] = in ? a.qty[l0 + i] : 0;
        id[i] = (in && a.order_state) ? a.order_id[l0 + i] : a.base;
      }
    }
    if (a.order_state) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (!(a.order_state[min((uint32_t)id[i] - (uint32_t)a.base, top)] & 1)) q[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t off = (uint32_t)pr[i] - (uint32_t)lo;       // a product outside [0, P) is in no range
      if (off < (uint32_t)len) {
        if ((uint32_t)q[i] < a.q_small)
          __hip_atomic_fetch_add(s_tab + off, (uint32_t)q[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
          atomicAdd(&a.prod_carry[pr[i]], (unsigned long long)(long long)q[i]);
      }
    }
  }
  __syncthreads();
  uint32_t* slab = a.prod_slab + (size_t)s * a.p_pad + lo;       // lo and p_pad are multiples of 4
  const int len4 = len & ~3;
  for (int i = (int)threadIdx.x * 4; i < len4; i += (int)blockDim.x * 4)
    *reinterpret_cast<uint4*>(slab + i) = *reinterpret_cast<const uint4*>(s_tab + i);
  for (int i = len4 + (int)threadIdx.x; i < len; i += (int)blockDim.x) slab[i] = s_tab[i];
}

// ---------------------------------------------------------------------------------------------
// k_customers / k_custmerge
// ---------------------------------------------------------------------------------------------
struct CustArgs {
  const int32_t* order_customer;   // [M]
  const int64_t* order_total;      // [M]
  int64_t n_orders;
  int64_t n_tiles;                 // ceil(M / 256)
  int64_t tiles_per_stream;
  int32_t n_customers;
  int32_t c_pad;                   // slab row pitch
  int32_t span;                    // customers per range
  int32_t n_ranges;                // R
  int32_t n_streams;
  uint32_t* cust_slab;             // [n_streams][c_pad]
  unsigned long long* cust_carry;  // [C] high-word corrections (LDS mode) / the totals (global mode)
  uint32_t* status;
  int32_t ablate;                  // developer switch: 1 = skip the LDS adds (timing experiments only)
  const unsigned char* order_state;   // [M] or null (kLife kernels): bit 0 = the order counts
};

// One tile of 256 orders: lane l holds orders 4l..4l+3 in (cs, tv).  `s_scratch` is this lane's own scratch
// word; `check_ids` (uniform) says whether this workgroup is the one of its stream that reports customer ids
// outside the table.
// The pass is bound by instruction issue (a sibling workgroup runs this for every order of its stream, two
// thirds of which belong to other ranges), so the common path is kept to a subtract, a compare, an address, a
// select and the add per order, and one carry test afterwards; anything else sends the lane down the exact path.
template <bool kLdsTables>
__device__ __forceinline__ void customers_tile(const CustArgs& a, uint32_t* s_tab, uint32_t* s_scratch, int lo, int len,
                                               bool check_ids, const int (&cs)[4], const long long (&tv)[4],
                                               uint32_t& st) {
  if (!kLdsTables) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((uint32_t)cs[i] >= (uint32_t)a.n_customers) st |= kStCustomerRange;
      else if (tv[i] != 0) atomicAdd(&a.cust_carry[cs[i]], (unsigned long long)tv[i]);
    }
    return;
  }
  // 32-bit partial sums in LDS; the add returns the old word so that a carry out of bit 31 (and the
  // high half of a total of 2^32 cents or more) can be forwarded to a 64-bit global cell.  All four
  // adds are issued before the first result is needed; an order of another range adds to the lane's scratch
  // word, whose contents nobody reads.
  uint32_t old[4];
  bool inr[4];
  if (a.ablate & 1) {
    if ((cs[0] ^ cs[1] ^ cs[2] ^ cs[3] ^ (int)tv[0] ^ (int)tv[1] ^ (int)tv[2] ^ (int)tv[3]) == 0x7ffffff1) st |= 8u;
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t off = (uint32_t)(cs[i] - lo);
    inr[i] = off < (uint32_t)len;
    uint32_t* cell = inr[i] ? (s_tab + off) : s_scratch;
    old[i] = __hip_atomic_fetch_add(cell, (uint32_t)tv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  bool rare = false;
  uint32_t high = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rare |= inr[i] && (uint32_t)(old[i] + (uint32_t)tv[i]) < old[i];
    high |= (uint32_t)((unsigned long long)tv[i] >> 32);
  }
  rare |= high != 0u;
  if (check_ids) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rare |= (uint32_t)cs[i] >= (uint32_t)a.n_customers;
  }
  if (rare) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((uint32_t)cs[i] >= (uint32_t)a.n_customers) st |= kStCustomerRange;
      const long long carry = ((uint32_t)(old[i] + (uint32_t)tv[i]) < old[i]) ? 1 : 0;
      const long long hi = (tv[i] >> 32) + carry;
      if (inr[i] && hi != 0) atomicAdd(&a.cust_carry[cs[i]], ((unsigned long long)hi) << 32);
    }
  }
}

// Workgroup b: XCD label x = b % 8 (workgroups with equal label share an L2), slot j = b / 8,
// range r = j % R, stream s = x + 8 * (j / R).  The R siblings of a stream read the same orders:
// one of them brings a line in from HBM, the others find it in their XCD's L2.
template <bool kLdsTables, bool kLife>
__device__ __forceinline__ void customers_block(const CustArgs& a, unsigned char* smem, int bid);

template <bool kLdsTables, bool kLife = false>
__global__ void __launch_bounds__(1024) k_customers(CustArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  customers_block<kLdsTables, kLife>(a, smem, (int)blockIdx.x);
}

// The customer pass with the record block of k_fixup riding along: workgroup 0 finishes the orders cut
// by wave boundaries in deferred mode (see FixArgs), workgroups 1..lead-1 merge the stock-delta slabs,
// and workgroup lead + b is customer workgroup b (the customer pass maps workgroups to XCDs by index
// modulo 8, so lead is a multiple of 8).
struct CustFusedArgs {
  CustArgs c;
  FixArgs f;
  int32_t lead;             // leading workgroups (a multiple of 8): one record block, lead - 1 merge blocks
};

// kLife: the order-state column is given (a cancelled order adds nothing to its customer).
template <bool kLife>
__global__ void __launch_bounds__(1024) k_customers_fused(CustFusedArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  if ((int)blockIdx.x < a.lead) {
    if (blockIdx.x == 0) { fixup_records(a.f, smem); return; }
    // the other leading workgroups share the stock-delta merge (64 products per step)
    const int n_groups = (a.f.n_products + kMergeProducts - 1) / kMergeProducts;
    for (int g = (int)blockIdx.x - 1; g < n_groups; g += a.lead - 1) {
      merge_product_group(a.f, g, (int)(blockDim.x >> 6), reinterpret_cast<unsigned long long*>(smem));
      __syncthreads();
    }
    return;
  }
  customers_block<true, kLife>(a.c, smem, (int)blockIdx.x - a.lead);
}

template <bool kLdsTables, bool kLife>
__device__ __forceinline__ void customers_block(const CustArgs& a, unsigned char* smem, int bid) {
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(smem);
  const int lane = lane_id();
  uint32_t* s_scratch = s_tab + a.span + lane;
  // (the wave's number in a scalar register: tile numbers and addresses are then scalar arithmetic)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int waves_per_block = (int)(blockDim.x >> 6);
  const int x = bid & 7, jslot = bid >> 3;
  const int r = jslot % a.n_ranges;
  const bool check_ids = r == 0;                           // one workgroup of a stream reports ids outside the table
  const int s = x + 8 * (jslot / a.n_ranges);
  const int lo = r * a.span;
  int len = a.n_customers - lo;
  if (len > a.span) len = a.span;
  if (len < 0) len = 0;

  if (kLdsTables) {
    // span is a multiple of 4: clear whole 16-byte groups (cells past len are never read)
    for (int i = (int)threadIdx.x * 4; i < a.span; i += (int)blockDim.x * 4)
      *reinterpret_cast<uint4*>(s_tab + i) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
  }
  uint32_t st = 0;
  const long long t0 = (long long)s * a.tiles_per_stream;
  long long t1 = t0 + a.tiles_per_stream;
  if (t1 > a.n_tiles) t1 = a.n_tiles;
  const long long n_full = a.n_orders / kTile;
  const long long tf = t1 < n_full ? t1 : n_full;         // complete tiles of this stream: [t0, tf)

  if (t0 + wave < tf && !(a.ablate & 2)) {
    // Software pipeline over this wave's tiles (t, t + W, t + 2W, ...) with kCustDepth tiles in
    // flight: most of these loads are L2 hits shared with the sibling workgroups, and a CU needs
    // far more bytes in flight to stream from L2 than to stream its share of HBM.  Loads past the
    // wave's last tile re-read that tile and are ignored.
    // A tile is four loads, each of them contiguous over the wave: lane l takes orders 2l, 2l + 1 of the
    // tile's first half and of its second half (ids 8 bytes, totals 16 bytes at a time).  With one lane's four
    // orders side by side instead, a 16-byte load of totals covers half of every 64-byte piece it touches and
    // the vector cache asks L2 for every piece twice.
    // (tile numbers fit 32 bits: at most 2^31 orders; kept in scalar registers, the wave's number being one)
    const int W = waves_per_block;
    int t = (int)t0 + wave;
    const int last = t + (((int)tf - 1 - t) / W) * W;         // this wave's last tile
    int2 rc0[kCustDepth], rc1[kCustDepth];
    longlong2 ra[kCustDepth], rb[kCustDepth];
    uint32_t rs[kCustDepth];                                  // four state bytes per lane (lifecycle)
    const bool stated = kLife && a.order_state != nullptr;
    auto request = [&](int tile, int d) {
      const size_t e = (size_t)(tile < last ? tile : last) * kTile + (size_t)lane * 2;
      rc0[d] = *reinterpret_cast<const int2*>(a.order_customer + e);
      rc1[d] = *reinterpret_cast<const int2*>(a.order_customer + e + kTile / 2);
      ra[d] = *reinterpret_cast<const longlong2*>(a.order_total + e);
      rb[d] = *reinterpret_cast<const longlong2*>(a.order_total + e + kTile / 2);
      if (stated)
        rs[d] = (uint32_t)*reinterpret_cast<const unsigned short*>(a.order_state + e) |
                ((uint32_t)*reinterpret_cast<const unsigned short*>(a.order_state + e + kTile / 2) << 16);
      // (kept where it stands: the tiles are asked for in the order in which the loop takes them, and a tile's
      // registers are read before they are asked for again.  Moved up by the scheduler, old and new contents
      // would be alive together in different registers and the end of the loop would move every set back into
      // place, waiting for every load in flight; asked for out of order, the first tile taken may be the last
      // one asked for, with the same wait at the top of the loop.)
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int d = 0; d < kCustDepth; ++d) { rs[d] = 0x01010101u; request(t + d * W, d); }
    for (; t <= last; t += kCustDepth * W) {
#pragma unroll
      for (int d = 0; d < kCustDepth; ++d) {
        const int tc = t + d * W;
        if (tc <= last) {
          const int cs[4] = {rc0[d].x, rc0[d].y, rc1[d].x, rc1[d].y};
          long long tv[4] = {ra[d].x, ra[d].y, rb[d].x, rb[d].y};
          if (kLife) {
            const uint32_t s4 = rs[d];
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = ((s4 >> (8 * i)) & 1u) ? tv[i] : 0ll;
          }
          customers_tile<kLdsTables>(a, s_tab, s_scratch, lo, len, check_ids, cs, tv, st);
        }
        __builtin_amdgcn_sched_barrier(0);
        request(tc + kCustDepth * W, d);
      }
    }
  }
  if (wave == 0 && n_full < a.n_tiles && t0 <= n_full && n_full < t1) {
    // the last, partial tile of the order columns (it belongs to exactly one stream): orders past
    // the end are customer 0 with total 0
    const long long o0 = n_full * kTile + (long long)lane * 4;
    int cs[4];
    long long tv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = o0 + i < a.n_orders;
      cs[i] = in ? a.order_customer[o0 + i] : 0;
      tv[i] = in ? a.order_total[o0 + i] : 0;
      if (kLife && a.order_state && in && !(a.order_state[o0 + i] & 1)) tv[i] = 0;
    }
    customers_tile<kLdsTables>(a, s_tab, s_scratch, lo, len, check_ids, cs, tv, st);
  }
  if (st && r == 0) atomicOr(a.status, st);
  if (kLdsTables) {
    __syncthreads();
    // lo and c_pad are multiples of 4, so the slab slice is 16-byte aligned; the few cells past len
    // (inside the row's padding or the next range's start) are rewritten by their owner or unused
    uint32_t* slab = a.cust_slab + (size_t)s * a.c_pad + lo;
    const int len4 = len & ~3;
    for (int i = (int)threadIdx.x * 4; i < len4; i += (int)blockDim.x * 4)
      *reinterpret_cast<uint4*>(slab + i) = *reinterpret_cast<const uint4*>(s_tab + i);
    for (int i = len4 + (int)threadIdx.x; i < len; i += (int)blockDim.x) slab[i] = s_tab[i];
  }
}

struct CustMergeArgs {
  const uint32_t* cust_slab;
  int32_t n_streams;
  int32_t c_pad;                   // multiple of 4
  int32_t n_customers;
  unsigned long long* cust_carry;
  int64_t* customer_total;
};

// 256 threads = 4 waves; the block owns 256 consecutive customers (4 per lane, 16-byte loads) and
// wave w sums streams w, w+4, ...; the four partial sums meet in LDS.
__device__ __forceinline__ void custmerge_block(const CustMergeArgs& a, int bid, unsigned long long (*s_part)[256]);

__global__ void __launch_bounds__(256) k_custmerge(CustMergeArgs a) {
  __shared__ unsigned long long s_part[4][256];
  custmerge_block(a, (int)blockIdx.x, s_part);
}

__device__ __forceinline__ void custmerge_block(const CustMergeArgs& a, int bid, unsigned long long (*s_part)[256]) {
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int c0 = bid * 256 + lane * 4;
  unsigned long long acc[4] = {0, 0, 0, 0};
  if (c0 < a.c_pad) {
#pragma unroll 4
    for (int s = wave; s < a.n_streams; s += 4) {
      const uint4 w = *reinterpret_cast<const uint4*>(a.cust_slab + (size_t)s * a.c_pad + c0);
      acc[0] += w.x; acc[1] += w.y; acc[2] += w.z; acc[3] += w.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s_part[wave][lane * 4 + i] = acc[i];
  __syncthreads();
  const int c = bid * 256 + (int)threadIdx.x;
  if (c < a.n_customers) {
    const unsigned long long sum = a.cust_carry[c] + s_part[0][threadIdx.x] + s_part[1][threadIdx.x] +
                                   s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    a.customer_total[c] = (int64_t)sum;
    a.cust_carry[c] = 0ull;           // leave the workspace clean for the next call
  }
}

// unless_moved: when not null, the word in which the order-restoring path says what it did; 1 means that its
// reduce wrote the totals, net of the discounts already.
__global__ void __launch_bounds__(256) k_order_adjust(int64_t* order_total, const int32_t* descuento, int64_t n_orders,
                                                      int mode, const unsigned long long* unless_moved) {
  if (unless_moved && *unless_moved == 1ull) return;       // uniform
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o < n_orders) {
    const long long g = order_total[o];
    order_total[o] = (int64_t)((unsigned long long)g - (unsigned long long)descuento_cents(g, descuento[o], mode));
  }
}

// Everything that is left after the fused customer pass, in one launch of 256-thread workgroups:
//   [0, n_cust_blocks)                       customer merge (k_custmerge's work)
//   [n_cust_blocks, + n_prod_blocks)         stock-delta merge, 64 products per workgroup
//   the last ceil(n_rec / 256) workgroups    totals of the orders cut by wave boundaries -> order_total
struct TailArgs {
  CustMergeArgs m;
  FixArgs f;
  int32_t n_cust_blocks, n_prod_blocks;
};

__global__ void __launch_bounds__(256) k_tail(TailArgs a) {
  __shared__ unsigned long long s_part[4][256];
  const int b = (int)blockIdx.x;
  if (b < a.n_cust_blocks) { custmerge_block(a.m, b, s_part); return; }
  if (b < a.n_cust_blocks + a.n_prod_blocks) {             // (n_prod_blocks is 0 when the merge ran earlier)
    merge_product_group(a.f, b - a.n_cust_blocks, 4, &s_part[0][0]);
    return;
  }
  const int i = (b - a.n_cust_blocks - a.n_prod_blocks) * 256 + (int)threadIdx.x;
  if (i < a.f.n_rec) {
    const uint32_t idx = a.f.edge_idx[i];
    const int64_t total = a.f.edge_total[i];
    if (idx != 0xffffffffu) a.f.order_total[idx] = total;
  }
}

// Global-table mode epilogue: move the 64-bit accumulators into the output and clear them.
__global__ void __launch_bounds__(256) k_move_clear(unsigned long long* acc, int64_t* out, int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) { out[i] = (int64_t)acc[i]; acc[i] = 0ull; }
}

// ---------------------------------------------------------------------------------------------
// synthetic data
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kGolden = 0x9E3779B97F4A7C15ull;
constexpr unsigned long long kSeedLen = 0xA11CEull, kSeedProd = 0xB0Bull, kSeedQty = 0x51DEull,
                             kSeedPrice = 0xCAFEull, kSeedCust = 0xD00Dull;
constexpr int kSynthOrdersPerBlock = 1024;

__host__ __device__ __forceinline__ unsigned long long synth_hash(unsigned long long seed, unsigned long long i) {
  unsigned long long z = (seed ^ (i * kGolden)) + kGolden;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct SynthArgs {
  int64_t n_lines_total;
  int64_t order_lo, order_hi;
  int64_t line_lo;
  const int64_t* block_line_off;   // global line index of the first order of every 1024-order block
  const unsigned long long* cdf;   // Zipf cumulative weights or nullptr
  unsigned long long wtot;
  int32_t n_products, n_customers, base, qty_mode;
  int32_t* order_id;
  int32_t* product_idx;
  int32_t* qty;
  int32_t* order_customer;
};

__global__ void __launch_bounds__(kSynthOrdersPerBlock) k_synth(SynthArgs a) {
  __shared__ int s_w[kSynthOrdersPerBlock / 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long o = a.order_lo + (long long)blockIdx.x * kSynthOrdersPerBlock + tid;
  const bool in = o < a.order_hi;
  const int len = in ? 1 + (int)(synth_hash(kSeedLen, (unsigned long long)o) % 7ull) : 0;
  int incl = len;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_w[w];
  if (!in) return;
  const long long start = a.block_line_off[blockIdx.x] + woff + incl - len;
  a.order_customer[o - a.order_lo] = (int32_t)(synth_hash(kSeedCust, (unsigned long long)o) % (unsigned long long)a.n_customers);
  for (int i = 0; i < len; ++i) {
    const long long l = start + i;
    if (l >= a.n_lines_total) break;                 // the last order is truncated
    const unsigned long long hp = synth_hash(kSeedProd, (unsigned long long)l);
    int p;
    if (a.cdf == nullptr) {
      p = (int)(hp % (unsigned long long)a.n_products);
    } else {
      const unsigned long long u = hp % a.wtot;
      int lo = 0, hi = a.n_products;                 // first k with cdf[k] > u
      while (lo < hi) { const int m = lo + (hi - lo) / 2; if (a.cdf[m] > u) hi = m; else lo = m + 1; }
      p = lo;
    }
    const long long k = l - a.line_lo;
    a.order_id[k] = (int32_t)(a.base + o);
    a.product_idx[k] = p;
    a.qty[k] = a.qty_mode ? 1 + (int)(synth_hash(kSeedQty, (unsigned long long)l) % 5ull) : 1;
  }
}

}  // namespace acmpc
