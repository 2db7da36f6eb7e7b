// acmpc.hip — host side of libacmpc.so: context, workspace, kernel launches, the C ABI of
// include/acmpc.h.  Device code is in acmpc_kernels.hpp.  gfx950 only; no CPU backend.
#include "acmpc_kernels.hpp"
#include "acmpc_ingest.hpp"
#include "acmpc_emit.hpp"
#include "acmpc_comm.hpp"
#include "acmpc_partition.hpp"
#include "../../include/acmpc.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

using namespace acmpc;

#define ACMPC_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kMaxLds = 160 * 1024;
constexpr int kLdsReserve = 1024;       // static LDS some kernels declare besides the tables
// Size limits that keep the padded counts, table sizes and chunk tickets inside 32 bits.
constexpr int32_t kMaxProducts = 1 << 28;
constexpr int32_t kMaxCodes = 1 << 21;       // the code dictionary takes 64 bytes per slot, two slots per code
constexpr int32_t kMaxCustomers = 1 << 30;
constexpr int64_t kMaxTextBytes = 1ll << 44;

char g_init_error[512] = "no error";

// ---- workspace of the allocation-free entry point (acmpc_aggregate_device_noalloc) ----------------
// One fixed array in the code object; every GPU the library runs on holds its own copy, at its own
// address, until the process ends.  `scratch` is only a host-side stand-in for that array: the address a
// kernel can use is asked for once per device (acmpc_init, with that device current) and kept in
// g_scratch_ptr; host code uses that pointer, never the array's name.
constexpr int32_t kNaMaxProducts = 16380;                  // padded table (P + 1 rounded up to 4) <= 16384
constexpr int32_t kNaMaxCustomers = 131072;
constexpr size_t kNaProdSlabBytes = (size_t)256 * 16384 * 4;          // one 32-bit slab per workgroup of the line pass
constexpr size_t kNaCustSlabBytes = (size_t)44 << 20;                 // one 32-bit slab per stream of the customer pass
constexpr size_t kNaProdCarryBytes = (size_t)16384 * 8;
constexpr size_t kNaCustCarryBytes = (size_t)kNaMaxCustomers * 8;
constexpr size_t kNaRecBytes = (size_t)2 * kMaxWavesTotal * 8;        // each of the four record / edge arrays
constexpr size_t kNaOffProdSlab = 0;
constexpr size_t kNaOffCustSlab = kNaOffProdSlab + kNaProdSlabBytes;
constexpr size_t kNaOffProdCarry = kNaOffCustSlab + kNaCustSlabBytes;
constexpr size_t kNaOffCustCarry = kNaOffProdCarry + kNaProdCarryBytes;
constexpr size_t kNaOffRecId = kNaOffCustCarry + kNaCustCarryBytes;
constexpr size_t kNaOffRecSum = kNaOffRecId + kNaRecBytes;
constexpr size_t kNaOffEdgeIdx = kNaOffRecSum + kNaRecBytes;
constexpr size_t kNaOffEdgeTotal = kNaOffEdgeIdx + kNaRecBytes;
constexpr size_t kScratchBytes = kNaOffEdgeTotal + kNaRecBytes;       // 61.4 MiB
static_assert(kNaOffCustSlab % 256 == 0 && kNaOffProdCarry % 256 == 0 && kNaOffCustCarry % 256 == 0 &&
              kNaOffRecId % 256 == 0 && kNaRecBytes % 256 == 0, "scratch regions stay 256-byte aligned");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
// device address of `scratch` on every device seen.  Contexts may be created from different threads: the
// address of a device is looked up under the lock and published with an atomic store.
std::atomic<unsigned char*> g_scratch_ptr[kMaxDevices];
std::mutex g_scratch_lock;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;     // elements
};

}  // namespace

struct acmpc_ctx {
  int device = -1;
  int n_cu = 0;
  char err[512];
  hipStream_t copy_streams[4] = {nullptr, nullptr, nullptr, nullptr};   // host_copies: one per copying thread

  // product table
  int32_t n_products = 0;
  int32_t p_pad = 0;
  bool price_fits32 = false;
  DevBuf<int64_t> d_price;

  // workspace
  DevBuf<uint32_t> d_prod_slab;
  DevBuf<unsigned long long> d_prod_carry;
  DevBuf<int32_t> d_rec_id;
  DevBuf<int64_t> d_rec_sum;
  DevBuf<uint32_t> d_edge_idx;      // fused path: orders cut by wave boundaries, written after the customer pass
  DevBuf<int64_t> d_edge_total;
  int fuse = 1;                     // 0: always the four-launch sequence (ACMPC_FUSE=0)
  DevBuf<uint32_t> d_cust_slab;
  // order-restoring path for rows in any order: two sets of (order index, value) records, counters
  DevBuf<unsigned long long> d_part_rec[2];   // records of the order-restoring path after its two scatter passes
  DevBuf<unsigned long long> d_part_meta;
  DevBuf<uint32_t> d_part_state_bits;         // one bit per order (lifecycle: order states for rows in any order)
  int partition = 1;                // 0: rows in any order always go through the atomic kernel (ACMPC_PARTITION=0)
  bool part_ran = false;            // the last aggregate call queued the order-restoring path (its mode word says what it did)
  unsigned long long* d_part_mode = nullptr;   // that word (inside d_part_meta)
  DevBuf<unsigned long long> d_cust_carry;
  uint32_t* d_status = nullptr;
  bool carry_dirty = false;     // a launch failed half way: clear the carries before the next call

  // staging buffers of the host-pointer entry point
  DevBuf<int32_t> s_order_id, s_product_idx, s_qty, s_order_customer;
  DevBuf<int64_t> s_order_total, s_product_qty, s_customer_total;
  DevBuf<int64_t> s_merged;        // [product_qty | customer_total_cents] of the sharded host entry point
  DevBuf<int64_t> s_line_price;
  DevBuf<unsigned char> s_order_state;
  DevBuf<int32_t> s_order_descuento;
  DevBuf<int64_t> s_synth_off;
  DevBuf<unsigned long long> s_synth_cdf;

  // text ingest: dictionary of the product codes, chunk row counts, scalars, staging
  int32_t dict_products = 0;
  uint32_t dict_mask = 0;
  DevBuf<IngSlot> d_dict;
  DevBuf<unsigned short> d_dict_disp;
  uint32_t dict_seed = 0, dict_bucket_shift = 0;
  DevBuf<unsigned long long> d_codes;
  DevBuf<unsigned long long> d_chunk_rows;   // per-chunk row-count entries of the ingest, then its ticket
  DevBuf<unsigned long long> d_ing_tables;   // row automaton: transition entries, then the byte classes
  long long* d_ing_scalars = nullptr;      // [0] rows found, [1] malformed rows
  DevBuf<unsigned char> s_text;
  int32_t code_sql_bad = -1, code_csv_bad = -1;   // first code a text style cannot carry, or -1

  // text emit: chunk sizes, byte counter
  DevBuf<long long> d_chunk_bytes;
  DevBuf<int64_t> s_emit_value;

  // tuning (environment overrides, read once)
  int k1_threads = 1024;
  int k1_blocks_per_cu = 1;
  int k1_lpl = 4;
  int k2_threads = 1024;
  int k1_burst = 10;            // tiles between synchronised output bursts of k_lines (0 = off)
  int k1_tpw = 0;               // developer switch: at least this many tiles per wave in k_lines (ACMPC_K1_TPW)
  int k1_stage = 0;             // developer switch: staging slots per wave in k_lines (ACMPC_K1_STAGE; 0 = 1024)
  int k1_cus = 0, k2_cus = 0;   // developer switches: pretend the device has this many CUs (0 = all)
  // developer switches for timing experiments: parts of a kernel are skipped and its results are WRONG.
  // Read once at init, and only when ACMPC_DEV_ABLATE=1 is set as well.
  int ablate_fix = 0, ablate_k2 = 0, ablate_ing = 0;
  int force_global_tables = 0;
  int part_shift = 0;           // developer switch: at least 2^this orders per bucket of the order-restoring path (ACMPC_PART_SHIFT)
  int ing_blocks = 0;           // developer switch: workgroups of the text ingest (ACMPC_ING_BLOCKS; 0 = what fits the chip)
  int line_order = ACMPC_ORDER_GROUPED;

  // page-locked host buffers handed out by acmpc_host_alloc
  std::vector<void*> host_blocks;

  // profiling
  bool profiling = false;
  hipEvent_t ev[ACMPC_N_KERNELS + 1] = {};
  bool ev_recorded = false;
};

namespace {

int fail(acmpc_ctx* ctx, int code, const char* fmt, ...) {
  char* dst = ctx ? ctx->err : g_init_error;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(ctx, e_ == hipErrorOutOfMemory ? ACMPC_E_NOMEM : ACMPC_E_HIP, "%s failed: %s", \
                  #call, hipGetErrorString(e_));                                               \
  } while (0)

// Makes the context's device current for the duration of an entry point and puts the caller's device
// back afterwards: the library must not change the calling thread's current device as a side effect.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev != dev) {
      e = hipSetDevice(dev);
      switched = e == hipSuccess;
    }
    return e;
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
#define ENTER_DEVICE(ctx)   \
  DeviceScope device_scope_; \
  HIPCHK(ctx, device_scope_.enter((ctx)->device))

// Entry points whose host side uses standard containers run inside this guard: no exception may
// cross the C boundary, and running out of host memory is an error code like any other.
template <typename F>
int guarded(acmpc_ctx* ctx, const char* who, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(ctx, ACMPC_E_NOMEM, "%s: out of host memory", who);
  } catch (const std::system_error& e) {
    return fail(ctx, ACMPC_E_NOMEM, "%s: %s", who, e.what());     // (a thread or lock the system would not give)
  } catch (...) {
    return fail(ctx, ACMPC_E_INVALID, "%s: unexpected exception", who);
  }
}

// zero_new: a newly allocated buffer starts at zero.  The fill is queued on `st`, the stream of the call
// that will use the buffer, so that it is ordered before that call's kernels whatever kind of stream it is
// (a fill on the null stream is not ordered against a non-blocking stream).
template <typename T>
int ensure(acmpc_ctx* ctx, DevBuf<T>& b, size_t n, bool zero_new = false, hipStream_t st = nullptr) {
  if (n == 0) n = 1;
  if (b.cap >= n) return ACMPC_OK;
  if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  HIPCHK(ctx, hipMalloc((void**)&b.p, n * sizeof(T)));
  b.cap = n;
  if (zero_new) {
    HIPCHK(ctx, hipMemsetAsync(b.p, 0, n * sizeof(T), st));
    if (!st) HIPCHK(ctx, hipStreamSynchronize(nullptr));
  }
  return ACMPC_OK;
}

template <typename T>
void release(DevBuf<T>& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename K>
int set_dyn_lds(acmpc_ctx* ctx, K kernel, size_t bytes) {
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return ACMPC_OK;
}

struct LinesPlan {
  bool lds_tables;
  bool price_lds;          // the price table is in LDS (not with line prices, not when only the histogram fits)
  bool hist_lds;           // the stock-delta histogram is in LDS (not when only the price table fits)
  bool price32;
  size_t lds_bytes;
  int threads, blocks, waves_per_block;
  int lpl;                 // lines per lane per tile (4 or 8)
  int tile_lines;
  uint32_t stage_cap, flush_reserve;
  int burst;               // tiles between two output bursts of a workgroup (0: none)
  int64_t n_tiles, tiles_per_wave;
  int n_active_waves;
};

// line_prices: the grouped kernel reads a line-price column (lifecycle): no price table in LDS, and a
// 2 KB slice per wave for re-ordering the price loads.
int plan_lines(const acmpc_ctx* ctx, int64_t n_lines, LinesPlan* p, bool line_prices = false) {
  // tables hold P + 1 entries (the null product): int32 or int64 prices + 32-bit histogram
  const size_t need32 = (size_t)ctx->p_pad * 8, need64 = (size_t)ctx->p_pad * 12;
  p->price32 = ctx->price_fits32;
  p->price_lds = !line_prices;
  p->hist_lds = true;
  p->lds_bytes = line_prices ? (size_t)ctx->p_pad * 4 : p->price32 ? need32 : need64;
  p->threads = ctx->k1_threads;
  p->waves_per_block = p->threads / kWave;
  p->lpl = (p->price32 && ctx->k1_lpl == 8) ? 8 : 4;
  p->tile_lines = kWave * p->lpl;
  // besides the tables every wave stages finished order totals: stage_cap slots + 64 scratch slots
  auto stage_bytes = [&](uint32_t cap) {
    return (size_t)p->waves_per_block * ((cap + kWave) * sizeof(StageT) + (line_prices ? 2048 : 0));
  };
  auto fits = [&](size_t table_bytes) { return table_bytes + stage_bytes(kTile) + kLdsReserve <= (size_t)kMaxLds; };
  p->lds_tables = !ctx->force_global_tables && fits(p->lds_bytes);
  if (!p->lds_tables && !ctx->force_global_tables && !line_prices) {
    const size_t price_bytes = (size_t)ctx->p_pad * (p->price32 ? 4 : 8), hist_bytes = (size_t)ctx->p_pad * 4;
    if (fits(price_bytes)) {
      // Both tables do not fit, the price table alone does: keep it (a price is read for every line in the
      // middle of the dependent work of a tile) and leave the stock deltas to k_stock.
      p->lds_tables = true;
      p->hist_lds = false;
      p->lds_bytes = price_bytes;
      p->lpl = 4;
      p->tile_lines = kTile;
    } else if (fits(hist_bytes)) {
      // 64-bit prices: only the histogram fits.  It stays in LDS and the prices are gathered from global
      // memory, where a table of this size stays in L2.
      p->lds_tables = true;
      p->price_lds = false;
      p->price32 = false;
      p->lds_bytes = hist_bytes;
      p->lpl = 4;
      p->tile_lines = kTile;
    }
  }
  if (!p->lds_tables) {
    p->lds_bytes = 0; p->price32 = false; p->price_lds = false; p->hist_lds = false; p->lpl = 4; p->tile_lines = kTile;
  }
  // as many staging slots as the tables leave room for (in steps of 64, at most 1024, at least one tile's worth)
  p->stage_cap = ctx->k1_stage >= kTile ? (uint32_t)ctx->k1_stage : 1024;
  while (p->stage_cap > (uint32_t)kTile && p->lds_bytes + stage_bytes(p->stage_cap) + kLdsReserve > (size_t)kMaxLds)
    p->stage_cap -= kWave;
  p->flush_reserve = std::min<uint32_t>((uint32_t)p->tile_lines, p->stage_cap / 2);
  // the workgroup writes its finished totals together every `burst` tiles: as long a stretch as the slots hold at
  // the 64 orders a tile of the reference's shape closes (a wave that runs out of slots earlier flushes on its own)
  p->burst = ctx->k1_burst <= 0 ? 0 : std::max(1, std::min(ctx->k1_burst, (int)((p->stage_cap - p->flush_reserve) / 76)));
  p->lds_bytes += stage_bytes(p->stage_cap);
  int bpc = ctx->k1_blocks_per_cu;
  {
    const int fit = (int)((size_t)kMaxLds / (p->lds_bytes + kLdsReserve));
    bpc = std::max(1, std::min(bpc, fit));
  }
  bpc = std::max(1, std::min(bpc, 2048 / p->threads));
  int max_blocks = (ctx->k1_cus > 0 ? std::min(ctx->k1_cus, ctx->n_cu) : ctx->n_cu) * bpc;
  max_blocks = std::min(max_blocks, kMaxWavesTotal / p->waves_per_block);
  p->n_tiles = (n_lines + p->tile_lines - 1) / p->tile_lines;
  // Small inputs: at least eight tiles per wave while that leaves a thousand waves.  Every wave costs the record
  // block two records, and a launch that short is paid for in latency, not in bytes (1 M lines: 0.044 -> 0.039 ms
  // per call, 4 M: 0.054 -> 0.050 ms; from 16 M lines on the cap below is what binds).
  const int64_t max_waves = std::min<int64_t>((int64_t)max_blocks * p->waves_per_block,
                                              std::max<int64_t>(1024, p->n_tiles / 8));
  p->tiles_per_wave = std::max<int64_t>(1, (p->n_tiles + max_waves - 1) / max_waves);
  if (ctx->k1_tpw > 0) p->tiles_per_wave = std::max<int64_t>(p->tiles_per_wave, ctx->k1_tpw);   // developer switch
  p->n_active_waves = (int)((p->n_tiles + p->tiles_per_wave - 1) / p->tiles_per_wave);
  p->blocks = std::max(1, (p->n_active_waves + p->waves_per_block - 1) / p->waves_per_block);
  return ACMPC_OK;
}

struct CustPlan {
  bool lds_tables;
  int n_ranges, span, n_streams, blocks, threads;
  size_t lds_bytes;
  int64_t n_tiles, tiles_per_stream;
  int c_pad;
};

// leave_slots: workgroup slots per XCD label that are left to other workgroups of the same launch (the record
// and merge blocks that ride along with the customer pass): a launch that does not fit the CUs in one go runs
// its last workgroups after the first have finished, which nearly doubles its time.
void plan_customers(const acmpc_ctx* ctx, int64_t n_orders, int32_t n_customers, CustPlan* p, int leave_slots = 0) {
  p->threads = ctx->k2_threads;
  const int cus = ctx->k2_cus > 0 ? std::min(ctx->k2_cus, ctx->n_cu) : ctx->n_cu;
  const int slots = std::max(1, cus / 8);                   // workgroups per XCD label, one per CU
  // Every sibling workgroup of a stream re-reads the stream's orders through L2, so the number of
  // customer ranges R is kept as small as the LDS allows: 32-bit partial sums, one workgroup's LDS.
  const int64_t max_span = ((int64_t)kMaxLds - kLdsReserve - kWave * 4) / 4;
  const int64_t r64 = std::max<int64_t>(1, (n_customers + max_span - 1) / max_span);
  p->lds_tables = !ctx->force_global_tables && r64 <= slots;
  p->n_tiles = (n_orders + kTile - 1) / kTile;
  p->c_pad = (n_customers + 3) & ~3;
  if (!p->lds_tables) {
    p->n_ranges = 1;
    p->span = n_customers;
    p->lds_bytes = 0;
    const int64_t want = std::max<int64_t>(1, (p->n_tiles + 63) / 64);
    const int groups = (int)std::min<int64_t>(slots, want);
    p->n_streams = 8 * groups;
  } else {
    const int r = (int)r64;
    p->n_ranges = r;
    p->span = (n_customers + r - 1) / r;
    p->span = (p->span + 3) & ~3;
    p->lds_bytes = (size_t)p->span * 4 + kWave * 4;      // partial sums + one scratch word per lane
    const int max_groups = std::max(1, (slots - leave_slots) / r);
    // small inputs: fewer streams, so that every stream still has a few tiles per wave
    const int64_t want = std::max<int64_t>(1, (p->n_tiles + 8 * 32 - 1) / (8 * 32));
    const int groups = (int)std::min<int64_t>(max_groups, want);
    p->n_streams = 8 * groups;
  }
  p->blocks = p->n_streams * p->n_ranges;
  p->tiles_per_stream = std::max<int64_t>(1, (p->n_tiles + p->n_streams - 1) / p->n_streams);
}

// The stock-delta pass of its own (k_stock), for product tables whose histogram does not fit LDS.
struct StockPlan {
  bool use;
  int n_ranges, span, n_streams, blocks;
  size_t lds_bytes;
  int64_t n_tiles, tiles_per_stream;
  uint32_t q_small;
};

void plan_stock(const acmpc_ctx* ctx, int64_t n_lines, StockPlan* p) {
  const int64_t max_span = (((int64_t)kMaxLds - kLdsReserve) / 4) & ~(int64_t)3;
  const int64_t r64 = std::max<int64_t>(1, (ctx->n_products + max_span - 1) / max_span);
  p->use = !ctx->force_global_tables && n_lines > 0 && r64 <= 64;
  if (!p->use) return;
  const int slots = std::max(1, ctx->n_cu / 8);
  p->n_ranges = (int)r64;
  p->span = (int)((ctx->n_products + r64 - 1) / r64);
  p->span = (p->span + 3) & ~3;
  p->lds_bytes = (size_t)p->span * 4;
  p->n_tiles = (n_lines + kTile - 1) / kTile;
  int groups = std::max(1, slots / p->n_ranges);
  // small inputs: every stream keeps a few tiles per wave; large tables: at most 512 MB of slabs
  groups = (int)std::min<int64_t>(groups, std::max<int64_t>(1, (p->n_tiles + 8 * 32 - 1) / (8 * 32)));
  groups = (int)std::min<int64_t>(groups, std::max<int64_t>(1, ((int64_t)512 << 20) / ((int64_t)ctx->p_pad * 4 * 8)));
  p->n_streams = 8 * groups;
  p->blocks = p->n_streams * p->n_ranges;
  p->tiles_per_stream = std::max<int64_t>(1, (p->n_tiles + p->n_streams - 1) / p->n_streams);
  // a workgroup adds at most the lines of its stream, each below q_small, into a 32-bit cell
  const uint64_t lim = 0xffffffffull / std::max<uint64_t>(1, (uint64_t)p->tiles_per_stream * kTile);
  p->q_small = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(65536, lim));
}

int clear_carries(acmpc_ctx* ctx, hipStream_t st) {
  if (ctx->d_prod_carry.p)
    HIPCHK(ctx, hipMemsetAsync(ctx->d_prod_carry.p, 0, ctx->d_prod_carry.cap * 8, st));
  if (ctx->d_cust_carry.p)
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cust_carry.p, 0, ctx->d_cust_carry.cap * 8, st));
  ctx->carry_dirty = false;
  return ACMPC_OK;
}

template <typename PriceT, bool kLds, int LPL, bool kLife = false, bool kSnap = false, bool kPriceGlobal = false,
          bool kNoHist = false>
int launch_lines(acmpc_ctx* ctx, const LinesPlan& pl, const LinesArgs& a, hipStream_t st) {
  auto kern = k_lines<PriceT, kLds, LPL, kLife, kSnap, kPriceGlobal, kNoHist>;
  {
    int rc = set_dyn_lds(ctx, kern, pl.lds_bytes);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(pl.threads), pl.lds_bytes, st, a);
  HIPCHK(ctx, hipGetLastError());
  return ACMPC_OK;
}

template <typename PriceT, bool kLds, bool kLife = false>
int launch_lines_any(acmpc_ctx* ctx, const LinesPlan& pl, size_t lds_bytes, const LinesArgs& a, hipStream_t st) {
  auto kern = k_lines_any<PriceT, kLds, kLife>;
  if (lds_bytes) {
    int rc = set_dyn_lds(ctx, kern, lds_bytes);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(pl.threads), lds_bytes, st, a);
  HIPCHK(ctx, hipGetLastError());
  return ACMPC_OK;
}

// The order-restoring path (acmpc_partition.hpp): queues count, scan, two scatter passes and the per-bucket
// reduce.  On return order_total holds every order's total; with !totals_only the stock-delta slabs of
// `blocks` workgroups are in prod_slab as after k_lines.
constexpr size_t kPartLdsA = (size_t)2 * kPartCntPitch * 4 + 34 * 8 + (size_t)kPartRound * (8 + 2);
constexpr size_t kPartLdsB = (size_t)2 * (kPartMaxBuckets + 4) * 4 + (size_t)kPartMaxBuckets * 8 + (size_t)kPartRoundB * (8 + 2);

bool partition_fits(const acmpc_ctx* ctx, const LinesPlan& lp, int64_t n_orders) {
  return ctx->partition && !ctx->force_global_tables && lp.lds_tables && lp.price_lds && lp.hist_lds && lp.price32 &&
         n_orders <= ((int64_t)kPartMaxBuckets << kPartMaxShift) &&
         (size_t)ctx->p_pad * 8 + kPartLdsA + kLdsReserve <= (size_t)kMaxLds;
}

int launch_partition(acmpc_ctx* ctx, PartArgs& a, int blocks, hipStream_t st) {
  int rc;
  const size_t n = ((size_t)std::max<int64_t>(a.n_lines, 1) + 1) & ~(size_t)1;     // (the reduce reads whole pairs of records)
  for (int k = 0; k < 2; ++k)
    if ((rc = ensure(ctx, ctx->d_part_rec[k], n))) return rc;
  const size_t n_meta = (size_t)kPartMaxBuckets * 3 + kPartCoarse + 24;   // (+ 8 of bucket_off, changes, descents, mode)
  if ((rc = ensure(ctx, ctx->d_part_meta, n_meta))) return rc;
  a.counts = ctx->d_part_meta.p;
  a.bucket_off = a.counts + kPartMaxBuckets;
  a.cursor_b = a.bucket_off + kPartMaxBuckets + 8;
  a.cursor_a = a.cursor_b + kPartMaxBuckets;
  a.changes = a.cursor_a + kPartCoarse;
  a.descents = a.changes + 1;
  a.mode = a.changes + 2;
  ctx->d_part_mode = a.mode;
  a.rec_a = ctx->d_part_rec[0].p;
  a.rec_b = ctx->d_part_rec[1].p;
  // at most 2048 buckets: 16384 orders each up to 2^25 orders, 32768 up to 2^26, 65536 up to 2^27 (partition_fits)
  a.shift = std::max((int)kPartShift, ctx->part_shift);     // (the switch lets small inputs exercise the larger buckets)
  while (((int64_t)kPartMaxBuckets << a.shift) < a.n_orders) ++a.shift;
  a.val_bits = 64 - 11 - a.shift;
  a.n_buckets = (int32_t)(((int64_t)a.n_orders + ((int64_t)1 << a.shift) - 1) >> a.shift);
  HIPCHK(ctx, hipMemsetAsync(a.counts, 0, (size_t)kPartMaxBuckets * 8, st));
  HIPCHK(ctx, hipMemsetAsync(a.changes, 0, 16, st));     // changes and descents
  a.state_bits = nullptr;
  if (a.order_state && !a.totals_only) {                 // (only the stock deltas look at the states)
    const size_t n_words = ((size_t)a.n_orders + 31) / 32;
    if ((rc = ensure(ctx, ctx->d_part_state_bits, n_words))) return rc;
    a.state_bits = ctx->d_part_state_bits.p;
    hipLaunchKernelGGL(k_part_state_bits, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n_words + 255) / 256, (size_t)ctx->n_cu * 8))), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(k_part_count, dim3(blocks), dim3(kPartThreads), 0, st, a);
  hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(kPartThreads), 0, st, a);
  const size_t lds_a = (size_t)ctx->p_pad * 8 + kPartLdsA;
  if (a.line_price || a.order_state) {
    if ((rc = set_dyn_lds(ctx, k_part_scatter_a<true>, lds_a))) return rc;
    hipLaunchKernelGGL(k_part_scatter_a<true>, dim3(blocks), dim3(kPartThreads), lds_a, st, a);
  } else {
    if ((rc = set_dyn_lds(ctx, k_part_scatter_a<false>, lds_a))) return rc;
    hipLaunchKernelGGL(k_part_scatter_a<false>, dim3(blocks), dim3(kPartThreads), lds_a, st, a);
  }
  if ((rc = set_dyn_lds(ctx, k_part_scatter_b, kPartLdsB))) return rc;
  hipLaunchKernelGGL(k_part_scatter_b, dim3(blocks), dim3(kPartThreads), kPartLdsB, st, a);
  const size_t lds_c = (size_t)kPartBucket * 8;
  if ((rc = set_dyn_lds(ctx, k_part_reduce, lds_c))) return rc;
  hipLaunchKernelGGL(k_part_reduce, dim3(std::max(1, a.n_buckets) << (a.shift - kPartShift)), dim3(kPartThreads), lds_c, st, a);
  HIPCHK(ctx, hipGetLastError());
  return ACMPC_OK;
}

int aggregate_device_impl(acmpc_ctx* ctx, const int32_t* d_order_id, const int32_t* d_product_idx,
                          const int32_t* d_qty, int64_t n_lines, const int32_t* d_order_customer,
                          int64_t n_orders, int32_t base, int32_t n_customers, int64_t* d_tot,
                          int64_t* d_pq, int64_t* d_ct, hipStream_t st, const acmpc_lifecycle* life = nullptr,
                          unsigned char* fixed_ws = nullptr) {
  // fixed_ws: null = the context's own workspace, grown with hipMalloc as needed; otherwise the device
  // address of the fixed scratch array, and this call allocates nothing (and refuses what does not fit).
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_aggregate_device: null context");
  // a lifecycle argument without any column is no lifecycle argument (and takes the plain path's three launches)
  if (life && !life->line_price_cents && !life->order_state && !life->order_descuento) life = nullptr;
  const int64_t* d_line_price = life ? life->line_price_cents : nullptr;
  const uint8_t* d_state = life ? life->order_state : nullptr;
  const int32_t* d_desc = life ? life->order_descuento : nullptr;
  if (life) {
    if (d_desc && life->descuento_mode != ACMPC_DESCUENTO_UNITS && life->descuento_mode != ACMPC_DESCUENTO_PERCENT)
      return fail(ctx, ACMPC_E_INVALID, "unknown descuento_mode %d", life->descuento_mode);
    if (!aligned16(d_line_price) || !aligned16(d_state) || !aligned16(d_desc))
      return fail(ctx, ACMPC_E_ALIGN, "lifecycle columns must be 16-byte aligned");
  }
  // In the grouped mode the discounts are taken off where the totals are stored (k_lines, k_fixup); the
  // any-order kernels add partial sums with atomics, so there k_order_adjust goes over the finished totals.
  const bool desc_in_lines = d_desc && ctx->line_order == ACMPC_ORDER_GROUPED && n_lines > 0;
  const bool life_lines = d_line_price || d_state || desc_in_lines;
  if (ctx->n_products <= 0) return fail(ctx, ACMPC_E_STATE, "acmpc_set_products has not been called");
  if (n_lines < 0 || n_orders < 0 || n_customers <= 0)
    return fail(ctx, ACMPC_E_INVALID, "negative size or n_customers <= 0");
  if (n_orders > 0x7fffffffll || n_lines > (1ll << 40))
    return fail(ctx, ACMPC_E_INVALID, "n_orders must fit int32 (order ids are INT, store.sql:98,159)");
  if (n_customers > kMaxCustomers) return fail(ctx, ACMPC_E_INVALID, "more than %d customers", kMaxCustomers);
  if ((n_lines > 0 && (!d_order_id || !d_product_idx || !d_qty)) || (n_orders > 0 && (!d_order_customer || !d_tot)) ||
      !d_pq || !d_ct)
    return fail(ctx, ACMPC_E_INVALID, "null device pointer");
  if (n_lines > 0 && n_orders == 0)
    return fail(ctx, ACMPC_E_INVALID, "n_lines > 0 with n_orders == 0");
  if (!aligned16(d_order_id) || !aligned16(d_product_idx) || !aligned16(d_qty) ||
      !aligned16(d_order_customer) || !aligned16(d_tot))
    return fail(ctx, ACMPC_E_ALIGN, "device columns must be 16-byte aligned");
  ENTER_DEVICE(ctx);

  const int32_t P = ctx->n_products;
  // the workspace this call uses
  struct {
    uint32_t* prod_slab; unsigned long long* prod_carry; int32_t* rec_id; int64_t* rec_sum;
    uint32_t* edge_idx; int64_t* edge_total; uint32_t* cust_slab; unsigned long long* cust_carry; uint32_t* status;
  } w = {};
  w.status = ctx->d_status;
  if (fixed_ws) {
    if (life || ctx->line_order != ACMPC_ORDER_GROUPED || ctx->force_global_tables)
      return fail(ctx, ACMPC_E_INVALID, "the allocation-free entry point serves the grouped path without lifecycle columns only");
    if (P > kNaMaxProducts || n_customers > kNaMaxCustomers)
      return fail(ctx, ACMPC_E_INVALID, "the allocation-free entry point takes at most %d products and %d customers (have %d, %d)",
                  kNaMaxProducts, kNaMaxCustomers, P, n_customers);
    w.prod_slab = reinterpret_cast<uint32_t*>(fixed_ws + kNaOffProdSlab);
    w.cust_slab = reinterpret_cast<uint32_t*>(fixed_ws + kNaOffCustSlab);
    w.prod_carry = reinterpret_cast<unsigned long long*>(fixed_ws + kNaOffProdCarry);
    w.cust_carry = reinterpret_cast<unsigned long long*>(fixed_ws + kNaOffCustCarry);
    w.rec_id = reinterpret_cast<int32_t*>(fixed_ws + kNaOffRecId);
    w.rec_sum = reinterpret_cast<int64_t*>(fixed_ws + kNaOffRecSum);
    w.edge_idx = reinterpret_cast<uint32_t*>(fixed_ws + kNaOffEdgeIdx);
    w.edge_total = reinterpret_cast<int64_t*>(fixed_ws + kNaOffEdgeTotal);
    // The array is shared by every call on this GPU and keeps the last call's contents: the 64-bit carry
    // cells must start at zero (everything else is written before it is read).
    HIPCHK(ctx, hipMemsetAsync(w.prod_carry, 0, kNaProdCarryBytes, st));
    HIPCHK(ctx, hipMemsetAsync(w.cust_carry, 0, kNaCustCarryBytes, st));
  } else if (ctx->carry_dirty) { int rc = clear_carries(ctx, st); if (rc) return rc; }
  // one workspace buffer: from the context (allocating) or checked against its fixed region
  auto need = [&](auto& buf, auto*& ptr, size_t n, size_t fixed_bytes, bool zero_new) -> int {
    if (fixed_ws) {
      if (n * sizeof(*ptr) > fixed_bytes)
        return fail(ctx, ACMPC_E_INVALID, "the allocation-free entry point: a workspace region of %zu bytes is too small for this input (%zu needed)",
                    fixed_bytes, n * sizeof(*ptr));
      return ACMPC_OK;
    }
    const int rc_ = ensure(ctx, buf, n, zero_new, st);
    ptr = buf.p;
    return rc_;
  };

  if (ctx->profiling) HIPCHK(ctx, hipEventRecord(ctx->ev[0], st));

  // ---- order-line pass ------------------------------------------------------------------
  LinesPlan lp;
  const bool any_order = ctx->line_order == ACMPC_ORDER_ANY;
  const bool xpose_prices = d_line_price && !any_order;
  plan_lines(ctx, n_lines, &lp, xpose_prices);
  if ((any_order || life_lines) && lp.lpl != 4) {   // these kernels work on 256-line tiles
    const int keep = ctx->k1_lpl;
    ctx->k1_lpl = 4;
    plan_lines(ctx, n_lines, &lp, xpose_prices);
    ctx->k1_lpl = keep;
  }
  int rc;
  // The grouped path (lifecycle columns or not) runs in three launches: the record block of k_fixup rides along
  // with the customer pass, and one tail kernel does the merges.  Rows in any order keep the four-launch sequence.
  CustPlan cp;
  if (n_orders > 0) plan_customers(ctx, n_orders, n_customers, &cp);
  // (The record block needs some 40 us next to the customer pass: below a few million orders that pass is
  // shorter and the four-launch sequence is quicker.  ACMPC_FUSE=2 forces the fused path for tests.)
  const bool fused = ctx->fuse && ctx->line_order == ACMPC_ORDER_GROUPED && n_lines > 0 &&
                     (n_orders >= 4000000 || (ctx->fuse == 2 && n_orders > 0)) &&
                     lp.lds_tables && lp.hist_lds && cp.lds_tables && cp.threads == kFixThreads;
  if (fused && cp.blocks + 8 > ctx->n_cu) plan_customers(ctx, n_orders, n_customers, &cp, 1);   // room for the leading blocks
  FixArgs f;
  memset(&f, 0, sizeof(f));
  const size_t fix_lds = (size_t)kFixThreads * kFixStrip * 12;
  if (n_lines == 0) {
    if (n_orders > 0) HIPCHK(ctx, hipMemsetAsync(d_tot, 0, (size_t)n_orders * 8, st));
    HIPCHK(ctx, hipMemsetAsync(d_pq, 0, (size_t)P * 8, st));
    if (d_desc && n_orders > 0) {
      hipLaunchKernelGGL(k_order_adjust, dim3((unsigned)((n_orders + 255) / 256)), dim3(256), 0, st, d_tot, d_desc,
                         n_orders, life->descuento_mode, (const unsigned long long*)nullptr);
      HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->profiling) { HIPCHK(ctx, hipEventRecord(ctx->ev[1], st)); HIPCHK(ctx, hipEventRecord(ctx->ev[2], st)); }
  } else {
    if ((rc = need(ctx->d_prod_carry, w.prod_carry, (size_t)ctx->p_pad, kNaProdCarryBytes, true))) return rc;
    if ((rc = need(ctx->d_rec_id, w.rec_id, (size_t)2 * kMaxWavesTotal, kNaRecBytes, false))) return rc;
    if ((rc = need(ctx->d_rec_sum, w.rec_sum, (size_t)2 * kMaxWavesTotal, kNaRecBytes, false))) return rc;
    if (lp.hist_lds && (rc = need(ctx->d_prod_slab, w.prod_slab, (size_t)lp.blocks * ctx->p_pad, kNaProdSlabBytes, false))) return rc;
    // neither product table fits LDS: the line pass gathers prices from global memory and leaves the
    // stock deltas to a pass of their own (rather than one global atomic per line)
    StockPlan sp;
    sp.use = false;
    if (!lp.hist_lds) plan_stock(ctx, n_lines, &sp);
    if (fixed_ws && !lp.hist_lds)
      return fail(ctx, ACMPC_E_INVALID, "the allocation-free entry point needs a product table that fits LDS");
    if (sp.use && (rc = need(ctx->d_prod_slab, w.prod_slab, (size_t)sp.n_streams * ctx->p_pad, kNaProdSlabBytes, false))) return rc;

    LinesArgs a;
    a.order_id = d_order_id; a.product_idx = d_product_idx; a.qty = d_qty;
    a.n_lines = n_lines; a.n_tiles = lp.n_tiles; a.tiles_per_wave = lp.tiles_per_wave;
    a.base = base; a.n_orders = (int32_t)n_orders;
    a.price = ctx->d_price.p; a.n_products = P; a.p_pad = ctx->p_pad;
    {
      // a workgroup adds at most lines_per_block quantities below q_small into a 32-bit cell
      const uint64_t lines_per_block = (uint64_t)lp.tiles_per_wave * lp.tile_lines * lp.waves_per_block;
      const uint64_t lim = 0xffffffffull / std::max<uint64_t>(1, lines_per_block);
      a.q_small = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(65536, lim));
      a.stage_cap = lp.stage_cap;
      a.flush_reserve = lp.flush_reserve;
      a.tile_lines = lp.tile_lines;
    }
    a.order_total = d_tot; a.prod_slab = w.prod_slab; a.prod_carry = w.prod_carry;
    a.rec_id = w.rec_id; a.rec_sum = w.rec_sum; a.status = w.status;
    a.line_price = reinterpret_cast<const long long*>(d_line_price);
    a.price_in_lds = lp.price_lds ? 1 : 0;
    ctx->part_ran = false;
    a.only_if_unsorted = 0;
    a.totals_only = 0;
    a.fill_edges = fused ? 1 : 0;
    a.burst = lp.burst;
    a.order_descuento = desc_in_lines ? d_desc : nullptr;
    a.descuento_mode = d_desc ? life->descuento_mode : 0;
    a.order_state = d_state;
    a.no_hist = sp.use ? 1 : 0;
    a.run_if_zero = nullptr;
    a.run_if_sorted = nullptr;
    const unsigned long long* sorted_gate = nullptr;   // any-order rows: where the device says that they were grouped

    if (!fixed_ws) ctx->carry_dirty = true;      // cleared again once the merge kernels are queued
    // rows in any order: the order-restoring path when the sizes allow it, else one global atomic per run
    // (with a line-price column the grouped kernel that takes rows which turn out grouped needs a plan of its
    // own, with the LDS through which it re-orders the price loads; the path is only taken if both plans agree
    // on the workgroups, whose stock-delta slabs are merged whichever kernel wrote them)
    LinesPlan lpx = lp;
    if (any_order && d_line_price) plan_lines(ctx, n_lines, &lpx, true);
    const bool part_ok = !fixed_ws && partition_fits(ctx, lp, n_orders) && lpx.blocks == lp.blocks &&
                         lpx.n_active_waves == lp.n_active_waves && lpx.tiles_per_wave == lp.tiles_per_wave && lpx.lds_tables;
    const int part_blocks = lp.blocks;
    PartArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.order_id = d_order_id; pa.product_idx = d_product_idx; pa.qty = d_qty; pa.n_lines = n_lines;
    pa.base = base; pa.n_orders = (int32_t)n_orders; pa.price = ctx->d_price.p; pa.n_products = P; pa.p_pad = ctx->p_pad;
    {
      const uint64_t rounds = ((uint64_t)n_lines + kPartRound - 1) / kPartRound;
      const uint64_t per_block = ((rounds + part_blocks - 1) / part_blocks) * kPartRound;
      pa.q_small = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(65536, 0xffffffffull / std::max<uint64_t>(1, per_block)));
    }
    pa.prod_carry = w.prod_carry; pa.order_total = d_tot; pa.status = w.status;
    pa.line_price = reinterpret_cast<const long long*>(d_line_price);
    pa.order_state = d_state;
    pa.order_descuento = d_desc;                          // (taken off where the reduce writes the totals)
    pa.descuento_mode = d_desc ? life->descuento_mode : 0;
    const unsigned long long* moved_gate = nullptr;       // where the device says that the reduce wrote net totals
    if (any_order && part_ok) {
      // count + decide on the device; then either the moving passes or the atomic kernel does the work
      pa.prod_slab = w.prod_slab;                       // (lp.blocks slabs, whichever path fills them)
      if ((rc = launch_partition(ctx, pa, lp.blocks, st))) return rc;
      hipLaunchKernelGGL(k_zero_if_unsorted, dim3(ctx->n_cu * 4), dim3(256), 0, st, d_tot, n_orders, w.status, 0, pa.mode);
      HIPCHK(ctx, hipGetLastError());
      ctx->part_ran = true;
      moved_gate = pa.mode;
      a.run_if_zero = pa.mode;
      const size_t tables = (size_t)ctx->p_pad * 8;     // (partition_fits: both tables in LDS, 32-bit prices)
      if (life_lines) rc = launch_lines_any<int32_t, true, true>(ctx, lp, tables, a, st);
      else rc = launch_lines_any<int32_t, true>(ctx, lp, tables, a, st);
      if (rc) return rc;
      a.run_if_zero = nullptr;
      // ... or, if the ids never went down, the grouped kernel (its wave-boundary records are finished by k_fixup)
      a.run_if_sorted = pa.mode;
      if (d_line_price) {
        LinesArgs ax = a;                                 // the same launch with the plan that holds the price slices
        ax.stage_cap = lpx.stage_cap; ax.flush_reserve = lpx.flush_reserve; ax.burst = lpx.burst;
        rc = launch_lines<int64_t, true, 4, true, true>(ctx, lpx, ax, st);
      }
      else if (life_lines) rc = launch_lines<int32_t, true, 4, true>(ctx, lp, a, st);
      else rc = launch_lines<int32_t, true, 4>(ctx, lp, a, st);
      if (rc) return rc;
      a.run_if_sorted = nullptr;
      sorted_gate = pa.mode;
    }
    else if (any_order) {
      HIPCHK(ctx, hipMemsetAsync(d_tot, 0, (size_t)n_orders * 8, st));
      const size_t tables = !lp.lds_tables ? 0 : (size_t)ctx->p_pad * ((lp.price_lds ? (lp.price32 ? 4 : 8) : 0) + (lp.hist_lds ? 4 : 0));
      if (life_lines) {
        if (!lp.lds_tables) rc = launch_lines_any<int64_t, false, true>(ctx, lp, tables, a, st);
        else if (lp.price32) rc = launch_lines_any<int32_t, true, true>(ctx, lp, tables, a, st);
        else rc = launch_lines_any<int64_t, true, true>(ctx, lp, tables, a, st);
      }
      else if (!lp.lds_tables) rc = launch_lines_any<int64_t, false>(ctx, lp, tables, a, st);
      else if (lp.price32) rc = launch_lines_any<int32_t, true>(ctx, lp, tables, a, st);
      else rc = launch_lines_any<int64_t, true>(ctx, lp, tables, a, st);
    }
    else if (life_lines) {
      // with a line-price column the price table is not read: one price type serves
      if (d_line_price) rc = lp.lds_tables ? launch_lines<int64_t, true, 4, true, true>(ctx, lp, a, st)
                                           : launch_lines<int64_t, false, 4, true, true>(ctx, lp, a, st);
      else if (!lp.lds_tables) rc = launch_lines<int64_t, false, 4, true>(ctx, lp, a, st);
      else if (!lp.price_lds) rc = launch_lines<int64_t, true, 4, true, false, true>(ctx, lp, a, st);
      else if (!lp.hist_lds) rc = lp.price32 ? launch_lines<int32_t, true, 4, true, false, false, true>(ctx, lp, a, st)
                                             : launch_lines<int64_t, true, 4, true, false, false, true>(ctx, lp, a, st);
      else if (lp.price32) rc = launch_lines<int32_t, true, 4, true>(ctx, lp, a, st);
      else rc = launch_lines<int64_t, true, 4, true>(ctx, lp, a, st);
    }
    else if (!lp.lds_tables) rc = launch_lines<int64_t, false, 4>(ctx, lp, a, st);
    else if (!lp.price_lds) rc = launch_lines<int64_t, true, 4, false, false, true>(ctx, lp, a, st);
    else if (!lp.hist_lds) rc = lp.price32 ? launch_lines<int32_t, true, 4, false, false, false, true>(ctx, lp, a, st)
                                           : launch_lines<int64_t, true, 4, false, false, false, true>(ctx, lp, a, st);
    else if (lp.price32 && lp.lpl == 8) rc = launch_lines<int32_t, true, 8>(ctx, lp, a, st);
    else if (lp.price32) rc = launch_lines<int32_t, true, 4>(ctx, lp, a, st);
    else rc = launch_lines<int64_t, true, 4>(ctx, lp, a, st);
    if (rc) return rc;
    if (sp.use) {
      StockArgs k;
      k.order_id = d_order_id; k.product_idx = d_product_idx; k.qty = d_qty;
      k.n_lines = n_lines; k.n_tiles = sp.n_tiles; k.tiles_per_stream = sp.tiles_per_stream;
      k.n_products = P; k.p_pad = ctx->p_pad; k.span = sp.span; k.n_ranges = sp.n_ranges; k.n_streams = sp.n_streams;
      k.q_small = sp.q_small; k.prod_slab = w.prod_slab; k.prod_carry = w.prod_carry;
      k.base = base; k.n_orders = (int32_t)n_orders; k.order_state = d_state;
      if ((rc = set_dyn_lds(ctx, k_stock, sp.lds_bytes))) return rc;
      hipLaunchKernelGGL(k_stock, dim3(sp.blocks), dim3(1024), sp.lds_bytes, st, k);
      HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->profiling) HIPCHK(ctx, hipEventRecord(ctx->ev[1], st));

    f.rec_id = w.rec_id; f.rec_sum = w.rec_sum; f.n_rec = 2 * lp.n_active_waves;
    f.base = base; f.n_orders = (int32_t)n_orders; f.order_total = d_tot; f.status = w.status;
    f.prod_slab = w.prod_slab; f.n_slabs = sp.use ? sp.n_streams : lp.blocks; f.p_pad = ctx->p_pad; f.n_products = P;
    f.merge_products = (lp.hist_lds || sp.use) ? 1 : 0;
    f.fix_orders = (any_order && !sorted_gate) ? 0 : 1;
    f.fix_if_sorted = sorted_gate;
    f.prod_carry = w.prod_carry; f.product_qty = d_pq;
    f.ablate = ctx->ablate_fix;
    f.order_descuento = a.order_descuento; f.descuento_mode = a.descuento_mode;
    if (!fused) {
      if ((rc = set_dyn_lds(ctx, k_fixup, fix_lds))) return rc;
      const int fix_blocks = 1 + (f.merge_products ? (P + kMergeProducts - 1) / kMergeProducts : 0);
      hipLaunchKernelGGL(k_fixup, dim3(fix_blocks), dim3(kFixThreads), fix_lds, st, f);
      HIPCHK(ctx, hipGetLastError());
    }
    if (!f.merge_products) {
      hipLaunchKernelGGL(k_move_clear, dim3((P + 255) / 256), dim3(256), 0, st, w.prod_carry, d_pq, P);
      HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->line_order == ACMPC_ORDER_AUTO) {
      // If the grouped pass met descending ids its order totals are wrong (the stock deltas are not):
      // zero them and redo them with the any-order kernel.  All three launches leave at once otherwise.
      LinesPlan ap;
      plan_lines(ctx, n_lines, &ap, false);
      if (ap.lpl != 4) { const int keep = ctx->k1_lpl; ctx->k1_lpl = 4; plan_lines(ctx, n_lines, &ap, false); ctx->k1_lpl = keep; }
      LinesArgs b = a;
      b.n_tiles = ap.n_tiles; b.tiles_per_wave = ap.tiles_per_wave; b.tile_lines = ap.tile_lines;
      b.price_in_lds = ap.price_lds ? 1 : 0; b.only_if_unsorted = 1; b.totals_only = 1;
      if (part_ok) {
        // count how far from grouped the rows are and let the order-restoring path redo the totals if they are
        // far (every kernel of it leaves at once if the rows were grouped after all)
        pa.totals_only = 1; pa.only_if_unsorted = 1;
        if ((rc = launch_partition(ctx, pa, lp.blocks, st))) return rc;
        ctx->part_ran = true;
        moved_gate = pa.mode;
        b.run_if_zero = pa.mode;
      }
      {
      hipLaunchKernelGGL(k_zero_if_unsorted, dim3(ctx->n_cu * 4), dim3(256), 0, st, d_tot, n_orders, w.status, 1, b.run_if_zero);
      HIPCHK(ctx, hipGetLastError());
      const size_t tables = !ap.lds_tables ? 0 : (size_t)ctx->p_pad * ((ap.price_lds ? (ap.price32 ? 4 : 8) : 0) + (ap.hist_lds ? 4 : 0));
      if (life_lines) {
        if (!ap.lds_tables) rc = launch_lines_any<int64_t, false, true>(ctx, ap, tables, b, st);
        else if (ap.price32) rc = launch_lines_any<int32_t, true, true>(ctx, ap, tables, b, st);
        else rc = launch_lines_any<int64_t, true, true>(ctx, ap, tables, b, st);
      }
      else if (!ap.lds_tables) rc = launch_lines_any<int64_t, false>(ctx, ap, tables, b, st);
      else if (ap.price32) rc = launch_lines_any<int32_t, true>(ctx, ap, tables, b, st);
      else rc = launch_lines_any<int64_t, true>(ctx, ap, tables, b, st);
      if (rc) return rc;
      }
      hipLaunchKernelGGL(k_clear_unsorted, dim3(1), dim3(64), 0, st, w.status);
      HIPCHK(ctx, hipGetLastError());
    }
    if (d_desc && !desc_in_lines) {   // totals are final after k_fixup: take the discounts off (unless the reduce of the
                                      // order-restoring path wrote them, net already)
      hipLaunchKernelGGL(k_order_adjust, dim3((unsigned)((n_orders + 255) / 256)), dim3(256), 0, st, d_tot, d_desc,
                         n_orders, life->descuento_mode, moved_gate);
      HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->profiling) HIPCHK(ctx, hipEventRecord(ctx->ev[2], st));
  }

  // ---- per-customer pass ----------------------------------------------------------------
  if (n_orders == 0) {
    HIPCHK(ctx, hipMemsetAsync(d_ct, 0, (size_t)n_customers * 8, st));
    if (ctx->profiling) { HIPCHK(ctx, hipEventRecord(ctx->ev[3], st)); HIPCHK(ctx, hipEventRecord(ctx->ev[4], st)); }
  } else {
    if ((rc = need(ctx->d_cust_carry, w.cust_carry, (size_t)n_customers, kNaCustCarryBytes, true))) return rc;
    if (cp.lds_tables && (rc = need(ctx->d_cust_slab, w.cust_slab, (size_t)cp.n_streams * cp.c_pad, kNaCustSlabBytes, false))) return rc;
    CustArgs c;
    c.order_customer = d_order_customer; c.order_total = d_tot; c.n_orders = n_orders;
    c.n_tiles = cp.n_tiles; c.tiles_per_stream = cp.tiles_per_stream;
    c.n_customers = n_customers; c.c_pad = cp.c_pad; c.span = cp.span; c.n_ranges = cp.n_ranges;
    c.n_streams = cp.n_streams; c.cust_slab = w.cust_slab; c.cust_carry = w.cust_carry;
    c.status = w.status;
    c.ablate = ctx->ablate_k2;
    c.order_state = d_state;
    if (!fixed_ws) ctx->carry_dirty = true;
    if (fused) {
      if ((rc = need(ctx->d_edge_idx, w.edge_idx, (size_t)2 * kMaxWavesTotal, kNaRecBytes, false))) return rc;
      if ((rc = need(ctx->d_edge_total, w.edge_total, (size_t)2 * kMaxWavesTotal, kNaRecBytes, false))) return rc;
      f.deferred = 1;
      f.order_customer = d_order_customer; f.cust_carry = w.cust_carry; f.n_customers = n_customers;
      f.edge_idx = w.edge_idx; f.edge_total = w.edge_total; f.order_state = d_state;
      CustFusedArgs fa;
      fa.c = c;
      fa.f = f;
      const size_t lds = std::max(cp.lds_bytes, fix_lds);
      // as many leading workgroups as there are CUs the customer pass leaves idle (8 .. 64)
      fa.lead = std::max(8, std::min(64, (ctx->n_cu - cp.blocks) / 8 * 8));
      if (d_state) {
        if ((rc = set_dyn_lds(ctx, k_customers_fused<true>, lds))) return rc;
        hipLaunchKernelGGL(k_customers_fused<true>, dim3(cp.blocks + fa.lead), dim3(cp.threads), lds, st, fa);
      } else {
        if ((rc = set_dyn_lds(ctx, k_customers_fused<false>, lds))) return rc;
        hipLaunchKernelGGL(k_customers_fused<false>, dim3(cp.blocks + fa.lead), dim3(cp.threads), lds, st, fa);
      }
    } else if (cp.lds_tables && d_state) {
      if ((rc = set_dyn_lds(ctx, k_customers<true, true>, cp.lds_bytes))) return rc;
      hipLaunchKernelGGL((k_customers<true, true>), dim3(cp.blocks), dim3(cp.threads), cp.lds_bytes, st, c);
    } else if (cp.lds_tables) {
      if ((rc = set_dyn_lds(ctx, k_customers<true>, cp.lds_bytes))) return rc;
      hipLaunchKernelGGL(k_customers<true>, dim3(cp.blocks), dim3(cp.threads), cp.lds_bytes, st, c);
    } else if (d_state) {
      hipLaunchKernelGGL((k_customers<false, true>), dim3(cp.blocks), dim3(cp.threads), 0, st, c);
    } else {
      hipLaunchKernelGGL(k_customers<false>, dim3(cp.blocks), dim3(cp.threads), 0, st, c);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->profiling) HIPCHK(ctx, hipEventRecord(ctx->ev[3], st));
    if (fused) {
      TailArgs t;
      t.m.cust_slab = w.cust_slab; t.m.n_streams = cp.n_streams; t.m.c_pad = cp.c_pad;
      t.m.n_customers = n_customers; t.m.cust_carry = w.cust_carry; t.m.customer_total = d_ct;
      t.f = f;
      t.n_cust_blocks = (n_customers + 255) / 256;
      t.n_prod_blocks = 0;          // the leading workgroups of the fused launch merged the stock deltas
      hipLaunchKernelGGL(k_tail, dim3(t.n_cust_blocks + t.n_prod_blocks + (f.n_rec + 255) / 256), dim3(256), 0, st, t);
    } else if (cp.lds_tables) {
      CustMergeArgs m;
      m.cust_slab = w.cust_slab; m.n_streams = cp.n_streams; m.c_pad = cp.c_pad;
      m.n_customers = n_customers; m.cust_carry = w.cust_carry; m.customer_total = d_ct;
      hipLaunchKernelGGL(k_custmerge, dim3((n_customers + 255) / 256), dim3(256), 0, st, m);
    } else {
      hipLaunchKernelGGL(k_move_clear, dim3((n_customers + 255) / 256), dim3(256), 0, st,
                         w.cust_carry, d_ct, n_customers);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->profiling) HIPCHK(ctx, hipEventRecord(ctx->ev[4], st));
  }
  if (!fixed_ws) ctx->carry_dirty = false;
  if (ctx->profiling) ctx->ev_recorded = true;
  return ACMPC_OK;
}

// ---- synthetic data, host part ---------------------------------------------------------

int synth_check(acmpc_ctx* ctx, const acmpc_synth_cfg* c) {
  if (!c) return fail(ctx, ACMPC_E_INVALID, "null synth config");
  if (c->n_lines_total < 0 || c->n_products <= 0 || c->n_customers <= 0 || c->world <= 0 || c->rank < 0 ||
      c->rank >= c->world)
    return fail(ctx, ACMPC_E_INVALID, "bad synth config");
  return ACMPC_OK;
}

inline int synth_len(int64_t o) { return 1 + (int)(synth_hash(kSeedLen, (unsigned long long)o) % 7ull); }

int64_t synth_count_orders(int64_t n_lines_total) {
  int64_t acc = 0, o = 0;
  while (acc < n_lines_total) { acc += synth_len(o); ++o; }
  return o;
}

// Line offsets of orders lo and hi, and (optionally) of every 1024th order from lo on.
void synth_offsets(int64_t n_lines_total, int64_t lo, int64_t hi, int64_t* line_lo, int64_t* line_hi,
                   std::vector<int64_t>* block_off) {
  int64_t acc = 0;
  for (int64_t o = 0; o < hi; ++o) {
    if (o == lo) *line_lo = acc;
    if (block_off && o >= lo && ((o - lo) % kSynthOrdersPerBlock) == 0) block_off->push_back(acc);
    acc += synth_len(o);
    if (acc > n_lines_total) acc = n_lines_total;
  }
  if (lo >= hi) *line_lo = acc;
  *line_hi = acc;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================

ACMPC_API const char* acmpc_version(void) { return "acmpc 0.1 gfx950"; }

ACMPC_API const char* acmpc_last_error(const acmpc_ctx* ctx) { return ctx ? ctx->err : g_init_error; }

ACMPC_API int acmpc_init(acmpc_ctx** out, int device) {
  if (!out) return fail(nullptr, ACMPC_E_INVALID, "acmpc_init: null out pointer");
  *out = nullptr;
  if (device < 0) return fail(nullptr, ACMPC_E_INVALID, "acmpc_init: device must be >= 0 (there is no CPU backend)");
  int n = 0;
  HIPCHK(nullptr, hipGetDeviceCount(&n));
  if (device >= n) return fail(nullptr, ACMPC_E_INVALID, "acmpc_init: device %d of %d", device, n);
  DeviceScope init_scope;
  HIPCHK(nullptr, init_scope.enter(device));
  hipDeviceProp_t prop;
  HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, ACMPC_E_INVALID, "acmpc_init: device %d is %s, this library is built for gfx950 only",
                device, prop.gcnArchName);
  acmpc_ctx* ctx = new (std::nothrow) acmpc_ctx();
  if (!ctx) return fail(nullptr, ACMPC_E_NOMEM, "acmpc_init: out of host memory");
  strcpy(ctx->err, "no error");
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  ctx->k1_threads = env_int("ACMPC_K1_THREADS", 1024);
  if (ctx->k1_threads != 256 && ctx->k1_threads != 512 && ctx->k1_threads != 1024) ctx->k1_threads = 1024;
  ctx->k1_blocks_per_cu = std::max(1, env_int("ACMPC_K1_BPC", 1));
  ctx->k1_lpl = env_int("ACMPC_K1_LPL", 4);
  ctx->k2_threads = env_int("ACMPC_K2_THREADS", 1024);
  ctx->fuse = env_int("ACMPC_FUSE", 1);
  ctx->partition = env_int("ACMPC_PARTITION", 1);
  if (env_int("ACMPC_DEV_ABLATE", 0) == 1) {
    ctx->ablate_fix = env_int("ACMPC_FIX_ABLATE", 0);
    ctx->ablate_k2 = env_int("ACMPC_K2_ABLATE", 0);
    ctx->ablate_ing = env_int("ACMPC_ING_ABLATE", 0);
  }
  ctx->k1_burst = std::max(0, env_int("ACMPC_K1_BURST", 10));
  ctx->k1_cus = env_int("ACMPC_K1_CUS", 0);
  ctx->k1_tpw = env_int("ACMPC_K1_TPW", 0);
  ctx->k1_stage = std::min(env_int("ACMPC_K1_STAGE", 0), 4096);
  ctx->k2_cus = env_int("ACMPC_K2_CUS", 0);
  if (ctx->k2_threads != 256 && ctx->k2_threads != 512 && ctx->k2_threads != 1024) ctx->k2_threads = 1024;
  ctx->force_global_tables = env_int("ACMPC_FORCE_GLOBAL_TABLES", 0);
  ctx->part_shift = std::max(0, std::min(env_int("ACMPC_PART_SHIFT", 0), (int)kPartMaxShift));
  ctx->ing_blocks = std::max(0, std::min(env_int("ACMPC_ING_BLOCKS", 0), 1 << 16));
  if (device < kMaxDevices && !g_scratch_ptr[device].load(std::memory_order_acquire)) {
    // `device` is current here.  The array's name is not a device address: this call returns the one the
    // kernels can use on this device.  A failure only matters to the allocation-free entry point, which then
    // finds no address and says so; every other call works without the array.
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[device].load(std::memory_order_relaxed)) {
      unsigned char* sp = nullptr;
      const hipError_t es = hipGetSymbolAddress((void**)&sp, HIP_SYMBOL(scratch));
      if (es == hipSuccess && sp) g_scratch_ptr[device].store(sp, std::memory_order_release);
      else (void)hipGetLastError();
    }
  }
  hipError_t e = hipMalloc((void**)&ctx->d_status, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(ctx->d_status, 0, sizeof(uint32_t));
  for (int i = 0; i <= ACMPC_N_KERNELS && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
  if (e != hipSuccess) {
    fail(nullptr, ACMPC_E_HIP, "acmpc_init: %s", hipGetErrorString(e));
    acmpc_destroy(ctx);
    return ACMPC_E_HIP;
  }
  *out = ctx;
  return ACMPC_OK;
}

ACMPC_API void acmpc_destroy(acmpc_ctx* ctx) {
  if (!ctx) return;
  DeviceScope destroy_scope;
  (void)destroy_scope.enter(ctx->device);
  (void)hipDeviceSynchronize();
  release(ctx->d_price); release(ctx->d_prod_slab); release(ctx->d_prod_carry);
  release(ctx->d_rec_id); release(ctx->d_rec_sum); release(ctx->d_cust_slab); release(ctx->d_cust_carry);
  release(ctx->s_order_id); release(ctx->s_product_idx); release(ctx->s_qty); release(ctx->s_order_customer);
  release(ctx->s_order_total); release(ctx->s_product_qty); release(ctx->s_customer_total); release(ctx->s_merged);
  release(ctx->s_synth_off); release(ctx->s_synth_cdf);
  release(ctx->d_edge_idx); release(ctx->d_edge_total);
  release(ctx->d_part_rec[0]); release(ctx->d_part_rec[1]);
  release(ctx->d_part_meta); release(ctx->d_part_state_bits);
  release(ctx->d_dict); release(ctx->d_dict_disp); release(ctx->d_codes); release(ctx->d_chunk_rows); release(ctx->d_ing_tables);
  release(ctx->s_text); release(ctx->d_chunk_bytes); release(ctx->s_emit_value);
  release(ctx->s_line_price); release(ctx->s_order_state); release(ctx->s_order_descuento);
  for (void* p : ctx->host_blocks) (void)hipHostFree(p);
  ctx->host_blocks.clear();
  if (ctx->d_ing_scalars) (void)hipFree(ctx->d_ing_scalars);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  for (int i = 0; i <= ACMPC_N_KERNELS; ++i)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  for (hipStream_t cs : ctx->copy_streams)
    if (cs) (void)hipStreamDestroy(cs);
  delete ctx;
}

ACMPC_API int acmpc_set_products(acmpc_ctx* ctx, const int64_t* price_cents, int32_t n_products) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_set_products: null context");
  if (!price_cents || n_products <= 0) return fail(ctx, ACMPC_E_INVALID, "acmpc_set_products: empty product table");
  if (n_products > kMaxProducts) return fail(ctx, ACMPC_E_INVALID, "acmpc_set_products: more than %d products", kMaxProducts);
  ENTER_DEVICE(ctx);
  HIPCHK(ctx, hipDeviceSynchronize());          // nothing may still be reading the old table
  int rc = ensure(ctx, ctx->d_price, (size_t)n_products + 1);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpy(ctx->d_price.p, price_cents, (size_t)n_products * 8, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemset(ctx->d_price.p + n_products, 0, 8));      // the null product: price 0
  bool fits = true;
  for (int32_t i = 0; i < n_products; ++i)
    if (price_cents[i] < INT32_MIN || price_cents[i] > INT32_MAX) { fits = false; break; }
  ctx->price_fits32 = fits;
  ctx->n_products = n_products;
  ctx->p_pad = (n_products + 1 + 3) & ~3;                         // tables hold the null product too
  // the carry workspace is indexed by product: start it clean at the new size
  release(ctx->d_prod_carry);
  if ((rc = ensure(ctx, ctx->d_prod_carry, (size_t)ctx->p_pad, true))) return rc;
  return ACMPC_OK;
}

ACMPC_API int acmpc_aggregate_device(acmpc_ctx* ctx, const int32_t* d_order_id, const int32_t* d_product_idx,
                                     const int32_t* d_qty, int64_t n_lines, const int32_t* d_order_customer,
                                     int64_t n_orders, int32_t order_id_base, int32_t n_customers,
                                     int64_t* d_out_order_total_cents, int64_t* d_out_product_qty,
                                     int64_t* d_out_customer_total_cents, void* hip_stream) {
  return aggregate_device_impl(ctx, d_order_id, d_product_idx, d_qty, n_lines, d_order_customer, n_orders,
                               order_id_base, n_customers, d_out_order_total_cents, d_out_product_qty,
                               d_out_customer_total_cents, (hipStream_t)hip_stream);
}

ACMPC_API int acmpc_aggregate_device_noalloc(acmpc_ctx* ctx, const int32_t* d_order_id, const int32_t* d_product_idx,
                                             const int32_t* d_qty, int64_t n_lines, const int32_t* d_order_customer,
                                             int64_t n_orders, int32_t order_id_base, int32_t n_customers,
                                             int64_t* d_out_order_total_cents, int64_t* d_out_product_qty,
                                             int64_t* d_out_customer_total_cents, void* hip_stream) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_aggregate_device_noalloc: null context");
  unsigned char* ws = ctx->device < kMaxDevices ? g_scratch_ptr[ctx->device].load(std::memory_order_acquire) : nullptr;   // set by acmpc_init
  if (!ws) return fail(ctx, ACMPC_E_STATE, "acmpc_aggregate_device_noalloc: no scratch address for device %d", ctx->device);
  return aggregate_device_impl(ctx, d_order_id, d_product_idx, d_qty, n_lines, d_order_customer, n_orders,
                               order_id_base, n_customers, d_out_order_total_cents, d_out_product_qty,
                               d_out_customer_total_cents, (hipStream_t)hip_stream, nullptr, ws);
}

ACMPC_API int acmpc_aggregate_lifecycle_device(acmpc_ctx* ctx, const int32_t* d_order_id, const int32_t* d_product_idx,
                                               const int32_t* d_qty, int64_t n_lines, const int32_t* d_order_customer,
                                               int64_t n_orders, int32_t order_id_base, int32_t n_customers,
                                               const acmpc_lifecycle* life, int64_t* d_out_order_total_cents,
                                               int64_t* d_out_product_qty, int64_t* d_out_customer_total_cents,
                                               void* hip_stream) {
  return aggregate_device_impl(ctx, d_order_id, d_product_idx, d_qty, n_lines, d_order_customer, n_orders,
                               order_id_base, n_customers, d_out_order_total_cents, d_out_product_qty,
                               d_out_customer_total_cents, (hipStream_t)hip_stream, life);
}

ACMPC_API int acmpc_status(acmpc_ctx* ctx, void* hip_stream, uint32_t* flags) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_status: null context");
  ENTER_DEVICE(ctx);
  hipStream_t st = (hipStream_t)hip_stream;
  uint32_t f = 0;
  HIPCHK(ctx, hipMemcpyAsync(&f, ctx->d_status, sizeof(f), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(f), st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  if (flags) *flags = f;
  if (f)
    return fail(ctx, ACMPC_E_DATA, "data contract broken:%s%s%s%s%s%s%s",
                (f & ACMPC_ST_ORDER_RANGE) ? " order_id outside [base, base+M);" : "",
                (f & ACMPC_ST_ORDER_UNSORTED) ? " order_id not non-decreasing;" : "",
                (f & ACMPC_ST_CUSTOMER_RANGE) ? " order_customer outside [0, C);" : "",
                (f & ACMPC_ST_INGEST_MALFORMED) ? " malformed detalle row in the text;" : "",
                (f & ACMPC_ST_INGEST_CAPACITY) ? " more detalle rows than capacity_rows;" : "",
                (f & ACMPC_ST_EMIT_CAPACITY) ? " emitted text needs more than capacity_bytes;" : "",
                (f & ACMPC_ST_INTERNAL) ? " internal: a chunk waited too long for its predecessors;" : "");
  return ACMPC_OK;
}

// Copies between host and device memory for the host-pointer entry points, waited for before returning.
// A copy from or to pageable memory goes through the runtime's staging buffers and holds its calling thread;
// cut into pieces that four threads move on streams of their own, the same bytes cross the link about one and a
// half times as fast (1.2 GB of pageable columns: 28 GB/s from one thread, 43 GB/s from three or more).  Small
// jobs are queued on `st` as they are.  `st` is waited for first: what the copies read or overwrite may still
// be in use by work queued there.
namespace {
struct HostCopy { void* dst; const void* src; size_t bytes; hipMemcpyKind kind; };
int host_copies(acmpc_ctx* ctx, const std::vector<HostCopy>& jobs, hipStream_t st) {
  constexpr size_t kMiB = (size_t)1 << 20;
  constexpr int kThreads = 4;
  size_t total = 0;
  for (const HostCopy& j : jobs) total += j.bytes;
  if (total < 32 * kMiB) {
    for (const HostCopy& j : jobs)
      if (j.bytes) HIPCHK(ctx, hipMemcpyAsync(j.dst, j.src, j.bytes, j.kind, st));
    return ACMPC_OK;
  }
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (hipStream_t& cs : ctx->copy_streams)
    if (!cs) HIPCHK(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  const size_t piece = std::max<size_t>(4 * kMiB, ((total / (2 * kThreads)) + kMiB - 1) / kMiB * kMiB);
  std::vector<HostCopy> pieces;
  for (const HostCopy& j : jobs)
    for (size_t off = 0; off < j.bytes; off += piece)
      pieces.push_back({static_cast<char*>(j.dst) + off, static_cast<const char*>(j.src) + off, std::min(piece, j.bytes - off), j.kind});
  hipError_t errs[kThreads];
  std::thread threads[kThreads];
  auto share = [&](int t) {
    hipError_t e = hipSetDevice(ctx->device);
    for (size_t k = (size_t)t; k < pieces.size() && e == hipSuccess; k += kThreads)
      e = hipMemcpyAsync(pieces[k].dst, pieces[k].src, pieces[k].bytes, pieces[k].kind, ctx->copy_streams[t]);
    const hipError_t w = hipStreamSynchronize(ctx->copy_streams[t]);
    errs[t] = e != hipSuccess ? e : w;
  };
  for (int t = 0; t < kThreads; ++t) {
    // a thread the system will not start (process or thread limit): its share is copied from this thread
    try { threads[t] = std::thread(share, t); } catch (const std::system_error&) { share(t); }
  }
  for (std::thread& t : threads)
    if (t.joinable()) t.join();
  for (hipError_t e : errs) HIPCHK(ctx, e);
  return ACMPC_OK;
}
}  // namespace

ACMPC_API int acmpc_aggregate(acmpc_ctx* ctx, const int32_t* order_id, const int32_t* product_idx,
                              const int32_t* qty, int64_t n_lines, const int32_t* order_customer,
                              int64_t n_orders, int32_t order_id_base, int32_t n_customers,
                              int64_t* out_order_total_cents, int64_t* out_product_qty,
                              int64_t* out_customer_total_cents) {
  return acmpc_aggregate_lifecycle(ctx, order_id, product_idx, qty, n_lines, order_customer, n_orders, order_id_base,
                                   n_customers, nullptr, out_order_total_cents, out_product_qty,
                                   out_customer_total_cents);
}

ACMPC_API int acmpc_aggregate_lifecycle(acmpc_ctx* ctx, const int32_t* order_id, const int32_t* product_idx,
                                        const int32_t* qty, int64_t n_lines, const int32_t* order_customer,
                                        int64_t n_orders, int32_t order_id_base, int32_t n_customers,
                                        const acmpc_lifecycle* life, int64_t* out_order_total_cents,
                                        int64_t* out_product_qty, int64_t* out_customer_total_cents) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_aggregate: null context");
  if (ctx->n_products <= 0) return fail(ctx, ACMPC_E_STATE, "acmpc_set_products has not been called");
  if (n_lines < 0 || n_orders < 0 || n_customers <= 0) return fail(ctx, ACMPC_E_INVALID, "negative size or n_customers <= 0");
  if (n_orders > 0x7fffffffll || n_lines > (1ll << 40) || n_customers > kMaxCustomers)
    return fail(ctx, ACMPC_E_INVALID, "size beyond the limits (n_orders int32, n_lines 2^40, n_customers %d)", kMaxCustomers);
  if ((n_lines > 0 && (!order_id || !product_idx || !qty)) || (n_orders > 0 && (!order_customer || !out_order_total_cents)) ||
      !out_product_qty || !out_customer_total_cents)
    return fail(ctx, ACMPC_E_INVALID, "null host pointer");
  return guarded(ctx, "acmpc_aggregate", [&]() -> int {
  ENTER_DEVICE(ctx);
  const int32_t P = ctx->n_products;
  int rc;
  if ((rc = ensure(ctx, ctx->s_order_id, (size_t)n_lines))) return rc;
  if ((rc = ensure(ctx, ctx->s_product_idx, (size_t)n_lines))) return rc;
  if ((rc = ensure(ctx, ctx->s_qty, (size_t)n_lines))) return rc;
  if ((rc = ensure(ctx, ctx->s_order_customer, (size_t)n_orders))) return rc;
  if ((rc = ensure(ctx, ctx->s_order_total, (size_t)n_orders))) return rc;
  if ((rc = ensure(ctx, ctx->s_product_qty, (size_t)P))) return rc;
  if ((rc = ensure(ctx, ctx->s_customer_total, (size_t)n_customers))) return rc;
  hipStream_t st = nullptr;
  // this call reports on its own input only: flags left by earlier device calls are dropped
  HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), st));
  std::vector<HostCopy> up;
  if (n_lines > 0) {
    up.push_back({ctx->s_order_id.p, order_id, (size_t)n_lines * 4, hipMemcpyHostToDevice});
    up.push_back({ctx->s_product_idx.p, product_idx, (size_t)n_lines * 4, hipMemcpyHostToDevice});
    up.push_back({ctx->s_qty.p, qty, (size_t)n_lines * 4, hipMemcpyHostToDevice});
  }
  if (n_orders > 0) up.push_back({ctx->s_order_customer.p, order_customer, (size_t)n_orders * 4, hipMemcpyHostToDevice});
  acmpc_lifecycle dl = {nullptr, nullptr, nullptr, 0};
  if (life) {
    dl.descuento_mode = life->descuento_mode;
    if (life->line_price_cents && n_lines > 0) {
      if ((rc = ensure(ctx, ctx->s_line_price, (size_t)n_lines))) return rc;
      up.push_back({ctx->s_line_price.p, life->line_price_cents, (size_t)n_lines * 8, hipMemcpyHostToDevice});
      dl.line_price_cents = ctx->s_line_price.p;
    }
    if (life->order_state && n_orders > 0) {
      if ((rc = ensure(ctx, ctx->s_order_state, (size_t)n_orders))) return rc;
      up.push_back({ctx->s_order_state.p, life->order_state, (size_t)n_orders, hipMemcpyHostToDevice});
      dl.order_state = ctx->s_order_state.p;
    }
    if (life->order_descuento && n_orders > 0) {
      if ((rc = ensure(ctx, ctx->s_order_descuento, (size_t)n_orders))) return rc;
      up.push_back({ctx->s_order_descuento.p, life->order_descuento, (size_t)n_orders * 4, hipMemcpyHostToDevice});
      dl.order_descuento = ctx->s_order_descuento.p;
    }
  }
  if ((rc = host_copies(ctx, up, st))) return rc;
  rc = aggregate_device_impl(ctx, ctx->s_order_id.p, ctx->s_product_idx.p, ctx->s_qty.p, n_lines,
                             ctx->s_order_customer.p, n_orders, order_id_base, n_customers, ctx->s_order_total.p,
                             ctx->s_product_qty.p, ctx->s_customer_total.p, st, life ? &dl : nullptr);
  if (rc) return rc;
  std::vector<HostCopy> down;
  if (n_orders > 0) down.push_back({out_order_total_cents, ctx->s_order_total.p, (size_t)n_orders * 8, hipMemcpyDeviceToHost});
  down.push_back({out_product_qty, ctx->s_product_qty.p, (size_t)P * 8, hipMemcpyDeviceToHost});
  down.push_back({out_customer_total_cents, ctx->s_customer_total.p, (size_t)n_customers * 8, hipMemcpyDeviceToHost});
  if ((rc = host_copies(ctx, down, st))) return rc;
  uint32_t flags = 0;
  return acmpc_status(ctx, st, &flags);
  });
}

ACMPC_API int acmpc_set_option(acmpc_ctx* ctx, int option, int64_t value) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_set_option: null context");
  if (option == ACMPC_OPT_LINE_ORDER &&
      (value == ACMPC_ORDER_GROUPED || value == ACMPC_ORDER_ANY || value == ACMPC_ORDER_AUTO)) {
    ctx->line_order = (int)value;
    return ACMPC_OK;
  }
  return fail(ctx, ACMPC_E_INVALID, "acmpc_set_option: unknown option %d or value %lld", option, (long long)value);
}

ACMPC_API int acmpc_last_line_order_path(acmpc_ctx* ctx, void* hip_stream, int* path) {
  if (!ctx || !path) return fail(ctx, ACMPC_E_INVALID, "acmpc_last_line_order_path: null argument");
  *path = ACMPC_PATH_NONE;
  if (!ctx->part_ran || !ctx->d_part_mode) return ACMPC_OK;
  ENTER_DEVICE(ctx);
  hipStream_t st = (hipStream_t)hip_stream;
  unsigned long long mode = 0;
  HIPCHK(ctx, hipMemcpyAsync(&mode, ctx->d_part_mode, sizeof(mode), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  *path = (int)mode;
  return ACMPC_OK;
}

ACMPC_API int acmpc_set_profiling(acmpc_ctx* ctx, int enabled) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_set_profiling: null context");
  ctx->profiling = enabled != 0;
  ctx->ev_recorded = false;
  return ACMPC_OK;
}

ACMPC_API int acmpc_last_kernel_ms(acmpc_ctx* ctx, float* ms) {
  if (!ctx || !ms) return fail(ctx, ACMPC_E_INVALID, "acmpc_last_kernel_ms: null argument");
  if (!ctx->profiling || !ctx->ev_recorded)
    return fail(ctx, ACMPC_E_STATE, "no profiled acmpc_aggregate_device call to report");
  ENTER_DEVICE(ctx);
  HIPCHK(ctx, hipEventSynchronize(ctx->ev[ACMPC_N_KERNELS]));
  for (int i = 0; i < ACMPC_N_KERNELS; ++i) HIPCHK(ctx, hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]));
  return ACMPC_OK;
}

ACMPC_API int acmpc_lines_geometry(acmpc_ctx* ctx, int32_t* n_workgroups, int32_t* waves_per_workgroup) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_lines_geometry: null context");
  LinesPlan lp;
  plan_lines(ctx, (int64_t)1 << 40, &lp);
  if (n_workgroups) *n_workgroups = lp.blocks;
  if (waves_per_workgroup) *waves_per_workgroup = lp.waves_per_block;
  return ACMPC_OK;
}

ACMPC_API int acmpc_host_alloc(acmpc_ctx* ctx, int64_t n_bytes, void** out) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_host_alloc: null context");
  if (!out) return fail(ctx, ACMPC_E_INVALID, "acmpc_host_alloc: null out pointer");
  *out = nullptr;
  if (n_bytes < 0) return fail(ctx, ACMPC_E_INVALID, "acmpc_host_alloc: negative size");
  ENTER_DEVICE(ctx);
  void* p = nullptr;
  HIPCHK(ctx, hipHostMalloc(&p, (size_t)std::max<int64_t>(n_bytes, 16), hipHostMallocDefault));
  const int rc = guarded(ctx, "acmpc_host_alloc", [&] { ctx->host_blocks.push_back(p); return (int)ACMPC_OK; });
  if (rc) { (void)hipHostFree(p); return rc; }
  *out = p;
  return ACMPC_OK;
}

ACMPC_API int acmpc_host_free(acmpc_ctx* ctx, void* p) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_host_free: null context");
  if (!p) return ACMPC_OK;
  auto it = std::find(ctx->host_blocks.begin(), ctx->host_blocks.end(), p);
  if (it == ctx->host_blocks.end())
    return fail(ctx, ACMPC_E_INVALID, "acmpc_host_free: not a block of this context");
  ENTER_DEVICE(ctx);
  ctx->host_blocks.erase(it);
  HIPCHK(ctx, hipHostFree(p));          // waits for copies that still use the block
  return ACMPC_OK;
}

// ---- synthetic data ----------------------------------------------------------------------

ACMPC_API int acmpc_synth_plan(const acmpc_synth_cfg* cfg, acmpc_synth_shard* out) {
  int rc = synth_check(nullptr, cfg);
  if (rc) return rc;
  if (!out) return fail(nullptr, ACMPC_E_INVALID, "acmpc_synth_plan: null out");
  const int64_t M = synth_count_orders(cfg->n_lines_total);
  out->n_orders_total = M;
  out->order_lo = (int64_t)(((__int128)M * cfg->rank) / cfg->world);
  out->order_hi = (int64_t)(((__int128)M * (cfg->rank + 1)) / cfg->world);
  synth_offsets(cfg->n_lines_total, out->order_lo, out->order_hi, &out->line_lo, &out->line_hi, nullptr);
  return ACMPC_OK;
}

ACMPC_API int acmpc_synth_prices(const acmpc_synth_cfg* cfg, int64_t* price_cents) {
  int rc = synth_check(nullptr, cfg);
  if (rc) return rc;
  if (!price_cents) return fail(nullptr, ACMPC_E_INVALID, "acmpc_synth_prices: null out");
  static const int64_t seed_rows[3] = {25000, 45000, 150000};   // store.sql:126-128
  for (int32_t p = 0; p < cfg->n_products; ++p)
    price_cents[p] = p < 3 ? seed_rows[p] : 1 + (int64_t)(synth_hash(kSeedPrice, (unsigned long long)p) % 500000ull);
  return ACMPC_OK;
}

namespace {
int synth_generate_impl(acmpc_ctx* ctx, const acmpc_synth_cfg* cfg, int32_t* d_order_id, int32_t* d_product_idx,
                        int32_t* d_qty, int32_t* d_order_customer, void* hip_stream);
}

ACMPC_API int acmpc_synth_generate_device(acmpc_ctx* ctx, const acmpc_synth_cfg* cfg, int32_t* d_order_id,
                                          int32_t* d_product_idx, int32_t* d_qty, int32_t* d_order_customer,
                                          void* hip_stream) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_synth_generate_device: null context");
  return guarded(ctx, "acmpc_synth_generate_device", [&] {
    return synth_generate_impl(ctx, cfg, d_order_id, d_product_idx, d_qty, d_order_customer, hip_stream);
  });
}

namespace {
int synth_generate_impl(acmpc_ctx* ctx, const acmpc_synth_cfg* cfg, int32_t* d_order_id, int32_t* d_product_idx,
                        int32_t* d_qty, int32_t* d_order_customer, void* hip_stream) {
  int rc = synth_check(ctx, cfg);
  if (rc) return rc;
  ENTER_DEVICE(ctx);
  hipStream_t st = (hipStream_t)hip_stream;
  const int64_t M = synth_count_orders(cfg->n_lines_total);
  const int64_t lo = (int64_t)(((__int128)M * cfg->rank) / cfg->world);
  const int64_t hi = (int64_t)(((__int128)M * (cfg->rank + 1)) / cfg->world);
  if (hi <= lo) return ACMPC_OK;
  if (!d_order_id || !d_product_idx || !d_qty || !d_order_customer)
    return fail(ctx, ACMPC_E_INVALID, "acmpc_synth_generate_device: null device pointer");
  std::vector<int64_t> off;
  int64_t line_lo = 0, line_hi = 0;
  synth_offsets(cfg->n_lines_total, lo, hi, &line_lo, &line_hi, &off);
  if ((rc = ensure(ctx, ctx->s_synth_off, off.size()))) return rc;
  // the previous generate call may still be reading the staging buffers
  HIPCHK(ctx, hipStreamSynchronize(st));
  HIPCHK(ctx, hipMemcpy(ctx->s_synth_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
  SynthArgs a;
  a.n_lines_total = cfg->n_lines_total; a.order_lo = lo; a.order_hi = hi; a.line_lo = line_lo;
  a.block_line_off = ctx->s_synth_off.p; a.cdf = nullptr; a.wtot = 0;
  a.n_products = cfg->n_products; a.n_customers = cfg->n_customers; a.base = cfg->order_id_base;
  a.qty_mode = cfg->qty_mode;
  a.order_id = d_order_id; a.product_idx = d_product_idx; a.qty = d_qty; a.order_customer = d_order_customer;
  if (cfg->zipf) {
    std::vector<unsigned long long> cdf((size_t)cfg->n_products);
    unsigned long long w = 0;
    for (int32_t k = 0; k < cfg->n_products; ++k) { w += (1ull << 32) / (unsigned long long)(k + 1); cdf[k] = w; }
    if ((rc = ensure(ctx, ctx->s_synth_cdf, cdf.size()))) return rc;
    HIPCHK(ctx, hipMemcpy(ctx->s_synth_cdf.p, cdf.data(), cdf.size() * 8, hipMemcpyHostToDevice));
    a.cdf = ctx->s_synth_cdf.p;
    a.wtot = w;
  }
  const int blocks = (int)((hi - lo + kSynthOrdersPerBlock - 1) / kSynthOrdersPerBlock);
  hipLaunchKernelGGL(k_synth, dim3(blocks), dim3(kSynthOrdersPerBlock), 0, st, a);
  HIPCHK(ctx, hipGetLastError());
  return ACMPC_OK;
}
}  // namespace

// ---- text ingest -------------------------------------------------------------------------

namespace {
int set_product_codes_impl(acmpc_ctx* ctx, const char* codes32, int32_t n_products);
}

ACMPC_API int acmpc_set_product_codes(acmpc_ctx* ctx, const char* codes32, int32_t n_products) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_set_product_codes: null context");
  return guarded(ctx, "acmpc_set_product_codes", [&] { return set_product_codes_impl(ctx, codes32, n_products); });
}

namespace {
int set_product_codes_impl(acmpc_ctx* ctx, const char* codes32, int32_t n_products) {
  if (!codes32 || n_products <= 0) return fail(ctx, ACMPC_E_INVALID, "acmpc_set_product_codes: empty code list");
  if (n_products > kMaxCodes) return fail(ctx, ACMPC_E_INVALID, "acmpc_set_product_codes: more than %d codes", kMaxCodes);
  ENTER_DEVICE(ctx);
  HIPCHK(ctx, hipDeviceSynchronize());
  uint32_t size = 16;
  while (size < 2u * (uint32_t)n_products + 2u) size <<= 1;
  std::vector<unsigned long long> codes((size_t)n_products * 4, 0ull);
  int32_t sql_bad = -1, csv_bad = -1;
  for (int32_t i = 0; i < n_products; ++i) {
    const char* c = codes32 + (size_t)i * 32;
    size_t len = 0;
    while (len < 32 && c[len]) ++len;
    if (len > 30) return fail(ctx, ACMPC_E_INVALID, "product code %d is longer than 30 bytes (varchar(30), store.sql:110)", i);
    if (sql_bad < 0 && (memchr(c, '\'', len) || memchr(c, '\\', len))) sql_bad = i;
    if (csv_bad < 0) {
      bool bad = len == 0 || c[0] == '\'' || c[0] == '"';
      for (size_t k = 0; k < len && !bad; ++k) bad = strchr(",\t \r\n)", c[k]) != nullptr;
      if (bad) csv_bad = i;
    }
    for (size_t k = 0; k < len; ++k)
      codes[(size_t)i * 4 + (k >> 3)] |= (unsigned long long)(unsigned char)c[k] << (8 * (k & 7));
  }
  {
    // CodigoProd is a primary key (store.sql:200): equal codes are refused
    std::vector<int32_t> order((size_t)n_products);
    for (int32_t i = 0; i < n_products; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
      const int c = memcmp(&codes[(size_t)x * 4], &codes[(size_t)y * 4], 32);
      return c != 0 ? c < 0 : x < y;
    });
    for (int32_t i = 1; i < n_products; ++i)
      if (memcmp(&codes[(size_t)order[(size_t)i - 1] * 4], &codes[(size_t)order[(size_t)i] * 4], 32) == 0)
        return fail(ctx, ACMPC_E_INVALID, "product codes %d and %d are equal (CodigoProd is a primary key, store.sql:200)",
                    order[(size_t)i - 1], order[(size_t)i]);
  }
  // Perfect hash by hash-and-displace: buckets (about four codes each) are placed largest first; a bucket's
  // displacement is the first d that sends all its codes to distinct free slots.  A seed that cannot be
  // completed (two codes with equal hash pairs) is replaced by the next one.
  uint32_t n_buckets = 2;
  while (n_buckets * 4u < (uint32_t)n_products) n_buckets <<= 1;
  uint32_t bucket_bits = 0;
  while ((1u << bucket_bits) < n_buckets) ++bucket_bits;
  const uint32_t bucket_shift = 32u - bucket_bits;
  IngSlot empty_slot;
  memset(&empty_slot, 0, sizeof(empty_slot));
  empty_slot.idx = -1;
  empty_slot.short_idx = -1;
  std::vector<IngSlot> dict;
  std::vector<unsigned short> disp;
  uint32_t seed = 0;
  bool built = false;
  for (; seed < 16 && !built; ++seed) {
    dict.assign(size, empty_slot);
    disp.assign(n_buckets, 0);
    std::vector<IngHash> hs((size_t)n_products);
    std::vector<std::vector<int32_t>> members(n_buckets);
    for (int32_t i = 0; i < n_products; ++i) {
      uint32_t kw[8];
      memcpy(kw, &codes[(size_t)i * 4], 32);         // little-endian host: the same eight words the kernel builds
      hs[(size_t)i] = ing_code_hash(kw, seed);
      members[hs[(size_t)i].a >> bucket_shift].push_back(i);
    }
    std::vector<uint32_t> by_size(n_buckets);
    for (uint32_t b = 0; b < n_buckets; ++b) by_size[b] = b;
    std::sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) {
      return members[x].size() != members[y].size() ? members[x].size() > members[y].size() : x < y;
    });
    built = true;
    std::vector<uint32_t> slots;
    for (uint32_t bi = 0; bi < n_buckets && built; ++bi) {
      const std::vector<int32_t>& m = members[by_size[bi]];
      if (m.empty()) break;
      uint32_t d = 0;
      for (; d < 65536; ++d) {
        slots.clear();
        bool fits = true;
        for (int32_t i : m) {
          const uint32_t s = ing_slot_of(hs[(size_t)i], d, size - 1);
          if (dict[s].idx >= 0 || std::find(slots.begin(), slots.end(), s) != slots.end()) { fits = false; break; }
          slots.push_back(s);
        }
        if (fits) break;
      }
      if (d == 65536) { built = false; break; }
      disp[by_size[bi]] = (unsigned short)d;
      for (size_t k = 0; k < m.size(); ++k) {
        IngSlot& slot = dict[slots[k]];
        uint32_t kw[8];
        memcpy(kw, &codes[(size_t)m[k] * 4], 32);
        memcpy(slot.head, kw, 12);
        memcpy(slot.tail, kw + 3, 20);
        slot.idx = m[k];
        slot.short_idx = (kw[3] | kw[4] | kw[5] | kw[6] | kw[7]) == 0u ? m[k] : -1;
      }
    }
    if (built) break;
  }
  if (!built) return fail(ctx, ACMPC_E_INVALID, "could not index the product codes (hash construction failed for 16 seeds)");
  int rc;
  if ((rc = ensure(ctx, ctx->d_dict, size))) return rc;
  if ((rc = ensure(ctx, ctx->d_dict_disp, n_buckets))) return rc;
  HIPCHK(ctx, hipMemcpy(ctx->d_dict_disp.p, disp.data(), (size_t)n_buckets * 2, hipMemcpyHostToDevice));
  if ((rc = ensure(ctx, ctx->d_codes, codes.size()))) return rc;
  HIPCHK(ctx, hipMemcpy(ctx->d_dict.p, dict.data(), (size_t)size * sizeof(IngSlot), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->d_codes.p, codes.data(), codes.size() * 8, hipMemcpyHostToDevice));
  if (!ctx->d_ing_scalars) HIPCHK(ctx, hipMalloc((void**)&ctx->d_ing_scalars, 2 * sizeof(long long)));
  if (!ctx->d_ing_tables.p) {
    std::vector<unsigned long long> tab(kIngTableWords + 256 / 8, 0ull);
    ing_build_tables(tab.data(), reinterpret_cast<unsigned char*>(tab.data() + kIngTableWords));
    if ((rc = ensure(ctx, ctx->d_ing_tables, tab.size()))) return rc;
    HIPCHK(ctx, hipMemcpy(ctx->d_ing_tables.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  }
  ctx->dict_mask = size - 1;
  ctx->dict_seed = seed;
  ctx->dict_bucket_shift = bucket_shift;
  ctx->dict_products = n_products;
  ctx->code_sql_bad = sql_bad;
  ctx->code_csv_bad = csv_bad;
  return ACMPC_OK;
}
}  // namespace

ACMPC_API int acmpc_ingest_detalle_device(acmpc_ctx* ctx, const char* d_text, int64_t n_bytes, int32_t* d_order_id,
                                          int32_t* d_product_idx, int32_t* d_qty, int64_t capacity_rows,
                                          int64_t* d_n_rows, void* hip_stream) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_ingest_detalle_device: null context");
  if (ctx->dict_products <= 0) return fail(ctx, ACMPC_E_STATE, "acmpc_set_product_codes has not been called");
  if (n_bytes < 0 || capacity_rows < 0 || !d_n_rows) return fail(ctx, ACMPC_E_INVALID, "bad size or null row counter");
  if (n_bytes > kMaxTextBytes) return fail(ctx, ACMPC_E_INVALID, "more than 2^44 bytes of text");
  if (n_bytes > 0 && !d_text) return fail(ctx, ACMPC_E_INVALID, "null text pointer");
  if (capacity_rows > 0 && (!d_order_id || !d_product_idx || !d_qty)) return fail(ctx, ACMPC_E_INVALID, "null column pointer");
  if (!aligned16(d_text)) return fail(ctx, ACMPC_E_ALIGN, "device text must be 16-byte aligned");
  ENTER_DEVICE(ctx);
  hipStream_t st = (hipStream_t)hip_stream;
  IngestArgs a;
  a.text = reinterpret_cast<const unsigned char*>(d_text);
  a.n_bytes = n_bytes;
  a.n_chunks = (n_bytes + kIngChunk - 1) / kIngChunk;
  int rc;
  constexpr size_t kTicketWords64 = (size_t)kIngStripes * kIngTicketPitch / 2;    // the ticket counters, behind the entries
  if ((rc = ensure(ctx, ctx->d_chunk_rows, (size_t)a.n_chunks + kTicketWords64))) return rc;
  a.chunk_state = ctx->d_chunk_rows.p;
  a.ticket = reinterpret_cast<unsigned int*>(ctx->d_chunk_rows.p + a.n_chunks);
  a.dict = ctx->d_dict.p; a.dict_disp = ctx->d_dict_disp.p; a.dict_mask = ctx->dict_mask;
  a.dict_seed = ctx->dict_seed; a.dict_bucket_shift = ctx->dict_bucket_shift;
  a.codes = ctx->d_codes.p;
  a.tables = ctx->d_ing_tables.p;
  a.ablate = ctx->ablate_ing;
  a.order_id = d_order_id; a.product_idx = d_product_idx; a.qty = d_qty; a.capacity = capacity_rows;
  a.n_rows = reinterpret_cast<long long*>(d_n_rows);
  a.n_malformed = reinterpret_cast<unsigned long long*>(ctx->d_ing_scalars + 1);
  a.status = ctx->d_status;
  HIPCHK(ctx, hipMemsetAsync(ctx->d_ing_scalars, 0, 2 * sizeof(long long), st));
  if (a.n_chunks == 0) {
    HIPCHK(ctx, hipMemsetAsync(d_n_rows, 0, sizeof(int64_t), st));
    return ACMPC_OK;
  }
  // entries and tickets start at zero
  HIPCHK(ctx, hipMemsetAsync(ctx->d_chunk_rows.p, 0, ((size_t)a.n_chunks + kTicketWords64) * 8, st));
  // (any grid size gives the same rows: fewer workgroups than stripes, or more than the chip holds at once —
  // see the forward-progress note in acmpc_ingest.hpp; the switch exists so that a test can run both)
  const int blocks = (int)std::min<int64_t>(a.n_chunks, ctx->ing_blocks > 0 ? (int64_t)ctx->ing_blocks
                                                                             : (int64_t)ctx->n_cu * kIngBlocksPerCu);
  a.n_stripes = std::min(blocks, kIngStripes);
  if (a.ablate) hipLaunchKernelGGL(k_ingest_parse<true>, dim3(blocks), dim3(kIngThreads), 0, st, a);
  else hipLaunchKernelGGL(k_ingest_parse<false>, dim3(blocks), dim3(kIngThreads), 0, st, a);
  HIPCHK(ctx, hipGetLastError());
  return ACMPC_OK;
}

ACMPC_API int acmpc_ingest_detalle(acmpc_ctx* ctx, const char* text, int64_t n_bytes, int32_t* order_id,
                                   int32_t* product_idx, int32_t* qty, int64_t capacity_rows, int64_t* n_rows,
                                   int64_t* n_malformed) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_ingest_detalle: null context");
  if (ctx->dict_products <= 0) return fail(ctx, ACMPC_E_STATE, "acmpc_set_product_codes has not been called");
  if (n_bytes < 0 || capacity_rows < 0 || !n_rows || (n_bytes > 0 && !text))
    return fail(ctx, ACMPC_E_INVALID, "acmpc_ingest_detalle: bad argument");
  if (capacity_rows > 0 && (!order_id || !product_idx || !qty)) return fail(ctx, ACMPC_E_INVALID, "null column pointer");
  return guarded(ctx, "acmpc_ingest_detalle", [&]() -> int {
  ENTER_DEVICE(ctx);
  int rc;
  if ((rc = ensure(ctx, ctx->s_text, (size_t)n_bytes + 16))) return rc;
  if ((rc = ensure(ctx, ctx->s_order_id, (size_t)capacity_rows))) return rc;
  if ((rc = ensure(ctx, ctx->s_product_idx, (size_t)capacity_rows))) return rc;
  if ((rc = ensure(ctx, ctx->s_qty, (size_t)capacity_rows))) return rc;
  hipStream_t st = nullptr;
  HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), st));   // this call reports on its own text only
  if (n_bytes > 0 && (rc = host_copies(ctx, {{ctx->s_text.p, text, (size_t)n_bytes, hipMemcpyHostToDevice}}, st))) return rc;
  rc = acmpc_ingest_detalle_device(ctx, reinterpret_cast<const char*>(ctx->s_text.p), n_bytes, ctx->s_order_id.p,
                                   ctx->s_product_idx.p, ctx->s_qty.p, capacity_rows,
                                   reinterpret_cast<int64_t*>(ctx->d_ing_scalars), st);
  if (rc) return rc;
  long long scal[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(scal, ctx->d_ing_scalars, sizeof(scal), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  *n_rows = scal[0];
  if (n_malformed) *n_malformed = scal[1];
  const int64_t got = std::min<int64_t>(scal[0], capacity_rows);
  if (got > 0 && (rc = host_copies(ctx, {{order_id, ctx->s_order_id.p, (size_t)got * 4, hipMemcpyDeviceToHost},
                                        {product_idx, ctx->s_product_idx.p, (size_t)got * 4, hipMemcpyDeviceToHost},
                                        {qty, ctx->s_qty.p, (size_t)got * 4, hipMemcpyDeviceToHost}}, st))) return rc;
  if (got > 0) HIPCHK(ctx, hipStreamSynchronize(st));      // (small results are only queued by host_copies)
  uint32_t flags = 0;
  return acmpc_status(ctx, st, &flags);
  });
}

// ---- text emit ---------------------------------------------------------------------------

namespace {

template <int Kind, int Style>
int launch_emit(acmpc_ctx* ctx, const EmitArgs& a, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>(a.n_chunks, (int64_t)ctx->n_cu * 16);
  hipLaunchKernelGGL((k_emit_count<Kind, Style>), dim3(blocks), dim3(kEmitThreads), 0, st, a);
  HIPCHK(ctx, hipGetLastError());
  ScanArgs sc{a.chunk_bytes, a.n_chunks, a.n_bytes, a.capacity, a.status, kStEmitCapacity};
  hipLaunchKernelGGL(k_chunk_scan, dim3(1), dim3(1024), 0, st, sc);
  HIPCHK(ctx, hipGetLastError());
  hipLaunchKernelGGL((k_emit_write<Kind, Style>), dim3(blocks), dim3(kEmitThreads), 0, st, a);
  HIPCHK(ctx, hipGetLastError());
  return ACMPC_OK;
}

int emit_device_impl(acmpc_ctx* ctx, const char* who, int kind, const int64_t* d_value, int64_t n_rows,
                     int64_t order_id_base, int style, char* d_text, int64_t capacity, int64_t* d_n_bytes,
                     hipStream_t st) {
  if (style != ACMPC_EMIT_SQL && style != ACMPC_EMIT_CSV && !(style == ACMPC_EMIT_DISPLAY && kind == kEmitTotals))
    return fail(ctx, ACMPC_E_INVALID, "%s: unknown style %d", who, style);
  if (n_rows < 0 || capacity < 0 || !d_n_bytes) return fail(ctx, ACMPC_E_INVALID, "%s: bad size or null byte counter", who);
  if (n_rows > 0 && !d_value) return fail(ctx, ACMPC_E_INVALID, "%s: null value pointer", who);
  if (capacity > 0 && !d_text) return fail(ctx, ACMPC_E_INVALID, "%s: null text pointer", who);
  if (!aligned16(d_text)) return fail(ctx, ACMPC_E_ALIGN, "%s: device text must be 16-byte aligned", who);
  ENTER_DEVICE(ctx);
  EmitArgs a;
  a.value = reinterpret_cast<const long long*>(d_value);
  a.n_rows = n_rows;
  a.n_chunks = (n_rows + kEmitChunkRows - 1) / kEmitChunkRows;
  a.order_id_base = order_id_base;
  a.codes = ctx->d_codes.p;
  int rc;
  if ((rc = ensure(ctx, ctx->d_chunk_bytes, (size_t)a.n_chunks + 1))) return rc;
  a.chunk_bytes = ctx->d_chunk_bytes.p;
  a.text = reinterpret_cast<unsigned char*>(d_text);
  a.capacity = capacity;
  a.n_bytes = reinterpret_cast<long long*>(d_n_bytes);
  a.status = ctx->d_status;
  if (a.n_chunks == 0) {
    HIPCHK(ctx, hipMemsetAsync(d_n_bytes, 0, sizeof(int64_t), st));
    return ACMPC_OK;
  }
  if (kind == kEmitTotals && style == ACMPC_EMIT_DISPLAY) return launch_emit<kEmitTotals, kEmitDisplay>(ctx, a, st);
  if (kind == kEmitTotals)
    return style == ACMPC_EMIT_SQL ? launch_emit<kEmitTotals, kEmitSql>(ctx, a, st) : launch_emit<kEmitTotals, kEmitCsv>(ctx, a, st);
  return style == ACMPC_EMIT_SQL ? launch_emit<kEmitStock, kEmitSql>(ctx, a, st) : launch_emit<kEmitStock, kEmitCsv>(ctx, a, st);
}

int emit_host_impl(acmpc_ctx* ctx, const char* who, int kind, const int64_t* value, int64_t n_rows, int64_t base,
                   int style, char* text, int64_t capacity, int64_t* n_bytes) {
  if (!n_bytes || n_rows < 0 || capacity < 0 || (n_rows > 0 && !value) || (capacity > 0 && !text))
    return fail(ctx, ACMPC_E_INVALID, "%s: bad argument", who);
  return guarded(ctx, who, [&]() -> int {
  ENTER_DEVICE(ctx);
  int rc;
  if ((rc = ensure(ctx, ctx->s_emit_value, (size_t)n_rows))) return rc;
  if ((rc = ensure(ctx, ctx->s_text, (size_t)capacity + 16))) return rc;
  if (!ctx->d_ing_scalars) HIPCHK(ctx, hipMalloc((void**)&ctx->d_ing_scalars, 2 * sizeof(long long)));
  hipStream_t st = nullptr;
  HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), st));   // this call reports on its own rows only
  if (n_rows > 0 && (rc = host_copies(ctx, {{ctx->s_emit_value.p, value, (size_t)n_rows * 8, hipMemcpyHostToDevice}}, st))) return rc;
  rc = kind == kEmitTotals
           ? acmpc_emit_totals_device(ctx, ctx->s_emit_value.p, n_rows, (int32_t)base, style,
                                      reinterpret_cast<char*>(ctx->s_text.p), capacity,
                                      reinterpret_cast<int64_t*>(ctx->d_ing_scalars), st)
           : acmpc_emit_stock_device(ctx, ctx->s_emit_value.p, (int32_t)n_rows, style,
                                     reinterpret_cast<char*>(ctx->s_text.p), capacity,
                                     reinterpret_cast<int64_t*>(ctx->d_ing_scalars), st);
  if (rc) return rc;
  long long need = 0;
  HIPCHK(ctx, hipMemcpyAsync(&need, ctx->d_ing_scalars, sizeof(need), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  *n_bytes = need;
  const int64_t got = std::min<int64_t>(need, capacity);
  if (got > 0) {
    if ((rc = host_copies(ctx, {{text, ctx->s_text.p, (size_t)got, hipMemcpyDeviceToHost}}, st))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(st));                 // (a short text is only queued by host_copies)
  }
  uint32_t flags = 0;
  return acmpc_status(ctx, st, &flags);
  });
}

}  // namespace

ACMPC_API int acmpc_emit_totals_device(acmpc_ctx* ctx, const int64_t* d_order_total_cents, int64_t n_orders,
                                       int32_t order_id_base, int style, char* d_text, int64_t capacity_bytes,
                                       int64_t* d_n_bytes, void* hip_stream) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_emit_totals_device: null context");
  if (n_orders > 0 && (int64_t)order_id_base + n_orders - 1 > INT32_MAX)
    return fail(ctx, ACMPC_E_INVALID, "acmpc_emit_totals_device: order ids would leave int32 (NumPedido is INT, store.sql:159)");
  return emit_device_impl(ctx, "acmpc_emit_totals_device", kEmitTotals, d_order_total_cents, n_orders, order_id_base,
                          style, d_text, capacity_bytes, d_n_bytes, (hipStream_t)hip_stream);
}

ACMPC_API int acmpc_emit_stock_device(acmpc_ctx* ctx, const int64_t* d_product_qty, int32_t n_products, int style,
                                      char* d_text, int64_t capacity_bytes, int64_t* d_n_bytes, void* hip_stream) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_emit_stock_device: null context");
  if (ctx->dict_products <= 0) return fail(ctx, ACMPC_E_STATE, "acmpc_set_product_codes has not been called");
  if (n_products != ctx->dict_products)
    return fail(ctx, ACMPC_E_INVALID, "acmpc_emit_stock_device: %d deltas for a code list of %d", n_products, ctx->dict_products);
  const int32_t bad = style == ACMPC_EMIT_SQL ? ctx->code_sql_bad : style == ACMPC_EMIT_CSV ? ctx->code_csv_bad : -1;
  if (bad >= 0) return fail(ctx, ACMPC_E_DATA, "product code %d cannot be written unescaped in this style", bad);
  return emit_device_impl(ctx, "acmpc_emit_stock_device", kEmitStock, d_product_qty, n_products, 0, style, d_text,
                          capacity_bytes, d_n_bytes, (hipStream_t)hip_stream);
}

ACMPC_API int acmpc_emit_totals(acmpc_ctx* ctx, const int64_t* order_total_cents, int64_t n_orders,
                                int32_t order_id_base, int style, char* text, int64_t capacity_bytes, int64_t* n_bytes) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_emit_totals: null context");
  return emit_host_impl(ctx, "acmpc_emit_totals", kEmitTotals, order_total_cents, n_orders, order_id_base, style, text,
                        capacity_bytes, n_bytes);
}

ACMPC_API int acmpc_emit_stock(acmpc_ctx* ctx, const int64_t* product_qty, int32_t n_products, int style, char* text,
                               int64_t capacity_bytes, int64_t* n_bytes) {
  if (!ctx) return fail(nullptr, ACMPC_E_INVALID, "acmpc_emit_stock: null context");
  if (n_products < 0) return fail(ctx, ACMPC_E_INVALID, "acmpc_emit_stock: negative size");
  return emit_host_impl(ctx, "acmpc_emit_stock", kEmitStock, product_qty, n_products, 0, style, text, capacity_bytes,
                        n_bytes);
}

// ---- multi-GPU: shards by order-id range, one all-reduce ---------------------------------------

namespace {

#define NCCLCHK(ctx, api, call)                                                                \
  do {                                                                                         \
    ncclResult_t r_ = (call);                                                                  \
    if (r_ != ncclSuccess)                                                                     \
      return fail(ctx, ACMPC_E_HIP, "%s failed: %s", #call, (api)->GetErrorString(r_));        \
  } while (0)

const RcclApi* need_rccl(acmpc_ctx* ctx, int* rc) {
  const char* why = "";
  const RcclApi* api = rccl_api(&why);
  if (!api) *rc = fail(ctx, ACMPC_E_STATE, "RCCL is not available: %s", why);
  return api;
}

}  // namespace

ACMPC_API int acmpc_comm_id(char* id128) {
  if (!id128) return fail(nullptr, ACMPC_E_INVALID, "acmpc_comm_id: null buffer");
  int rc = 0;
  const RcclApi* api = need_rccl(nullptr, &rc);
  if (!api) return rc;
  static_assert(sizeof(ncclUniqueId) == ACMPC_COMM_ID_BYTES, "the id travels as ACMPC_COMM_ID_BYTES opaque bytes");
  ncclUniqueId id;
  NCCLCHK(nullptr, api, api->GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return ACMPC_OK;
}

ACMPC_API int acmpc_comm_init_rank(acmpc_comm** out, acmpc_ctx* ctx, const char* id128, int rank, int world) {
  if (!out) return fail(ctx, ACMPC_E_INVALID, "acmpc_comm_init_rank: null out pointer");
  *out = nullptr;
  if (!ctx || !id128 || world <= 0 || rank < 0 || rank >= world)
    return fail(ctx, ACMPC_E_INVALID, "acmpc_comm_init_rank: bad argument (rank %d of %d)", rank, world);
  int rc = 0;
  const RcclApi* api = need_rccl(ctx, &rc);
  if (!api) return rc;
  ENTER_DEVICE(ctx);                                   // a communicator rank belongs to the current device
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  acmpc_comm* c = new (std::nothrow) acmpc_comm();
  if (!c) return fail(ctx, ACMPC_E_NOMEM, "acmpc_comm_init_rank: out of host memory");
  const ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(ctx, ACMPC_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
  }
  c->device = ctx->device; c->rank = rank; c->world = world;
  *out = c;
  return ACMPC_OK;
}

ACMPC_API int acmpc_comm_init_all(acmpc_comm** out, acmpc_ctx* const* ctxs, int n_devices) {
  if (!out || !ctxs || n_devices <= 0 || n_devices > kMaxDevices)
    return fail(nullptr, ACMPC_E_INVALID, "acmpc_comm_init_all: bad argument");
  for (int g = 0; g < n_devices; ++g) out[g] = nullptr;
  for (int g = 0; g < n_devices; ++g) {
    if (!ctxs[g]) return fail(nullptr, ACMPC_E_INVALID, "acmpc_comm_init_all: null context %d", g);
    for (int h = 0; h < g; ++h)
      if (ctxs[h]->device == ctxs[g]->device)
        return fail(ctxs[0], ACMPC_E_INVALID, "acmpc_comm_init_all: contexts %d and %d share device %d", h, g, ctxs[g]->device);
  }
  acmpc_ctx* ctx0 = ctxs[0];
  int rc = 0;
  const RcclApi* api = need_rccl(ctx0, &rc);
  if (!api) return rc;
  return guarded(ctx0, "acmpc_comm_init_all", [&]() -> int {
    std::vector<int> devs(n_devices);
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    for (int g = 0; g < n_devices; ++g) devs[g] = ctxs[g]->device;
    DeviceScope keep;                                  // ncclCommInitAll moves the current device about
    HIPCHK(ctx0, keep.enter(ctx0->device));
    NCCLCHK(ctx0, api, api->CommInitAll(comms.data(), n_devices, devs.data()));
    (void)hipSetDevice(ctx0->device);
    for (int g = 0; g < n_devices; ++g) {
      acmpc_comm* c = new acmpc_comm();
      c->comm = comms[g]; c->device = devs[g]; c->rank = g; c->world = n_devices;
      out[g] = c;
    }
    return ACMPC_OK;
  });
}

ACMPC_API void acmpc_comm_destroy(acmpc_comm* comm) {
  if (!comm) return;
  const RcclApi* api = rccl_api(nullptr);
  if (api && comm->comm) {
    DeviceScope scope;
    (void)scope.enter(comm->device);
    (void)api->CommDestroy(comm->comm);
  }
  delete comm;
}

ACMPC_API int acmpc_merge_device(acmpc_ctx* ctx, acmpc_comm* comm, int64_t* d_values, int64_t count, void* hip_stream) {
  if (!ctx || !comm || count < 0 || (count > 0 && !d_values))
    return fail(ctx, ACMPC_E_INVALID, "acmpc_merge_device: bad argument");
  if (comm->device != ctx->device)
    return fail(ctx, ACMPC_E_INVALID, "acmpc_merge_device: the communicator rank lives on device %d, the context on %d",
                comm->device, ctx->device);
  if (count == 0) return ACMPC_OK;
  int rc = 0;
  const RcclApi* api = need_rccl(ctx, &rc);
  if (!api) return rc;
  ENTER_DEVICE(ctx);
  NCCLCHK(ctx, api, api->AllReduce(d_values, d_values, (size_t)count, ncclInt64, ncclSum, comm->comm, (hipStream_t)hip_stream));
  return ACMPC_OK;
}

ACMPC_API int acmpc_merge_group_device(acmpc_ctx* const* ctxs, acmpc_comm* const* comms, int n_devices,
                                       int64_t* const* d_values, int64_t count, void* const* hip_streams) {
  if (!ctxs || !comms || !d_values || n_devices <= 0 || count < 0)
    return fail(nullptr, ACMPC_E_INVALID, "acmpc_merge_group_device: bad argument");
  for (int g = 0; g < n_devices; ++g)
    if (!ctxs[g] || !comms[g] || (count > 0 && !d_values[g]) || comms[g]->device != ctxs[g]->device)
      return fail(ctxs[0], ACMPC_E_INVALID, "acmpc_merge_group_device: bad context, communicator or buffer %d", g);
  if (count == 0) return ACMPC_OK;
  acmpc_ctx* ctx0 = ctxs[0];
  int rc = 0;
  const RcclApi* api = need_rccl(ctx0, &rc);
  if (!api) return rc;
  // one thread drives every rank: the calls must sit in one group, or the first would wait for the others
  NCCLCHK(ctx0, api, api->GroupStart());
  ncclResult_t first_bad = ncclSuccess;
  for (int g = 0; g < n_devices; ++g) {
    const ncclResult_t r = api->AllReduce(d_values[g], d_values[g], (size_t)count, ncclInt64, ncclSum, comms[g]->comm,
                                          hip_streams ? (hipStream_t)hip_streams[g] : nullptr);
    if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
  }
  const ncclResult_t rend = api->GroupEnd();
  if (first_bad != ncclSuccess) return fail(ctx0, ACMPC_E_HIP, "ncclAllReduce failed: %s", api->GetErrorString(first_bad));
  if (rend != ncclSuccess) return fail(ctx0, ACMPC_E_HIP, "ncclGroupEnd failed: %s", api->GetErrorString(rend));
  return ACMPC_OK;
}

namespace {

// Order-index cuts of the shards: shard g owns [o_cut[g], o_cut[g + 1]).
void shard_order_cuts(int64_t n_orders, int n_devices, int64_t* o_cut) {
  for (int g = 0; g <= n_devices; ++g) o_cut[g] = (int64_t)(((__int128)n_orders * g) / n_devices);
}

// The shard that owns the line with this order id.  An id below the base goes to the first shard that has
// orders and an id at or past base + M to the last one: the device pass of that shard reports it.
inline int shard_of_id(int32_t id, int32_t base, int64_t n_orders, int n_devices, const int64_t* o_cut, int first_owner) {
  const int64_t idx = (int64_t)id - (int64_t)base;
  if (idx < 0) return first_owner;
  if (idx >= n_orders) return n_devices - 1;
  return (int)(std::upper_bound(o_cut, o_cut + n_devices + 1, idx) - o_cut) - 1;
}

int shard_plan_check(const int32_t* order_id, int64_t n_lines, int64_t n_orders, int n_devices, const void* o_cut,
                     const void* l_cut) {
  if (n_devices <= 0 || n_devices > kMaxDevices) return fail(nullptr, ACMPC_E_INVALID, "shard plan: %d devices (1 to %d)", n_devices, kMaxDevices);
  if (n_lines < 0 || n_orders < 0 || n_orders > 0x7fffffffll) return fail(nullptr, ACMPC_E_INVALID, "shard plan: bad size");
  if (n_lines > 0 && !order_id) return fail(nullptr, ACMPC_E_INVALID, "shard plan: null order_id");
  if (n_lines > 0 && n_orders == 0) return fail(nullptr, ACMPC_E_INVALID, "n_lines > 0 with n_orders == 0");
  if (!o_cut || !l_cut) return fail(nullptr, ACMPC_E_INVALID, "shard plan: null output");
  return ACMPC_OK;
}

}  // namespace

ACMPC_API int acmpc_shard_plan(const int32_t* order_id, int64_t n_lines, int64_t n_orders, int32_t order_id_base,
                               int n_devices, int line_order, int64_t* o_cut, int64_t* l_cut) {
  int rc = shard_plan_check(order_id, n_lines, n_orders, n_devices, o_cut, l_cut);
  if (rc) return rc;
  if (line_order != ACMPC_ORDER_GROUPED && line_order != ACMPC_ORDER_ANY && line_order != ACMPC_ORDER_AUTO)
    return fail(nullptr, ACMPC_E_INVALID, "shard plan: unknown line order %d", line_order);
  shard_order_cuts(n_orders, n_devices, o_cut);
  int first_owner = 0;                                   // the first shard that owns any order
  while (first_owner < n_devices - 1 && o_cut[first_owner + 1] == o_cut[first_owner]) ++first_owner;
  const bool sorted = line_order == ACMPC_ORDER_GROUPED || std::is_sorted(order_id, order_id + n_lines);
  if (sorted) {
    // Grouped rows: the cut in front of shard g is the first line whose id reaches the shard's first order.
    // The cuts are forced to ascend, so that every line lands in exactly one shard whatever the column holds:
    // on a column that is not sorted after all, a descent either lies inside a shard or leaves one of its two
    // lines outside its shard's id range, and the device pass reports either.
    l_cut[0] = 0;
    for (int g = 1; g < n_devices; ++g) {
      const int64_t first_id = (int64_t)order_id_base + o_cut[g];
      const int64_t at = g <= first_owner ? 0     // ids below the base stay with the first shard that has orders
                                          : std::lower_bound(order_id, order_id + n_lines, first_id,
                                                             [](int32_t a, int64_t b) { return (int64_t)a < b; }) - order_id;
      l_cut[g] = std::max(l_cut[g - 1], at);
    }
    l_cut[n_devices] = n_lines;
    return ACMPC_OK;
  }
  // Rows in any order: shard g gets the lines of its orders, wherever they stand; l_cut are their places after
  // a stable partition of the rows by shard (acmpc_shard_partition).
  int64_t count[kMaxDevices] = {};
  for (int64_t l = 0; l < n_lines; ++l) ++count[shard_of_id(order_id[l], order_id_base, n_orders, n_devices, o_cut, first_owner)];
  l_cut[0] = 0;
  for (int g = 0; g < n_devices; ++g) l_cut[g + 1] = l_cut[g] + count[g];
  return ACMPC_OK;
}

ACMPC_API int acmpc_shard_partition(const int32_t* order_id, const int32_t* product_idx, const int32_t* qty,
                                    int64_t n_lines, int64_t n_orders, int32_t order_id_base, int n_devices,
                                    int32_t* out_order_id, int32_t* out_product_idx, int32_t* out_qty, int64_t* l_cut) {
  int64_t o_cut[kMaxDevices + 1];
  int rc = shard_plan_check(order_id, n_lines, n_orders, n_devices, o_cut, l_cut);
  if (rc) return rc;
  if (n_lines > 0 && (!product_idx || !qty || !out_order_id || !out_product_idx || !out_qty))
    return fail(nullptr, ACMPC_E_INVALID, "acmpc_shard_partition: null column");
  if ((rc = acmpc_shard_plan(order_id, n_lines, n_orders, order_id_base, n_devices, ACMPC_ORDER_ANY, o_cut, l_cut))) return rc;
  if (std::is_sorted(order_id, order_id + n_lines)) {    // (the plan then made contiguous cuts: the rows stay where they are)
    std::copy(order_id, order_id + n_lines, out_order_id);
    std::copy(product_idx, product_idx + n_lines, out_product_idx);
    std::copy(qty, qty + n_lines, out_qty);
    return ACMPC_OK;
  }
  int first_owner = 0;
  while (first_owner < n_devices - 1 && o_cut[first_owner + 1] == o_cut[first_owner]) ++first_owner;
  int64_t at[kMaxDevices];
  for (int g = 0; g < n_devices; ++g) at[g] = l_cut[g];
  for (int64_t l = 0; l < n_lines; ++l) {
    const int64_t k = at[shard_of_id(order_id[l], order_id_base, n_orders, n_devices, o_cut, first_owner)]++;
    out_order_id[k] = order_id[l]; out_product_idx[k] = product_idx[l]; out_qty[k] = qty[l];
  }
  return ACMPC_OK;
}

ACMPC_API int acmpc_aggregate_sharded(acmpc_ctx* const* ctxs, acmpc_comm* const* comms, int n_devices,
                                      const int32_t* order_id, const int32_t* product_idx, const int32_t* qty,
                                      int64_t n_lines, const int32_t* order_customer, int64_t n_orders,
                                      int32_t order_id_base, int32_t n_customers, int64_t* out_order_total_cents,
                                      int64_t* out_product_qty, int64_t* out_customer_total_cents) {
  if (!ctxs || !comms || n_devices <= 0 || n_devices > kMaxDevices)
    return fail(nullptr, ACMPC_E_INVALID, "acmpc_aggregate_sharded: bad device list");
  for (int g = 0; g < n_devices; ++g)
    if (!ctxs[g] || !comms[g] || comms[g]->device != ctxs[g]->device || comms[g]->world != n_devices)
      return fail(ctxs[0], ACMPC_E_INVALID, "acmpc_aggregate_sharded: bad context or communicator %d", g);
  acmpc_ctx* ctx0 = ctxs[0];
  const int32_t P = ctx0->n_products;
  if (P <= 0) return fail(ctx0, ACMPC_E_STATE, "acmpc_set_products has not been called");
  for (int g = 1; g < n_devices; ++g) {
    if (ctxs[g]->n_products != P)
      return fail(ctx0, ACMPC_E_STATE, "acmpc_aggregate_sharded: context %d holds %d products, context 0 holds %d", g,
                  ctxs[g]->n_products, P);
    if (ctxs[g]->line_order != ctx0->line_order)
      return fail(ctx0, ACMPC_E_STATE, "acmpc_aggregate_sharded: context %d has another ACMPC_OPT_LINE_ORDER than context 0", g);
  }
  if (n_lines < 0 || n_orders < 0 || n_customers <= 0) return fail(ctx0, ACMPC_E_INVALID, "negative size or n_customers <= 0");
  if (n_orders > 0x7fffffffll || n_lines > (1ll << 40) || n_customers > kMaxCustomers)
    return fail(ctx0, ACMPC_E_INVALID, "size beyond the limits (n_orders int32, n_lines 2^40, n_customers %d)", kMaxCustomers);
  if ((n_lines > 0 && (!order_id || !product_idx || !qty)) || (n_orders > 0 && (!order_customer || !out_order_total_cents)) ||
      !out_product_qty || !out_customer_total_cents)
    return fail(ctx0, ACMPC_E_INVALID, "null host pointer");
  if (n_lines > 0 && n_orders == 0) return fail(ctx0, ACMPC_E_INVALID, "n_lines > 0 with n_orders == 0");
  return guarded(ctx0, "acmpc_aggregate_sharded", [&]() -> int {
  const int64_t n_merged = (int64_t)P + n_customers;
  DeviceScope keep;                                    // the loop below moves the current device about
  HIPCHK(ctx0, keep.enter(ctx0->device));
  // The shards (acmpc_shard_plan).  Grouped rows are cut where they lie; rows in any order that are not sorted
  // are first brought together shard by shard in host memory (a counting pass and a stable scatter), since a
  // shard's lines then lie anywhere in the columns.
  int64_t o_cut[kMaxDevices + 1], l_cut[kMaxDevices + 1];
  int rc = acmpc_shard_plan(order_id, n_lines, n_orders, order_id_base, n_devices, ctx0->line_order, o_cut, l_cut);
  if (rc) return fail(ctx0, rc, "%s", g_init_error);
  std::vector<int32_t> moved;                            // the three partitioned columns, one after the other
  const int32_t *src_id = order_id, *src_pr = product_idx, *src_q = qty;
  if (ctx0->line_order != ACMPC_ORDER_GROUPED && n_devices > 1 && !std::is_sorted(order_id, order_id + n_lines)) {
    moved.resize((size_t)n_lines * 3);
    int32_t* m = moved.data();
    if ((rc = acmpc_shard_partition(order_id, product_idx, qty, n_lines, n_orders, order_id_base, n_devices, m, m + n_lines,
                                    m + 2 * n_lines, l_cut)))
      return fail(ctx0, rc, "%s", g_init_error);
    src_id = m; src_pr = m + n_lines; src_q = m + 2 * n_lines;
  }
  // One host thread per device for the two phases that move host memory: a copy from or to pageable memory
  // holds its calling thread until the data has gone through the staging buffers, so issued from one thread the
  // shards would cross their PCIe links one after the other.  (A thread sets its own current device.)  A thread
  // the system will not start has its shard served from the calling thread instead.
  std::vector<int> rcs((size_t)n_devices, ACMPC_OK);
  auto on_every_device = [&](auto&& body) {
    std::vector<std::thread> threads;
    threads.reserve((size_t)n_devices);
    for (int g = 0; g < n_devices; ++g) {
      try { threads.emplace_back([&, g] { rcs[(size_t)g] = body(g); }); }
      catch (const std::system_error&) { rcs[(size_t)g] = body(g); }
    }
    for (std::thread& t : threads) t.join();
    (void)hipSetDevice(ctx0->device);                  // (a shard served from this thread moved the current device)
    for (int g = 0; g < n_devices; ++g)
      if (rcs[(size_t)g] != ACMPC_OK) {
        if (ctxs[g] != ctx0) fail(ctx0, rcs[(size_t)g], "shard %d: %s", g, ctxs[g]->err);
        return rcs[(size_t)g];
      }
    return (int)ACMPC_OK;
  };
  rc = on_every_device([&](int g) -> int {
    acmpc_ctx* c = ctxs[g];
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t nl = l_cut[g + 1] - l_cut[g], no = o_cut[g + 1] - o_cut[g];
    int r;
    if ((r = ensure(c, c->s_order_id, (size_t)nl))) return r;
    if ((r = ensure(c, c->s_product_idx, (size_t)nl))) return r;
    if ((r = ensure(c, c->s_qty, (size_t)nl))) return r;
    if ((r = ensure(c, c->s_order_customer, (size_t)no))) return r;
    if ((r = ensure(c, c->s_order_total, (size_t)no))) return r;
    if ((r = ensure(c, c->s_merged, (size_t)n_merged))) return r;
    hipStream_t st = nullptr;
    HIPCHK(c, hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), st));
    if (nl > 0) {
      HIPCHK(c, hipMemcpyAsync(c->s_order_id.p, src_id + l_cut[g], (size_t)nl * 4, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(c->s_product_idx.p, src_pr + l_cut[g], (size_t)nl * 4, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(c->s_qty.p, src_q + l_cut[g], (size_t)nl * 4, hipMemcpyHostToDevice, st));
    }
    if (no > 0) HIPCHK(c, hipMemcpyAsync(c->s_order_customer.p, order_customer + o_cut[g], (size_t)no * 4, hipMemcpyHostToDevice, st));
    return aggregate_device_impl(c, c->s_order_id.p, c->s_product_idx.p, c->s_qty.p, nl, c->s_order_customer.p, no,
                                 (int32_t)((int64_t)order_id_base + o_cut[g]), n_customers, c->s_order_total.p, c->s_merged.p,
                                 c->s_merged.p + P, st);
  });
  if (rc) return rc;
  // the one exchange: [product_qty | customer_total_cents], summed in place on every device
  {
    int64_t* bufs[kMaxDevices];
    for (int g = 0; g < n_devices; ++g) bufs[g] = ctxs[g]->s_merged.p;
    if ((rc = acmpc_merge_group_device(ctxs, comms, n_devices, bufs, n_merged, nullptr))) return rc;
  }
  // results back, and every device's verdict on its data (acmpc_status waits for the device)
  std::vector<int> verdict((size_t)n_devices, ACMPC_OK);
  rc = on_every_device([&](int g) -> int {
    acmpc_ctx* c = ctxs[g];
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t no = o_cut[g + 1] - o_cut[g];
    if (no > 0)
      HIPCHK(c, hipMemcpyAsync(out_order_total_cents + o_cut[g], c->s_order_total.p, (size_t)no * 8, hipMemcpyDeviceToHost, nullptr));
    if (g == 0) {
      HIPCHK(c, hipMemcpyAsync(out_product_qty, c->s_merged.p, (size_t)P * 8, hipMemcpyDeviceToHost, nullptr));
      HIPCHK(c, hipMemcpyAsync(out_customer_total_cents, c->s_merged.p + P, (size_t)n_customers * 8, hipMemcpyDeviceToHost, nullptr));
    }
    uint32_t flags = 0;
    verdict[(size_t)g] = acmpc_status(c, nullptr, &flags);
    return ACMPC_OK;
  });
  if (rc) return rc;
  int worst = ACMPC_OK;
  for (int g = 0; g < n_devices; ++g)
    if (verdict[(size_t)g] != ACMPC_OK && worst == ACMPC_OK) {
      worst = verdict[(size_t)g];
      if (ctxs[g] != ctx0) fail(ctx0, worst, "shard %d: %s", g, ctxs[g]->err);
    }
  return worst;
  });
}
