// acmpc_check — developer tool: generate a synthetic data set on the device, run the engine
// through the C ABI, compare every output (and the generated columns) with the CPU oracle, and
// print per-kernel times.  Not part of the product; tests and bench.py are the judged harness.
//
//   acmpc_check N [P C zipf qty_mode reps check]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../include/acmpc.h"

extern "C" {
int oracle_aggregate(const int32_t*, const int32_t*, const int32_t*, int64_t, const int32_t*, int64_t, int32_t,
                     int32_t, const int64_t*, int32_t, int64_t*, int64_t*, int64_t*);
void oracle_synth_prices(int32_t, int64_t*);
int64_t oracle_synth_n_orders(int64_t);
void oracle_synth_fill(int64_t, int32_t, int32_t, int32_t, int32_t, int32_t, int64_t, int64_t, int32_t*, int32_t*,
                       int32_t*, int32_t*);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 2; } } while (0)
#define AK(x) do { int r = (x); if (r != 0) { printf("acmpc error %d: %s (%s:%d)\n", r, acmpc_last_error(ctx), __FILE__, __LINE__); return 3; } } while (0)

template <typename T>
static long long diff_count(const std::vector<T>& a, const std::vector<T>& b, const char* name) {
  long long bad = 0;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i] != b[i]) { if (bad < 5) printf("  %s[%zu]: got %lld want %lld\n", name, i, (long long)a[i], (long long)b[i]); ++bad; }
  printf("%s: %zu entries, %lld mismatches\n", name, a.size(), bad);
  return bad;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1000000;
  const int32_t P = argc > 2 ? atoi(argv[2]) : 10000;
  const int32_t C = argc > 3 ? atoi(argv[3]) : 100000;
  const int zipf = argc > 4 ? atoi(argv[4]) : 0;
  const int qty_mode = argc > 5 ? atoi(argv[5]) : 0;
  const int reps = argc > 6 ? atoi(argv[6]) : 20;
  const int check = argc > 7 ? atoi(argv[7]) : 1;

  acmpc_ctx* ctx = nullptr;
  if (acmpc_init(&ctx, 0) != 0) { printf("init failed: %s\n", acmpc_last_error(nullptr)); return 1; }
  acmpc_synth_cfg cfg = {N, P, C, 5, zipf, qty_mode, 0, 1};
  acmpc_synth_shard sh;
  AK(acmpc_synth_plan(&cfg, &sh));
  const int64_t M = sh.n_orders_total;
  printf("N=%lld M=%lld P=%d C=%d zipf=%d qty_mode=%d\n", (long long)N, (long long)M, P, C, zipf, qty_mode);
  std::vector<int64_t> price(P);
  AK(acmpc_synth_prices(&cfg, price.data()));
  AK(acmpc_set_products(ctx, price.data(), P));

  int32_t *d_oid, *d_pidx, *d_qty, *d_cust;
  int64_t *d_tot, *d_pq, *d_ct;
  CK(hipMalloc(&d_oid, std::max<int64_t>(N, 1) * 4)); CK(hipMalloc(&d_pidx, std::max<int64_t>(N, 1) * 4));
  CK(hipMalloc(&d_qty, std::max<int64_t>(N, 1) * 4)); CK(hipMalloc(&d_cust, std::max<int64_t>(M, 1) * 4));
  CK(hipMalloc(&d_tot, std::max<int64_t>(M, 1) * 8)); CK(hipMalloc(&d_pq, (size_t)P * 8)); CK(hipMalloc(&d_ct, (size_t)C * 8));
  CK(hipMemset(d_tot, 0xAB, std::max<int64_t>(M, 1) * 8)); CK(hipMemset(d_pq, 0xAB, (size_t)P * 8)); CK(hipMemset(d_ct, 0xAB, (size_t)C * 8));
  AK(acmpc_synth_generate_device(ctx, &cfg, d_oid, d_pidx, d_qty, d_cust, nullptr));
  CK(hipDeviceSynchronize());

  // warm-up + correctness run
  AK(acmpc_aggregate_device(ctx, d_oid, d_pidx, d_qty, N, d_cust, M, 5, C, d_tot, d_pq, d_ct, nullptr));
  uint32_t flags = 0;
  AK(acmpc_status(ctx, nullptr, &flags));

  long long bad = 0;
  if (check) {
    std::vector<int32_t> oid(N), pidx(N), qty(N), cust(M), roid(N), rpidx(N), rqty(N), rcust(M);
    CK(hipMemcpy(oid.data(), d_oid, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(pidx.data(), d_pidx, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(qty.data(), d_qty, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cust.data(), d_cust, M * 4, hipMemcpyDeviceToHost));
    if (oracle_synth_n_orders(N) != M) { printf("order count mismatch\n"); ++bad; }
    oracle_synth_fill(N, P, C, 5, zipf, qty_mode, 0, M, roid.data(), rpidx.data(), rqty.data(), rcust.data());
    bad += diff_count(oid, roid, "synth order_id"); bad += diff_count(pidx, rpidx, "synth product_idx");
    bad += diff_count(qty, rqty, "synth qty"); bad += diff_count(cust, rcust, "synth order_customer");
    std::vector<int64_t> rprice(P);
    oracle_synth_prices(P, rprice.data());
    bad += diff_count(price, rprice, "synth price");
    std::vector<int64_t> tot(M), pq(P), ct(C), rtot(M), rpq(P), rct(C);
    auto t0 = std::chrono::steady_clock::now();
    int of = oracle_aggregate(oid.data(), pidx.data(), qty.data(), N, cust.data(), M, 5, C, price.data(), P, rtot.data(), rpq.data(), rct.data());
    auto t1 = std::chrono::steady_clock::now();
    const double osec = std::chrono::duration<double>(t1 - t0).count();
    printf("oracle: flags=%d  %.3f s  %.1f M lines/s (1 thread)\n", of, osec, N / osec / 1e6);
    CK(hipMemcpy(tot.data(), d_tot, M * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(pq.data(), d_pq, (size_t)P * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ct.data(), d_ct, (size_t)C * 8, hipMemcpyDeviceToHost));
    bad += diff_count(tot, rtot, "order_total_cents"); bad += diff_count(pq, rpq, "product_qty");
    bad += diff_count(ct, rct, "customer_total_cents");
  }

  // timing: whole call (events around reps calls) and per kernel (library events)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) AK(acmpc_aggregate_device(ctx, d_oid, d_pidx, d_qty, N, d_cust, M, 5, C, d_tot, d_pq, d_ct, nullptr));
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; ++i) AK(acmpc_aggregate_device(ctx, d_oid, d_pidx, d_qty, N, d_cust, M, 5, C, d_tot, d_pq, d_ct, nullptr));
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double per = ms / reps;
  const double balg = 12.0 * N + 12.0 * M + 16.0 * P + 8.0 * C;
  printf("whole call: %.3f ms  %.2f G lines/s  %.2f TB/s algorithmic (%.1f%% of 8 TB/s)\n", per, N / per / 1e6,
         balg / per / 1e9, balg / per / 1e9 / 8.0 * 100);
  AK(acmpc_set_profiling(ctx, 1));
  double ks[ACMPC_N_KERNELS] = {0, 0, 0, 0};
  for (int i = 0; i < reps; ++i) {
    AK(acmpc_aggregate_device(ctx, d_oid, d_pidx, d_qty, N, d_cust, M, 5, C, d_tot, d_pq, d_ct, nullptr));
    float k[ACMPC_N_KERNELS];
    AK(acmpc_last_kernel_ms(ctx, k));
    for (int j = 0; j < ACMPC_N_KERNELS; ++j) ks[j] += k[j];
  }
  for (int j = 0; j < ACMPC_N_KERNELS; ++j) ks[j] /= reps;
  const double b_lines = 12.0 * N + 8.0 * M + 16.0 * P;
  printf("kernels (ms): lines %.4f  fixup %.4f  customers %.4f  custmerge %.4f\n", ks[0], ks[1], ks[2], ks[3]);
  printf("k_lines: %.2f TB/s algorithmic = %.1f%% of 8 TB/s\n", b_lines / ks[0] / 1e9, b_lines / ks[0] / 1e9 / 8 * 100);
  AK(acmpc_status(ctx, nullptr, &flags));
  acmpc_destroy(ctx);
  printf(bad ? "FAIL (%lld mismatches)\n" : "PASS\n", bad);
  return bad ? 1 : 0;
}
