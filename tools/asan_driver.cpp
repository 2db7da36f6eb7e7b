// Developer tool: walks the host-pointer entry points of libacmpc (built with AddressSanitizer on the
// host side only) through ordinary calls and argument errors; the sanitizer reports what the host code
// does wrong with memory.  Results are checked for plausibility only (parity is the tests' business).
//   hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -Xarch_host -fsanitize=address \
//         -o tools/asan/libacmpc.so ac_mpc_amd/csrc/acmpc.hip
//   hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -Xarch_host -fsanitize=address -o tools/asan/asan_driver \
//         tools/asan_driver.cpp -Ltools/asan -lacmpc -Wl,-rpath,'$ORIGIN'
#include "../include/acmpc.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define REQ(x) do { if (!(x)) { printf("FAILED: %s (line %d): %s\n", #x, __LINE__, acmpc_last_error(ctx)); return 1; } } while (0)

int main() {
  acmpc_ctx* ctx = nullptr;
  if (acmpc_init(&ctx, 0) != ACMPC_OK) { printf("init: %s\n", acmpc_last_error(nullptr)); return 1; }
  for (int round = 0; round < 3; ++round) {
    const int P = round == 0 ? 7 : round == 1 ? 20000 : 50000, C = 1000 + round, M = 5000 * (round + 1);
    std::vector<int64_t> price(P);
    for (int p = 0; p < P; ++p) price[p] = 100 + 37 * p;
    REQ(acmpc_set_products(ctx, price.data(), P) == ACMPC_OK);
    std::vector<int32_t> oid, pidx, qty, cust(M);
    for (int o = 0; o < M; ++o) {
      cust[o] = (o * 7) % C;
      for (int k = 0; k < (o % 5); ++k) { oid.push_back(5 + o); pidx.push_back((o + k) % (P + 1) - (k == 3)); qty.push_back(1 + k); }
    }
    const int64_t N = (int64_t)oid.size();
    std::vector<int64_t> tot(M), pq(P), ct(C);
    REQ(acmpc_aggregate(ctx, oid.data(), pidx.data(), qty.data(), N, cust.data(), M, 5, C, tot.data(), pq.data(), ct.data()) == ACMPC_OK);
    long long s1 = 0, s2 = 0;
    for (auto v : tot) s1 += v;
    for (auto v : ct) s2 += v;
    REQ(s1 == s2);
    // lifecycle columns
    std::vector<int64_t> lp(N, 250);
    std::vector<uint8_t> state(M, 1);
    std::vector<int32_t> desc(M, 10);
    acmpc_lifecycle life = {lp.data(), state.data(), desc.data(), ACMPC_DESCUENTO_PERCENT};
    REQ(acmpc_aggregate_lifecycle(ctx, oid.data(), pidx.data(), qty.data(), N, cust.data(), M, 5, C, &life, tot.data(), pq.data(), ct.data()) == ACMPC_OK);
    for (int mode = ACMPC_ORDER_ANY; mode <= ACMPC_ORDER_AUTO; ++mode) {
      REQ(acmpc_set_option(ctx, ACMPC_OPT_LINE_ORDER, mode) == ACMPC_OK);
      REQ(acmpc_aggregate(ctx, oid.data(), pidx.data(), qty.data(), N, cust.data(), M, 5, C, tot.data(), pq.data(), ct.data()) == ACMPC_OK);
    }
    REQ(acmpc_set_option(ctx, ACMPC_OPT_LINE_ORDER, ACMPC_ORDER_GROUPED) == ACMPC_OK);
    // contract violation: reported, not fatal
    if (N > 2) {
      std::swap(oid[0], oid[N - 1]);
      REQ(acmpc_aggregate(ctx, oid.data(), pidx.data(), qty.data(), N, cust.data(), M, 5, C, tot.data(), pq.data(), ct.data()) == ACMPC_E_DATA);
      std::swap(oid[0], oid[N - 1]);
    }
    // text: codes, ingest, emit
    std::vector<char> codes((size_t)P * 32, 0);
    for (int p = 0; p < P; ++p) snprintf(&codes[(size_t)p * 32], 31, "K%d", p);
    REQ(acmpc_set_product_codes(ctx, codes.data(), P) == ACMPC_OK);
    std::string text = "INSERT INTO `detalle` (`NumPedido`, `CodigoProd`, `CantidadProductos`) VALUES\n";
    for (int64_t i = 0; i < N; ++i)
      text += "(" + std::to_string(oid[i]) + ", 'K" + std::to_string(pidx[i] < 0 ? 0 : pidx[i] % P) + "', " + std::to_string(qty[i]) + "),\n";
    std::vector<int32_t> a(N + 1), b(N + 1), c(N + 1);
    int64_t rows = -1, bad = -1;
    REQ(acmpc_ingest_detalle(ctx, text.data(), (int64_t)text.size(), a.data(), b.data(), c.data(), N + 1, &rows, &bad) == ACMPC_OK);
    REQ(rows == N && bad == 0);
    REQ(acmpc_ingest_detalle(ctx, text.data(), (int64_t)text.size(), a.data(), b.data(), c.data(), N / 2, &rows, &bad) == ACMPC_E_DATA);
    std::vector<char> out((size_t)M * 83 + 1);
    int64_t nb = -1;
    REQ(acmpc_emit_totals(ctx, tot.data(), M, 5, ACMPC_EMIT_SQL, out.data(), (int64_t)out.size(), &nb) == ACMPC_OK);
    REQ(nb > 0 && nb <= (int64_t)out.size());
    REQ(acmpc_emit_totals(ctx, tot.data(), M, 5, ACMPC_EMIT_CSV, out.data(), 10, &nb) == ACMPC_E_DATA);
    std::vector<char> out2((size_t)P * 103 + 1);
    REQ(acmpc_emit_stock(ctx, pq.data(), P, ACMPC_EMIT_SQL, out2.data(), (int64_t)out2.size(), &nb) == ACMPC_OK);
    // page-locked buffers
    void* h = nullptr;
    REQ(acmpc_host_alloc(ctx, N * 4 + 64, &h) == ACMPC_OK && h);
    memcpy(h, oid.data(), (size_t)N * 4);
    REQ(acmpc_aggregate(ctx, (const int32_t*)h, pidx.data(), qty.data(), N, cust.data(), M, 5, C, tot.data(), pq.data(), ct.data()) == ACMPC_OK);
    REQ(acmpc_host_free(ctx, h) == ACMPC_OK);
    REQ(acmpc_host_free(ctx, h) == ACMPC_E_INVALID);
    REQ(acmpc_host_alloc(ctx, 100, &h) == ACMPC_OK);        // left for acmpc_destroy to release
    // argument errors
    REQ(acmpc_aggregate(ctx, nullptr, pidx.data(), qty.data(), N, cust.data(), M, 5, C, tot.data(), pq.data(), ct.data()) == ACMPC_E_INVALID);
    REQ(acmpc_set_products(ctx, price.data(), 0) == ACMPC_E_INVALID);
    codes[32] = 'K'; codes[33] = '0'; codes[34] = 0;          // duplicate of code 0
    if (P > 1) REQ(acmpc_set_product_codes(ctx, codes.data(), P) == ACMPC_E_INVALID);
    printf("round %d ok: %lld lines, %d orders, %d products\n", round, (long long)N, M, P);
  }
  acmpc_destroy(ctx);
  printf("asan driver done\n");
  return 0;
}
