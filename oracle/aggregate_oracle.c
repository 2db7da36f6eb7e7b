/*
 * aggregate_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain, single-threaded C restatement of the reference checkout arithmetic,
 * applied row by row to the `detalle` table exactly in table order.  Only the
 * tests, the smoke check and the benchmark's cpu_baseline leg may load this
 * library; the product (libacmpc.so) never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, fixtures or
 * golden vectors, and neither PHP nor MySQL/MariaDB exists in the build image,
 * so the reference path cannot be executed.  What pins this file instead:
 *   - the hand-derived known-answer cases anchored to the seed rows of
 *     store.sql:89,126-128,232 (tests/golden/checkout_cases.json),
 *   - oracle/checkout_oracle.py, which restates the PHP double accumulate,
 *     the PHP float->string step and the MariaDB DECIMAL(30,2) rounding and is
 *     asserted equal to the integer-cent sums computed here,
 *   - an independent sqlite3 GROUP BY cross-check in the tests.
 *
 * What each function follows:
 *   oracle_aggregate     process/carrito.php:12-18  (for every cart line: look
 *                        up the product, add its Precio to $suma; an unknown
 *                        code returns no row and adds nothing),
 *                        process/carrito.php:21 + process/confirmcompra.php:24
 *                        ($suma becomes venta.TotalPagar of the order),
 *                        process/confirmcompra.php:35-37 (one detalle row per
 *                        cart entry, CantidadProductos = 1), and the two
 *                        aggregates the schema stores the inputs for:
 *                        SUM(CantidadProductos) GROUP BY CodigoProd
 *                        (store.sql:97-101,116) and SUM(TotalPagar) GROUP BY NIT
 *                        (store.sql:158-165,215).
 *   oracle_synth_*       no reference counterpart: the integer-only synthetic
 *                        data recipe documented in include/acmpc.h, written a
 *                        second time here, independently of the device code.
 *
 * Sums are kept in uint64_t so that overflow wraps (two's complement) instead
 * of being undefined; results are reinterpreted as int64_t.
 */
#define _POSIX_C_SOURCE 200809L      /* pthread barriers under -std=c11 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* Returns 0, or a bit mask: 1 = order id out of range, 2 = order ids not
 * non-decreasing, 4 = customer out of range.  Out-of-range rows are skipped. */
ORACLE_API int oracle_aggregate(const int32_t* order_id, const int32_t* product_idx,
                                const int32_t* qty, int64_t n_lines,
                                const int32_t* order_customer, int64_t n_orders,
                                int32_t order_id_base, int32_t n_customers,
                                const int64_t* price_cents, int32_t n_products,
                                int64_t* order_total_cents, int64_t* product_qty,
                                int64_t* customer_total_cents)
{
  int flags = 0;
  uint64_t* tot = (uint64_t*)order_total_cents;
  uint64_t* pq = (uint64_t*)product_qty;
  uint64_t* ct = (uint64_t*)customer_total_cents;
  memset(tot, 0, (size_t)n_orders * sizeof(uint64_t));
  memset(pq, 0, (size_t)n_products * sizeof(uint64_t));
  memset(ct, 0, (size_t)n_customers * sizeof(uint64_t));

  /* carrito.php:12-18 — walk the cart lines in stored order. */
  for (int64_t l = 0; l < n_lines; ++l) {
    if (l > 0 && order_id[l] < order_id[l - 1]) flags |= 2;
    int64_t o = (int64_t)order_id[l] - (int64_t)order_id_base;
    if (o < 0 || o >= n_orders) { flags |= 1; continue; }
    int32_t p = product_idx[l];
    if (p < 0 || p >= n_products) continue;          /* unknown CodigoProd: no row, adds nothing */
    uint64_t q = (uint64_t)(int64_t)qty[l];           /* detalle.CantidadProductos */
    tot[o] += (uint64_t)price_cents[p] * q;           /* $suma += $fila['Precio']  */
    pq[p] += q;                                       /* stock delta of product p  */
  }
  /* confirmcompra.php:24 — TotalPagar is booked on the order's NIT. */
  for (int64_t o = 0; o < n_orders; ++o) {
    int32_t c = order_customer[o];
    if (c < 0 || c >= n_customers) { flags |= 4; continue; }
    ct[c] += tot[o];
  }
  return flags;
}

/*
 * The same walk with the order-lifecycle columns (each may be NULL):
 *   line_price_cents[l]  price at the time of sale: process/updateProduct.php:15 may change
 *                        producto.Precio later, venta.TotalPagar keeps the old sum
 *                        (process/confirmcompra.php:24);
 *   order_state[o]       bit 0 clear: cancelled through venta.Estado (process/updatePedido.php:11)
 *                        or deleted (process/delPedido.php:12-13): the order keeps its total but
 *                        adds nothing to stock deltas and customer totals;
 *   order_descuento[o]   venta.Descuento (store.sql:162): mode 0 = whole currency units,
 *                        mode 1 = per cent of the gross total, rounded half away from zero to a cent.
 * The reference has no code for any of this; the definitions are the build's own.
 */
ORACLE_API int oracle_aggregate_lifecycle(const int32_t* order_id, const int32_t* product_idx,
                                          const int32_t* qty, int64_t n_lines,
                                          const int32_t* order_customer, int64_t n_orders,
                                          int32_t order_id_base, int32_t n_customers,
                                          const int64_t* price_cents, int32_t n_products,
                                          const int64_t* line_price_cents, const uint8_t* order_state,
                                          const int32_t* order_descuento, int32_t descuento_mode,
                                          int64_t* order_total_cents, int64_t* product_qty,
                                          int64_t* customer_total_cents)
{
  int flags = 0;
  uint64_t* tot = (uint64_t*)order_total_cents;
  uint64_t* pq = (uint64_t*)product_qty;
  uint64_t* ct = (uint64_t*)customer_total_cents;
  memset(tot, 0, (size_t)n_orders * sizeof(uint64_t));
  memset(pq, 0, (size_t)n_products * sizeof(uint64_t));
  memset(ct, 0, (size_t)n_customers * sizeof(uint64_t));
  for (int64_t l = 0; l < n_lines; ++l) {
    if (l > 0 && order_id[l] < order_id[l - 1]) flags |= 2;
    int64_t o = (int64_t)order_id[l] - (int64_t)order_id_base;
    if (o < 0 || o >= n_orders) { flags |= 1; continue; }
    int32_t p = product_idx[l];
    if (p < 0 || p >= n_products) continue;
    uint64_t q = (uint64_t)(int64_t)qty[l];
    uint64_t price = (uint64_t)(line_price_cents ? line_price_cents[l] : price_cents[p]);
    tot[o] += price * q;
    if (!order_state || (order_state[o] & 1)) pq[p] += q;
  }
  for (int64_t o = 0; o < n_orders; ++o) {
    if (order_descuento) {
      int64_t gross = (int64_t)tot[o];
      uint64_t disc;
      if (descuento_mode == 0) {
        disc = (uint64_t)(100 * (int64_t)order_descuento[o]);
      } else {
        __int128 x = (__int128)gross * (__int128)order_descuento[o];
        __int128 r = x >= 0 ? (x + 50) / 100 : -((-x + 50) / 100);
        disc = (uint64_t)r;                               /* low 64 bits: wrap-around */
      }
      tot[o] -= disc;
    }
    int32_t c = order_customer[o];
    if (c < 0 || c >= n_customers) { flags |= 4; continue; }
    if (!order_state || (order_state[o] & 1)) ct[c] += tot[o];
  }
  return flags;
}

/*
 * oracle_aggregate on n_threads host threads: the orders are cut into n_threads contiguous ranges on
 * order boundaries (the same cut the multi-GPU path makes), every thread walks the lines of its
 * range with private per-product and per-customer sums, and the private sums are added up at the end.
 * Requires non-decreasing order ids (a descent found by any thread sends the whole call to the
 * one-thread walk).
 * Integer sums: the result is identical to the one-thread walk.
 */
#include <pthread.h>

typedef struct {
  const int32_t *order_id, *product_idx, *qty, *order_customer;
  int64_t line_lo, line_hi, order_lo, order_hi, n_orders;
  int32_t base, n_customers, n_products;
  const int64_t* price;
  uint64_t *tot, *pq, *ct;       /* tot: shared (disjoint ranges); pq, ct: private */
  /* second phase: every thread adds up one slice of the private sums of all threads */
  uint64_t *scratch, *out_pq, *out_ct;
  size_t priv;
  int t, n_threads;
  pthread_barrier_t* barrier;
  int flags;
} mt_job;

static void* mt_worker(void* arg)
{
  mt_job* j = (mt_job*)arg;
  memset(j->pq, 0, j->priv * sizeof(uint64_t));
  memset(j->tot + j->order_lo, 0, (size_t)(j->order_hi - j->order_lo) * sizeof(uint64_t));
  for (int64_t l = j->line_lo; l < j->line_hi; ++l) {
    if (l > 0 && j->order_id[l] < j->order_id[l - 1]) { j->flags |= 2; break; }   /* redone on one thread */
    int64_t o = (int64_t)j->order_id[l] - (int64_t)j->base;
    if (o < 0 || o >= j->n_orders) { j->flags |= 1; continue; }
    int32_t p = j->product_idx[l];
    if (p < 0 || p >= j->n_products) continue;
    uint64_t q = (uint64_t)(int64_t)j->qty[l];
    j->tot[o] += (uint64_t)j->price[p] * q;
    j->pq[p] += q;
  }
  if (!(j->flags & 2)) {
    for (int64_t o = j->order_lo; o < j->order_hi; ++o) {
      int32_t c = j->order_customer[o];
      if (c < 0 || c >= j->n_customers) { j->flags |= 4; continue; }
      j->ct[c] += j->tot[o];
    }
  }
  pthread_barrier_wait(j->barrier);
  /* slice [lo, hi) of the concatenated [product | customer] sums */
  size_t lo = j->priv * (size_t)j->t / (size_t)j->n_threads, hi = j->priv * (size_t)(j->t + 1) / (size_t)j->n_threads;
  for (size_t i = lo; i < hi; ++i) {
    uint64_t acc = 0;
    for (int t = 0; t < j->n_threads; ++t) acc += j->scratch[j->priv * (size_t)t + i];
    if (i < (size_t)j->n_products) j->out_pq[i] = acc; else j->out_ct[i - (size_t)j->n_products] = acc;
  }
  return NULL;
}

ORACLE_API int oracle_aggregate_mt(const int32_t* order_id, const int32_t* product_idx,
                                   const int32_t* qty, int64_t n_lines,
                                   const int32_t* order_customer, int64_t n_orders,
                                   int32_t order_id_base, int32_t n_customers,
                                   const int64_t* price_cents, int32_t n_products,
                                   int64_t* order_total_cents, int64_t* product_qty,
                                   int64_t* customer_total_cents, int32_t n_threads)
{
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  uint64_t* tot = (uint64_t*)order_total_cents;
  size_t priv = (size_t)n_products + (size_t)n_customers;
  uint64_t* scratch = (uint64_t*)malloc(priv * (size_t)n_threads * sizeof(uint64_t));
  mt_job* jobs = (mt_job*)calloc((size_t)n_threads, sizeof(mt_job));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  pthread_barrier_t barrier;
  pthread_barrier_init(&barrier, NULL, (unsigned)n_threads);
  int64_t line = 0;
  for (int t = 0; t < n_threads; ++t) {
    mt_job* j = &jobs[t];
    j->order_id = order_id; j->product_idx = product_idx; j->qty = qty; j->order_customer = order_customer;
    j->order_lo = n_orders * t / n_threads;
    j->order_hi = n_orders * (t + 1) / n_threads;
    j->n_orders = n_orders; j->base = order_id_base; j->n_customers = n_customers; j->n_products = n_products;
    j->price = price_cents; j->tot = tot;
    j->pq = scratch + priv * (size_t)t;
    j->ct = j->pq + n_products;
    j->scratch = scratch; j->priv = priv; j->t = t; j->n_threads = n_threads; j->barrier = &barrier;
    j->out_pq = (uint64_t*)product_qty; j->out_ct = (uint64_t*)customer_total_cents;
    /* lines of this range: ids below base + order_hi (the last thread also takes any id beyond the range;
     * an id below the base lands in thread 0; both are flagged by the walk) */
    j->line_lo = line;
    if (t == n_threads - 1) line = n_lines;
    else {
      int64_t lo = line, hi = n_lines, key = (int64_t)order_id_base + j->order_hi;
      while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if ((int64_t)order_id[mid] < key) lo = mid + 1; else hi = mid; }
      line = lo;
    }
    j->line_hi = line;
  }
  for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
  int flags = 0;
  for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); flags |= jobs[t].flags; }
  pthread_barrier_destroy(&barrier);
  free(th); free(jobs); free(scratch);
  if (flags & 2)                 /* the cut needs sorted ids: walk the table on one thread instead */
    return oracle_aggregate(order_id, product_idx, qty, n_lines, order_customer, n_orders, order_id_base,
                            n_customers, price_cents, n_products, order_total_cents, product_qty,
                            customer_total_cents);
  return flags;
}

/* ---------------- synthetic data (see include/acmpc.h for the recipe) -------- */

#define GOLDEN 0x9E3779B97F4A7C15ull
#define SEED_LEN   0xA11CEull
#define SEED_PROD  0xB0Bull
#define SEED_QTY   0x51DEull
#define SEED_PRICE 0xCAFEull
#define SEED_CUST  0xD00Dull

static uint64_t h64(uint64_t seed, uint64_t i)
{
  uint64_t z = (seed ^ (i * GOLDEN)) + GOLDEN;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

ORACLE_API uint64_t oracle_hash(uint64_t seed, uint64_t i) { return h64(seed, i); }

/* Number of orders needed to hold n_lines_total lines (last order truncated). */
ORACLE_API int64_t oracle_synth_n_orders(int64_t n_lines_total)
{
  int64_t acc = 0, o = 0;
  while (acc < n_lines_total) { acc += 1 + (int64_t)(h64(SEED_LEN, (uint64_t)o) % 7u); ++o; }
  return o;
}

ORACLE_API void oracle_synth_prices(int32_t n_products, int64_t* price_cents)
{
  static const int64_t seed_rows[3] = {25000, 45000, 150000}; /* store.sql:126-128 */
  for (int32_t p = 0; p < n_products; ++p)
    price_cents[p] = p < 3 ? seed_rows[p] : 1 + (int64_t)(h64(SEED_PRICE, (uint64_t)p) % 500000u);
}

/* Fill the slice of the data set owned by orders [order_lo, order_hi).  The
 * caller sizes the line arrays with oracle_synth_count_lines(). */
ORACLE_API int64_t oracle_synth_count_lines(int64_t n_lines_total, int64_t order_lo, int64_t order_hi,
                                            int64_t* line_lo_out)
{
  int64_t acc = 0, lo = 0;
  for (int64_t o = 0; o < order_hi; ++o) {
    if (o == order_lo) lo = acc;
    acc += 1 + (int64_t)(h64(SEED_LEN, (uint64_t)o) % 7u);
    if (acc > n_lines_total) acc = n_lines_total;
  }
  if (order_lo >= order_hi) lo = acc;
  if (line_lo_out) *line_lo_out = lo;
  return acc - lo;
}

ORACLE_API void oracle_synth_fill(int64_t n_lines_total, int32_t n_products, int32_t n_customers,
                                  int32_t order_id_base, int32_t zipf, int32_t qty_mode,
                                  int64_t order_lo, int64_t order_hi,
                                  int32_t* order_id, int32_t* product_idx, int32_t* qty,
                                  int32_t* order_customer)
{
  uint64_t* cdf = NULL;
  uint64_t wtot = 0;
  if (zipf) {
    cdf = (uint64_t*)malloc((size_t)n_products * sizeof(uint64_t));
    for (int32_t k = 0; k < n_products; ++k) { wtot += (1ull << 32) / (uint64_t)(k + 1); cdf[k] = wtot; }
  }
  int64_t line = 0;          /* global line index */
  int64_t out = 0;           /* index into the shard's arrays */
  for (int64_t o = 0; o < order_hi; ++o) {
    int64_t len = 1 + (int64_t)(h64(SEED_LEN, (uint64_t)o) % 7u);
    if (line + len > n_lines_total) len = n_lines_total - line;
    if (o >= order_lo) {
      order_customer[o - order_lo] = (int32_t)(h64(SEED_CUST, (uint64_t)o) % (uint64_t)n_customers);
      for (int64_t i = 0; i < len; ++i) {
        uint64_t l = (uint64_t)(line + i);
        uint64_t hp = h64(SEED_PROD, l);
        int32_t p;
        if (!zipf) {
          p = (int32_t)(hp % (uint64_t)n_products);
        } else {
          uint64_t u = hp % wtot;
          int32_t a = 0, b = n_products;           /* first k with cdf[k] > u */
          while (a < b) { int32_t m = a + (b - a) / 2; if (cdf[m] > u) b = m; else a = m + 1; }
          p = a;
        }
        order_id[out] = (int32_t)(order_id_base + o);
        product_idx[out] = p;
        qty[out] = qty_mode ? 1 + (int32_t)(h64(SEED_QTY, l) % 5u) : 1;
        ++out;
      }
    }
    line += len;
  }
  free(cdf);
}
