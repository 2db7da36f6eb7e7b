/*
 * emit_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Row-at-a-time CPU restatement of the text emit: per-order totals and per-product stock deltas ->
 * the statements that would store them in the shop's tables.  Only tests and benchmark baseline legs
 * load it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference writes venta.TotalPagar once
 * (process/confirmcompra.php:24) and never touches producto.Stock; it ships no such statements.  The
 * shapes follow the reference's own writer, consultasSQL::UpdateSQL (library/consulSQL.php:43-44:
 * "update $tabla set $campos where $condicion") with values quoted as in process/updatePedido.php:11.
 * An amount is integer cents printed as [-]units.cc, MariaDB's text for DECIMAL(30,2)
 * (store.sql:163).
 */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static uint64_t mag(int64_t v) { return v < 0 ? 0u - (uint64_t)v : (uint64_t)v; }

static int amount(char* dst, int64_t cents)
{
  uint64_t m = mag(cents);
  return sprintf(dst, "%s%" PRIu64 ".%02u", cents < 0 ? "-" : "", m / 100u, (unsigned)(m % 100u));
}

/* number_format($suma, 2) of process/carrito.php:19 for an amount held exactly in cents. */
static int grouped(char* dst, int64_t cents)
{
  char digits[32];
  uint64_t m = mag(cents);
  int nd = sprintf(digits, "%" PRIu64, m / 100u), n = 0;
  if (cents < 0) dst[n++] = '-';
  for (int i = 0; i < nd; ++i) {
    if (i && (nd - i) % 3 == 0) dst[n++] = ',';
    dst[n++] = digits[i];
  }
  n += sprintf(dst + n, ".%02u", (unsigned)(m % 100u));
  return n;
}

/* Appends n bytes of src at offset *pos if they fit; always advances *pos. */
static void put(char* out, int64_t cap, int64_t* pos, const char* src, int n)
{
  for (int i = 0; i < n; ++i)
    if (*pos + i < cap) out[*pos + i] = src[i];
  *pos += n;
}

/* style 0: SQL update statements, 1: CSV, 2: the subtotal fragment of process/carrito.php:19.  Returns the
 * bytes the full text needs. */
ORACLE_API int64_t oracle_emit_totals(const int64_t* total_cents, int64_t n_orders, int32_t order_id_base, int style,
                                      char* out, int64_t capacity)
{
  int64_t pos = 0;
  char row[160], am[32];
  for (int64_t i = 0; i < n_orders; ++i) {
    amount(am, total_cents[i]);
    int64_t id = (int64_t)order_id_base + i;
    int n;
    if (style == 2) { char g[48]; g[grouped(g, total_cents[i])] = 0; n = sprintf(row, "<tr><td>Subtotal</td><td>$%s</td></tr>\n", g); }
    else n = style == 0 ? sprintf(row, "update venta set TotalPagar='%s' where NumPedido='%" PRId64 "';\n", am, id)
                        : sprintf(row, "%" PRId64 ",%s\n", id, am);
    put(out, capacity, &pos, row, n);
  }
  return pos;
}

/* codes: n_products entries of 32 bytes, NUL padded.  Products with a zero delta get no row. */
ORACLE_API int64_t oracle_emit_stock(const int64_t* product_qty, int32_t n_products, const char* codes, int style,
                                     char* out, int64_t capacity)
{
  int64_t pos = 0;
  char row[200], code[33];
  for (int32_t p = 0; p < n_products; ++p) {
    int64_t d = product_qty[p];
    if (d == 0) continue;
    memcpy(code, codes + (size_t)p * 32, 32);
    code[32] = 0;
    int n = style == 0 ? sprintf(row, "update producto set Stock=Stock%c%" PRIu64 " where CodigoProd='%s';\n",
                                 d < 0 ? '+' : '-', mag(d), code)
                       : sprintf(row, "%s,%s%" PRIu64 "\n", code, d < 0 ? "-" : "", mag(d));
    put(out, capacity, &pos, row, n);
  }
  return pos;
}
