/*
 * ingest_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Row-at-a-time CPU restatement of the `detalle` text ingest: SQL-dump / CSV text -> the three int32
 * columns the aggregation engine reads.  Only tests and benchmarks' cpu_baseline legs load it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference never reads `detalle` back and ships no dump of it.
 * The text shapes accepted here are the ones the reference itself writes:
 *   - the statement of process/confirmcompra.php:36 through library/consulSQL.php:29-30,
 *       insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('5', '1', '1')
 *     (all three values quoted; one statement per line, as a query log holds them),
 *   - the phpMyAdmin dump style of the reference's own store.sql:88-89,125-128: an
 *       INSERT INTO `t` (...) VALUES
 *     header line followed by one "(v, 'v', v)," row per line, the last one ending in ";",
 *   - plain comma- or tab-separated rows, one per line (SELECT ... INTO OUTFILE).
 *
 * Grammar (shared with the device kernels, which must produce identical columns):
 *   A line starts at offset 0 or after '\n'.  It holds a DATA row iff its first byte is a digit, '-'
 *   followed by a digit, '(' or a quote (the row starts with the line), or it is a STATEMENT line: its
 *   first six bytes are "insert" in any letter case and, at most 100 bytes from the line start and before
 *   the '\n', there is a '(' directly preceded (spaces allowed between) by "values" in any letter case;
 *   the row then starts at the first such '('.  Every other line (the INSERT header of the dump style,
 *   "--" comments, blanks) is skipped.
 *   row   := ['('] sp int sp sep sp code sp sep sp int sp [')'] [',' | ';'] ['\r'] ('\n' | end)
 *   int   := [q] ['-'] digit{1,10} [q]      value must fit int32; q is ' or " and must match
 *   code  := q bytes-until-q q  |  bytes until one of , \t space \r \n )      0..30 bytes
 *   sep   := ',' | '\t'         sp := ' '*
 *   A data row longer than 120 bytes (from its first byte to before its '\n') or that does not match is MALFORMED: it yields (0, -1, 0) and is counted.
 *   CodigoProd is looked up by exact bytes in the product code list; unknown codes give -1
 *   (an unmatched code returns no row at process/carrito.php:13-14).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static int code_len(const char* c) { int n = 0; while (n < 32 && c[n]) ++n; return n; }

static int is_digit(unsigned char c) { return c >= '0' && c <= '9'; }

static int lower_is(const unsigned char* t, const char* word)
{
  for (int i = 0; word[i]; ++i) if ((t[i] | 0x20) != (unsigned char)word[i]) return 0;
  return 1;
}

/* Offset of the data row of the line [p, e), or -1 when the line holds none. */
static int64_t data_row_start(const unsigned char* t, int64_t p, int64_t e)
{
  unsigned char c = t[p];
  if (is_digit(c) || c == '(' || c == '\'' || c == '"') return p;
  if (c == '-' && p + 1 < e && is_digit(t[p + 1])) return p;
  if (e - p >= 6 && lower_is(t + p, "insert")) {      /* the statement of library/consulSQL.php:30 */
    for (int64_t j = p + 6; j < e && j - p <= 100; ++j) {
      if (t[j] != '(') continue;
      int64_t k = j - 1;
      while (k > p && t[k] == ' ') --k;
      if (k - 5 >= p + 6 && lower_is(t + k - 5, "values")) return j;
    }
  }
  return -1;
}

/* Parses one int field at *pp; returns 0 on success. */
static int parse_int(const unsigned char* t, int64_t n, int64_t* pp, int32_t* out)
{
  int64_t p = *pp;
  unsigned char q = 0;
  if (p < n && (t[p] == '\'' || t[p] == '"')) q = t[p++];
  int neg = 0;
  if (p < n && t[p] == '-') { neg = 1; ++p; }
  int nd = 0;
  int64_t v = 0;
  while (p < n && is_digit(t[p])) { v = v * 10 + (t[p] - '0'); ++p; if (++nd > 10) return 1; }
  if (nd == 0) return 1;
  if (neg) v = -v;
  if (v < INT32_MIN || v > INT32_MAX) return 1;
  if (q) { if (p >= n || t[p] != q) return 1; ++p; }
  *out = (int32_t)v;
  *pp = p;
  return 0;
}

static void skip_sp(const unsigned char* t, int64_t n, int64_t* pp) { while (*pp < n && t[*pp] == ' ') ++*pp; }

static int expect_sep(const unsigned char* t, int64_t n, int64_t* pp)
{
  if (*pp < n && (t[*pp] == ',' || t[*pp] == '\t')) { ++*pp; return 0; }
  return 1;
}

/* codes: n_products entries of 32 bytes, NUL padded (codes are at most 30 bytes, store.sql:99,110). */
/* Returns the number of data rows (which may exceed capacity: rows past it are counted, not stored);
 * *n_malformed receives the number of malformed data rows. */
ORACLE_API int64_t oracle_ingest_detalle(const char* text, int64_t n_bytes, const char* codes, int32_t n_products,
                                         int32_t* order_id, int32_t* product_idx, int32_t* qty, int64_t capacity,
                                         int64_t* n_malformed)
{
  const unsigned char* t = (const unsigned char*)text;
  int64_t rows = 0, bad = 0, p = 0;
  /* a tiny open hash over the codes keeps the oracle usable on millions of rows */
  int32_t hsize = 1;
  while (hsize < 4 * n_products + 4) hsize <<= 1;
  int32_t* htab = (int32_t*)malloc((size_t)hsize * sizeof(int32_t));
  for (int32_t i = 0; i < hsize; ++i) htab[i] = -1;
  for (int32_t i = 0; i < n_products; ++i) {
    const char* c = codes + (size_t)i * 32;
    uint64_t h = 1469598103934665603ull;
    for (int k = 0; k < 32 && c[k]; ++k) h = (h ^ (unsigned char)c[k]) * 1099511628211ull;
    int32_t s = (int32_t)(h & (uint64_t)(hsize - 1));
    while (htab[s] >= 0) s = (s + 1) & (hsize - 1);
    htab[s] = i;
  }
  while (p < n_bytes) {
    int64_t start = p;
    int64_t e = p;
    while (e < n_bytes && t[e] != '\n') ++e;          /* e: the row's '\n' or end of text */
    const int64_t rs = e > start ? data_row_start(t, start, e) : -1;
    if (rs >= 0) {
      int32_t a = 0, q = 0, pi = -1;
      int ok = 1;
      int64_t c = rs;
      if (t[c] == '(') ++c;
      skip_sp(t, e, &c);
      if (parse_int(t, e, &c, &a)) ok = 0;
      if (ok) { skip_sp(t, e, &c); if (expect_sep(t, e, &c)) ok = 0; }
      const unsigned char* cs = NULL;
      int clen = 0;
      if (ok) {
        skip_sp(t, e, &c);
        if (c < e && (t[c] == '\'' || t[c] == '"')) {
          unsigned char qt = t[c++];
          int64_t b = c;
          while (c < e && t[c] != qt) ++c;
          if (c >= e) ok = 0;
          else { cs = t + b; clen = (int)(c - b); ++c; if (c < e && t[c] == qt) ok = 0; }
        } else {
          int64_t b = c;
          while (c < e && t[c] != ',' && t[c] != '\t' && t[c] != ' ' && t[c] != '\r' && t[c] != ')') ++c;
          cs = t + b; clen = (int)(c - b);
        }
        if (clen > 30) ok = 0;
      }
      if (ok) { skip_sp(t, e, &c); if (expect_sep(t, e, &c)) ok = 0; }
      if (ok) { skip_sp(t, e, &c); if (parse_int(t, e, &c, &q)) ok = 0; }
      if (ok) {
        skip_sp(t, e, &c);
        if (c < e && t[c] == ')') ++c;
        if (c < e && (t[c] == ',' || t[c] == ';')) ++c;
        if (c < e && t[c] == '\r') ++c;
        if (c != e) ok = 0;
      }
      if (e - rs > 120) ok = 0;                    /* longest row the grammar admits */
      if (ok) {
        uint64_t h = 1469598103934665603ull;
        for (int k = 0; k < clen; ++k) h = (h ^ cs[k]) * 1099511628211ull;
        int32_t s = (int32_t)(h & (uint64_t)(hsize - 1));
        while (htab[s] >= 0) {
          const char* cc = codes + (size_t)htab[s] * 32;
          if (code_len(cc) == clen && memcmp(cc, cs, (size_t)clen) == 0) { pi = htab[s]; break; }
          s = (s + 1) & (hsize - 1);
        }
      } else {
        ++bad; a = 0; q = 0; pi = -1;
      }
      if (rows < capacity) { order_id[rows] = a; product_idx[rows] = pi; qty[rows] = q; }
      ++rows;
    }
    p = e + 1;
  }
  free(htab);
  if (n_malformed) *n_malformed = bad;
  return rows;
}
