/*
 * acmpc.h — C ABI of the MI355X (gfx950) checkout-aggregation engine.
 *
 * The engine recomputes, in bulk and over columnar arrays, what the reference
 * storefront's checkout produces one row at a time:
 *
 *   order_total_cents[o]    = sum over the `detalle` rows of order o of
 *                             price_cents[product_idx] * qty
 *                             (process/carrito.php:12-21 accumulates Precio per
 *                             cart line; process/confirmcompra.php:24 stores the
 *                             sum as venta.TotalPagar)
 *   product_qty[p]          = sum of detalle.CantidadProductos over rows whose
 *                             CodigoProd is product p (store.sql:97-101,116 —
 *                             the stock decrement the schema implies)
 *   customer_total_cents[c] = sum of venta.TotalPagar over the orders whose
 *                             venta.NIT is customer c (store.sql:158-165,215)
 *
 * All arithmetic is two's-complement int64 in cents (wrap-around on overflow,
 * like unsigned 64-bit addition), so every result is independent of the order
 * in which partial sums are merged.
 *
 * Every entry point is plain C: pointers, sizes, integer return codes.  There
 * is no CPU backend: a context can only be created on a HIP device, and every
 * call fails with a negative code (never abort()/exit()) when something is
 * wrong.  The reference's convention is die("<message>") at
 * library/consulSQL.php:12,19,31,38,45; a caller that wants that behaviour
 * maps a non-zero return to die(acmpc_last_error(ctx)).
 *
 * Threading: one context is used by one thread at a time.  Many processes may
 * each hold their own context on the same GPU.  The host-pointer entry points
 * start a few short-lived threads of their own while they copy more than 32 MiB
 * (copies from and to pageable memory hold their calling thread; several at once
 * cross the link faster), and acmpc_aggregate_sharded one per device; all have
 * ended when the call returns.
 *
 * Limits (larger counts return ACMPC_E_INVALID): n_orders < 2^31 (order ids are
 * INT, store.sql:98,159), n_lines <= 2^40, n_products <= 2^28 (2^21 for a code
 * list), n_customers <= 2^30, 2^44 bytes of text per ingest call.
 */
#ifndef ACMPC_H
#define ACMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct acmpc_ctx acmpc_ctx;

/* Return codes. */
#define ACMPC_OK            0
#define ACMPC_E_INVALID    -1  /* bad argument (null pointer, negative size, ...)      */
#define ACMPC_E_HIP        -2  /* a HIP runtime call failed; see acmpc_last_error      */
#define ACMPC_E_ALIGN      -3  /* a device column is not 16-byte aligned               */
#define ACMPC_E_DATA       -4  /* input breaks the data contract; see acmpc_status     */
#define ACMPC_E_STATE      -5  /* call order wrong (e.g. aggregate before products)    */
#define ACMPC_E_NOMEM      -6  /* host or device allocation failed                     */

/* Bits reported by acmpc_status(): which part of the data contract was broken. */
#define ACMPC_ST_ORDER_RANGE     1u /* some order_id outside [base, base+M)             */
#define ACMPC_ST_ORDER_UNSORTED  2u /* order_id column is not non-decreasing            */
#define ACMPC_ST_CUSTOMER_RANGE  4u /* some order_customer outside [0, C)               */
#define ACMPC_ST_INGEST_MALFORMED 8u /* acmpc_ingest_*: a data row did not match the grammar */
#define ACMPC_ST_INGEST_CAPACITY 16u /* acmpc_ingest_*: more data rows than capacity_rows    */
#define ACMPC_ST_INTERNAL        64u /* a device-side wait gave up (never expected): results of that call are invalid */

/* Kernel slots of acmpc_last_kernel_ms(). */
#define ACMPC_K_LINES      0   /* order-line pass: totals, stock deltas               */
#define ACMPC_K_FIXUP      1   /* orders cut by a wave boundary + stock-delta merge    */
#define ACMPC_K_CUSTOMERS  2   /* per-customer pass                                    */
#define ACMPC_K_CUSTMERGE  3   /* customer partial merge                               */
#define ACMPC_N_KERNELS    4

/*
 * Create / destroy a context bound to HIP device `device` (>= 0).
 * Replaces: ejecutarSQL::conectar(), library/consulSQL.php:5-16 (one connection
 * per statement there; one long-lived context per worker process here).
 * acmpc_last_error(NULL) returns the message of a failed acmpc_init.
 */
int  acmpc_init(acmpc_ctx** out, int device);
void acmpc_destroy(acmpc_ctx* ctx);

/* Static or context-owned message of the last failure; never NULL.
 * Replaces the text handed to die() at library/consulSQL.php:12,19,31,38,45. */
const char* acmpc_last_error(const acmpc_ctx* ctx);

/* "acmpc <major>.<minor> gfx950".  No counterpart in the reference. */
const char* acmpc_version(void);

/*
 * Load the product table: price_cents[p] is producto.Precio (DECIMAL(30,2),
 * store.sql:113) of dictionary entry p, in integer cents.  Host pointer.
 * Replaces the per-line "select * from producto where CodigoProd=..." lookup of
 * process/carrito.php:13.
 */
int acmpc_set_products(acmpc_ctx* ctx, const int64_t* price_cents, int32_t n_products);

/*
 * Aggregate N order lines (the `detalle` table, store.sql:97-101, one row per
 * cart entry as written by process/confirmcompra.php:35-37) held in HOST memory.
 *
 *   order_id[N]       detalle.NumPedido; non-decreasing (rows of one order are
 *                     contiguous), each in [order_id_base, order_id_base + M)
 *   product_idx[N]    dictionary index of detalle.CodigoProd into the product
 *                     table; any value outside [0, n_products) is an unknown
 *                     code: it adds nothing to any sum, as an unmatched
 *                     CodigoProd yields no row at process/carrito.php:13-14
 *   qty[N]            detalle.CantidadProductos (always 1 in the reference,
 *                     process/confirmcompra.php:36)
 *   order_customer[M] venta.NIT (= cliente.id_Cliente) of order
 *                     order_id_base + i, as a dictionary index in [0, C)
 *   out_order_total_cents[M], out_product_qty[n_products],
 *   out_customer_total_cents[C]   results, fully overwritten
 *
 * Copies in, runs the device path, copies out, and returns ACMPC_E_DATA if the
 * data contract was broken (results are then unspecified but no memory outside
 * the output arrays is touched).
 * Replaces: the loops of process/carrito.php:12-18 and
 * process/confirmcompra.php:24,35-37 over one cart, for all carts at once.
 */
int acmpc_aggregate(acmpc_ctx* ctx,
                    const int32_t* order_id, const int32_t* product_idx, const int32_t* qty,
                    int64_t n_lines,
                    const int32_t* order_customer, int64_t n_orders,
                    int32_t order_id_base, int32_t n_customers,
                    int64_t* out_order_total_cents,
                    int64_t* out_product_qty,
                    int64_t* out_customer_total_cents);

/*
 * Same computation on DEVICE-resident columns and outputs (device pointers on
 * the context's device; the four input columns and d_out_order_total_cents
 * 16-byte aligned, the other two outputs 8-byte aligned).  Work is enqueued on `hip_stream`
 * (a hipStream_t; NULL = the default stream) and the call returns without
 * waiting.  Uses a workspace owned by the context: calls on one context must
 * not overlap in time.  Contract violations are reported by acmpc_status().
 * Replaces the same loops as acmpc_aggregate (process/carrito.php:12-18,
 * process/confirmcompra.php:24,35-37) for callers that keep the columns in HBM.
 */
int acmpc_aggregate_device(acmpc_ctx* ctx,
                           const int32_t* d_order_id, const int32_t* d_product_idx,
                           const int32_t* d_qty, int64_t n_lines,
                           const int32_t* d_order_customer, int64_t n_orders,
                           int32_t order_id_base, int32_t n_customers,
                           int64_t* d_out_order_total_cents,
                           int64_t* d_out_product_qty,
                           int64_t* d_out_customer_total_cents,
                           void* hip_stream);

/*
 * acmpc_aggregate_device for callers that cannot allocate device memory while they call it: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in
 * the reference.  Same arguments, same results bit for bit; the call itself allocates and frees
 * nothing.  Its workspace (per-workgroup stock-delta and per-stream customer partials, 64-bit carry
 * cells, the records of orders cut by wave boundaries) is a fixed 61.4 MiB array inside the library's
 * code object, of which every GPU the library runs on holds its own copy until the process ends.
 * Limits (ACMPC_E_INVALID beyond them, with the limit in acmpc_last_error): at most 16380 products
 * whose tables fit LDS (about 17.7 k with prices below 2^31 cents, 11.8 k otherwise) and at most
 * 131072 customers; ACMPC_ORDER_GROUPED only, no lifecycle columns.  acmpc_set_products must have been
 * called before (it is the call that allocates the price table).
 * Every call on one GPU, from any context of the process, uses that GPU's one array: calls that
 * overlap in time on one GPU race, also on different streams or contexts.  Serialise them.
 */
int acmpc_aggregate_device_noalloc(acmpc_ctx* ctx,
                                   const int32_t* d_order_id, const int32_t* d_product_idx,
                                   const int32_t* d_qty, int64_t n_lines,
                                   const int32_t* d_order_customer, int64_t n_orders,
                                   int32_t order_id_base, int32_t n_customers,
                                   int64_t* d_out_order_total_cents,
                                   int64_t* d_out_product_qty,
                                   int64_t* d_out_customer_total_cents,
                                   void* hip_stream);

/*
 * Order lifecycle: optional columns that make the sums right over time, not only for a frozen
 * snapshot of the tables.  Every pointer may be NULL (= the column is absent).
 *
 *   line_price_cents[N]  the price each line was sold at, in cents.  process/updateProduct.php:15
 *                        can change producto.Precio after a sale, while venta.TotalPagar keeps the
 *                        sum at sale time (process/confirmcompra.php:24).  When given, a line of a
 *                        known product contributes line_price_cents * qty instead of the table
 *                        price; an unknown product still adds nothing.
 *   order_state[M]       bit 0 (ACMPC_ORDER_COUNTS) set: the order counts.  Clear it for an order
 *                        whose venta.Estado (process/updatePedido.php:11) says it was cancelled, or
 *                        whose venta row was removed (process/delPedido.php:12-13).  Such an order
 *                        still gets its total (the audit value of its TotalPagar) but adds nothing
 *                        to the stock deltas or to its customer's total.
 *   order_descuento[M]   venta.Descuento (INT, store.sql:162; the reference always stores 0,
 *                        process/confirmcompra.php:24), taken off the order's total:
 *                          ACMPC_DESCUENTO_UNITS    an amount in whole currency units (100 cents each)
 *                          ACMPC_DESCUENTO_PERCENT  per cent of the gross total, the discount rounded
 *                                                   half away from zero to a cent
 *                        Customer totals add the discounted totals.
 * Arithmetic wraps in 64 bits like everything else.  Device pointers must be 16-byte aligned.
 * No counterpart in the reference, which never recomputes a total.
 */
#define ACMPC_ORDER_COUNTS       1u
#define ACMPC_DESCUENTO_UNITS    0
#define ACMPC_DESCUENTO_PERCENT  1
typedef struct acmpc_lifecycle {
  const int64_t* line_price_cents;
  const uint8_t* order_state;
  const int32_t* order_descuento;
  int32_t        descuento_mode;
} acmpc_lifecycle;

/* acmpc_aggregate / acmpc_aggregate_device with the lifecycle columns (`life` may be NULL; its
 * pointers are host pointers for the first, device pointers for the second). */
int acmpc_aggregate_lifecycle(acmpc_ctx* ctx,
                              const int32_t* order_id, const int32_t* product_idx, const int32_t* qty,
                              int64_t n_lines,
                              const int32_t* order_customer, int64_t n_orders,
                              int32_t order_id_base, int32_t n_customers,
                              const acmpc_lifecycle* life,
                              int64_t* out_order_total_cents,
                              int64_t* out_product_qty,
                              int64_t* out_customer_total_cents);
int acmpc_aggregate_lifecycle_device(acmpc_ctx* ctx,
                                     const int32_t* d_order_id, const int32_t* d_product_idx,
                                     const int32_t* d_qty, int64_t n_lines,
                                     const int32_t* d_order_customer, int64_t n_orders,
                                     int32_t order_id_base, int32_t n_customers,
                                     const acmpc_lifecycle* life,
                                     int64_t* d_out_order_total_cents,
                                     int64_t* d_out_product_qty,
                                     int64_t* d_out_customer_total_cents,
                                     void* hip_stream);

/*
 * Wait for `hip_stream`, then store in *flags the OR of ACMPC_ST_* bits raised
 * by device calls since the previous acmpc_status(), and clear them.  The flags
 * of all *_device calls of a context accumulate in one word until they are
 * collected here; the host-pointer entry points start from a clean word and
 * report on their own input only.
 * Returns ACMPC_OK when *flags == 0, ACMPC_E_DATA otherwise.
 * Stands where the reference tests the result of mysqli_query and dies
 * (library/consulSQL.php:18-19): the device calls only enqueue work, so their
 * verdict on the data is collected here.
 */
int acmpc_status(acmpc_ctx* ctx, void* hip_stream, uint32_t* flags);

/*
 * Options.  ACMPC_OPT_LINE_ORDER says what the caller promises about the `detalle` rows:
 *   ACMPC_ORDER_GROUPED (default)  rows of one order contiguous, order ids non-decreasing — the
 *                                  order in which process/confirmcompra.php:35-37 inserts them when
 *                                  checkouts do not overlap; a violation is reported, not summed
 *   ACMPC_ORDER_ANY                no promise: concurrent checkouts interleave their rows
 *                                  (process/confirmcompra.php:27,35-37; `detalle` has no primary key,
 *                                  store.sql:192-194).  Same results.  The device decides how: rows far
 *                                  from grouped are moved next to their order's neighbourhood (a two-level
 *                                  partition by order id) and summed there, rows near it are summed with one
 *                                  64-bit global atomic per run of equal ids, and rows whose ids never go down
 *                                  go through the grouped pass after one look at the ids.  1.5 (grouped after
 *                                  all) to 3.7 times (no order at all) the grouped time.  The lifecycle columns
 *                                  take the same route; so do up to 2^27 orders (beyond 2^25 the buckets hold
 *                                  32768 or 65536 orders and are read twice or four times by the last pass).
 *   ACMPC_ORDER_AUTO               the grouped pass runs first; only if it meets a descending order id
 *                                  are the order totals redone by the any-order pass (decided on the
 *                                  device, without a host round trip).  Costs a few near-empty kernel
 *                                  launches (8 % of a step) when the rows were grouped after all.  An input that is
 *                                  not non-decreasing contains a descent, so none slips through.
 * No counterpart in the reference, which never reads `detalle` back.
 */
#define ACMPC_OPT_LINE_ORDER   1
#define ACMPC_ORDER_GROUPED    0
#define ACMPC_ORDER_ANY        1
#define ACMPC_ORDER_AUTO       2
int acmpc_set_option(acmpc_ctx* ctx, int option, int64_t value);

/*
 * Which way the last aggregate call of this context summed rows whose order was not promised (ACMPC_ORDER_ANY, or
 * ACMPC_ORDER_AUTO after the grouped pass met a descending id).  The device decides, so the answer is read back
 * from it: the call waits for hip_stream.  No counterpart in the reference (diagnostics and tests).
 *   ACMPC_PATH_NONE     the order-restoring path was not part of the call (grouped mode; or its limits: both product
 *                       tables in LDS with prices below 2^31 cents, at most 2^27 orders)
 *   ACMPC_PATH_ATOMIC   rows near enough to grouped: one 64-bit atomic per run of equal ids (also what is reported
 *                       when the path was queued but left at once: ACMPC_ORDER_AUTO on rows that were grouped)
 *   ACMPC_PATH_MOVED    rows moved next to their order's neighbourhood and summed there
 *   ACMPC_PATH_MOVED_STOCK  a line's value did not fit a record: stock deltas from the moving pass, totals by atomics
 *   ACMPC_PATH_GROUPED  the ids never went down: the grouped pass summed them
 */
#define ACMPC_PATH_NONE        -1
#define ACMPC_PATH_ATOMIC       0
#define ACMPC_PATH_MOVED        1
#define ACMPC_PATH_MOVED_STOCK  2
#define ACMPC_PATH_GROUPED      3
int acmpc_last_line_order_path(acmpc_ctx* ctx, void* hip_stream, int* path);

/*
 * Kernel timing.  With profiling enabled every acmpc_aggregate_device() call
 * brackets each of its kernels with HIP events on the call's stream;
 * acmpc_last_kernel_ms() waits for them and returns the milliseconds of the
 * last call's kernels in ms[ACMPC_N_KERNELS].  No counterpart in the reference
 * (measurement only).
 */
int acmpc_set_profiling(acmpc_ctx* ctx, int enabled);
int acmpc_last_kernel_ms(acmpc_ctx* ctx, float* ms);

/* Launch geometry of the order-line pass: workgroups and waves per workgroup.
 * No counterpart in the reference (reported by the benchmark). */
int acmpc_lines_geometry(acmpc_ctx* ctx, int32_t* n_workgroups, int32_t* waves_per_workgroup);

/*
 * Page-locked host memory for the columns, texts and results handed to the
 * host-pointer entry points (acmpc_aggregate, acmpc_ingest_detalle, acmpc_emit_*).
 * Those calls accept any host pointer; from a buffer obtained here the copies to
 * and from the GPU run as direct DMA at the link's rate, from ordinary memory the
 * runtime stages them through its own buffers at a fraction of it.
 * *out is 16-byte aligned, belongs to the context and is released by
 * acmpc_host_free() or, at the latest, by acmpc_destroy().
 * Replaces the PHP arrays in which the reference holds a cart
 * ($_SESSION['producto'], process/carrito.php:7; process/confirmcompra.php:35).
 */
int acmpc_host_alloc(acmpc_ctx* ctx, int64_t n_bytes, void** out);
int acmpc_host_free(acmpc_ctx* ctx, void* p);

/* ------------------------------------------------------------------------
 * More than one GPU.  Orders are independent (each venta row owns its detalle rows,
 * process/confirmcompra.php:24,35-37), so shard g of G owns the contiguous order-index range
 * [floor(g*M/G), floor((g+1)*M/G)) and exactly the lines of those orders; per-order totals stay
 * with their shard.  Only the per-product and per-customer vectors cross shards: ONE in-place
 * all-reduce (sum, int64) of the concatenated [product_qty | customer_total_cents] vector through
 * RCCL over xGMI.  Integer sums do not depend on the order of reduction, so the merged result is
 * bit-identical for any number of shards.  RCCL (librccl.so.1) is loaded when the first of these
 * functions is called; without it they return ACMPC_E_STATE.
 * Stands where the reference opens one connection per statement (ejecutarSQL::conectar,
 * library/consulSQL.php:5-16): one context and one communicator rank per GPU, either all in one
 * process (acmpc_comm_init_all, the route of a PHP worker) or one process per GPU
 * (acmpc_comm_id + acmpc_comm_init_rank, the id carried to the other processes by any channel).
 * ---------------------------------------------------------------------- */
typedef struct acmpc_comm acmpc_comm;
#define ACMPC_COMM_ID_BYTES 128

/* One process per GPU: rank 0 obtains an id, every rank (its context's device current or not)
 * joins with the same id.  Collective: returns when all `world` ranks have called it. */
int acmpc_comm_id(char* id128);
int acmpc_comm_init_rank(acmpc_comm** out, acmpc_ctx* ctx, const char* id128, int rank, int world);

/* One process, n_devices GPUs: out[g] becomes the rank of ctxs[g] (contexts on distinct devices). */
int acmpc_comm_init_all(acmpc_comm** out, acmpc_ctx* const* ctxs, int n_devices);
void acmpc_comm_destroy(acmpc_comm* comm);

/* The exchange step for one rank: d_values[count] (device, int64) becomes the element-wise sum over all
 * ranks, in place, enqueued on hip_stream.  With one process per GPU every process calls it once per
 * step; a single process that drives several ranks must use acmpc_merge_group_device instead (the
 * ranks' calls have to be issued together). */
int acmpc_merge_device(acmpc_ctx* ctx, acmpc_comm* comm, int64_t* d_values, int64_t count, void* hip_stream);
int acmpc_merge_group_device(acmpc_ctx* const* ctxs, acmpc_comm* const* comms, int n_devices,
                             int64_t* const* d_values, int64_t count, void* const* hip_streams);

/*
 * Where acmpc_aggregate_sharded cuts.  Host only: no GPU and no context are needed, and nothing is
 * allocated.  No counterpart in the reference.  o_cut[n_devices + 1] and l_cut[n_devices + 1] receive
 * the shards: shard g owns the order indices [o_cut[g], o_cut[g + 1]) = [floor(g*M/G), floor((g+1)*M/G))
 * and the lines [l_cut[g], l_cut[g + 1]).
 *   line_order ACMPC_ORDER_GROUPED: l_cut[g] is the first line whose id reaches order_id_base + o_cut[g]
 *     (a binary search of the ascending column).  Lines whose id lies below the base stay with the first
 *     shard that owns an order, lines past base + M with the last one; the device pass there reports
 *     them.  The cuts never descend, so that every line belongs to exactly one shard even when the
 *     column is not sorted after all; such a column is then reported by a shard's device pass
 *     (ACMPC_ST_ORDER_UNSORTED or ACMPC_ST_ORDER_RANGE), never mis-sized.
 *   line_order ACMPC_ORDER_ANY / ACMPC_ORDER_AUTO: a column that is sorted is cut as above; otherwise
 *     shard g's lines lie anywhere, and l_cut are their places after acmpc_shard_partition has brought
 *     the rows together shard by shard.
 * A shard without orders (n_devices > M) gets no lines.  n_lines > 0 with n_orders == 0 is
 * ACMPC_E_INVALID, as in acmpc_aggregate.  Errors are reported through acmpc_last_error(NULL).
 */
int acmpc_shard_plan(const int32_t* order_id, int64_t n_lines, int64_t n_orders, int32_t order_id_base,
                     int n_devices, int line_order, int64_t* o_cut, int64_t* l_cut);

/*
 * Host only, like acmpc_shard_plan: the stable partition of the three line columns by shard, for rows in
 * any order.  out_*[n_lines] receive the rows of shard 0, then shard 1, ... each shard's rows in the order
 * in which they stood; l_cut[n_devices + 1] the places where the shards begin.  No counterpart in the
 * reference (concurrent checkouts interleave their rows: process/confirmcompra.php:27,35-37).
 */
int acmpc_shard_partition(const int32_t* order_id, const int32_t* product_idx, const int32_t* qty,
                          int64_t n_lines, int64_t n_orders, int32_t order_id_base, int n_devices,
                          int32_t* out_order_id, int32_t* out_product_idx, int32_t* out_qty, int64_t* l_cut);

/*
 * acmpc_aggregate over n_devices GPUs from one process: cuts the HOST columns on order boundaries
 * (acmpc_shard_plan), copies every shard to its device, runs the device path there, merges the product
 * and customer vectors with the one all-reduce, and copies the results back (order totals from their
 * shards).  Every context must hold the same product table (acmpc_set_products on each) and the same
 * ACMPC_OPT_LINE_ORDER (ACMPC_E_STATE otherwise).  With ACMPC_ORDER_ANY or ACMPC_ORDER_AUTO and more
 * than one device, rows that are not sorted are first brought together shard by shard in host memory
 * (acmpc_shard_partition: one counting pass and one scatter over the columns, 12 bytes per line of
 * temporary host memory).  Same arguments, results and error behaviour as acmpc_aggregate — an unsorted
 * column under ACMPC_ORDER_GROUPED is ACMPC_E_DATA here too; with n_devices == 1 the results equal
 * acmpc_aggregate's bit for bit (as they do for any n_devices).
 * Replaces the loops of process/carrito.php:12-18 and process/confirmcompra.php:24,35-37 like
 * acmpc_aggregate, for all carts at once and all GPUs of the node.
 */
int acmpc_aggregate_sharded(acmpc_ctx* const* ctxs, acmpc_comm* const* comms, int n_devices,
                            const int32_t* order_id, const int32_t* product_idx, const int32_t* qty,
                            int64_t n_lines,
                            const int32_t* order_customer, int64_t n_orders,
                            int32_t order_id_base, int32_t n_customers,
                            int64_t* out_order_total_cents,
                            int64_t* out_product_qty,
                            int64_t* out_customer_total_cents);

/* ------------------------------------------------------------------------
 * Text ingest: `detalle` rows as SQL-dump / CSV text -> the three int32 columns.
 *
 * Accepted text (the shapes the reference itself writes):
 *   - the statement process/confirmcompra.php:36 sends through library/consulSQL.php:29-30, one
 *     statement per line (a query log of checkouts):
 *       insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('5', '1', '1')
 *   - the phpMyAdmin dump style of store.sql:88-89,125-128: an "INSERT INTO `t` (...) VALUES" header
 *     line followed by one "(5, '1', 1)," row per line, the last ending in ";"
 *   - plain comma- or tab-separated rows, one per line
 * Grammar.  A line starts at offset 0 or after '\n'.  It holds a DATA row iff its first byte is a
 * digit, '-' followed by a digit, '(' or a quote (the row starts with the line), or it is a
 * STATEMENT line: its first six bytes are "insert" in any letter case and, at most 100 bytes from
 * the line start and before the '\n', there is a '(' directly preceded (spaces allowed between) by
 * "values" in any letter case; the row then starts at the first such '('.  Any other line (the INSERT
 * header of the dump style, "--" comment, blank) is skipped.
 *   row  := ['('] sp int sp sep sp code sp sep sp int sp [')'] [','|';'] ['\r'] ('\n' | end of text)
 *   int  := [q] ['-'] digit{1,10} [q]        must fit int32; q is ' or " and must match
 *   code := q bytes q  (no q, no '\n' inside; a doubled q is not accepted)
 *         | bytes up to one of , \t space \r \n )           0..30 bytes (varchar(30), store.sql:99)
 *   sep  := ',' | '\t'      sp := ' '*      a row is at most 120 bytes from its first byte to its '\n'
 * A data row that does not match yields (0, -1, 0), is counted, and raises
 * ACMPC_ST_INGEST_MALFORMED.  CodigoProd is matched byte for byte against the code list; an
 * unknown code becomes product_idx -1 (it adds nothing: process/carrito.php:13-14).
 * Rows keep their order in the text.  Several rows on one line (mysqldump's extended inserts) are
 * not accepted: a statement line carries one row, and a second row behind it makes that row
 * malformed (phpMyAdmin and mysqldump --skip-extended-insert write one row per line).
 * ---------------------------------------------------------------------- */

/* Product code list: n_products entries of 32 bytes each, NUL padded, codes of at most 30 bytes
 * (producto.CodigoProd, varchar(30) and primary key: store.sql:110,200), in the same order as the
 * price table of acmpc_set_products.  Host pointer.  Replaces the match on CodigoProd that the
 * reference leaves to MariaDB (process/carrito.php:13). */
int acmpc_set_product_codes(acmpc_ctx* ctx, const char* codes32, int32_t n_products);

/* Reads back what consultasSQL::InsertSQL("detalle", ...) wrote (process/confirmcompra.php:36 through
 * library/consulSQL.php:29-30; the reference itself has no reader of `detalle`).
 * Parse n_bytes of DEVICE text (16-byte aligned) into DEVICE columns of capacity_rows entries.
 * *d_n_rows (device int64) receives the number of data rows found (rows beyond capacity_rows are
 * counted but not stored, and raise ACMPC_ST_INGEST_CAPACITY).  Enqueued on hip_stream. */
int acmpc_ingest_detalle_device(acmpc_ctx* ctx, const char* d_text, int64_t n_bytes,
                                int32_t* d_order_id, int32_t* d_product_idx, int32_t* d_qty,
                                int64_t capacity_rows, int64_t* d_n_rows, void* hip_stream);

/* Same from HOST text into HOST columns; waits, and returns ACMPC_E_DATA if a row was malformed
 * or capacity_rows was too small.  n_malformed may be NULL. */
int acmpc_ingest_detalle(acmpc_ctx* ctx, const char* text, int64_t n_bytes,
                         int32_t* order_id, int32_t* product_idx, int32_t* qty,
                         int64_t capacity_rows, int64_t* n_rows, int64_t* n_malformed);

/* ------------------------------------------------------------------------
 * Text emit: results -> the statements that would store them back in the shop's tables.
 *
 * The reference keeps `venta.TotalPagar` (DECIMAL(30,2), store.sql:163, written once by
 * process/confirmcompra.php:24) and `producto.Stock` (INT, store.sql:116, never decremented).
 * Every statement has the shape consultasSQL::UpdateSQL builds, library/consulSQL.php:43-44
 * ("update $tabla set $campos where $condicion"), with values quoted the way
 * process/updatePedido.php:11 quotes them; one row per line, rows in index order.
 *
 *   totals, ACMPC_EMIT_SQL   update venta set TotalPagar='2200.00' where NumPedido='5';\n
 *   totals, ACMPC_EMIT_CSV   5,2200.00\n
 *   totals, ACMPC_EMIT_DISPLAY   <tr><td>Subtotal</td><td>$2,200.00</td></tr>\n
 *                            the fragment process/carrito.php:19 echoes for a cart: number_format($suma, 2),
 *                            i.e. a comma between groups of three digits and '.' before the cents; one
 *                            line per order, in order-index order (totals only)
 *   stock,  ACMPC_EMIT_SQL   update producto set Stock=Stock-3 where CodigoProd='1';\n
 *                            (Stock=Stock+3 when the delta is negative)
 *   stock,  ACMPC_EMIT_CSV   1,3\n                    (code, signed delta)
 *
 * An amount is integer cents written as [-]units.cc, the text MariaDB prints for a DECIMAL(30,2).
 * Every order gets a totals row (order id = order_id_base + index).  A product whose delta is zero
 * gets no stock row.  Stock rows need the code list of acmpc_set_product_codes; a code that the
 * style cannot carry unescaped (SQL: ' or \ ; CSV: , tab space CR LF ) or a leading quote, or an
 * empty code) makes the call fail with ACMPC_E_DATA before anything is written.
 * Upper bounds for sizing: 83 bytes per totals row, 103 per stock row.
 * ---------------------------------------------------------------------- */
#define ACMPC_EMIT_SQL 0
#define ACMPC_EMIT_CSV 1
#define ACMPC_EMIT_DISPLAY 2
#define ACMPC_ST_EMIT_CAPACITY 32u /* acmpc_emit_*: the text needs more than capacity_bytes */

/* Write the audit rows of n_orders DEVICE totals into DEVICE text (16-byte aligned) of
 * capacity_bytes.  *d_n_bytes (device int64) receives the bytes the full text needs; when that
 * exceeds capacity_bytes only the first capacity_bytes are written and ACMPC_ST_EMIT_CAPACITY is
 * raised.  Enqueued on hip_stream. */
int acmpc_emit_totals_device(acmpc_ctx* ctx, const int64_t* d_order_total_cents, int64_t n_orders,
                             int32_t order_id_base, int style, char* d_text, int64_t capacity_bytes,
                             int64_t* d_n_bytes, void* hip_stream);

/* Write the stock rows of the DEVICE deltas product_qty[n_products] (n_products must equal the
 * length of the code list). */
int acmpc_emit_stock_device(acmpc_ctx* ctx, const int64_t* d_product_qty, int32_t n_products,
                            int style, char* d_text, int64_t capacity_bytes, int64_t* d_n_bytes,
                            void* hip_stream);

/* Same from HOST results into HOST text; they wait, and return ACMPC_E_DATA when capacity_bytes
 * was too small (*n_bytes then still holds the size needed). */
int acmpc_emit_totals(acmpc_ctx* ctx, const int64_t* order_total_cents, int64_t n_orders,
                      int32_t order_id_base, int style, char* text, int64_t capacity_bytes,
                      int64_t* n_bytes);
int acmpc_emit_stock(acmpc_ctx* ctx, const int64_t* product_qty, int32_t n_products, int style,
                     char* text, int64_t capacity_bytes, int64_t* n_bytes);

/* ------------------------------------------------------------------------
 * Synthetic store.sql-schema data, integer-only and reproducible bit for bit
 * on host and device.  These have no counterpart in the reference (which ships
 * three products and one customer, store.sql:89,126-128); they exist so that
 * benchmarks can build their input directly in HBM.
 *
 *   h(seed, i)          = splitmix64 step applied to (seed XOR i*0x9E3779B97F4A7C15)
 *   lines of order o    = 1 + h(0xA11CE, o) mod 7; orders are laid out in
 *                         ascending order id; the last order is truncated so
 *                         that exactly n_lines_total lines exist
 *   order_id            = order_id_base + o
 *   product_idx[l]      = h(0xB0B, l) mod P                      (zipf == 0)
 *                         upper_bound(cdf, h(0xB0B, l) mod W)    (zipf != 0),
 *                         cdf[k] = sum_{j<=k} floor(2^32/(j+1)), W = cdf[P-1]
 *   qty[l]              = 1 (qty_mode 0) or 1 + h(0x51DE, l) mod 5 (qty_mode 1)
 *   price_cents[p]      = 1 + h(0xCAFE, p) mod 500000, p = 0,1,2 overridden by
 *                         the seed rows 25000, 45000, 150000 (store.sql:126-128)
 *   order_customer[o]   = h(0xD00D, o) mod C
 * l and o are GLOBAL indices, so a shard holds exactly the slice of the
 * single-device data set that it owns.
 * ---------------------------------------------------------------------- */
typedef struct acmpc_synth_cfg {
  int64_t n_lines_total;   /* lines in the whole data set                  */
  int32_t n_products;      /* P                                            */
  int32_t n_customers;     /* C                                            */
  int32_t order_id_base;   /* 5 = venta AUTO_INCREMENT start, store.sql:232 */
  int32_t zipf;            /* 0 uniform product popularity, 1 Zipf(s=1)    */
  int32_t qty_mode;        /* 0: always 1; 1: 1..5                          */
  int32_t rank;            /* shard index in [0, world)                    */
  int32_t world;           /* number of shards (order-id ranges)           */
} acmpc_synth_cfg;

typedef struct acmpc_synth_shard {
  int64_t n_orders_total;  /* M of the whole data set                       */
  int64_t order_lo;        /* first global order index of this shard        */
  int64_t order_hi;        /* one past the last                             */
  int64_t line_lo;         /* first global line index of this shard         */
  int64_t line_hi;         /* one past the last                             */
} acmpc_synth_shard;

/* Host only: where shard cfg->rank of cfg->world begins and ends.  Shards are
 * contiguous order-index ranges [floor(g*M/G), floor((g+1)*M/G)), cut on order
 * boundaries so that no order straddles two shards. */
int acmpc_synth_plan(const acmpc_synth_cfg* cfg, acmpc_synth_shard* out);

/* Host only: fill price_cents[cfg->n_products]. */
int acmpc_synth_prices(const acmpc_synth_cfg* cfg, int64_t* price_cents);

/* Fill DEVICE arrays with this shard's columns: d_order_id, d_product_idx, d_qty
 * hold (line_hi - line_lo) entries, d_order_customer holds (order_hi - order_lo).
 * Enqueued on hip_stream; returns after the work is queued. */
int acmpc_synth_generate_device(acmpc_ctx* ctx, const acmpc_synth_cfg* cfg,
                                int32_t* d_order_id, int32_t* d_product_idx, int32_t* d_qty,
                                int32_t* d_order_customer, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* ACMPC_H */
