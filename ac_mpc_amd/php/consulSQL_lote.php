<?php
/*
 * consulSQL_lote.php — the batch class a maintainer would add next to ejecutarSQL / consultasSQL
 * (library/consulSQL.php:3-50 of the reference).  It binds libacmpc.so through PHP FFI.
 *
 * NOT EXECUTED IN THIS REPOSITORY: the build image has no PHP interpreter, so this file has only
 * been checked by eye against include/acmpc.h.  It needs PHP >= 7.4 with ext-ffi (the project
 * already targets 7.4: nbproject/project.properties:2) and ffi.enable=1 for the CLI / worker SAPI.
 *
 * Error convention: the reference ends the request with die("...") when a statement fails
 * (library/consulSQL.php:19,31); the C library returns a negative code and a message, and this
 * shim turns that back into die().
 *
 * Creating a context initialises the HIP runtime (hundreds of milliseconds), so this belongs in a
 * long-lived CLI / queue worker (nightly re-aggregation, audit of venta.TotalPagar against detalle),
 * not in a per-request page such as process/confirmcompra.php.
 */
class consultasLote {

    private static $ffi = null;
    private static $ctx = null;
    private static $nProductos = 0;

    const CDEF = "
        typedef struct acmpc_ctx acmpc_ctx;
        int  acmpc_init(acmpc_ctx** out, int device);
        void acmpc_destroy(acmpc_ctx* ctx);
        const char* acmpc_last_error(const acmpc_ctx* ctx);
        int acmpc_set_products(acmpc_ctx* ctx, const int64_t* price_cents, int32_t n_products);
        int acmpc_aggregate(acmpc_ctx* ctx,
                            const int32_t* order_id, const int32_t* product_idx, const int32_t* qty, int64_t n_lines,
                            const int32_t* order_customer, int64_t n_orders,
                            int32_t order_id_base, int32_t n_customers,
                            int64_t* out_order_total_cents, int64_t* out_product_qty,
                            int64_t* out_customer_total_cents);
        typedef struct acmpc_lifecycle {
            const int64_t* line_price_cents; const uint8_t* order_state; const int32_t* order_descuento;
            int32_t descuento_mode;
        } acmpc_lifecycle;
        int acmpc_aggregate_lifecycle(acmpc_ctx* ctx,
                            const int32_t* order_id, const int32_t* product_idx, const int32_t* qty, int64_t n_lines,
                            const int32_t* order_customer, int64_t n_orders,
                            int32_t order_id_base, int32_t n_customers, const acmpc_lifecycle* life,
                            int64_t* out_order_total_cents, int64_t* out_product_qty,
                            int64_t* out_customer_total_cents);
        int acmpc_set_product_codes(acmpc_ctx* ctx, const char* codes32, int32_t n_products);
        int acmpc_ingest_detalle(acmpc_ctx* ctx, const char* text, int64_t n_bytes,
                            int32_t* order_id, int32_t* product_idx, int32_t* qty,
                            int64_t capacity_rows, int64_t* n_rows, int64_t* n_malformed);
        int acmpc_emit_totals(acmpc_ctx* ctx, const int64_t* order_total_cents, int64_t n_orders,
                            int32_t order_id_base, int style, char* text, int64_t capacity_bytes, int64_t* n_bytes);
        int acmpc_emit_stock(acmpc_ctx* ctx, const int64_t* product_qty, int32_t n_products, int style,
                            char* text, int64_t capacity_bytes, int64_t* n_bytes);
        int acmpc_host_alloc(acmpc_ctx* ctx, int64_t n_bytes, void** out);
        int acmpc_host_free(acmpc_ctx* ctx, void* p);
        typedef struct acmpc_comm acmpc_comm;
        int acmpc_comm_init_all(acmpc_comm** out, acmpc_ctx* const* ctxs, int n_devices);
        void acmpc_comm_destroy(acmpc_comm* comm);
        int acmpc_aggregate_sharded(acmpc_ctx* const* ctxs, acmpc_comm* const* comms, int n_devices,
                            const int32_t* order_id, const int32_t* product_idx, const int32_t* qty, int64_t n_lines,
                            const int32_t* order_customer, int64_t n_orders,
                            int32_t order_id_base, int32_t n_customers,
                            int64_t* out_order_total_cents, int64_t* out_product_qty,
                            int64_t* out_customer_total_cents);
        int acmpc_set_option(acmpc_ctx* ctx, int option, int64_t value);
        int acmpc_shard_plan(const int32_t* order_id, int64_t n_lines, int64_t n_orders, int32_t order_id_base,
                            int n_devices, int line_order, int64_t* o_cut, int64_t* l_cut);
    ";

    /* Every GPU of the node: contexts [0] = the one of conectar(), communicator ranks beside them. */
    private static $ctxs = null;
    private static $coms = null;
    private static $nGpus = 1;

    /* Counterpart of ejecutarSQL::conectar() (library/consulSQL.php:5-16). */
    public static function conectar($lib = "libacmpc.so", $device = 0) {
        if (self::$ctx !== null) return self::$ctx;
        self::$ffi = FFI::cdef(self::CDEF, $lib);
        $out = self::$ffi->new("acmpc_ctx*");
        $rc = self::$ffi->acmpc_init(FFI::addr($out), $device);
        if ($rc != 0) die("Error en el servidor, verifique sus datos: " . FFI::string(self::$ffi->acmpc_last_error(null)));
        self::$ctx = $out;
        return self::$ctx;
    }

    private static function comprobar($rc, $donde) {
        if ($rc != 0) die("Problemas al " . $donde . ": " . FFI::string(self::$ffi->acmpc_last_error(self::$ctx)));
    }

    private static function columna($tipo, $valores) {
        $n = count($valores);
        $col = self::$ffi->new($tipo . "[" . max($n, 1) . "]");
        for ($i = 0; $i < $n; $i++) $col[$i] = $valores[$i];
        return $col;
    }

    /*
     * A column of $n elements of $tipo ("int32_t", "int64_t") in page-locked memory of the context: a
     * worker that fills such columns in place (instead of PHP arrays, as $_SESSION['producto'] is at
     * process/carrito.php:7) and passes them to acmpc_aggregate gets its copies at the PCIe rate.
     * Index it like an FFI array; release it with LiberarColumna.
     */
    public static function ReservarColumna($tipo, $n) {
        self::conectar();
        $p = self::$ffi->new("void*");
        self::comprobar(self::$ffi->acmpc_host_alloc(self::$ctx, max($n, 1) * FFI::sizeof(self::$ffi->type($tipo)),
                                                     FFI::addr($p)), "reservar memoria");
        return self::$ffi->cast($tipo . "*", $p);
    }

    public static function LiberarColumna($col) {
        self::comprobar(self::$ffi->acmpc_host_free(self::$ctx, self::$ffi->cast("void*", $col)), "liberar memoria");
    }

    /* producto.Precio in integer cents, in dictionary order (store.sql:110,113). */
    public static function CargarProductos($preciosCentavos) {
        self::conectar();
        $p = self::columna("int64_t", $preciosCentavos);
        self::comprobar(self::$ffi->acmpc_set_products(self::$ctx, $p, count($preciosCentavos)), "cargar productos");
        for ($g = 1; self::$ctxs !== null && $g < self::$nGpus; $g++)      // every GPU holds the same table
            self::comprobar(self::$ffi->acmpc_set_products(self::$ctxs[$g], $p, count($preciosCentavos)), "cargar productos");
        self::$nProductos = count($preciosCentavos);
    }

    /*
     * $NumPedido, $IndiceProd, $Cantidad: the detalle columns (store.sql:97-101), rows of one order
     * contiguous and NumPedido ascending; $ClientePorPedido[i] is the customer index of order
     * $primerPedido + i.  Returns [TotalPagar[], DeltaStock[], TotalCliente[]] in integer cents / units:
     * what process/carrito.php:12-21 and process/confirmcompra.php:24,35-37 produce one cart at a time.
     */
    public static function AgregarDetalle($NumPedido, $IndiceProd, $Cantidad, $ClientePorPedido, $primerPedido, $nClientes) {
        self::conectar();
        $n = count($NumPedido);
        $m = count($ClientePorPedido);
        $oid = self::columna("int32_t", $NumPedido);
        $pid = self::columna("int32_t", $IndiceProd);
        $qty = self::columna("int32_t", $Cantidad);
        $cus = self::columna("int32_t", $ClientePorPedido);
        $tot = self::$ffi->new("int64_t[" . max($m, 1) . "]");
        $pq  = self::$ffi->new("int64_t[" . max(self::$nProductos, 1) . "]");
        $ct  = self::$ffi->new("int64_t[" . max($nClientes, 1) . "]");
        self::comprobar(self::$ffi->acmpc_aggregate(self::$ctx, $oid, $pid, $qty, $n, $cus, $m,
                                                    $primerPedido, $nClientes, $tot, $pq, $ct), "agregar el lote");
        $TotalPagar = []; $DeltaStock = []; $TotalCliente = [];
        for ($i = 0; $i < $m; $i++) $TotalPagar[] = $tot[$i];
        for ($i = 0; $i < self::$nProductos; $i++) $DeltaStock[] = $pq[$i];
        for ($i = 0; $i < $nClientes; $i++) $TotalCliente[] = $ct[$i];
        return [$TotalPagar, $DeltaStock, $TotalCliente];
    }

    /*
     * One context and one RCCL communicator rank per GPU, all in this process: where the reference
     * opens one connection per statement (ejecutarSQL::conectar, library/consulSQL.php:5-16), a worker
     * opens the node's GPUs once.  Call before CargarProductos (the price table goes to every GPU).
     */
    public static function conectarTodas($nGpus, $lib = "libacmpc.so") {
        self::conectar($lib, 0);
        self::$ctxs = self::$ffi->new("acmpc_ctx*[$nGpus]");
        self::$coms = self::$ffi->new("acmpc_comm*[$nGpus]");
        self::$ctxs[0] = self::$ctx;
        for ($g = 1; $g < $nGpus; $g++) {
            $out = self::$ffi->new("acmpc_ctx*");
            $rc = self::$ffi->acmpc_init(FFI::addr($out), $g);
            if ($rc != 0) die("Error en el servidor, verifique sus datos: " . FFI::string(self::$ffi->acmpc_last_error(null)));
            self::$ctxs[$g] = $out;
        }
        self::comprobar(self::$ffi->acmpc_comm_init_all(self::$coms, self::$ctxs, $nGpus), "abrir el comunicador");
        self::$nGpus = $nGpus;
    }

    /*
     * AgregarDetalle over every GPU opened by conectarTodas: orders are cut into contiguous NumPedido
     * ranges, one per GPU, and the stock deltas and customer totals are merged by one all-reduce.
     * Same arguments and results as AgregarDetalle.
     */
    public static function AgregarDetalleRepartido($NumPedido, $IndiceProd, $Cantidad, $ClientePorPedido, $primerPedido, $nClientes) {
        if (self::$ctxs === null) die("Problemas al agregar el lote: conectarTodas no se ha llamado");
        $n = count($NumPedido);
        $m = count($ClientePorPedido);
        $oid = self::columna("int32_t", $NumPedido);
        $pid = self::columna("int32_t", $IndiceProd);
        $qty = self::columna("int32_t", $Cantidad);
        $cus = self::columna("int32_t", $ClientePorPedido);
        $tot = self::$ffi->new("int64_t[" . max($m, 1) . "]");
        $pq  = self::$ffi->new("int64_t[" . max(self::$nProductos, 1) . "]");
        $ct  = self::$ffi->new("int64_t[" . max($nClientes, 1) . "]");
        self::comprobar(self::$ffi->acmpc_aggregate_sharded(self::$ctxs, self::$coms, self::$nGpus, $oid, $pid, $qty, $n,
                                                            $cus, $m, $primerPedido, $nClientes, $tot, $pq, $ct),
                        "agregar el lote repartido");
        $TotalPagar = []; $DeltaStock = []; $TotalCliente = [];
        for ($i = 0; $i < $m; $i++) $TotalPagar[] = $tot[$i];
        for ($i = 0; $i < self::$nProductos; $i++) $DeltaStock[] = $pq[$i];
        for ($i = 0; $i < $nClientes; $i++) $TotalCliente[] = $ct[$i];
        return [$TotalPagar, $DeltaStock, $TotalCliente];
    }

    /*
     * What the caller promises about the detalle rows, on every GPU that is open: 0 = rows of one order
     * contiguous and NumPedido ascending (the order of process/confirmcompra.php:35-37 when checkouts do not
     * overlap; the default), 1 = no promise (concurrent checkouts interleave their rows,
     * process/confirmcompra.php:27,35-37), 2 = try the grouped pass first.  Same results in every mode.
     */
    public static function OrdenLineas($modo) {
        self::conectar();
        self::comprobar(self::$ffi->acmpc_set_option(self::$ctx, 1 /* ACMPC_OPT_LINE_ORDER */, $modo), "fijar el orden de las lineas");
        for ($g = 1; self::$ctxs !== null && $g < self::$nGpus; $g++)
            self::comprobar(self::$ffi->acmpc_set_option(self::$ctxs[$g], 1, $modo), "fijar el orden de las lineas");
    }

    /*
     * Where AgregarDetalleRepartido cuts: [[first order, ...], [first line, ...]] of every GPU plus the ends.
     * Host arithmetic only (no GPU work): a worker can size its exports with it.
     */
    public static function PlanReparto($NumPedido, $nPedidos, $primerPedido, $nGpus, $modo = 0) {
        self::conectar();
        $oid = self::columna("int32_t", $NumPedido);
        $oc = self::$ffi->new("int64_t[" . ($nGpus + 1) . "]");
        $lc = self::$ffi->new("int64_t[" . ($nGpus + 1) . "]");
        $rc = self::$ffi->acmpc_shard_plan($oid, count($NumPedido), $nPedidos, $primerPedido, $nGpus, $modo, $oc, $lc);
        if ($rc != 0) die("Problemas al repartir el lote: " . FFI::string(self::$ffi->acmpc_last_error(null)));
        $pedidos = []; $lineas = [];
        for ($g = 0; $g <= $nGpus; $g++) { $pedidos[] = $oc[$g]; $lineas[] = $lc[$g]; }
        return [$pedidos, $lineas];
    }

    /* CodigoProd of every product, in the order of CargarProductos (varchar(30), store.sql:110). */
    public static function CargarCodigos($codigos) {
        self::conectar();
        $p = count($codigos);
        $tabla = self::$ffi->new("char[" . max(32 * $p, 1) . "]");
        FFI::memset($tabla, 0, max(32 * $p, 1));
        for ($i = 0; $i < $p; $i++) {
            $c = (string)$codigos[$i];
            if (strlen($c) > 30) die("Problemas al cargar codigos: CodigoProd de mas de 30 bytes");
            FFI::memcpy(FFI::addr($tabla[32 * $i]), $c, strlen($c));
        }
        self::comprobar(self::$ffi->acmpc_set_product_codes(self::$ctx, $tabla, $p), "cargar codigos");
    }

    /*
     * Text of detalle rows (a dump in the style of store.sql:88-89, the statement of
     * process/confirmcompra.php:36, or comma / tab separated rows) -> [NumPedido[], IndiceProd[], Cantidad[]].
     */
    public static function LeerDetalle($texto) {
        self::conectar();
        $cap = substr_count($texto, "\n") + 1;
        $oid = self::$ffi->new("int32_t[$cap]");
        $pid = self::$ffi->new("int32_t[$cap]");
        $qty = self::$ffi->new("int32_t[$cap]");
        $n = self::$ffi->new("int64_t");
        $malas = self::$ffi->new("int64_t");
        self::comprobar(self::$ffi->acmpc_ingest_detalle(self::$ctx, $texto, strlen($texto), $oid, $pid, $qty, $cap,
                                                         FFI::addr($n), FFI::addr($malas)), "leer detalle");
        $NumPedido = []; $IndiceProd = []; $Cantidad = [];
        for ($i = 0; $i < $n->cdata; $i++) { $NumPedido[] = $oid[$i]; $IndiceProd[] = $pid[$i]; $Cantidad[] = $qty[$i]; }
        return [$NumPedido, $IndiceProd, $Cantidad];
    }

    /*
     * The statements consultasSQL::UpdateSQL would send (library/consulSQL.php:43-44), one per line:
     * "update venta set TotalPagar='2200.00' where NumPedido='5';" for every order.
     */
    public static function SentenciasTotalPagar($TotalPagar, $primerPedido) {
        self::conectar();
        $m = count($TotalPagar);
        $tot = self::columna("int64_t", $TotalPagar);
        $cap = max(83 * $m, 1);
        $out = self::$ffi->new("char[$cap]");
        $len = self::$ffi->new("int64_t");
        self::comprobar(self::$ffi->acmpc_emit_totals(self::$ctx, $tot, $m, $primerPedido, 0, $out, $cap, FFI::addr($len)),
                        "escribir TotalPagar");
        return FFI::string($out, $len->cdata);
    }

    /*
     * The line process/carrito.php:19 echoes for a cart, for every order:
     * "<tr><td>Subtotal</td><td>$2,200.00</td></tr>" (number_format($suma, 2) of the total).
     */
    public static function FilasSubtotal($TotalPagar) {
        self::conectar();
        $m = count($TotalPagar);
        $tot = self::columna("int64_t", $TotalPagar);
        $cap = max(70 * $m, 1);
        $out = self::$ffi->new("char[$cap]");
        $len = self::$ffi->new("int64_t");
        self::comprobar(self::$ffi->acmpc_emit_totals(self::$ctx, $tot, $m, 0, 2 /* ACMPC_EMIT_DISPLAY */, $out, $cap, FFI::addr($len)),
                        "escribir subtotales");
        return FFI::string($out, $len->cdata);
    }

    /* "update producto set Stock=Stock-3 where CodigoProd='1';" for every product with a non-zero delta. */
    public static function SentenciasStock($DeltaStock) {
        self::conectar();
        $p = count($DeltaStock);
        $dq = self::columna("int64_t", $DeltaStock);
        $cap = max(103 * $p, 1);
        $out = self::$ffi->new("char[$cap]");
        $len = self::$ffi->new("int64_t");
        self::comprobar(self::$ffi->acmpc_emit_stock(self::$ctx, $dq, $p, 0, $out, $cap, FFI::addr($len)), "escribir Stock");
        return FFI::string($out, $len->cdata);
    }

    public static function desconectar() {
        for ($g = 0; self::$coms !== null && $g < self::$nGpus; $g++) self::$ffi->acmpc_comm_destroy(self::$coms[$g]);
        for ($g = 1; self::$ctxs !== null && $g < self::$nGpus; $g++) self::$ffi->acmpc_destroy(self::$ctxs[$g]);
        self::$coms = null; self::$ctxs = null; self::$nGpus = 1;
        if (self::$ctx !== null) self::$ffi->acmpc_destroy(self::$ctx);
        self::$ctx = null;
    }
}
