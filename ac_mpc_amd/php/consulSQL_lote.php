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
    ";

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

    /* producto.Precio in integer cents, in dictionary order (store.sql:110,113). */
    public static function CargarProductos($preciosCentavos) {
        self::conectar();
        $p = self::columna("int64_t", $preciosCentavos);
        self::comprobar(self::$ffi->acmpc_set_products(self::$ctx, $p, count($preciosCentavos)), "cargar productos");
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

    public static function desconectar() {
        if (self::$ctx !== null) self::$ffi->acmpc_destroy(self::$ctx);
        self::$ctx = null;
    }
}
