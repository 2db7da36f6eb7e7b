"""Batch backend of the query layer: the host-side mirror of what library/consulSQL.php gains.

The reference funnels every statement through two static classes, ``ejecutarSQL`` and
``consultasSQL`` (library/consulSQL.php:3-50), and ends the request with ``die("...")`` when a
statement fails (library/consulSQL.php:12,19,31,38,45).  The checkout walks one cart at a time
through them (process/carrito.php:12-18, process/confirmcompra.php:24,35-37).  This module keeps
that shape — a static class, ``consultasLote``, sitting beside the other two — but its one
operation takes whole columns of `detalle` / `venta` rows and runs on the GPU through the C ABI of
include/acmpc.h.  PHP is not installed in the build image, so the host layer is Python; the PHP
shim a maintainer would add is ac_mpc_amd/php/consulSQL_lote.php.

``MotorLote`` is the handle (one HIP context + workspace); ``consultasLote`` is the static facade.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib


class ErrorLote(RuntimeError):
    """Raised where the reference would ``die("...")`` (library/consulSQL.php:19,31)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"Problemas al agregar el lote: {message} (codigo {code})")
        self.code = code
        self.detail = message


def _np(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class MotorLote:
    """One engine context on one GPU.  Not thread-safe; one per worker process."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.acmpc_init(C.byref(self._ctx), int(device))
        if rc != 0:
            raise ErrorLote(rc, self._lib.acmpc_last_error(None).decode())
        self.device = int(device)
        self.n_products = 0
        self._host_blocks: Dict[int, int] = {}     # address of a page-locked array -> its block

    # -- lifecycle ------------------------------------------------------------------------
    def cerrar(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.acmpc_destroy(self._ctx)          # also releases the page-locked blocks
            self._ctx = C.c_void_p()
            self._host_blocks = {}

    close = cerrar

    def __del__(self):
        try:
            self.cerrar()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise ErrorLote(rc, self._lib.acmpc_last_error(self._ctx).decode())

    # -- product table (producto.Precio in cents, store.sql:113) --------------------------------
    def cargar_productos(self, precio_centavos) -> None:
        price = _np(precio_centavos, np.int64)
        if price.ndim != 1 or price.size == 0:
            raise ErrorLote(_lib.E_INVALID, "la tabla de productos esta vacia")
        self._check(self._lib.acmpc_set_products(self._ctx, price.ctypes.data, price.size))
        self.n_products = int(price.size)

    # -- page-locked host memory ---------------------------------------------------------------
    def reservar_host(self, n: int, dtype) -> np.ndarray:
        """A numpy array of `n` elements in page-locked memory owned by this context: columns filled in
        place there (and results received there, see `agregar(salida=...)`) cross to the GPU by direct
        DMA.  The array is valid until `liberar_host(array)` or `cerrar()`; using it afterwards is a
        use-after-free, as with any buffer of a C library."""
        dt = np.dtype(dtype)
        nbytes = max(1, int(n) * dt.itemsize)
        p = C.c_void_p()
        self._check(self._lib.acmpc_host_alloc(self._ctx, nbytes, C.byref(p)))
        arr = np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dt, count=int(n))
        self._host_blocks[p.value] = p.value
        return arr

    def liberar_host(self, arr: np.ndarray) -> None:
        addr = arr.ctypes.data if arr.size else None
        block = self._host_blocks.pop(addr, None) if addr is not None else None
        if block is None:
            raise ErrorLote(_lib.E_INVALID, "el arreglo no viene de reservar_host (o esta vacio)")
        self._check(self._lib.acmpc_host_free(self._ctx, block))

    # -- host columns in, host results out ----------------------------------------------------
    def agregar(self, order_id, product_idx, qty, order_customer, order_id_base: int, n_customers: int,
                precio_linea=None, estado_pedido=None, descuento=None, descuento_porcentaje: bool = False,
                salida: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None
                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Return (order_total_cents[M], product_qty[P], customer_total_cents[C]).

        `order_id`, `product_idx`, `qty` are the `detalle` columns (store.sql:97-101) with the rows of
        each order contiguous and order ids non-decreasing; `order_customer[i]` is the customer index of
        order ``order_id_base + i``.

        Order lifecycle (all optional): `precio_linea[N]` int64 cents, the price each line was sold at
        (process/updateProduct.php:15 changes Precio later); `estado_pedido[M]` uint8, bit 0 set when the
        order counts (cleared for an Estado that means cancelled, process/updatePedido.php:11, or a
        deleted order, process/delPedido.php:12-13); `descuento[M]` int32, venta.Descuento (store.sql:162)
        in whole currency units, or per cent with `descuento_porcentaje`.

        `salida`: three int64 arrays (M, P, C) to receive the results instead of new ones, for instance
        arrays from `reservar_host`.
        """
        oid, pidx, q = _np(order_id, np.int32), _np(product_idx, np.int32), _np(qty, np.int32)
        cust = _np(order_customer, np.int32)
        if not (oid.shape == pidx.shape == q.shape) or oid.ndim != 1 or cust.ndim != 1:
            raise ErrorLote(_lib.E_INVALID, "las columnas de detalle no tienen la misma longitud")
        if self.n_products <= 0:
            raise ErrorLote(_lib.E_STATE, "no se han cargado productos")
        n, m = oid.size, cust.size
        if salida is not None:
            tot, pq, ct = salida
            for arr, size in ((tot, m), (pq, self.n_products), (ct, int(n_customers))):
                if not (isinstance(arr, np.ndarray) and arr.dtype == np.int64 and arr.ndim == 1 and arr.size == size
                        and arr.flags.c_contiguous and arr.flags.writeable):
                    raise ErrorLote(_lib.E_INVALID, "arreglo de salida de tipo o longitud incorrectos")
        else:
            tot = np.empty(m, dtype=np.int64)
            pq = np.empty(self.n_products, dtype=np.int64)
            ct = np.empty(int(n_customers), dtype=np.int64)
        keep = []
        life = None
        if precio_linea is not None or estado_pedido is not None or descuento is not None:
            life = _lib.Lifecycle(None, None, None, _lib.DESCUENTO_PERCENT if descuento_porcentaje else _lib.DESCUENTO_UNITS)
            for name, col, dt, size in (("line_price_cents", precio_linea, np.int64, n),
                                        ("order_state", estado_pedido, np.uint8, m),
                                        ("order_descuento", descuento, np.int32, m)):
                if col is None:
                    continue
                arr = _np(col, dt)
                if arr.ndim != 1 or arr.size != size:
                    raise ErrorLote(_lib.E_INVALID, f"columna {name} de longitud incorrecta")
                keep.append(arr)
                setattr(life, name, arr.ctypes.data if arr.size else None)
        self._check(self._lib.acmpc_aggregate_lifecycle(
            self._ctx, oid.ctypes.data, pidx.ctypes.data, q.ctypes.data, n, cust.ctypes.data, m,
            int(order_id_base), int(n_customers), C.byref(life) if life is not None else None,
            tot.ctypes.data, pq.ctypes.data, ct.ctypes.data))
        return tot, pq, ct

    # -- device-resident columns (torch tensors on this context's GPU) --------------------------
    def agregar_device(self, order_id, product_idx, qty, order_customer, order_id_base: int, n_customers: int,
                       out_order_total, out_product_qty, out_customer_total, stream: Optional[int] = None,
                       precio_linea=None, estado_pedido=None, descuento=None, descuento_porcentaje: bool = False,
                       sin_reservas: bool = False) -> None:
        """Enqueue the aggregation of device-resident columns; outputs are torch int64 CUDA tensors.
        The lifecycle columns (see `agregar`) are torch CUDA tensors of int64 / uint8 / int32.
        sin_reservas: go through acmpc_aggregate_device_noalloc, which allocates nothing (fixed workspace
        inside the library, shared by every call on the GPU; grouped rows, no lifecycle columns, sizes
        within that entry point's limits).

        Returns without waiting.  Call `estado()` to wait and check the data contract.
        """
        import torch
        n, m = order_id.numel(), order_customer.numel()
        cols = [(order_id, torch.int32, n), (product_idx, torch.int32, n), (qty, torch.int32, n),
                (order_customer, torch.int32, m), (out_order_total, torch.int64, m),
                (out_product_qty, torch.int64, self.n_products), (out_customer_total, torch.int64, int(n_customers))]
        life = None
        if precio_linea is not None or estado_pedido is not None or descuento is not None:
            life = _lib.Lifecycle(None, None, None, _lib.DESCUENTO_PERCENT if descuento_porcentaje else _lib.DESCUENTO_UNITS)
            for name, t, dt, size in (("line_price_cents", precio_linea, torch.int64, n),
                                      ("order_state", estado_pedido, torch.uint8, m),
                                      ("order_descuento", descuento, torch.int32, m)):
                if t is not None:
                    cols.append((t, dt, size))
                    setattr(life, name, t.data_ptr() if size else None)
        for t, dt, size in cols:
            if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.device.index == self.device):
                raise ErrorLote(_lib.E_INVALID, "tensor de tipo, dispositivo o disposicion incorrectos")
            if t.numel() != size:
                raise ErrorLote(_lib.E_INVALID, "longitudes de columnas o resultados incoherentes")
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if sin_reservas:
            if life is not None:
                raise ErrorLote(_lib.E_INVALID, "sin_reservas no admite columnas de ciclo de vida")
            self._check(self._lib.acmpc_aggregate_device_noalloc(
                self._ctx, order_id.data_ptr(), product_idx.data_ptr(), qty.data_ptr(), n,
                order_customer.data_ptr(), m, int(order_id_base), int(n_customers),
                out_order_total.data_ptr(), out_product_qty.data_ptr(), out_customer_total.data_ptr(), stream))
            return
        self._check(self._lib.acmpc_aggregate_lifecycle_device(
            self._ctx, order_id.data_ptr(), product_idx.data_ptr(), qty.data_ptr(), n,
            order_customer.data_ptr(), m, int(order_id_base), int(n_customers),
            C.byref(life) if life is not None else None,
            out_order_total.data_ptr(), out_product_qty.data_ptr(), out_customer_total.data_ptr(), stream))

    def estado(self, stream: Optional[int] = None) -> int:
        """Wait for the stream; raise ErrorLote if a device call broke the data contract."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device).cuda_stream
        flags = C.c_uint32(0)
        self._check(self._lib.acmpc_status(self._ctx, stream, C.byref(flags)))
        return int(flags.value)

    # -- options ------------------------------------------------------------------------------
    def orden_lineas(self, cualquiera) -> None:
        """False (default): the `detalle` rows of an order are contiguous and order ids ascend, as
        process/confirmcompra.php:35-37 writes them.  True: no promise (concurrent checkouts interleave
        rows); same results through a slower atomic path.  "auto": the grouped pass first, and the
        atomic path only if it met a descending order id (decided on the GPU)."""
        modo = _lib.ORDER_AUTO if cualquiera == "auto" else _lib.ORDER_ANY if cualquiera else _lib.ORDER_GROUPED
        self._check(self._lib.acmpc_set_option(self._ctx, _lib.OPT_LINE_ORDER, modo))

    def camino_lineas(self, stream: Optional[int] = None) -> int:
        """Which way the last call summed rows whose order was not promised (`_lib.PATH_*`): the GPU decides,
        so this waits for the stream and reads the answer back."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device).cuda_stream
        path = C.c_int(0)
        self._check(self._lib.acmpc_last_line_order_path(self._ctx, stream, C.byref(path)))
        return int(path.value)

    # -- kernel timing --------------------------------------------------------------------------
    def perfilar(self, activo: bool) -> None:
        self._check(self._lib.acmpc_set_profiling(self._ctx, 1 if activo else 0))

    def ultimos_ms(self) -> Dict[str, float]:
        ms = (C.c_float * _lib.N_KERNELS)()
        self._check(self._lib.acmpc_last_kernel_ms(self._ctx, ms))
        return {"lines": ms[0], "fixup": ms[1], "customers": ms[2], "custmerge": ms[3]}

    def geometria_lineas(self) -> Tuple[int, int]:
        wg, wv = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.acmpc_lines_geometry(self._ctx, C.byref(wg), C.byref(wv)))
        return int(wg.value), int(wv.value)


class consultasLote:
    """Static facade in the style of ``consultasSQL`` (library/consulSQL.php:27-50)."""

    _motor: Optional[MotorLote] = None
    _codigos: Dict[str, int] = {}

    @staticmethod
    def conectar(device: int = 0) -> MotorLote:
        """Counterpart of ``ejecutarSQL::conectar()`` (library/consulSQL.php:5-16): one long-lived
        context per process instead of one connection per statement."""
        if consultasLote._motor is None:
            consultasLote._motor = MotorLote(device)
        return consultasLote._motor

    @staticmethod
    def desconectar() -> None:
        if consultasLote._motor is not None:
            consultasLote._motor.cerrar()
        consultasLote._motor = None
        consultasLote._codigos = {}

    @staticmethod
    def CargarProductos(codigos: Sequence[str], precios_centavos: Sequence[int]) -> None:
        """`producto.CodigoProd` -> dictionary index, `producto.Precio` -> cents (store.sql:110,113)."""
        if len(codigos) != len(precios_centavos):
            raise ErrorLote(_lib.E_INVALID, "codigos y precios de distinta longitud")
        consultasLote._codigos = {str(c): i for i, c in enumerate(codigos)}
        consultasLote.conectar().cargar_productos(precios_centavos)

    @staticmethod
    def AgregarDetalle(NumPedido, CodigoProd: Sequence[str], CantidadProductos, NIT_por_pedido,
                       primer_pedido: int, n_clientes: int):
        """Aggregate `detalle` rows given by value, CodigoProd still as strings.

        Returns a dict with ``TotalPagar`` (cents per order, what confirmcompra.php:24 stores),
        ``DeltaStock`` (units per product, in the order `CargarProductos` received them) and
        ``TotalCliente`` (cents per customer index).
        Unknown product codes map to -1 and add nothing (process/carrito.php:13-14).
        """
        idx = np.fromiter((consultasLote._codigos.get(str(c), -1) for c in CodigoProd), dtype=np.int32,
                          count=len(CodigoProd))
        tot, pq, ct = consultasLote.conectar().agregar(NumPedido, idx, CantidadProductos, NIT_por_pedido,
                                                       primer_pedido, n_clientes)
        return {"TotalPagar": tot, "DeltaStock": pq, "TotalCliente": ct}


def formato_subtotal(centavos: int) -> str:
    """``"$" . number_format($suma, 2)`` (process/carrito.php:19) from integer cents."""
    sign = "-" if centavos < 0 else ""
    whole, frac = divmod(abs(int(centavos)), 100)
    return f"{sign}${whole:,}.{frac:02d}"
