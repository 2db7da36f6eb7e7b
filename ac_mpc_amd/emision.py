"""Text emit of the results (wrappers over acmpc_emit_totals* / acmpc_emit_stock*).

Every row has the shape consultasSQL::UpdateSQL builds in the reference (library/consulSQL.php:43-44,
"update $tabla set $campos where $condicion") with values quoted as process/updatePedido.php:11 quotes them:

    update venta set TotalPagar='2200.00' where NumPedido='5';
    update producto set Stock=Stock-3 where CodigoProd='1';

or, in CSV style, `5,2200.00` and `1,3`.  The row formats are specified in include/acmpc.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .lote import ErrorLote, MotorLote

SQL, CSV, VISTA = _lib.EMIT_SQL, _lib.EMIT_CSV, _lib.EMIT_DISPLAY     # VISTA: the subtotal fragment of process/carrito.php:19
MAX_FILA_TOTALES, MAX_FILA_STOCK = 83, 103


def _host(motor: MotorLote, fn, valores: np.ndarray, extra, estilo: int, max_fila: int, salida):
    v = np.ascontiguousarray(valores, np.int64)
    if salida is not None:
        if not (isinstance(salida, np.ndarray) and salida.dtype == np.uint8 and salida.ndim == 1
                and salida.flags.c_contiguous and salida.flags.writeable):
            raise ErrorLote(_lib.E_INVALID, "la salida debe ser un arreglo uint8 contiguo")
        buf = salida
    else:
        buf = np.empty(max(int(v.size) * max_fila, 1), np.uint8)      # the bound on the rows' sizes: never too small
    n = C.c_int64(0)
    motor._check(fn(motor._ctx, v.ctypes.data, v.size, *extra, estilo, buf.ctypes.data, buf.size, C.byref(n)))
    return buf[:int(n.value)] if salida is not None else buf[:int(n.value)].tobytes()


def totales(motor: MotorLote, total_centavos, base_pedido: int, estilo: int = SQL, salida=None):
    """Host totals (int64 cents, one per order) -> the audit rows of venta.TotalPagar, as bytes.
    With `salida` (a uint8 numpy array, for instance from `MotorLote.reservar_host`) the text is written
    there and the filled part of that array is returned; too small an array raises ErrorLote (E_DATA)."""
    return _host(motor, motor._lib.acmpc_emit_totals, total_centavos, (int(base_pedido),), estilo, MAX_FILA_TOTALES,
                 salida)


def stock(motor: MotorLote, delta_stock, estilo: int = SQL, salida=None):
    """Host stock deltas (int64, one per product of the loaded code list) -> the rows that decrement
    producto.Stock; products whose delta is zero get no row.  `salida` as in `totales`."""
    return _host(motor, motor._lib.acmpc_emit_stock, delta_stock, (), estilo, MAX_FILA_STOCK, salida)


def totales_device(motor: MotorLote, total_centavos, base_pedido: int, estilo: int, texto, n_bytes, stream=None) -> None:
    """Device totals (torch int64 CUDA tensor) -> device text (torch uint8 CUDA tensor, 16-byte aligned);
    `n_bytes` is a one-element torch int64 CUDA tensor that receives the size needed.  Enqueues only."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream(motor.device).cuda_stream
    motor._check(motor._lib.acmpc_emit_totals_device(motor._ctx, total_centavos.data_ptr(), total_centavos.numel(),
                                                     int(base_pedido), estilo, texto.data_ptr(), texto.numel(),
                                                     n_bytes.data_ptr(), stream))


def stock_device(motor: MotorLote, delta_stock, estilo: int, texto, n_bytes, stream=None) -> None:
    import torch
    if stream is None:
        stream = torch.cuda.current_stream(motor.device).cuda_stream
    motor._check(motor._lib.acmpc_emit_stock_device(motor._ctx, delta_stock.data_ptr(), delta_stock.numel(), estilo,
                                                    texto.data_ptr(), texto.numel(), n_bytes.data_ptr(), stream))


__all__ = ["SQL", "CSV", "totales", "stock", "totales_device", "stock_device", "ErrorLote"]
