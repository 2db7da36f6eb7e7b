"""Text ingest of `detalle` rows (wrappers over acmpc_set_product_codes / acmpc_ingest_detalle*).

The accepted text is what the reference itself writes: the INSERT statement of
process/confirmcompra.php:36, the phpMyAdmin dump style of store.sql:88-89,125-128 and plain comma / tab
separated rows; the grammar is in include/acmpc.h.  `CodigoProd` (varchar(30), store.sql:99) is
dictionary-encoded on the GPU against the product code list.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np

from . import _lib
from .lote import ErrorLote, MotorLote


def empaquetar_codigos(codigos: Sequence[str]) -> np.ndarray:
    """Product codes -> the [P, 32] NUL-padded byte table the C ABI takes (latin-1, at most 30 bytes)."""
    out = np.zeros((len(codigos), 32), dtype=np.uint8)
    for i, c in enumerate(codigos):
        b = c if isinstance(c, bytes) else str(c).encode("latin-1")
        if len(b) > 30:
            raise ErrorLote(_lib.E_INVALID, f"CodigoProd de mas de 30 bytes: {c!r}")
        out[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return out


def cargar_codigos(motor: MotorLote, codigos: Sequence[str]) -> None:
    packed = empaquetar_codigos(codigos)
    motor._check(motor._lib.acmpc_set_product_codes(motor._ctx, packed.ctypes.data, len(codigos)))


def leer_detalle(motor: MotorLote, texto, capacidad: int = -1, salida=None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Host text in, host columns out: (order_id, product_idx, qty), rows in text order.
    `texto` is bytes or a one-dimensional uint8 numpy array (for instance one from
    `MotorLote.reservar_host`, which the GPU reads by direct DMA); `salida` optionally names three int32
    arrays of equal length (the row capacity) that receive the columns.
    Raises ErrorLote (E_DATA) on a malformed data row."""
    if isinstance(texto, np.ndarray):
        if texto.dtype != np.uint8 or texto.ndim != 1 or not texto.flags.c_contiguous:
            raise ErrorLote(_lib.E_INVALID, "el texto debe ser bytes o un arreglo uint8 contiguo")
        ptr, n_bytes = texto.ctypes.data, texto.size
    else:
        ptr, n_bytes = texto, len(texto)
    if salida is not None:
        oid, pidx, qty = salida
        for a in salida:
            if not (isinstance(a, np.ndarray) and a.dtype == np.int32 and a.ndim == 1 and a.size == oid.size
                    and a.flags.c_contiguous and a.flags.writeable):
                raise ErrorLote(_lib.E_INVALID, "columnas de salida de tipo o longitud incorrectos")
        if capacidad < 0 or capacidad > oid.size:
            capacidad = oid.size
    else:
        if capacidad < 0:       # one row per line at most
            capacidad = 1 + (int(np.count_nonzero(texto == 10)) if isinstance(texto, np.ndarray) else texto.count(b"\n"))
        oid = np.empty(max(capacidad, 1), np.int32)
        pidx = np.empty(max(capacidad, 1), np.int32)
        qty = np.empty(max(capacidad, 1), np.int32)
    n, bad = C.c_int64(0), C.c_int64(0)
    motor._check(motor._lib.acmpc_ingest_detalle(motor._ctx, ptr, n_bytes, oid.ctypes.data, pidx.ctypes.data,
                                                 qty.ctypes.data, capacidad, C.byref(n), C.byref(bad)))
    k = int(n.value)
    return oid[:k], pidx[:k], qty[:k]


def leer_detalle_device(motor: MotorLote, texto, order_id, product_idx, qty, n_filas, stream=None) -> None:
    """Device text (torch uint8 CUDA tensor) -> device columns (torch int32 CUDA tensors of equal length);
    `n_filas` is a one-element torch int64 CUDA tensor that receives the number of data rows.  Enqueues only."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream(motor.device).cuda_stream
    cap = order_id.numel()
    if product_idx.numel() != cap or qty.numel() != cap:
        raise ErrorLote(_lib.E_INVALID, "columnas de distinta longitud")
    motor._check(motor._lib.acmpc_ingest_detalle_device(motor._ctx, texto.data_ptr(), texto.numel(), order_id.data_ptr(),
                                                        product_idx.data_ptr(), qty.data_ptr(), cap, n_filas.data_ptr(),
                                                        stream))
