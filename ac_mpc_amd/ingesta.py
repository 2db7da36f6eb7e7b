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


def leer_detalle(motor: MotorLote, texto: bytes, capacidad: int = -1) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Host text in, host columns out: (order_id, product_idx, qty), rows in text order.
    Raises ErrorLote (E_DATA) on a malformed data row."""
    if capacidad < 0:
        capacidad = texto.count(b"\n") + 1
    oid = np.empty(max(capacidad, 1), np.int32)
    pidx = np.empty(max(capacidad, 1), np.int32)
    qty = np.empty(max(capacidad, 1), np.int32)
    n, bad = C.c_int64(0), C.c_int64(0)
    motor._check(motor._lib.acmpc_ingest_detalle(motor._ctx, texto, len(texto), oid.ctypes.data, pidx.ctypes.data,
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
