"""Synthetic store.sql-schema data built directly in HBM (wrappers over acmpc_synth_*).

The recipe is integer-only (see include/acmpc.h) so the device generator and the CPU oracle's
generator produce identical columns; tests compare the two.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .lote import ErrorLote, MotorLote

ORDER_ID_BASE = 5   # venta AUTO_INCREMENT start, store.sql:232


@dataclass
class ConjuntoSintetico:
    """One shard of a synthetic data set, resident on the GPU."""
    order_id: "object"
    product_idx: "object"
    qty: "object"
    order_customer: "object"
    price_cents: np.ndarray
    order_id_base: int          # base of THIS shard (global base + order_lo)
    n_customers: int
    n_lines: int
    n_orders: int
    n_lines_total: int
    n_orders_total: int


def _cfg(n_lines_total, n_products, n_customers, zipf, qty_mode, rank, world) -> _lib.SynthCfg:
    return _lib.SynthCfg(int(n_lines_total), int(n_products), int(n_customers), ORDER_ID_BASE,
                         1 if zipf else 0, int(qty_mode), int(rank), int(world))


def plan(n_lines_total: int, rank: int = 0, world: int = 1) -> _lib.SynthShard:
    lib = _lib.load()
    cfg = _cfg(n_lines_total, 1, 1, False, 0, rank, world)
    sh = _lib.SynthShard()
    rc = lib.acmpc_synth_plan(C.byref(cfg), C.byref(sh))
    if rc != 0:
        raise ErrorLote(rc, lib.acmpc_last_error(None).decode())
    return sh


def precios(n_products: int) -> np.ndarray:
    lib = _lib.load()
    cfg = _cfg(0, n_products, 1, False, 0, 0, 1)
    out = np.empty(n_products, dtype=np.int64)
    rc = lib.acmpc_synth_prices(C.byref(cfg), out.ctypes.data)
    if rc != 0:
        raise ErrorLote(rc, lib.acmpc_last_error(None).decode())
    return out


def generar(motor: MotorLote, n_lines_total: int, n_products: int = 10_000, n_customers: int = 100_000,
            zipf: bool = False, qty_mode: int = 0, rank: int = 0, world: int = 1) -> ConjuntoSintetico:
    """Generate shard `rank` of `world` on the motor's GPU and load the product table."""
    import torch
    lib = _lib.load()
    cfg = _cfg(n_lines_total, n_products, n_customers, zipf, qty_mode, rank, world)
    sh = _lib.SynthShard()
    rc = lib.acmpc_synth_plan(C.byref(cfg), C.byref(sh))
    if rc != 0:
        raise ErrorLote(rc, lib.acmpc_last_error(None).decode())
    n = int(sh.line_hi - sh.line_lo)
    m = int(sh.order_hi - sh.order_lo)
    dev = torch.device("cuda", motor.device)
    oid = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    pidx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    qty = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    cust = torch.empty(max(m, 1), dtype=torch.int32, device=dev)[:m]
    stream = torch.cuda.current_stream(motor.device).cuda_stream
    rc = lib.acmpc_synth_generate_device(motor._ctx, C.byref(cfg), oid.data_ptr(), pidx.data_ptr(),
                                         qty.data_ptr(), cust.data_ptr(), stream)
    if rc != 0:
        raise ErrorLote(rc, lib.acmpc_last_error(motor._ctx).decode())
    price = precios(n_products)
    motor.cargar_productos(price)
    return ConjuntoSintetico(oid, pidx, qty, cust, price, ORDER_ID_BASE + int(sh.order_lo), n_customers, n, m,
                             int(n_lines_total), int(sh.n_orders_total))
