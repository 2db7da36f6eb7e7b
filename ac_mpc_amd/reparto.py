"""Sharding of orders over GPUs and the one exchange step.

Orders are independent (each `venta` row owns its `detalle` rows, process/confirmcompra.php:24,35-37),
so shard g of G owns the contiguous order-index range [floor(g*M/G), floor((g+1)*M/G)) together with
exactly the lines of those orders: cuts always fall on order boundaries.  Per-order totals stay
local.  Only the per-product and per-customer vectors cross shards: one all-reduce (sum, int64) over
the concatenated [product_qty | customer_total_cents] buffer.  Integer sums are independent of the
reduction order, so the merged result is bit-identical for any number of shards.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np


@dataclass(frozen=True)
class Corte:
    """One shard: orders [order_lo, order_hi) and lines [line_lo, line_hi) of the global arrays."""
    order_lo: int
    order_hi: int
    line_lo: int
    line_hi: int

    @property
    def n_orders(self) -> int:
        return self.order_hi - self.order_lo

    @property
    def n_lines(self) -> int:
        return self.line_hi - self.line_lo


def rango_pedidos(n_orders: int, rank: int, world: int) -> tuple:
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("rank/world")
    return (n_orders * rank) // world, (n_orders * (rank + 1)) // world


def cortes(order_id: np.ndarray, order_id_base: int, n_orders: int, world: int) -> List[Corte]:
    """Cut a sorted `order_id` column into `world` shards on order boundaries."""
    oid = np.asarray(order_id)
    out = []
    for g in range(world):
        lo, hi = rango_pedidos(n_orders, g, world)
        l0 = int(np.searchsorted(oid, order_id_base + lo, side="left"))
        l1 = int(np.searchsorted(oid, order_id_base + hi, side="left"))
        out.append(Corte(lo, hi, l0, l1))
    return out


def fusionar(product_y_cliente, group=None) -> None:
    """In-place all-reduce (sum) of the concatenated [product_qty | customer_total_cents] tensor.

    `product_y_cliente` is a torch int64 tensor (CUDA with the RCCL backend, CPU with gloo).  A
    single-process run (no process group) leaves it unchanged.
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(product_y_cliente, op=dist.ReduceOp.SUM, group=group)
