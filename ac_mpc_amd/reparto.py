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


def fusionar(product_y_cliente, group=None, en_curso: bool = False):
    """In-place all-reduce (sum) of the concatenated [product_qty | customer_total_cents] tensor.

    `product_y_cliente` is a torch int64 tensor (CUDA with the RCCL backend, CPU with gloo).  A
    single-process run (no process group) leaves it unchanged.

    With `en_curso` the collective is only started and its work handle returned (None when there is
    nothing to exchange): the caller goes on with the next batch and calls `.wait()` on the handle before
    it reads, or reuses, the tensor.  With RCCL that wait orders the caller's stream behind the
    collective without blocking the host.
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(product_y_cliente, op=dist.ReduceOp.SUM, group=group, async_op=en_curso)
        return work if en_curso else None
    return None


class FusionEnCurso:
    """The exchange step of batch i overlapped with the kernels of batch i + 1: two result buffers used
    in turn; before a buffer is written again the all-reduce that last read it is waited for."""

    def __init__(self, buffers, group=None):
        self.buffers = list(buffers)
        self.group = group
        self._work = [None] * len(self.buffers)
        self._turn = 0

    def siguiente(self):
        """The buffer of the next batch, free to be overwritten."""
        i = self._turn % len(self.buffers)
        if self._work[i] is not None:
            self._work[i].wait()
            self._work[i] = None
        return self.buffers[i]

    def lanzar(self):
        """Start the all-reduce of the buffer handed out last; returns that buffer."""
        i = self._turn % len(self.buffers)
        self._work[i] = fusionar(self.buffers[i], self.group, en_curso=True)
        self._turn += 1
        return self.buffers[i]

    def esperar(self):
        """Wait for every exchange still in flight; returns the buffer of the last batch."""
        for i, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[i] = None
        return self.buffers[(self._turn - 1) % len(self.buffers)]
