"""Sharding of orders over GPUs and the one exchange step.

Orders are independent (each `venta` row owns its `detalle` rows, process/confirmcompra.php:24,35-37),
so shard g of G owns the contiguous order-index range [floor(g*M/G), floor((g+1)*M/G)) together with
exactly the lines of those orders: cuts always fall on order boundaries.  Per-order totals stay
local.  Only the per-product and per-customer vectors cross shards: one all-reduce (sum, int64) over
the concatenated [product_qty | customer_total_cents] buffer.  Integer sums are independent of the
reduction order, so the merged result is bit-identical for any number of shards.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


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


def cortes_c(order_id: np.ndarray, order_id_base: int, n_orders: int, world: int,
             line_order: int = _lib.ORDER_GROUPED) -> List[Corte]:
    """The same cuts from the library itself (`acmpc_shard_plan`, host only: no GPU is needed): what
    `acmpc_aggregate_sharded` uses.  With rows in any order the line ranges are the shards' places after
    `particion_c`."""
    lib = _lib.load()
    oid = np.ascontiguousarray(order_id, np.int32)
    o_cut = np.zeros(world + 1, np.int64)
    l_cut = np.zeros(world + 1, np.int64)
    rc = lib.acmpc_shard_plan(oid.ctypes.data, oid.size, int(n_orders), int(order_id_base), int(world), int(line_order),
                              o_cut.ctypes.data, l_cut.ctypes.data)
    if rc != 0:
        from .lote import ErrorLote
        raise ErrorLote(rc, lib.acmpc_last_error(None).decode())
    return [Corte(int(o_cut[g]), int(o_cut[g + 1]), int(l_cut[g]), int(l_cut[g + 1])) for g in range(world)]


def particion_c(order_id, product_idx, qty, order_id_base: int, n_orders: int, world: int):
    """Rows in any order brought together shard by shard (`acmpc_shard_partition`, host only): returns the
    three permuted columns and the line cuts."""
    lib = _lib.load()
    oid = np.ascontiguousarray(order_id, np.int32)
    pidx = np.ascontiguousarray(product_idx, np.int32)
    q = np.ascontiguousarray(qty, np.int32)
    out = [np.empty(oid.size, np.int32) for _ in range(3)]
    l_cut = np.zeros(world + 1, np.int64)
    rc = lib.acmpc_shard_partition(oid.ctypes.data, pidx.ctypes.data, q.ctypes.data, oid.size, int(n_orders),
                                   int(order_id_base), int(world), out[0].ctypes.data, out[1].ctypes.data,
                                   out[2].ctypes.data, l_cut.ctypes.data)
    if rc != 0:
        from .lote import ErrorLote
        raise ErrorLote(rc, lib.acmpc_last_error(None).decode())
    return out[0], out[1], out[2], l_cut


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


# ---- the same exchange behind the C ABI (RCCL called by libacmpc.so itself) ------------------------

ID_BYTES = 128


def nuevo_id() -> bytes:
    """A communicator id (rank 0 creates it; the other ranks receive it over any channel)."""
    lib = _lib.load()
    buf = C.create_string_buffer(ID_BYTES)
    rc = lib.acmpc_comm_id(buf)
    if rc != 0:
        from .lote import ErrorLote
        raise ErrorLote(rc, lib.acmpc_last_error(None).decode())
    return buf.raw


class ComunicadorC:
    """One rank of an RCCL communicator owned by the library: `acmpc_comm_init_rank` for one process per
    GPU (pass the id of `nuevo_id()`), or built by `comunicadores_locales` for several GPUs in one process."""

    def __init__(self, motor, rank: int = 0, world: int = 1, id_bytes: Optional[bytes] = None, _handle=None):
        self.motor, self.rank, self.world = motor, int(rank), int(world)
        self._lib = motor._lib
        if _handle is not None:
            self._comm = _handle
            return
        if id_bytes is None:
            if world != 1:
                raise ValueError("every rank needs the id that rank 0 created")
            id_bytes = nuevo_id()
        h = C.c_void_p()
        motor._check(self._lib.acmpc_comm_init_rank(C.byref(h), motor._ctx, id_bytes, self.rank, self.world))
        self._comm = h

    def fusionar(self, product_y_cliente, stream: Optional[int] = None) -> None:
        """In-place sum over all ranks of a torch int64 CUDA tensor, enqueued on `stream` (default: current)."""
        import torch
        t = product_y_cliente
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.device.index == self.motor.device):
            from .lote import ErrorLote
            raise ErrorLote(_lib.E_INVALID, "tensor de tipo, dispositivo o disposicion incorrectos")
        if stream is None:
            stream = torch.cuda.current_stream(self.motor.device).cuda_stream
        self.motor._check(self._lib.acmpc_merge_device(self.motor._ctx, self._comm, t.data_ptr(), t.numel(), stream))

    def cerrar(self) -> None:
        if self._comm is not None:
            self._lib.acmpc_comm_destroy(self._comm)
            self._comm = None


def comunicadores_locales(motores: Sequence) -> List[ComunicadorC]:
    """One communicator rank per motor, all in this process (`acmpc_comm_init_all`)."""
    n = len(motores)
    lib = motores[0]._lib
    ctxs = (C.c_void_p * n)(*[m._ctx for m in motores])
    out = (C.c_void_p * n)()
    motores[0]._check(lib.acmpc_comm_init_all(out, ctxs, n))
    return [ComunicadorC(m, g, n, _handle=C.c_void_p(out[g])) for g, m in enumerate(motores)]


def agregar_repartido(motores: Sequence, comunicadores: Sequence[ComunicadorC], order_id, product_idx, qty,
                      order_customer, order_id_base: int, n_customers: int):
    """`MotorLote.agregar` over several GPUs from one process (`acmpc_aggregate_sharded`): host columns in,
    (order_total, product_qty, customer_total) out."""
    oid = np.ascontiguousarray(order_id, np.int32)
    pidx = np.ascontiguousarray(product_idx, np.int32)
    q = np.ascontiguousarray(qty, np.int32)
    cust = np.ascontiguousarray(order_customer, np.int32)
    n = len(motores)
    m0 = motores[0]
    tot = np.empty(cust.size, np.int64)
    pq = np.empty(m0.n_products, np.int64)
    ct = np.empty(int(n_customers), np.int64)
    ctxs = (C.c_void_p * n)(*[m._ctx for m in motores])
    comms = (C.c_void_p * n)(*[c._comm for c in comunicadores])
    m0._check(m0._lib.acmpc_aggregate_sharded(ctxs, comms, n, oid.ctypes.data, pidx.ctypes.data, q.ctypes.data, oid.size,
                                              cust.ctypes.data, cust.size, int(order_id_base), int(n_customers),
                                              tot.ctypes.data, pq.ctypes.data, ct.ctypes.data))
    return tot, pq, ct


class FusionEnCursoC:
    """FusionEnCurso with the exchange issued by the library: the all-reduce of batch i runs on a side
    stream while the kernels of batch i + 1 run on the caller's stream; two result buffers in turn."""

    def __init__(self, buffers, comunicador: ComunicadorC):
        import torch
        self.buffers = list(buffers)
        self.com = comunicador
        dev = comunicador.motor.device
        self._side = torch.cuda.Stream(device=dev)
        self._done = [None] * len(self.buffers)
        self._turn = 0
        self._dev = dev

    def siguiente(self):
        import torch
        i = self._turn % len(self.buffers)
        if self._done[i] is not None:
            torch.cuda.current_stream(self._dev).wait_event(self._done[i])
            self._done[i] = None
        return self.buffers[i]

    def lanzar(self):
        import torch
        i = self._turn % len(self.buffers)
        self._side.wait_stream(torch.cuda.current_stream(self._dev))
        self.com.fusionar(self.buffers[i], self._side.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(self._side)
        self._done[i] = ev
        self._turn += 1
        return self.buffers[i]

    def esperar(self):
        import torch
        for i, ev in enumerate(self._done):
            if ev is not None:
                torch.cuda.current_stream(self._dev).wait_event(ev)
                self._done[i] = None
        return self.buffers[(self._turn - 1) % len(self.buffers)]
