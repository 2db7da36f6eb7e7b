"""Host logic of the multi-GPU path on CPU: order-range cuts and the one all-reduce (gloo, 2 ranks).

The GPU engine cannot run here, so each rank uses the CPU oracle as its local aggregator; what is
under test is the sharding plan and the merge, which are the same code the GPU path uses."""
import os
import socket
import sys

import numpy as np
import pytest

from ac_mpc_amd import formato_subtotal, reparto
from oracle import checkout_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cuts_fall_on_order_boundaries(oracle):
    d = oracle.synth(20_000, 100, 50)
    m = d["n_orders_total"]
    for world in (1, 2, 3, 8):
        cs = reparto.cortes(d["order_id"], 5, m, world)
        assert cs[0].order_lo == 0 and cs[0].line_lo == 0
        assert cs[-1].order_hi == m and cs[-1].line_hi == d["order_id"].size
        for a, b in zip(cs, cs[1:]):
            assert a.order_hi == b.order_lo and a.line_hi == b.line_lo
        for c in cs:
            ids = d["order_id"][c.line_lo:c.line_hi]
            if ids.size:
                assert ids[0] >= 5 + c.order_lo and ids[-1] < 5 + c.order_hi
            if c.line_lo > 0 and c.line_lo < d["order_id"].size:
                assert d["order_id"][c.line_lo - 1] != d["order_id"][c.line_lo]


def test_rango_pedidos_rejects_bad_rank():
    with pytest.raises(ValueError):
        reparto.rango_pedidos(10, 2, 2)


def test_formato_subtotal_matches_php_number_format():
    for cents in (0, 5, 99, 100, 220000, 150000, 123456789, -123450, 10**15 + 1):
        assert formato_subtotal(cents) == ("-" if cents < 0 else "") + "$" + co.php_number_format_2(abs(cents) / 100) \
            or cents >= 10**15       # beyond 15 significant digits the double path itself is inexact
    assert formato_subtotal(220000) == "$2,200.00"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_lines, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from tests.oracle_binding import Oracle
    from ac_mpc_amd import reparto as rp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle()
        p_n, c_n = 300, 700
        price = orc.synth_prices(p_n)
        m_total = int(orc.lib.oracle_synth_n_orders(n_lines))
        lo, hi = rp.rango_pedidos(m_total, rank, world)
        d = orc.synth(n_lines, p_n, c_n, zipf=1, qty_mode=1, order_lo=lo, order_hi=hi)
        tot, pq, ct, flags = orc.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                           d["order_id_base"], c_n, price)
        assert flags == 0
        merged = torch.from_numpy(np.concatenate([pq, ct]))
        rp.fusionar(merged)
        q.put((rank, lo, hi, tot, merged.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_merge_equals_unsharded(oracle, world):
    """One process per shard (two ranks; three, which cut the orders unevenly), the exchange over gloo."""
    import torch.multiprocessing as mp
    n_lines = 30_011
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_lines, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    p_n, c_n = 300, 700
    d = oracle.synth(n_lines, p_n, c_n, zipf=1, qty_mode=1)
    etot, epq, ect, _ = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, c_n,
                                         oracle.synth_prices(p_n))
    results.sort(key=lambda r: r[0])
    np.testing.assert_array_equal(np.concatenate([r[3] for r in results]), etot)     # totals stay local
    for r in results:
        np.testing.assert_array_equal(r[4], np.concatenate([epq, ect]))              # every rank holds the merge


def _pipeline_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ac_mpc_amd import reparto as rp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bufs = [torch.zeros(1000, dtype=torch.int64) for _ in range(2)]
        turnos = rp.FusionEnCurso(bufs)
        seen = []
        for step in range(5):
            buf = turnos.siguiente()                 # waits for the exchange of step - 2, which used this buffer
            if step >= 2:
                seen.append(buf.clone())             # ... so its merged values are all there
            buf.copy_(torch.arange(1000, dtype=torch.int64) * (rank + 1) + step)
            turnos.lanzar()
        last = turnos.esperar()
        q.put((rank, [s.numpy() for s in seen], last.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_overlapped_merge_keeps_every_step():
    """The exchange of step i stays in flight during step i + 1: two buffers in turn, each waited for
    before it is written again, all of them at the end."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(1000, dtype=np.int64)
    merged = lambda step: base * 3 + 2 * step        # rank 0: base + step, rank 1: 2 * base + step
    for _, seen, last in results:
        for k, got in enumerate(seen):               # buffers handed back at steps 2, 3, 4 hold steps 0, 1, 2
            np.testing.assert_array_equal(got, merged(k))
        np.testing.assert_array_equal(last, merged(4))


def test_overlapped_merge_without_a_process_group_is_a_no_op():
    import torch
    from ac_mpc_amd import reparto as rp
    bufs = [torch.arange(4, dtype=torch.int64), torch.zeros(4, dtype=torch.int64)]
    turnos = rp.FusionEnCurso(bufs)
    assert turnos.siguiente() is bufs[0]
    assert turnos.lanzar() is bufs[0]
    assert turnos.siguiente() is bufs[1]
    turnos.lanzar()
    assert turnos.esperar() is bufs[1] and bufs[0].tolist() == [0, 1, 2, 3]
