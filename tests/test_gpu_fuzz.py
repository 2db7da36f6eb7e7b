"""Seeded random small inputs through every launch sequence and option, bit-exact against the C oracle.

Small inputs hit what the large seeded data sets rarely do: orders without lines at the front, at the end and
between waves, one-line data sets, unknown products, negative and very large quantities, 64-bit prices, every
combination of lifecycle columns, rows in any order."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# A longer campaign on the GPU box: ACMPC_FUZZ_SEED shifts every seed, ACMPC_FUZZ_SCALE multiplies the trials.
SEED = int(os.environ.get("ACMPC_FUZZ_SEED", "0"))
SCALE = int(os.environ.get("ACMPC_FUZZ_SCALE", "1"))


def _case(rng):
    n_orders = int(rng.integers(1, 400))
    shape = rng.integers(0, 4)
    if shape == 0:                       # a few lines in many orders
        n_lines = int(rng.integers(0, 12))
    elif shape == 1:                     # about one wave tile
        n_lines = int(rng.integers(200, 300))
    else:
        n_lines = int(rng.integers(0, 6000))
    base = int(rng.choice([5, 0, -7, 2_000_000_000, -2_147_483_648]))
    if base > 0 and base + n_orders > 2**31 - 1:
        base = 2**31 - 1 - n_orders
    oid = np.sort(rng.integers(0, n_orders, size=n_lines)).astype(np.int64) + base
    if rng.random() < 0.3 and n_lines:   # everything in one order, or in the two outermost orders
        oid[:] = base + int(rng.integers(0, n_orders))
        if rng.random() < 0.5:
            oid[: n_lines // 2] = base
            oid[n_lines // 2:] = base + n_orders - 1
    p = int(rng.integers(1, 50))
    pidx = rng.integers(-2, p + 2, size=n_lines).astype(np.int32)
    mode = rng.integers(0, 3)
    if mode == 0:
        qty = np.ones(n_lines, np.int32)
    elif mode == 1:
        qty = rng.integers(-5, 70000, size=n_lines).astype(np.int32)
    else:
        qty = rng.integers(-2**31, 2**31, size=n_lines).astype(np.int32)
    if rng.random() < 0.7:
        price = rng.integers(1, 500_000, size=p).astype(np.int64)
    else:
        price = rng.integers(-2**62, 2**62, size=p).astype(np.int64)
    c = int(rng.integers(1, 300))
    cust = rng.integers(0, c, size=n_orders).astype(np.int32)
    life = {}
    if rng.random() < 0.5:
        if rng.random() < 0.6:
            life["line_price"] = rng.integers(-10**6, 10**9, size=n_lines).astype(np.int64)
        if rng.random() < 0.6:
            life["order_state"] = rng.integers(0, 4, size=n_orders).astype(np.uint8)
        if rng.random() < 0.6:
            life["order_descuento"] = rng.integers(-50, 150, size=n_orders).astype(np.int32)
    return dict(oid=oid.astype(np.int32), pidx=pidx, qty=qty, cust=cust, base=base, c=c, price=price, life=life,
                pct=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("fuse", ["1", "2"])
@pytest.mark.parametrize("order", [False, True, "auto"])
def test_random_small_inputs(oracle, fuse, order):
    from ac_mpc_amd import MotorLote
    old = os.environ.get("ACMPC_FUSE")
    os.environ["ACMPC_FUSE"] = fuse
    try:
        m = MotorLote(0)
    finally:
        if old is None:
            os.environ.pop("ACMPC_FUSE", None)
        else:
            os.environ["ACMPC_FUSE"] = old
    m.orden_lineas(order)
    rng = np.random.default_rng(SEED + 1000 + int(fuse) * 10 + (0 if order is False else 1 if order is True else 2))
    for trial in range(300 * SCALE):
        k = _case(rng)
        oid, pidx, qty = k["oid"], k["pidx"], k["qty"]
        lp = k["life"].get("line_price")
        if order is not False and oid.size:
            perm = rng.permutation(oid.size)
            oid, pidx, qty = oid[perm], pidx[perm], qty[perm]
            lp = None if lp is None else lp[perm]
        m.cargar_productos(k["price"])
        got = m.agregar(oid, pidx, qty, k["cust"], k["base"], k["c"], precio_linea=lp,
                        estado_pedido=k["life"].get("order_state"), descuento=k["life"].get("order_descuento"),
                        descuento_porcentaje=k["pct"])
        exp = oracle.aggregate_lifecycle(oid, pidx, qty, k["cust"], k["base"], k["c"], k["price"], lp,
                                         k["life"].get("order_state"), k["life"].get("order_descuento"), int(k["pct"]))
        assert exp[3] & ~2 == 0
        for g, e, name in zip(got, exp[:3], ("order_total", "product_qty", "customer_total")):
            assert np.array_equal(g, e), (trial, name, fuse, order, oid.size, k["cust"].size)
    m.cerrar()


@pytest.mark.parametrize("fuse", ["1", "2"])
def test_random_medium_inputs_with_poisoned_outputs(oracle, fuse):
    """Up to a few million lines over many waves and workgroups; the device outputs are filled with a
    poison pattern before every call, so a slot the kernels forget to write cannot pass by luck."""
    import torch
    from ac_mpc_amd import MotorLote
    old = os.environ.get("ACMPC_FUSE")
    os.environ["ACMPC_FUSE"] = fuse
    try:
        m = MotorLote(0)
    finally:
        if old is None:
            os.environ.pop("ACMPC_FUSE", None)
        else:
            os.environ["ACMPC_FUSE"] = old
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(SEED + 77 + int(fuse))
    for trial in range(30 * SCALE):
        n_lines = int(rng.choice([1, 255, 256, 257, 4096, 65_536, 300_001, 1_048_576, 3_000_000]))
        density = rng.choice([0.02, 0.25, 1.0, 30.0])            # orders per line
        n_orders = min(8_000_000, max(1, int(n_lines * density) + int(rng.integers(0, 50))))
        base = int(rng.choice([5, -2_147_483_648, 2_147_483_647 - n_orders, 0]))
        lo = int(rng.integers(0, max(1, n_orders // 3)))         # orders before `lo` and from `hi` on have no lines
        hi = n_orders - int(rng.integers(0, max(1, n_orders // 3)))
        hi = max(hi, lo + 1)
        oid = (np.sort(rng.integers(lo, hi, size=n_lines)) + base).astype(np.int32)
        p = int(rng.choice([3, 1000, 10_000, 20_000, 28_000, 70_000]))       # both tables in LDS / the histogram alone / neither
        c = int(rng.choice([1, 500, 100_000, 600_000]))
        pidx = rng.integers(-1, p + 1, size=n_lines).astype(np.int32)
        qty = (np.ones(n_lines) if rng.random() < 0.5 else rng.integers(-3, 100_000, size=n_lines)).astype(np.int32)
        price = rng.integers(1, 500_000, size=p).astype(np.int64)
        cust = rng.integers(0, c, size=n_orders).astype(np.int32)
        m.cargar_productos(price)
        # row order and lifecycle columns at random
        order = [False, True, "auto"][int(rng.integers(0, 3))]
        lp = rng.integers(1, 10**6, size=n_lines).astype(np.int64) if rng.random() < 0.3 else None
        stt = rng.integers(0, 2, size=n_orders).astype(np.uint8) if rng.random() < 0.3 else None
        de = rng.integers(0, 101, size=n_orders).astype(np.int32) if rng.random() < 0.3 else None
        pct = bool(rng.integers(0, 2))
        if order is not False and n_lines > 1:
            perm = rng.permutation(n_lines) if rng.random() < 0.5 else np.arange(n_lines)
            if rng.random() < 0.5:                                 # a single swapped pair, far apart
                i, j = sorted(rng.integers(0, n_lines, size=2).tolist())
                perm[i], perm[j] = perm[j], perm[i]
            oid, pidx, qty = oid[perm], pidx[perm], qty[perm]
            lp = None if lp is None else lp[perm]
        m.orden_lineas(order)
        t = [torch.from_numpy(x).to(dev) for x in (oid, pidx, qty, cust)]
        life = dict(precio_linea=None if lp is None else torch.from_numpy(lp).to(dev),
                    estado_pedido=None if stt is None else torch.from_numpy(stt).to(dev),
                    descuento=None if de is None else torch.from_numpy(de).to(dev), descuento_porcentaje=pct)
        poison = -0x2152_4111_2152_4111
        tot = torch.full((n_orders,), poison, dtype=torch.int64, device=dev)
        pq = torch.full((p,), poison, dtype=torch.int64, device=dev)
        ct = torch.full((c,), poison, dtype=torch.int64, device=dev)
        m.agregar_device(t[0], t[1], t[2], t[3], base, c, tot, pq, ct, **life)
        assert m.estado() == 0
        etot, epq, ect, flags = oracle.aggregate_lifecycle(oid, pidx, qty, cust, base, c, price, lp, stt, de, int(pct))
        assert flags & ~2 == 0
        for g, e, name in zip((tot, pq, ct), (etot, epq, ect), ("order_total", "product_qty", "customer_total")):
            assert np.array_equal(g.cpu().numpy(), e), (trial, name, fuse, order, n_lines, n_orders, p, c, base,
                                                        lp is not None, stt is not None, de is not None)
    m.cerrar()


def test_text_paths_with_poisoned_device_buffers(oracle):
    """Ingest and emit through their device entry points into poisoned buffers, several calls on one context."""
    import torch
    from ac_mpc_amd import MotorLote, emision, ingesta
    from tests import ingest_cases as ic
    m = MotorLote(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(SEED + 91)
    p = 500
    codes = [ic.code_of(i) for i in range(p)]
    ingesta.cargar_codigos(m, codes)
    for trial in range(12 * SCALE):
        n = int(rng.choice([0, 1, 700, 20_000, 250_000]))
        oid = np.sort(rng.integers(5, 5 + max(1, n // 3), size=n)).astype(np.int32)
        pidx = rng.integers(-1, p, size=n).astype(np.int32)
        qty = rng.integers(-9, 10_000, size=n).astype(np.int32)
        style = str(rng.choice(["dump", "csv", "tsv_crlf", "quoted", "statements"]))
        text = ic.render(oid.tolist(), pidx.tolist(), qty.tolist(), style) if n else b"-- nothing\n"
        tb = torch.frombuffer(bytearray(text + b"\0" * 16), dtype=torch.uint8).to(dev)[: len(text)]
        cap = n + 5
        cols = [torch.full((cap,), -77, dtype=torch.int32, device=dev) for _ in range(3)]
        cnt = torch.full((1,), -1, dtype=torch.int64, device=dev)
        ingesta.leer_detalle_device(m, tb, cols[0], cols[1], cols[2], cnt)
        assert m.estado() == 0 and int(cnt.item()) == n
        for got, exp in zip(cols, (oid, pidx, qty)):
            assert np.array_equal(got[:n].cpu().numpy(), exp)
            assert bool((got[n:] == -77).all())                       # nothing written past the rows
        # emit the totals of a random vector into a poisoned text buffer
        mo = int(rng.choice([1, 300, 5000, 70_000]))
        tot = rng.integers(-10**7, 10**12, size=mo).astype(np.int64)
        st = int(rng.integers(0, 2))
        exp_text = oracle.emit_totals(tot, 5, st)
        out = torch.full((len(exp_text) + 64,), 0x7e, dtype=torch.uint8, device=dev)
        nb = torch.full((1,), -1, dtype=torch.int64, device=dev)
        emision.totales_device(m, torch.from_numpy(tot).to(dev), 5, st, out, nb)
        assert m.estado() == 0 and int(nb.item()) == len(exp_text)
        got_text = bytes(out.cpu().numpy())
        assert got_text[: len(exp_text)] == exp_text and got_text[len(exp_text):] == b"\x7e" * 64
    m.cerrar()
