"""Seeded random small inputs through every launch sequence and option, bit-exact against the C oracle.

Small inputs hit what the large seeded data sets rarely do: orders without lines at the front, at the end and
between waves, one-line data sets, unknown products, negative and very large quantities, 64-bit prices, every
combination of lifecycle columns, rows in any order."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    rng = np.random.default_rng(1000 + int(fuse) * 10 + (0 if order is False else 1 if order is True else 2))
    for trial in range(300):
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
