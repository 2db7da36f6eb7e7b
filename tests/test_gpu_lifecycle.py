"""GPU parity of the order-lifecycle columns through the C ABI: bit-exact against the C oracle."""
import numpy as np
import pytest

from tests import lifecycle_cases as lc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "fused"])
def nuevo_motor(request):
    """A fresh context per test (options and product tables differ between them), closed afterwards.  Every case
    runs twice: with the library's own choice of launch sequence, and with the three-launch sequence forced (by
    default it is only taken from a few million orders on)."""
    import os
    from ac_mpc_amd import MotorLote
    old = os.environ.get("ACMPC_FUSE")
    if request.param == "fused":
        os.environ["ACMPC_FUSE"] = "2"
    try:
        m = MotorLote(0)
    finally:
        if old is None:
            os.environ.pop("ACMPC_FUSE", None)
        else:
            os.environ["ACMPC_FUSE"] = old
    yield m
    m.cerrar()


def _run_host(motor, d, n_customers, cols, mode):
    return motor.agregar(d["order_id"], d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"], n_customers,
                         precio_linea=cols["line_price"], estado_pedido=cols["order_state"],
                         descuento=cols["order_descuento"], descuento_porcentaje=bool(mode))


def _expect(oracle, d, n_customers, price, cols, mode, any_order=False):
    tot, pq, ct, flags = oracle.aggregate_lifecycle(d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                                    d["order_id_base"], n_customers, price, cols["line_price"],
                                                    cols["order_state"], cols["order_descuento"], mode)
    assert flags & ~(2 if any_order else 0) == 0          # (the oracle only notes that the ids are unsorted)
    return tot, pq, ct


@pytest.mark.parametrize("case", lc.fixture()["cases"], ids=lambda c: c["name"])
def test_known_answers(case, nuevo_motor):
    d = lc.fixture()
    m = nuevo_motor
    m.cargar_productos(np.array(case.get("price_cents_now", d["price_cents"]), np.int64))
    tot, pq, ct = m.agregar(d["order_id"], d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"],
                            d["n_customers"], precio_linea=case.get("line_price"), estado_pedido=case.get("order_state"),
                            descuento=case.get("order_descuento"), descuento_porcentaje=bool(case.get("descuento_mode", 0)))
    assert (tot.tolist(), pq.tolist(), ct.tolist()) == (case["order_total"], case["product_qty"], case["customer_total"])


@pytest.mark.parametrize("mask", lc.MASKS)
@pytest.mark.parametrize("n_lines,p,c", [(1, 3, 1), (255, 10, 7), (1000, 100, 50), (65_537, 10_000, 100_000), (1_000_003, 10_000, 100_000)])
def test_random_columns_all_masks(oracle, mask, n_lines, p, c, nuevo_motor):
    d = oracle.synth(n_lines, p, c, qty_mode=1)
    price = oracle.synth_prices(p)
    m = nuevo_motor
    m.cargar_productos(price)
    for mode in (0, 1):
        cols = lc.columns(d, n_lines + mode, mask)
        got = _run_host(m, d, c, cols, mode)
        exp = _expect(oracle, d, c, price, cols, mode)
        for g, e, name in zip(got, exp, ("order_total", "product_qty", "customer_total")):
            assert np.array_equal(g, e), (name, mask, mode)


def test_wide_values_wrap_like_the_oracle(oracle, nuevo_motor):
    d = oracle.synth(20_000, 64, 33, qty_mode=1)
    price = oracle.synth_prices(64)
    m = nuevo_motor
    m.cargar_productos(price)
    for mode in (0, 1):
        cols = lc.columns(d, 99 + mode, "psd", wide=True)
        got = _run_host(m, d, 33, cols, mode)
        exp = _expect(oracle, d, 33, price, cols, mode)
        assert all(np.array_equal(g, e) for g, e in zip(got, exp))


@pytest.mark.parametrize("big_tables", [False, True])
@pytest.mark.parametrize("any_order", [False, True])
def test_table_modes_and_any_order(oracle, big_tables, any_order, nuevo_motor):
    """Price table too large for LDS (global-table kernels) and rows in any order."""
    p, c = (60_000, 700_000) if big_tables else (500, 2_000)
    d = oracle.synth(300_000, p, c, qty_mode=1)
    price = oracle.synth_prices(p)
    cols = lc.columns(d, 5, "psd")
    exp = _expect(oracle, d, c, price, cols, 1)
    if any_order:
        perm = np.random.default_rng(1).permutation(d["order_id"].size)
        for k in ("order_id", "product_idx", "qty"):
            d[k] = d[k][perm]
        cols["line_price"] = cols["line_price"][perm]
    m = nuevo_motor
    m.cargar_productos(price)
    m.orden_lineas(any_order)
    got = _run_host(m, d, c, cols, 1)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    # shuffled rows with every lifecycle column are moved next to their orders (the order-restoring path), not
    # summed with one atomic per line; product tables too large for LDS keep the atomic kernel
    from ac_mpc_amd import _lib
    assert m.camino_lineas() == (_lib.PATH_NONE if (big_tables or not any_order) else _lib.PATH_MOVED)


@pytest.mark.parametrize("mask", ["p--", "-s-", "--d", "psd"])
@pytest.mark.parametrize("orden", [True, "auto"])
def test_every_column_on_every_way_through_rows_in_any_order(oracle, nuevo_motor, mask, orden):
    """The device picks one of four ways to sum rows whose order is not promised; each must honour the lifecycle
    columns: rows in no order (moved), rows with a few swapped neighbours (one atomic per run), values that do not
    fit a record (stock deltas from the moving pass, totals by atomics) and rows that were grouped after all."""
    from ac_mpc_amd import _lib
    n, p, c = 400_003, 700, 3_000
    d = oracle.synth(n, p, c, qty_mode=1)
    price = oracle.synth_prices(p)
    m = nuevo_motor
    m.cargar_productos(price)
    m.orden_lineas(orden)
    rng = np.random.default_rng(17)
    for way in ("shuffled", "swapped", "wide", "grouped"):
        cols = lc.columns(d, 31, mask)
        dd = dict(d)
        perm = np.arange(n)
        if way in ("shuffled", "wide"):
            perm = rng.permutation(n)
        elif way == "swapped":
            k = rng.choice(n - 1, 50, replace=False)
            perm[k], perm[k + 1] = perm[k + 1].copy(), perm[k].copy()
        for key in ("order_id", "product_idx", "qty"):
            dd[key] = d[key][perm]
        if cols["line_price"] is not None:
            cols["line_price"] = cols["line_price"][perm]
        if way == "wide":                                   # one line worth more than 2^39 cents (a quantity of 2^30)
            dd["qty"] = dd["qty"].copy()
            dd["qty"][n // 2] = 1 << 30
        for mode in (0, 1):
            exp = _expect(oracle, dd, c, price, cols, mode, any_order=True)
            got = _run_host(m, dd, c, cols, mode)
            for g, e, name in zip(got, exp, ("order_total", "product_qty", "customer_total")):
                assert np.array_equal(g, e), (way, name, mode)
            path = m.camino_lineas()
            if orden is True:
                assert path == {"shuffled": _lib.PATH_MOVED, "swapped": _lib.PATH_ATOMIC, "wide": _lib.PATH_MOVED_STOCK,
                                "grouped": _lib.PATH_GROUPED}[way], (way, path)


def test_orders_without_lines_still_get_their_discount(nuevo_motor):
    m = nuevo_motor
    m.cargar_productos(np.array([100, 200], np.int64))
    z = np.zeros(0, np.int32)
    tot, pq, ct = m.agregar(z, z, z, np.array([0, 1, 0], np.int32), 5, 2, descuento=np.array([1, 2, 3], np.int32))
    assert tot.tolist() == [-100, -200, -300] and pq.tolist() == [0, 0] and ct.tolist() == [-400, -200]
    # orders 5 and 7 have no lines, order 6 has one
    tot, pq, ct = m.agregar(np.array([6], np.int32), np.array([1], np.int32), np.array([3], np.int32),
                            np.array([0, 1, 0], np.int32), 5, 2, descuento=np.array([1, 2, 3], np.int32),
                            estado_pedido=np.array([1, 1, 0], np.uint8))
    assert tot.tolist() == [-100, 400, -300] and pq.tolist() == [0, 3] and ct.tolist() == [-100, 400]


def test_bad_lifecycle_arguments(nuevo_motor):
    from ac_mpc_amd import ErrorLote
    m = nuevo_motor
    m.cargar_productos(np.array([100], np.int64))
    one = np.array([5], np.int32)
    with pytest.raises(ErrorLote):
        m.agregar(one, np.zeros(1, np.int32), np.ones(1, np.int32), np.zeros(1, np.int32), 5, 1, estado_pedido=np.ones(2, np.uint8))
    with pytest.raises(ErrorLote):
        m.agregar(one, np.zeros(1, np.int32), np.ones(1, np.int32), np.zeros(1, np.int32), 5, 1, precio_linea=np.ones(3, np.int64))


def test_device_columns_at_bench_size(oracle, nuevo_motor):
    """100 M lines with every lifecycle column, device resident, against the C oracle."""
    import torch
    from ac_mpc_amd import sintetico
    m = nuevo_motor
    dev = torch.device("cuda", 0)
    n, p, c = 100_000_000, 10_000, 100_000
    price = oracle.synth_prices(p)
    m.cargar_productos(price)
    data = sintetico.generar(m, n, p, c)
    mo = data.order_customer.numel()
    g = torch.Generator(device=dev); g.manual_seed(11)
    lp = torch.randint(1, 500_000, (n,), dtype=torch.int64, device=dev, generator=g)
    stt = (torch.randint(0, 10, (mo,), device=dev, generator=g) != 0).to(torch.uint8)
    de = torch.randint(0, 101, (mo,), dtype=torch.int32, device=dev, generator=g)
    tot = torch.empty(mo, dtype=torch.int64, device=dev)
    pq = torch.empty(p, dtype=torch.int64, device=dev)
    ct = torch.empty(c, dtype=torch.int64, device=dev)
    m.agregar_device(data.order_id, data.product_idx, data.qty, data.order_customer, data.order_id_base, c, tot, pq, ct,
                     precio_linea=lp, estado_pedido=stt, descuento=de, descuento_porcentaje=True)
    assert m.estado() == 0
    d = dict(order_id=data.order_id.cpu().numpy(), product_idx=data.product_idx.cpu().numpy(), qty=data.qty.cpu().numpy(),
             order_customer=data.order_customer.cpu().numpy(), order_id_base=data.order_id_base)
    cols = dict(line_price=lp.cpu().numpy(), order_state=stt.cpu().numpy(), order_descuento=de.cpu().numpy())
    etot, epq, ect = _expect(oracle, d, c, price, cols, 1)
    assert np.array_equal(tot.cpu().numpy(), etot)
    assert np.array_equal(pq.cpu().numpy(), epq)
    assert np.array_equal(ct.cpu().numpy(), ect)
