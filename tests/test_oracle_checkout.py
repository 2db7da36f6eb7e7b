"""The Python checkout oracle against the hand-derived known-answer cases (tests/golden/checkout_cases.json)
and against independent restatements (decimal arithmetic, sqlite3 GROUP BY)."""
import json
import os
import random
import sqlite3
from decimal import Decimal

import numpy as np
import pytest

from oracle import checkout_oracle as co

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seed():
    return json.load(open(os.path.join(GOLDEN, "store_seed.json")))


def _cases():
    return json.load(open(os.path.join(GOLDEN, "checkout_cases.json"), encoding="utf-8"))["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_checkout_known_answers(case):
    store = co.Store.from_seed(_seed())
    chk = co.Checkout(store)
    ses = co.Session()
    display = chk.carrito(ses)                       # empty render first, like a page load
    for code in case["cart"]:
        display = chk.carrito(ses, code)
    assert display == case["subtotal_display"]
    assert ses.sumaTotal == case["sumaTotal"]
    if case["sql_total_text"] is not None:
        assert co.php_float_to_string(float(ses.sumaTotal)) == case["sql_total_text"]
    died = False
    msg = None
    try:
        msg = chk.confirmcompra(ses, case["user"], co.md5(case["password"]))
    except co.Die:
        died = True
    assert died == case["die"]
    if not died:
        assert msg == case["message"]
    got_venta = [{k: (f"{v[k]:.2f}" if k == "TotalPagar" else v[k]) for k in ("NumPedido", "NIT", "Descuento", "TotalPagar", "Estado")}
                 for v in store.venta]
    assert got_venta == case["venta"]
    assert [list(r) for r in store.detalle] == case["detalle"]


def test_seed_matches_the_values_of_store_sql():
    s = _seed()
    assert [(r["CodigoProd"], r["Precio"]) for r in s["producto"]] == [("1", "250.00"), ("2", "450.00"), ("3", "1500.00")]
    assert s["cliente"][0]["id_Cliente"] == "7" and s["cliente"][0]["Clave"] == co.md5("123")
    assert s["venta_auto_increment"] == 5


def test_php_float_string_and_number_format():
    assert co.php_float_to_string(2200.0) == "2200"
    assert co.php_float_to_string(1234.5) == "1234.5"
    assert co.php_float_to_string(0.1 + 0.2) == "0.3"
    assert co.php_float_to_string(1234567.89) == "1234567.89"
    # precision=14: fourteen digits print in full, the fifteenth switches to exponent notation
    assert co.php_float_to_string(99999999999999.0) == "99999999999999"
    assert co.php_float_to_string(1e14) == "1.0E+14"
    assert co.php_float_to_string(123456789012345.0) == "1.2345678901234E+14"
    assert co.php_float_to_string(-1e14) == "-1.0E+14"
    assert co.php_float_to_string(99999999999999.5) == "1.0E+14"
    assert co.php_float_to_string(0.00001) == "1.0E-5"
    assert co.php_number_format_2(2200.0) == "2,200.00"
    assert co.php_number_format_2(0) == "0.00"
    assert co.php_number_format_2(1234567.891) == "1,234,567.89"
    assert co.php_number_format_2(1.005) == "1.01"          # PHP pre-rounding: half away from zero
    assert co.php_number_format_2(-1234.5) == "-1,234.50"
    assert str(co.mariadb_decimal_30_2("0.125")) == "0.13"
    assert str(co.mariadb_decimal_30_2("-0.125")) == "-0.13"


def test_double_path_equals_integer_cents_inside_the_bound():
    """DEC2(str14(sum of doubles)) == sum of integer cents for carts far larger than the generator's."""
    rng = random.Random(1234)
    for trial in range(300):
        n = rng.randint(1, 1000)
        prices = [Decimal(rng.randint(1, 500_000_00)) / 100 for _ in range(n)]   # up to 500,000.00
        assert co.equivalence_bound_ok(prices)
        suma = 0
        for p in prices:
            suma = suma + float(f"{p:.2f}")
        total = co.mariadb_decimal_30_2(co.php_float_to_string(float(suma)))
        assert co.to_cents(total) == sum(co.to_cents(p) for p in prices)


def test_bound_rejects_huge_totals():
    assert not co.equivalence_bound_ok([Decimal("999999999999.99")] * 2)


def test_where_the_double_path_and_the_integer_cent_path_part():
    """A checkout whose total crosses 10^12 currency units.  Fourteen significant digits no longer reach the
    cent there, so what process/confirmcompra.php:24 hands to MariaDB loses the cents and, from 10^14 on, is in
    exponent notation and loses whole units too.  Documented, not asserted equal: the engine's integer-cent
    totals are exact, the reference's double path is not, which is why equivalence_bound_ok() stops at 10^12."""
    store = co.Store()
    store.producto["X"] = Decimal("600000000000.37")          # a DECIMAL(30,2) column holds it (store.sql:113)
    store.cliente[("u", co.md5("p"))] = 1
    chk = co.Checkout(store)
    ses = co.Session()
    chk.carrito(ses, "X")
    chk.carrito(ses, "X")                                     # 1,200,000,000,000.74
    assert not co.equivalence_bound_ok([store.producto["X"]] * 2)
    assert co.php_float_to_string(float(ses.sumaTotal)) == "1200000000000.7"      # 14 digits: the cent is gone
    assert chk.confirmcompra(ses, "u", co.md5("p")).startswith("El pedido")
    assert store.venta[0]["TotalPagar"] == Decimal("1200000000000.70")
    assert co.cents_total(store, ["X", "X"]) == 120000000000074                   # the integer-cent path keeps it
    # two orders of magnitude further the reference's text is in exponent notation and rounds to units of ten
    store.producto["Y"] = Decimal("61728394506172.50")
    ses = co.Session()
    chk.carrito(ses, "Y")
    chk.carrito(ses, "Y")                                     # 123,456,789,012,345.00
    assert co.php_float_to_string(float(ses.sumaTotal)) == "1.2345678901234E+14"
    assert co.mariadb_decimal_30_2("1.2345678901234E+14") == Decimal("123456789012340.00")
    assert co.cents_total(store, ["Y", "Y"]) == 12345678901234500


def test_aggregate_py_matches_sqlite_group_by():
    """Independent cross-check of the three aggregates in integer cents with sqlite3."""
    rng = np.random.default_rng(7)
    n_products, n_customers, m = 37, 11, 400
    price = rng.integers(1, 500000, n_products).astype(np.int64)
    lens = rng.integers(0, 8, m)                      # some orders have no lines
    oid = np.repeat(np.arange(m, dtype=np.int32) + 5, lens)
    pidx = rng.integers(-1, n_products, oid.size).astype(np.int32)       # -1 = unknown code
    qty = rng.integers(1, 6, oid.size).astype(np.int32)
    cust = rng.integers(0, n_customers, m).astype(np.int32)
    tot, pq, ct = co.aggregate_py(oid, pidx, qty, cust, 5, n_customers, price)

    db = sqlite3.connect(":memory:")
    db.execute("create table producto (CodigoProd integer primary key, cents integer)")
    db.execute("create table venta (NumPedido integer primary key, NIT integer)")
    db.execute("create table detalle (NumPedido integer, CodigoProd integer, CantidadProductos integer)")
    db.executemany("insert into producto values (?,?)", [(i, int(price[i])) for i in range(n_products)])
    db.executemany("insert into venta values (?,?)", [(5 + i, int(cust[i])) for i in range(m)])
    db.executemany("insert into detalle values (?,?,?)", [(int(a), int(b), int(c)) for a, b, c in zip(oid, pidx, qty)])
    per_order = dict(db.execute("select d.NumPedido, sum(p.cents*d.CantidadProductos) from detalle d join producto p "
                                "on p.CodigoProd=d.CodigoProd group by d.NumPedido").fetchall())
    assert [per_order.get(5 + i, 0) for i in range(m)] == tot
    per_prod = dict(db.execute("select CodigoProd, sum(CantidadProductos) from detalle where CodigoProd>=0 "
                               "group by CodigoProd").fetchall())
    assert [per_prod.get(i, 0) for i in range(n_products)] == pq
    per_cust = dict(db.execute("select v.NIT, sum(p.cents*d.CantidadProductos) from detalle d join producto p on "
                               "p.CodigoProd=d.CodigoProd join venta v on v.NumPedido=d.NumPedido group by v.NIT").fetchall())
    assert [per_cust.get(i, 0) for i in range(n_customers)] == ct


def test_columnar_view_of_checkout_equals_TotalPagar():
    """Replay several carts through the checkout oracle, then recompute venta.TotalPagar from detalle."""
    store = co.Store.from_seed(_seed())
    chk = co.Checkout(store)
    for cart in (["1", "2", "3"], ["1", "1"], ["3"], ["2", "2", "2", "1"]):
        ses = co.Session()
        for code in cart:
            chk.carrito(ses, code)
        assert chk.confirmcompra(ses, "wilow", co.md5("123")).startswith("El pedido")
    col = co.columnar(store)
    tot, pq, ct = co.aggregate_py(col["order_id"], col["product_idx"], col["qty"], col["order_customer"],
                                  col["order_id_base"], col["n_customers"], col["price_cents"])
    assert tot == [co.to_cents(v["TotalPagar"]) for v in store.venta]
    assert pq == [4, 4, 2]      # '1': 1+2+0+1, '2': 1+0+0+3, '3': 1+0+1+0
    assert ct == [sum(tot)]
