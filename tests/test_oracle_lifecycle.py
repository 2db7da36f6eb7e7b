"""The lifecycle columns in the two oracles: hand-derived known answers, C against Python on random data."""
import numpy as np
import pytest

from oracle import checkout_oracle as co
from tests import lifecycle_cases as lc


@pytest.mark.parametrize("case", lc.fixture()["cases"], ids=lambda c: c["name"])
def test_known_answers_both_oracles(oracle, case):
    d = lc.fixture()
    price = case.get("price_cents_now", d["price_cents"])
    args = (d["order_id"], d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"], d["n_customers"], price)
    life = (case.get("line_price"), case.get("order_state"), case.get("order_descuento"), case.get("descuento_mode", 0))
    exp = (case["order_total"], case["product_qty"], case["customer_total"])
    assert tuple(map(list, co.aggregate_py(*args, *life))) == exp
    tot, pq, ct, flags = oracle.aggregate_lifecycle(*args, *life)
    assert flags == 0 and (tot.tolist(), pq.tolist(), ct.tolist()) == exp


@pytest.mark.parametrize("mask", lc.MASKS)
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("wide", [False, True])
def test_c_oracle_equals_python_oracle(oracle, mask, mode, wide):
    d = oracle.synth(3000, 50, 40)
    cols = lc.columns(d, 7, mask, wide)
    price = oracle.synth_prices(50)
    args = (d["order_id"], d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"], 40, price)
    tot, pq, ct, flags = oracle.aggregate_lifecycle(*args, cols["line_price"], cols["order_state"], cols["order_descuento"], mode)
    ptot, ppq, pct = co.aggregate_py(*[a.tolist() if hasattr(a, "tolist") else a for a in args],
                                     None if cols["line_price"] is None else cols["line_price"].tolist(),
                                     None if cols["order_state"] is None else cols["order_state"].tolist(),
                                     None if cols["order_descuento"] is None else cols["order_descuento"].tolist(), mode)
    assert flags == 0
    assert tot.tolist() == ptot and pq.tolist() == ppq and ct.tolist() == pct


def test_absent_columns_equal_the_plain_aggregate(oracle):
    d = oracle.synth(5000, 100, 30)
    price = oracle.synth_prices(100)
    args = (d["order_id"], d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"], 30, price)
    a = oracle.aggregate(*args)
    b = oracle.aggregate_lifecycle(*args)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    # a state column of all ones, zero discounts and the table prices as line prices change nothing
    lp = price[np.clip(d["product_idx"], 0, 99)]
    c = oracle.aggregate_lifecycle(*args, lp, np.ones(d["order_customer"].size, np.uint8),
                                   np.zeros(d["order_customer"].size, np.int32), 1)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], c[:3]))


def test_per_cent_discount_is_symmetric_and_exact():
    """disc(-g) == -disc(g); 100 % leaves zero; the rounding point is half a cent."""
    for g, dsc, exp in ((1005, 50, 503), (-1005, 50, -503), (1, 50, 1), (1, 49, 0), (-1, 50, -1), (33333, 10, 3333),
                        (33335, 10, 3334), (10**17 + 1, 100, 10**17 + 1)):
        tot, _, _ = co.aggregate_py([5], [0], [1], [0], 5, 1, [g], None, None, [dsc], 1)
        assert tot[0] == g - exp
