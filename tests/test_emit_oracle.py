"""The C emit oracle against an independent Python restatement of the row formats and against the
known-answer checkouts (what MariaDB prints for the stored DECIMAL(30,2))."""
import json
import os

import numpy as np
import pytest

from oracle import checkout_oracle as co
from tests import emit_cases as ec
from tests import ingest_cases as ic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("style", [0, 1, 2])
def test_totals_edge_and_random(oracle, style):
    for vals, base in ((ec.edge_values(), 5), (ec.random_values(5000, 1), -2147483648), (ec.random_values(300, 2), 2147483347)):
        assert oracle.emit_totals(vals, base, style) == ec.totals_text(vals.tolist(), base, style)


@pytest.mark.parametrize("style", [0, 1])
def test_stock_edge_and_random(oracle, style):
    vals = ec.edge_values()
    codes = [ic.code_of(i) for i in range(vals.size)]
    assert oracle.emit_stock(vals, codes, style) == ec.stock_text(vals.tolist(), codes, style)
    vals = ec.random_values(4000, 3)
    vals[::7] = 0
    codes = [("c%d" % i) * (1 + i % 3) for i in range(vals.size)]
    codes[5] = "abcdefghijklmnopqrstuvwxyz0123"              # 30 bytes
    assert oracle.emit_stock(vals, codes, style) == ec.stock_text(vals.tolist(), codes, style)


def test_capacity_truncates_and_reports_need(oracle):
    vals = ec.edge_values()
    full = oracle.emit_totals(vals, 5, 0)
    assert oracle.emit_totals(vals, 5, 0, capacity=100) == full[:100]
    assert oracle.emit_totals(vals[:0], 5, 0) == b""


def test_known_answer_checkouts_print_like_the_stored_decimal(oracle):
    """The amount text equals str() of the DECIMAL(30,2) the checkout oracle stores in venta.TotalPagar."""
    seed = json.load(open(os.path.join(GOLDEN, "store_seed.json")))
    cases = json.load(open(os.path.join(GOLDEN, "checkout_cases.json"), encoding="utf-8"))["cases"]
    seen = 0
    for case in cases:
        for v in case["venta"]:
            cents = int(co.to_cents(co.mariadb_decimal_30_2(v["TotalPagar"])))
            row = oracle.emit_totals(np.array([cents]), v["NumPedido"], 0).decode()
            assert row == f"update venta set TotalPagar='{v['TotalPagar']}' where NumPedido='{v['NumPedido']}';\n"
            assert oracle.emit_totals(np.array([cents]), v["NumPedido"], 1).decode() == f"{v['NumPedido']},{v['TotalPagar']}\n"
            seen += 1
    assert seen >= 3


def test_amount_text_round_trips_through_decimal():
    for c in ec.edge_values().tolist():
        assert int(co.mariadb_decimal_30_2(ec.amount(c)) * 100) == c


def test_display_style_is_the_reference_fragment(oracle):
    """The display rows against the restatement of PHP's number_format in oracle/checkout_oracle.py (double
    arithmetic, as process/carrito.php:16,19 does it) and against the known-answer checkouts."""
    import json
    import os
    from oracle import checkout_oracle as co
    rng = np.random.default_rng(5)
    cents = np.concatenate([rng.integers(-10**13, 10**13, 3000), np.array([0, 5, 99, 100, 99999, 100000, 220000, -220000])]).astype(np.int64)
    rows = oracle.emit_totals(cents, 5, 2).decode().splitlines()
    for c, row in zip(cents.tolist(), rows):
        assert row == "<tr><td>Subtotal</td><td>$" + co.php_number_format_2(c / 100.0) + "</td></tr>"
    here = os.path.dirname(os.path.abspath(__file__))
    for case in json.load(open(os.path.join(here, "golden", "checkout_cases.json")))["cases"]:
        if "subtotal_display" in case and "sumaTotal" in case:      # e.g. 2200.0 -> "Subtotal $2,200.00"
            cents = int(round(case["sumaTotal"] * 100))
            assert oracle.emit_totals(np.array([cents], np.int64), 5, 2).decode().strip() == \
                "<tr><td>Subtotal</td><td>" + case["subtotal_display"].replace("Subtotal ", "") + "</td></tr>"
