"""The C ingest oracle against hand-written expectations, the committed dump fixtures and an independent
regex parser (CPU only)."""
import os

import numpy as np
import pytest

from tests import ingest_cases as ic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_edge_text_hand_expectations(oracle):
    oid, pidx, qty, n, bad = oracle.ingest(ic.EDGE_TEXT, ic.EDGE_CODES)
    assert list(zip(oid.tolist(), pidx.tolist(), qty.tolist())) == ic.EDGE_EXPECT
    assert (n, bad) == (len(ic.EDGE_EXPECT), ic.EDGE_BAD)
    roid, rpidx, rqty, rbad = ic.reference_parse(ic.EDGE_TEXT, ic.EDGE_CODES)
    assert list(zip(roid.tolist(), rpidx.tolist(), rqty.tolist())) == ic.EDGE_EXPECT and rbad == ic.EDGE_BAD


def test_committed_dump_reproduces_the_columns(oracle):
    v = np.load(os.path.join(GOLDEN, "synth_1k_zipf.npz"))
    codes = [ic.code_of(p) for p in range(v["price_cents"].size)]
    text = open(os.path.join(GOLDEN, "detalle_dump.sql"), "rb").read()
    oid, pidx, qty, n, bad = oracle.ingest(text, codes)
    assert (n, bad) == (v["order_id"].size, 0)
    np.testing.assert_array_equal(oid, v["order_id"])
    np.testing.assert_array_equal(pidx, v["product_idx"])
    np.testing.assert_array_equal(qty, v["qty"])
    stmt = open(os.path.join(GOLDEN, "detalle_insert_stmt.sql"), "rb").read()
    oid, pidx, qty, n, bad = oracle.ingest(stmt, codes)
    assert (n, bad) == (40, 0)
    np.testing.assert_array_equal(oid, v["order_id"][:40])
    np.testing.assert_array_equal(pidx, v["product_idx"][:40])
    assert qty.tolist() == [1] * 40                    # process/confirmcompra.php:36 always writes '1'


@pytest.mark.parametrize("style", ["dump", "csv", "tsv_crlf", "quoted", "statements"])
def test_styles_round_trip(oracle, style):
    d = oracle.synth(20_000, 500, 50, zipf=1, qty_mode=1)
    pidx = d["product_idx"].copy()
    pidx[::97] = -1
    codes = [ic.code_of(p) for p in range(500)]
    text = ic.render(d["order_id"].tolist(), pidx.tolist(), d["qty"].tolist(), style)
    oid, gp, qty, n, bad = oracle.ingest(text, codes)
    assert (n, bad) == (d["order_id"].size, 0)
    np.testing.assert_array_equal(oid, d["order_id"])
    np.testing.assert_array_equal(gp, pidx)
    np.testing.assert_array_equal(qty, d["qty"])


def test_fuzzed_rows_agree_with_regex_reference(oracle):
    rng = np.random.default_rng(41)
    alphabet = list(b"0123456789 ,'\"()-;\t\rabSKU-x")
    codes = ["1", "2", "3", "ab", "SKU-1", "a b"]
    lines = []
    for i in range(4000):
        if rng.random() < 0.5:
            a, q = int(rng.integers(-50, 10**7)), int(rng.integers(-3, 99))
            c = codes[int(rng.integers(0, len(codes)))] if rng.random() < 0.8 else "zz"
            ln = rng.choice([f"({a}, '{c}', {q}),", f"{a},{c},{q}", f"'{a}'\t\"{c}\"\t'{q}'\r", f"( {a} ,'{c}' , {q} );"])
            ln = ln.encode("latin-1")
            if rng.random() < 0.25:                    # the same row behind a statement's front part
                ln = ic.statement_prefix(rng).encode("latin-1") + ln
            if rng.random() < 0.3:                     # corrupt one byte
                k = int(rng.integers(0, len(ln)))
                ln = ln[:k] + bytes([alphabet[int(rng.integers(0, len(alphabet)))]]) + ln[k + 1:]
        else:
            ln = bytes(alphabet[int(j)] for j in rng.integers(0, len(alphabet), int(rng.integers(0, 40))))
        lines.append(ln.replace(b"\n", b" "))
    text = b"\n".join(lines)
    oid, pidx, qty, n, bad = oracle.ingest(text, codes)
    roid, rpidx, rqty, rbad = ic.reference_parse(text, codes)
    assert n == roid.size and bad == rbad
    np.testing.assert_array_equal(oid, roid)
    np.testing.assert_array_equal(pidx, rpidx)
    np.testing.assert_array_equal(qty, rqty)


def test_capacity_smaller_than_rows(oracle):
    text = b"1,1,1\n2,1,1\n3,1,1\n"
    oid, pidx, qty, n, bad = oracle.ingest(text, ["1"], capacity=2)
    assert n == 3 and oid.tolist() == [1, 2]
