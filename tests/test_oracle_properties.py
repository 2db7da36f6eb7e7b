"""Property tests of the CPU oracles (hypothesis): the checker itself is checked from several sides."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import checkout_oracle as co
from tests import emit_cases as ec
from tests import ingest_cases as ic
from tests.oracle_binding import Oracle

ORC = Oracle()
CODES = ["1", "2", "3", "SKU-00003", "a b", "x" * 30]

line_alphabet = st.sampled_from(list(b"0123456789 ,'\"()-;\t\rabSKU-x\x0b\x00\xe9"))
clean_row = st.builds(
    lambda a, c, q, shape: [f"({a}, '{c}', {q}),", f"{a},{c},{q}", f"'{a}'\t\"{c}\"\t'{q}'\r", f"( {a} ,'{c}' , {q} );"][shape].encode("latin-1"),
    st.integers(-2**31 - 5, 2**31 + 5), st.sampled_from(CODES + ["zz", ""]), st.integers(-12, 10**11), st.integers(0, 3))
any_line = st.one_of(clean_row, st.lists(line_alphabet, max_size=50).map(bytes),
                     st.tuples(clean_row, st.integers(0, 60), line_alphabet).map(
                         lambda t: t[0][:t[1]] + bytes([t[2]]) + t[0][t[1] + 1:]))


@settings(max_examples=150, deadline=None)
@given(st.lists(any_line, max_size=40), st.booleans())
def test_ingest_oracle_equals_the_regex_restatement(lines, trailing_newline):
    text = b"\n".join(ln.replace(b"\n", b" ") for ln in lines) + (b"\n" if trailing_newline else b"")
    oid, pidx, qty, n, bad = ORC.ingest(text, CODES)
    roid, rpidx, rqty, rbad = ic.reference_parse(text, CODES)
    assert (n, bad) == (roid.size, rbad)
    assert np.array_equal(oid, roid) and np.array_equal(pidx, rpidx) and np.array_equal(qty, rqty)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(-2**63, 2**63 - 1), max_size=60), st.integers(-2**31, 2**31 - 61), st.integers(0, 1))
def test_emit_oracle_equals_python_formatting(values, base, style):
    v = np.array(values, dtype=np.int64)
    assert ORC.emit_totals(v, base, style) == ec.totals_text(values, base, style)
    codes = [f"c{i}" for i in range(len(values))]
    if values:
        assert ORC.emit_stock(v, codes, style) == ec.stock_text(values, codes, style)


rows = st.lists(st.tuples(st.integers(0, 30), st.integers(-1, 6), st.integers(-2**31, 2**31 - 1)), max_size=80)


@settings(max_examples=100, deadline=None)
@given(rows, rows, st.lists(st.integers(-2**62, 2**62), min_size=6, max_size=6), st.lists(st.integers(0, 4), min_size=31, max_size=31))
def test_aggregate_is_additive_over_disjoint_row_sets_and_matches_python(rows_a, rows_b, price, cust):
    """aggregate(A ++ B) = aggregate(A) + aggregate(B) modulo 2^64, for all three outputs; C equals Python."""
    def run(rs):
        rs = sorted(rs)
        oid = np.array([r[0] + 5 for r in rs], np.int32)
        pidx = np.array([r[1] for r in rs], np.int32)
        qty = np.array([r[2] for r in rs], np.int32)
        out = ORC.aggregate(oid, pidx, qty, np.array(cust, np.int32), 5, 5, np.array(price, np.int64))
        py = co.aggregate_py(oid.tolist(), pidx.tolist(), qty.tolist(), cust, 5, 5, price)
        assert out[3] == 0 and all(a.tolist() == b for a, b in zip(out[:3], py))
        return [x.astype(np.uint64) for x in out[:3]]
    a, b, ab = run(rows_a), run(rows_b), run(rows_a + rows_b)
    for x, y, z in zip(a, b, ab):
        assert np.array_equal(x + y, z)            # uint64 addition wraps
