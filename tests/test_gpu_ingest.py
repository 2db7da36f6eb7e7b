"""Parity of the device text ingest (through the C ABI) with the C oracle: identical columns, row for row."""
import os

import numpy as np
import pytest

from ac_mpc_amd import ErrorLote, _lib, ingesta
from tests import ingest_cases as ic

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("ACMPC_FUZZ_SEED", "0"))       # shifts the seeds of the generated texts
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(motor, oracle, text, codes, expect_bad=0):
    ingesta.cargar_codigos(motor, codes)
    eoid, epidx, eqty, n, bad = oracle.ingest(text, codes)
    assert bad == expect_bad
    if bad:
        with pytest.raises(ErrorLote) as e:
            ingesta.leer_detalle(motor, text)
        assert e.value.code == _lib.E_DATA and "malformed" in e.value.detail
        return
    oid, pidx, qty = ingesta.leer_detalle(motor, text)
    assert oid.size == n
    np.testing.assert_array_equal(oid, eoid)
    np.testing.assert_array_equal(pidx, epidx)
    np.testing.assert_array_equal(qty, eqty)


def test_committed_dumps(motor, oracle):
    v = np.load(os.path.join(GOLDEN, "synth_1k_zipf.npz"))
    codes = [ic.code_of(p) for p in range(v["price_cents"].size)]
    for name in ("detalle_dump.sql", "detalle_insert_stmt.sql"):
        _same(motor, oracle, open(os.path.join(GOLDEN, name), "rb").read(), codes)
    oid, pidx, qty = ingesta.leer_detalle(motor, open(os.path.join(GOLDEN, "detalle_dump.sql"), "rb").read())
    np.testing.assert_array_equal(oid, v["order_id"])
    np.testing.assert_array_equal(pidx, v["product_idx"])
    np.testing.assert_array_equal(qty, v["qty"])


def test_edge_rows_on_device_match_hand_expectations(motor):
    """Malformed rows are reported; the stored columns still match the expectations row for row."""
    import ctypes as C
    ingesta.cargar_codigos(motor, ic.EDGE_CODES)
    cap = len(ic.EDGE_EXPECT)
    oid, pidx, qty = (np.empty(cap, np.int32) for _ in range(3))
    n, bad = C.c_int64(0), C.c_int64(0)
    rc = motor._lib.acmpc_ingest_detalle(motor._ctx, ic.EDGE_TEXT, len(ic.EDGE_TEXT), oid.ctypes.data, pidx.ctypes.data,
                                         qty.ctypes.data, cap, C.byref(n), C.byref(bad))
    assert rc == _lib.E_DATA and (n.value, bad.value) == (cap, ic.EDGE_BAD)
    assert list(zip(oid.tolist(), pidx.tolist(), qty.tolist())) == ic.EDGE_EXPECT


@pytest.mark.parametrize("style", ["dump", "csv", "tsv_crlf", "quoted", "statements"])
@pytest.mark.parametrize("n", [1, 63, 1500, 3900, 5000, 400_003])   # 1, 1, 3..6, 8..16, 10..21, ~800+ chunks of 12 KiB
def test_styles_and_sizes(motor, oracle, style, n):
    d = oracle.synth(n, 10_000, 50, zipf=1, qty_mode=1)
    pidx = d["product_idx"].copy()
    pidx[::97] = -1
    codes = [ic.code_of(p) for p in range(10_000)]
    text = ic.render(d["order_id"].tolist(), pidx.tolist(), d["qty"].tolist(), style)
    _same(motor, oracle, text, codes)


def test_empty_and_header_only_text(motor, oracle):
    _same(motor, oracle, b"", ["1"])
    _same(motor, oracle, b"-- nothing here\nINSERT INTO `detalle` (`NumPedido`) VALUES\n\n", ["1"])


def test_fuzzed_text_row_for_row(motor, oracle):
    """Random corrupted lines: the device must classify and parse every line exactly like the oracle."""
    import ctypes as C
    rng = np.random.default_rng(SEED + 43)
    alphabet = list(b"0123456789 ,'\"()-;\t\rabSKU-x")
    codes = ["1", "2", "3", "ab", "SKU-1", "a b", "abcdefghijklmnopqrstuvwxyz0123"]
    lines = []
    for i in range(60_000):
        r = rng.random()
        if r < 0.6:
            a, q = int(rng.integers(-50, 2**31)), int(rng.integers(-3, 99))
            c = codes[int(rng.integers(0, len(codes)))] if rng.random() < 0.8 else "zz"
            ln = rng.choice([f"({a}, '{c}', {q}),", f"{a},{c},{q}", f"'{a}'\t\"{c}\"\t'{q}'\r", f"( {a} ,'{c}' , {q} );"]).encode("latin-1")
            if rng.random() < 0.25:                    # the same row behind a statement's front part
                ln = ic.statement_prefix(rng).encode("latin-1") + ln
            if rng.random() < 0.3:
                k = int(rng.integers(0, len(ln)))
                ln = ln[:k] + bytes([alphabet[int(rng.integers(0, len(alphabet)))]]) + ln[k + 1:]
        elif r < 0.95:
            ln = bytes(alphabet[int(j)] for j in rng.integers(0, len(alphabet), int(rng.integers(0, 60))))
        else:
            ln = b"(1, '" + b"x" * int(rng.integers(20, 200)) + b"', 1),"       # long rows, some beyond 120 bytes
        lines.append(ln.replace(b"\n", b" "))
    text = b"\n".join(lines)
    ingesta.cargar_codigos(motor, codes)
    eoid, epidx, eqty, n, bad = oracle.ingest(text, codes)
    oid, pidx, qty = (np.empty(n, np.int32) for _ in range(3))
    gn, gbad = C.c_int64(0), C.c_int64(0)
    rc = motor._lib.acmpc_ingest_detalle(motor._ctx, text, len(text), oid.ctypes.data, pidx.ctypes.data, qty.ctypes.data,
                                         n, C.byref(gn), C.byref(gbad))
    assert rc == _lib.E_DATA and (gn.value, gbad.value) == (n, bad) and bad > 1000
    np.testing.assert_array_equal(oid, eoid)
    np.testing.assert_array_equal(pidx, epidx)
    np.testing.assert_array_equal(qty, eqty)


def test_fuzzed_bytes_full_alphabet(motor, oracle):
    """Every byte value (a vertical tab or 0x01 right after a newline must not look like a line start),
    leading zeros, 9 to 12 digit numbers around the int32 limits, rows of 119 to 122 bytes."""
    import ctypes as C
    rng = np.random.default_rng(SEED + 47)
    codes = ["1", "2", "ab", "SKU-1", "x" * 30, "\x0b5", "caf\xe9"]
    lines = []
    for i in range(80_000):
        r = rng.random()
        if r < 0.35:
            a = rng.choice(["0000000005", "2147483647", "2147483648", "-2147483648", "-2147483649", "4294967301",
                            "99999999999", "000000000000", "0", "-0", "12345678901", str(int(rng.integers(0, 10**10)))])
            q = rng.choice(["1", "007", "-3", "2147483647", "9999999999", "00000000001", ""])
            c = codes[int(rng.integers(0, len(codes)))]
            ln = rng.choice([f"({a}, '{c}', {q}),", f"{a},{c},{q}", f"\"{a}\",\"{c}\",\"{q}\"", f"({a},'{c}',{q});\r"]).encode("latin-1")
        elif r < 0.8:
            ln = bytes(int(b) for b in rng.integers(0, 256, int(rng.integers(0, 40))))
            if rng.random() < 0.5:
                ln = rng.choice([b"5,", b"(7, '", b"-1\t", b"'3',"]) + ln
        else:
            pad = int(rng.integers(100, 112))
            ln = b"(1, '1',  " + b" " * pad + b"1),"                  # 113 + pad - 100 ... around the 120-byte limit
            ln = ln[:int(rng.integers(117, 124))] if rng.random() < 0.5 else ln
        lines.append(ln.replace(b"\n", b"\x0b"))
    text = b"\n".join(lines)
    ingesta.cargar_codigos(motor, codes)
    eoid, epidx, eqty, n, bad = oracle.ingest(text, codes)
    roid, rpidx, rqty, rbad = ic.reference_parse(text, codes)
    assert (roid.size, rbad) == (n, bad) and np.array_equal(roid, eoid) and np.array_equal(rpidx, epidx) and np.array_equal(rqty, eqty)
    oid, pidx, qty = (np.empty(n, np.int32) for _ in range(3))
    gn, gbad = C.c_int64(0), C.c_int64(0)
    rc = motor._lib.acmpc_ingest_detalle(motor._ctx, text, len(text), oid.ctypes.data, pidx.ctypes.data, qty.ctypes.data,
                                         n, C.byref(gn), C.byref(gbad))
    assert rc == _lib.E_DATA and (gn.value, gbad.value) == (n, bad) and bad > 1000 and n - bad > 1000
    np.testing.assert_array_equal(oid, eoid)
    np.testing.assert_array_equal(pidx, epidx)
    np.testing.assert_array_equal(qty, eqty)


def test_dense_rows_empty_chunks_and_long_rows_keep_their_order(motor, oracle):
    """Row bases come from a look-back over the chunks before: mix chunks with thousands of tiny rows (more
    than a chunk can park), chunks without any data row and chunks of long rows, over many chunks."""
    rng = np.random.default_rng(SEED + 53)
    codes = ["1", "2", "3"]
    parts = []
    k = 0
    for block in range(120):
        kind = block % 4
        if kind == 0:
            n = int(rng.integers(3000, 9000))
            parts.append("".join(f"{(k + i) % 10},{1 + (k + i) % 3},{i % 7}\n" for i in range(n)))      # 6-byte rows
        elif kind == 1:
            parts.append("-- no data here\n" * int(rng.integers(1500, 4000)))                             # > 1 empty chunk
            n = 0
        elif kind == 2:
            n = int(rng.integers(200, 600))
            parts.append("".join(f"({k + i},  '{1 + i % 3}',      {i}    );\n" for i in range(n)))
        else:
            n = int(rng.integers(1, 40))
            parts.append("".join(f"( {k + i} , '{'x' * 30}' , {i} ),\r\n" for i in range(n)))
        k += n
    text = "".join(parts).encode()
    assert len(text) > 40 * 16384
    _same(motor, oracle, text, codes)


@pytest.mark.parametrize("blocks", [1, 3, 8, 20000])
def test_any_grid_size_gives_the_same_rows(monkeypatch, oracle, blocks):
    """Chunk tickets are drawn per stripe, so a chunk can wait for one that no workgroup has drawn yet; what
    keeps that safe is that every stripe always has a running workgroup.  Run the ingest with fewer workgroups
    than stripes (1, 3), exactly one per stripe (8) and far more than the chip holds at once (the grid is then
    one workgroup per chunk): the same rows as the oracle, and the bounded wait never gives up."""
    from ac_mpc_amd import MotorLote
    monkeypatch.setenv("ACMPC_ING_BLOCKS", str(blocks))
    m = MotorLote(0)
    try:
        n = 1_200_003                                      # about 30 MB of text: 2.5 k chunks of 12 KiB
        d = oracle.synth(n, 10_000, 50, zipf=1, qty_mode=1)
        codes = [ic.code_of(p) for p in range(10_000)]
        text = ic.render(d["order_id"].tolist(), d["product_idx"].tolist(), d["qty"].tolist(), "dump")
        assert len(text) // 12288 > 2000
        _same(m, oracle, text, codes)                      # (an ACMPC_ST_INTERNAL would surface as ErrorLote)
    finally:
        m.cerrar()


def test_chunks_either_side_of_the_rows_by_rank_limit(motor, oracle):
    """A 12 KiB chunk with at most 256 data rows hands them to the threads by rank; a chunk with more leaves every
    row to the thread in whose 48 bytes it starts.  Rows of exactly 48 bytes put 256 rows in every chunk; a byte
    more or less per row moves chunks to either side of the limit, and so do stretches of other row lengths
    between them.  Statement lines as the reference writes them (library/consulSQL.php:30) are mixed with
    statement lines whose front part is not the reference's, which take the general search."""
    rng = np.random.default_rng(SEED + 59)
    codes = ["1", "2", "3", "SKU-1"]
    parts, k = [], 0

    def padded(n_rows, width):
        nonlocal k
        out = []
        for i in range(n_rows):
            body = f"({k + i}, '{codes[(k + i) % 4]}', {1 + i % 9}"
            out.append(body + " " * (width - 2 - len(body)) + ")\n")      # (spaces may stand in front of the ')')
        k += n_rows
        return "".join(out)

    for block in range(60):
        kind = block % 6
        if kind == 0:
            parts.append(padded(int(rng.integers(300, 900)), 48))          # exactly 256 rows a chunk
        elif kind == 1:
            parts.append(padded(int(rng.integers(300, 900)), 47))          # 261 rows a chunk: every thread its own rows
        elif kind == 2:
            parts.append(padded(int(rng.integers(300, 900)), 49))          # 250 rows a chunk: rows by rank
        elif kind == 3:
            n = int(rng.integers(100, 400))
            parts.append("".join(f"insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('{k + i}', '{codes[i % 4]}', '1')\n"
                                 for i in range(n)))
            k += n
        elif kind == 4:
            n = int(rng.integers(100, 400))
            fronts = ["INSERT INTO detalle (NumPedido, CodigoProd, CantidadProductos) VALUES",
                      "insert into detalle (NumPedido, CodigoProd, CantidadProductos)  VALUES ",
                      "insert into detalle (NumPedido, CodigoProd, CantidadProductos) values",
                      "insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUE",      # no row: skipped
                      "insert into detalle (NumPedido, CodigoProd, CantidadProductoz) VALUES"]
            parts.append("".join(f"{fronts[i % 5]}('{k + i}', '{codes[i % 4]}', '1')\n" for i in range(n)))
            k += n
        else:
            parts.append("-- nothing\n" * int(rng.integers(1000, 3000)))   # chunks without any row
    text = "".join(parts).encode()
    assert len(text) > 100 * 12288
    _same(motor, oracle, text, codes)


def test_capacity_too_small_is_reported(motor):
    ingesta.cargar_codigos(motor, ["1"])
    with pytest.raises(ErrorLote) as e:
        ingesta.leer_detalle(motor, b"1,1,1\n2,1,1\n3,1,1\n", capacidad=2)
    assert e.value.code == _lib.E_DATA and "capacity" in e.value.detail


def test_ingest_then_aggregate_equals_oracle(motor, oracle):
    """Text -> columns -> totals, all on the device, against oracle(parse) -> oracle(aggregate)."""
    import torch
    n, p, c = 1_000_003, 10_000, 100_000
    d = oracle.synth(n, p, c, zipf=1, qty_mode=1)
    codes = [ic.code_of(i) for i in range(p)]
    text = ic.render(d["order_id"].tolist(), d["product_idx"].tolist(), d["qty"].tolist(), "dump")
    ingesta.cargar_codigos(motor, codes)
    price = oracle.synth_prices(p)
    motor.cargar_productos(price)
    dev = torch.device("cuda", motor.device)
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    oid, pidx, qty = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    nrows = torch.zeros(1, dtype=torch.int64, device=dev)
    ingesta.leer_detalle_device(motor, t, oid, pidx, qty, nrows)
    cust = torch.from_numpy(d["order_customer"]).to(dev)
    tot = torch.empty(cust.numel(), dtype=torch.int64, device=dev)
    pq = torch.empty(p, dtype=torch.int64, device=dev)
    ct = torch.empty(c, dtype=torch.int64, device=dev)
    motor.agregar_device(oid, pidx, qty, cust, 5, c, tot, pq, ct)
    assert motor.estado() == 0 and int(nrows.item()) == n
    etot, epq, ect, _ = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, c, price)
    np.testing.assert_array_equal(tot.cpu().numpy(), etot)
    np.testing.assert_array_equal(pq.cpu().numpy(), epq)
    np.testing.assert_array_equal(ct.cpu().numpy(), ect)


def test_page_locked_text_and_columns(motor, oracle):
    """Text held in a page-locked uint8 array and columns received in page-locked arrays: same rows."""
    codes = [ic.code_of(p) for p in range(50)]
    rng = np.random.default_rng(5)
    n = 20_000
    oid = np.sort(rng.integers(5, 4000, n)).tolist()
    text = ic.render(oid, rng.integers(0, 50, n).tolist(), rng.integers(1, 9, n).tolist(), "dump")
    ingesta.cargar_codigos(motor, codes)
    locked = motor.reservar_host(len(text), np.uint8)
    locked[:] = np.frombuffer(text, np.uint8)
    cols = tuple(motor.reservar_host(n + 3, np.int32) for _ in range(3))
    got = ingesta.leer_detalle(motor, locked, salida=cols)
    eoid, epidx, eqty, rows, bad = oracle.ingest(text, codes)
    assert rows == n and bad == 0 and got[0].size == n
    for g, e, c in zip(got, (eoid, epidx, eqty), cols):
        np.testing.assert_array_equal(g, e)
        assert g.ctypes.data == c.ctypes.data          # views of the caller's arrays, not copies
    with pytest.raises(ErrorLote):
        ingesta.leer_detalle(motor, locked.astype(np.int32))
    for a in (locked,) + cols:
        motor.liberar_host(a)
