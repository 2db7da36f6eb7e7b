"""GPU parity of the text emit through the C ABI: byte-identical to the C oracle."""
import numpy as np
import pytest

from tests import emit_cases as ec
from tests import ingest_cases as ic

pytestmark = pytest.mark.gpu


@pytest.fixture()
def em():
    from ac_mpc_amd import emision
    return emision


@pytest.mark.parametrize("style", [0, 1, 2])
def test_totals_edge_values(motor, oracle, em, style):
    vals = ec.edge_values()
    assert em.totales(motor, vals, 5, style) == oracle.emit_totals(vals, 5, style) == ec.totals_text(vals.tolist(), 5, style)


@pytest.mark.parametrize("style", [0, 1, 2])
@pytest.mark.parametrize("n,base", [(1, 5), (255, 5), (256, -7), (257, 0), (4096, 999_990), (4097, -2147483648), (100_003, 2147383644)])
def test_totals_sizes(motor, oracle, em, style, n, base):
    vals = ec.random_values(n, n)
    assert em.totales(motor, vals, base, style) == oracle.emit_totals(vals, base, style)


@pytest.mark.parametrize("style", [0, 1, 2])
@pytest.mark.parametrize("base", [0, 5, 9_990, 99_999_000, 99_999_999, 100_000_000])
def test_totals_common_rows_with_a_few_others_between(motor, oracle, em, style, base):
    """Amounts of 0 .. 2^32 - 1 cents with ids below 10^8 are formatted a word at a time, a whole wave of them
    together; a wave that holds any other row is formatted byte by byte.  Every digit count of both numbers, the
    limits of the common case, waves of either kind side by side in a batch, ids that cross 10^8 on the way."""
    rng = np.random.default_rng(41 + style)
    n = 20_011
    mag = rng.integers(0, 33, size=n)
    vals = (rng.integers(0, 2**32, size=n, dtype=np.int64) >> (32 - mag)).astype(np.int64)
    vals[:12] = [0, 1, 9, 10, 99, 100, 101, 999, 1000, 2**32 - 1, 2**32 - 2, 4_294_967_200]
    for k in range(2, 10):
        vals[20 + 3 * k: 23 + 3 * k] = [10**k - 1, 10**k, 10**k + 1]
    vals[[700, 701, 5_000, 12_345, 20_010]] = [2**32, -1, -220000, 2**40, 2**63 - 1]      # these waves go byte by byte
    text = em.totales(motor, vals, base, style)
    assert text == oracle.emit_totals(vals, base, style)
    assert text.startswith(ec.totals_text(vals[:200].tolist(), base, style))


def test_totals_empty_and_bad_arguments(motor, em):
    from ac_mpc_amd import ErrorLote
    assert em.totales(motor, np.zeros(0, np.int64), 5) == b""
    with pytest.raises(ErrorLote):
        em.totales(motor, np.zeros(4, np.int64), 5, 7)                 # unknown style
    with pytest.raises(ErrorLote):
        em.totales(motor, np.zeros(4, np.int64), 2147483645)           # ids would leave int32


@pytest.mark.parametrize("style", [0, 1])
def test_stock_rows(motor, oracle, em, style):
    from ac_mpc_amd import ingesta
    p = 10_000
    codes = [ic.code_of(i) for i in range(p)]
    codes[17] = "abcdefghijklmnopqrstuvwxyz0123"
    ingesta.cargar_codigos(motor, codes)
    vals = ec.random_values(p, 11)
    vals[::5] = 0
    vals[:ec.edge_values().size] = ec.edge_values()
    got = em.stock(motor, vals, style)
    assert got == oracle.emit_stock(vals, codes, style) == ec.stock_text(vals.tolist(), codes, style)
    assert em.stock(motor, np.zeros(p, np.int64), style) == b""


def test_stock_needs_codes_and_rejects_unquotable_ones(em):
    from ac_mpc_amd import ErrorLote, MotorLote, ingesta
    m = MotorLote(0)
    with pytest.raises(ErrorLote):
        em.stock(m, np.ones(3, np.int64))                               # no code list yet
    ingesta.cargar_codigos(m, ["a", "it's", "c d"])
    with pytest.raises(ErrorLote):
        em.stock(m, np.ones(3, np.int64), em.SQL)                       # a quote inside a code
    with pytest.raises(ErrorLote):
        em.stock(m, np.ones(3, np.int64), em.CSV)                       # a space inside a code
    with pytest.raises(ErrorLote):
        em.stock(m, np.ones(2, np.int64), em.SQL)                       # length differs from the code list
    ingesta.cargar_codigos(m, ["a", "b", "c d"])
    assert em.stock(m, np.array([1, 0, -2]), em.SQL) == ec.stock_text([1, 0, -2], ["a", "b", "c d"], 0)


def test_capacity_too_small_is_reported(motor, oracle):
    import ctypes as C
    from ac_mpc_amd import _lib
    vals = ec.random_values(3000, 5)
    full = oracle.emit_totals(vals, 5, 0)
    cap = len(full) // 2 + 3
    buf = C.create_string_buffer(cap + 64)
    n = C.c_int64(0)
    v = np.ascontiguousarray(vals)
    rc = motor._lib.acmpc_emit_totals(motor._ctx, v.ctypes.data, v.size, 5, 0, buf, cap, C.byref(n))
    assert rc == _lib.E_DATA and n.value == len(full)
    assert buf.raw[:cap] == full[:cap] and buf.raw[cap:] == b"\0" * 64
    assert motor.estado() == 0                                        # the flag was consumed by the call


def test_text_into_a_page_locked_array(motor, oracle, em):
    from ac_mpc_amd import ErrorLote, _lib
    vals = motor.reservar_host(5000, np.int64)
    vals[:] = ec.random_values(5000, 9)
    out = motor.reservar_host(5000 * em.MAX_FILA_TOTALES, np.uint8)
    got = em.totales(motor, vals, 5, em.SQL, salida=out)
    assert got.ctypes.data == out.ctypes.data and got.tobytes() == oracle.emit_totals(vals, 5, 0)
    small = motor.reservar_host(100, np.uint8)
    with pytest.raises(ErrorLote) as e:
        em.totales(motor, vals, 5, em.SQL, salida=small)
    assert e.value.code == _lib.E_DATA
    with pytest.raises(ErrorLote):
        em.totales(motor, vals, 5, em.SQL, salida=np.zeros(10, np.int32))
    for a in (vals, out, small):
        motor.liberar_host(a)


@pytest.mark.parametrize("style", [0, 1])
def test_device_totals_at_bench_size_properties(motor, oracle, em, style):
    """25 M orders (config-3 order count): size, row count and windows of the text against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    m = 25_000_000
    g = torch.Generator(device=dev); g.manual_seed(3)
    tot = torch.randint(100, 3_500_000, (m,), dtype=torch.int64, device=dev, generator=g)
    tot[::1000] = 0
    tot[7::5000] = -tot[7::5000] * 1_000_000_007
    text = torch.empty(m * 83, dtype=torch.uint8, device=dev)
    nb = torch.zeros(1, dtype=torch.int64, device=dev)
    em.totales_device(motor, tot, 5, style, text, nb)
    assert motor.estado() == 0
    n = int(nb.item())
    # expected size from digit counts computed independently with torch
    ids = torch.arange(5, 5 + m, dtype=torch.int64, device=dev)
    def digits(x):
        x = x.abs().clamp(min=1).to(torch.float64)
        d = torch.floor(torch.log10(x)).to(torch.int64) + 1
        # float log10 can be off by one next to powers of ten: fix up exactly
        p = torch.pow(torch.tensor(10, dtype=torch.int64, device=dev), d.clamp(max=18))
        d = torch.where((d <= 18) & (x.to(torch.int64) >= p), d + 1, d)
        p1 = torch.pow(torch.tensor(10, dtype=torch.int64, device=dev), (d - 1).clamp(max=18))
        return torch.where(x.to(torch.int64) < p1, d - 1, d)
    units = tot.abs() // 100
    fixed = (29 + 19 + 3) if style == 0 else 2
    expect = int((digits(units) + 3 + (tot < 0).to(torch.int64) + digits(ids) + fixed).sum().item())
    assert n == expect
    assert int((text[:n] == 10).sum().item()) == m
    head = oracle.emit_totals(tot[:5000].cpu().numpy(), 5, style)
    assert bytes(text[:len(head)].cpu().numpy()) == head
    tail = oracle.emit_totals(tot[-5000:].cpu().numpy(), 5 + m - 5000, style)
    assert bytes(text[n - len(tail):n].cpu().numpy()) == tail


@pytest.mark.parametrize("style", [0, 1, 2])
def test_device_totals_column_that_is_only_8_byte_aligned(motor, oracle, em, style):
    """The totals may start anywhere an int64 may: a view one element into a tensor (the CSV kernel reads a
    thread's four values 16 bytes at a time only when the column allows it)."""
    import torch
    dev = torch.device("cuda", 0)
    vals = ec.random_values(10_007, 77)
    whole = torch.zeros(len(vals) + 1, dtype=torch.int64, device=dev)
    col = whole[1:]
    col.copy_(torch.from_numpy(vals))
    assert col.data_ptr() % 16 == 8
    text = torch.empty(len(vals) * 96, dtype=torch.uint8, device=dev)
    nb = torch.zeros(1, dtype=torch.int64, device=dev)
    em.totales_device(motor, col, -3, style, text, nb)
    assert motor.estado() == 0
    n = int(nb.item())
    assert bytes(text[:n].cpu().numpy()) == oracle.emit_totals(vals, -3, style)
