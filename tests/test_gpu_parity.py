"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Bar: bit-exact, every output.

Run on the GPU box: python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread
"""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

from ac_mpc_amd import ErrorLote, _lib, consultasLote, sintetico
from oracle import checkout_oracle as co

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "fused"])
def motor(request):
    """Every parity case runs twice: with the library's own choice of launch sequence, and with the fused
    three-launch sequence forced (by default it is only taken from a few million orders on)."""
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

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VECTORS = sorted(glob.glob(os.path.join(GOLDEN, "synth_*.npz")))


def _check(motor, oracle, oid, pidx, qty, cust, base, c_n, price):
    motor.cargar_productos(price)
    tot, pq, ct = motor.agregar(oid, pidx, qty, cust, base, c_n)
    etot, epq, ect, flags = oracle.aggregate(oid, pidx, qty, cust, base, c_n, price)
    assert flags == 0
    np.testing.assert_array_equal(tot, etot)
    np.testing.assert_array_equal(pq, epq)
    np.testing.assert_array_equal(ct, ect)
    return tot, pq, ct


# ---- committed golden vectors --------------------------------------------------------------

@pytest.mark.parametrize("path", VECTORS, ids=lambda p: os.path.basename(p)[:-4])
def test_golden_vectors(motor, path):
    v = np.load(path)
    motor.cargar_productos(v["price_cents"])
    tot, pq, ct = motor.agregar(v["order_id"], v["product_idx"], v["qty"], v["order_customer"],
                                int(v["order_id_base"]), int(v["n_customers"]))
    np.testing.assert_array_equal(tot, v["order_total_cents"])
    np.testing.assert_array_equal(pq, v["product_qty"])
    np.testing.assert_array_equal(ct, v["customer_total_cents"])


def test_checkout_cases_through_the_engine(motor):
    """The hand-derived carts of the reference's seed data: venta.TotalPagar recomputed from detalle."""
    seed = json.load(open(os.path.join(GOLDEN, "store_seed.json")))
    store = co.Store.from_seed(seed)
    chk = co.Checkout(store)
    for cart in (["1", "2", "3"], ["1", "1"], ["3"], ["2", "2", "2", "1"]):
        ses = co.Session()
        for code in cart:
            chk.carrito(ses, code)
        chk.confirmcompra(ses, "wilow", co.md5("123"))
    codes = [r["CodigoProd"] for r in seed["producto"]]
    cents = [co.to_cents(store.producto[c]) for c in codes]
    consultasLote._motor = motor
    consultasLote.CargarProductos(codes, cents)
    det = store.detalle
    res = consultasLote.AgregarDetalle([r[0] for r in det], [r[1] for r in det], [r[2] for r in det],
                                       [0] * len(store.venta), 5, 1)
    consultasLote._motor = None
    assert res["TotalPagar"].tolist() == [co.to_cents(v["TotalPagar"]) for v in store.venta] == [220000, 50000, 150000, 160000]
    assert res["DeltaStock"].tolist() == [4, 4, 2]
    assert res["TotalCliente"].tolist() == [580000]


def test_unknown_code_adds_nothing(motor):
    consultasLote._motor = motor
    consultasLote.CargarProductos(["1", "2", "3"], [25000, 45000, 150000])
    res = consultasLote.AgregarDetalle([5, 5], ["3", "999"], [1, 1], [0], 5, 1)
    consultasLote._motor = None
    assert res["TotalPagar"].tolist() == [150000]          # process/carrito.php:13-14: no row, no add
    assert res["DeltaStock"].tolist() == [0, 0, 1]


# ---- seeded sizes against the C oracle --------------------------------------------------------

@pytest.mark.parametrize("n,p,c,zipf,qm", [
    (1, 3, 1, 0, 0), (255, 10, 7, 0, 1), (256, 10, 7, 1, 1), (257, 10, 7, 0, 0), (4097, 50, 9, 1, 1),
    (1_000_003, 10_000, 100_000, 0, 1), (1_000_000, 10_000, 100_000, 1, 0), (3_333_333, 12_000, 250_000, 1, 1),
])
def test_seeded_sizes(motor, oracle, n, p, c, zipf, qm):
    d = oracle.synth(n, p, c, zipf=zipf, qty_mode=qm)
    _check(motor, oracle, d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, c, oracle.synth_prices(p))


def test_large_product_and_customer_tables_use_global_path(motor, oracle):
    """P too large for the LDS tables, C too large for the LDS customer ranges."""
    n, p, c = 500_000, 60_000, 3_000_000
    d = oracle.synth(n, p, c, zipf=0, qty_mode=1)
    _check(motor, oracle, d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, c, oracle.synth_prices(p))


@pytest.mark.parametrize("c", [30_000, 39_000, 60_000, 78_001, 150_000, 131_073, 260_000])
def test_number_of_customer_ranges(motor, oracle, c):
    """One to seven customer ranges per stream (a workgroup's LDS holds about 39 k partial sums), under both
    launch sequences: with the record block riding along, the plan leaves room for it on every XCD."""
    n, p = 1_500_003, 3_000
    d = oracle.synth(n, p, c, zipf=1, qty_mode=1)
    _check(motor, oracle, d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, c, oracle.synth_prices(p))


# ---- edge cases of the data contract -----------------------------------------------------------

def test_empty_input(motor, oracle):
    price = np.array([100, 200, 300], np.int64)
    e = np.zeros(0, np.int32)
    _check(motor, oracle, e, e, e, e, 5, 4, price)                                   # no orders, no lines
    _check(motor, oracle, e, e, e, np.array([1, 2, 3], np.int32), 5, 4, price)       # orders without lines


def test_orders_without_lines_are_zero(motor, oracle):
    rng = np.random.default_rng(3)
    m = 50_000
    lens = rng.integers(0, 4, m)
    lens[rng.integers(0, m, 200)] = 0
    lens[1000:9000] = 0                                  # a long run of empty orders inside
    lens[:300] = 0                                       # and at both ends
    lens[-300:] = 0
    oid = np.repeat(np.arange(m, dtype=np.int32) + 5, lens)
    pidx = rng.integers(0, 100, oid.size).astype(np.int32)
    qty = rng.integers(1, 4, oid.size).astype(np.int32)
    cust = rng.integers(0, 77, m).astype(np.int32)
    _check(motor, oracle, oid, pidx, qty, cust, 5, 77, rng.integers(1, 10**6, 100).astype(np.int64))


def test_sparse_lines_in_a_huge_order_range(motor, oracle):
    """Three lines, four million orders: the zero fill of absent orders must be cooperative, not serial."""
    m = 4_000_000
    oid = np.array([5 + 10, 5 + 2_000_000, 5 + m - 2], np.int32)
    pidx = np.array([0, 1, 2], np.int32)
    qty = np.array([1, 2, 3], np.int32)
    cust = (np.arange(m) % 13).astype(np.int32)
    _check(motor, oracle, oid, pidx, qty, cust, 5, 13, np.array([7, 11, 13], np.int64))


def test_one_order_spanning_every_wave(motor, oracle):
    n = 2_000_000
    oid = np.full(n, 5, np.int32)
    rng = np.random.default_rng(5)
    pidx = rng.integers(0, 500, n).astype(np.int32)
    qty = np.ones(n, np.int32)
    _check(motor, oracle, oid, pidx, qty, np.array([0], np.int32), 5, 1, rng.integers(1, 10**5, 500).astype(np.int64))


def test_ragged_orders_long_and_short(motor, oracle):
    rng = np.random.default_rng(11)
    lens = np.concatenate([rng.integers(1, 8, 5000), [300_000], rng.integers(1, 3, 5000), [70_000, 1, 1, 90_000],
                           rng.integers(1, 2000, 300)])
    m = lens.size
    oid = np.repeat(np.arange(m, dtype=np.int32) + 1_000_000, lens)
    pidx = rng.integers(-1, 10_000, oid.size).astype(np.int32)
    qty = rng.integers(-2, 9, oid.size).astype(np.int32)
    cust = rng.integers(0, 100_000, m).astype(np.int32)
    _check(motor, oracle, oid, pidx, qty, cust, 1_000_000, 100_000, oracle.synth_prices(10_000))


def test_int64_prices_and_wraparound(motor, oracle):
    rng = np.random.default_rng(13)
    n, p = 300_000, 2000
    d = oracle.synth(n, p, 500, zipf=1, qty_mode=1)
    price = rng.integers(-2**62, 2**62, p).astype(np.int64)
    qty = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    _check(motor, oracle, d["order_id"], d["product_idx"], qty, d["order_customer"], 5, 500, price)


@pytest.mark.parametrize("dear_every", [1, 7, 5000])
def test_order_totals_either_side_of_2_to_the_32(motor, oracle, dear_every):
    """Finished totals wait in 32-bit slots; a tile in which one may not fit writes 64-bit totals directly.
    Prices fit 32 bits and quantities are small, so nothing else sends these tiles the other way: every
    product dear, one in seven, and one in 5000 (most tiles staged, a few direct, changing back and forth)."""
    rng = np.random.default_rng(29 + dear_every)
    n, p = 700_001, 3000
    d = oracle.synth(n, p, 800, zipf=0, qty_mode=1)
    price = rng.integers(1, 400_000, p).astype(np.int64)
    price[::dear_every] = rng.integers(2**30, 2**31 - 1, price[::dear_every].size)
    tot, _, _ = _check(motor, oracle, d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, 800, price)
    assert (tot >= 2**32).any() and (dear_every == 1 or (tot < 2**32).any())


def test_order_totals_of_exactly_2_to_the_32_and_its_neighbours(motor, oracle):
    price = np.array([2**31 - 1, 1, 2, 300], np.int64)
    kinds = [([0, 1], [2, 1]), ([0, 2], [2, 1]), ([0, 2, 1], [2, 1, 1]), ([3, 3], [1, 2])]   # 2^32-1, 2^32, 2^32+1, 900
    rng = np.random.default_rng(31)
    pick = rng.integers(0, 4, 40_000)
    pick[:3000] = 3                                        # a stretch of staged tiles first
    oid = np.concatenate([np.full(len(kinds[k][0]), 5 + i, np.int32) for i, k in enumerate(pick)])
    pidx = np.concatenate([np.array(kinds[k][0], np.int32) for k in pick])
    qty = np.concatenate([np.array(kinds[k][1], np.int32) for k in pick])
    tot, _, _ = _check(motor, oracle, oid, pidx, qty, np.zeros(pick.size, np.int32), 5, 1, price)
    assert set(np.unique(tot).tolist()) == {900, 2**32 - 1, 2**32, 2**32 + 1}


def test_hot_key_quantity_carries(motor, oracle):
    """One product takes every line with quantities near 2^31: the 32-bit LDS partials must carry."""
    n = 600_000
    oid = np.repeat(np.arange(n // 4, dtype=np.int32) + 5, 4)
    pidx = np.zeros(n, np.int32)
    qty = np.full(n, 2_000_000_000, np.int32)
    qty[::3] = -1_999_999_999
    cust = np.zeros(n // 4, np.int32)
    _check(motor, oracle, oid, pidx, qty, cust, 5, 3, np.array([3, 5], np.int64))


def test_negative_order_ids_and_base(motor, oracle):
    rng = np.random.default_rng(17)
    m = 3000
    lens = rng.integers(1, 6, m)
    oid = np.repeat(np.arange(m, dtype=np.int32) - 1500, lens)
    pidx = rng.integers(0, 9, oid.size).astype(np.int32)
    qty = np.ones(oid.size, np.int32)
    _check(motor, oracle, oid, pidx, qty, rng.integers(0, 5, m).astype(np.int32), -1500, 5,
           rng.integers(1, 999, 9).astype(np.int64))


def test_contract_violations_are_reported_not_fatal(motor):
    motor.cargar_productos(np.array([100, 200], np.int64))
    with pytest.raises(ErrorLote) as e:
        motor.agregar([5, 6, 9], [0, 1, 0], [1, 1, 1], [0, 0], 5, 1)       # order id outside [5, 7)
    assert e.value.code == _lib.E_DATA and "order_id outside" in e.value.detail
    with pytest.raises(ErrorLote) as e:
        motor.agregar([6, 5], [0, 1], [1, 1], [0, 0], 5, 1)                # unsorted
    assert e.value.code == _lib.E_DATA and "non-decreasing" in e.value.detail
    with pytest.raises(ErrorLote) as e:
        motor.agregar([5, 6], [0, 1], [1, 1], [0, 3], 5, 1)                # customer outside [0, 1)
    assert e.value.code == _lib.E_DATA and "order_customer" in e.value.detail
    tot, pq, ct = motor.agregar([5, 6], [0, 1], [1, 1], [0, 0], 5, 1)      # the context still works
    assert tot.tolist() == [100, 200] and pq.tolist() == [1, 1] and ct.tolist() == [300]


def test_aggregate_before_products_is_a_state_error():
    from ac_mpc_amd import MotorLote
    m = MotorLote(0)
    try:
        with pytest.raises(ErrorLote) as e:
            m.agregar([5], [0], [1], [0], 5, 1)
        assert e.value.code == _lib.E_STATE
    finally:
        m.cerrar()


def test_sizes_beyond_the_limits_are_refused_before_anything_is_allocated(motor):
    """Counts whose padded forms would leave 32 bits are argument errors, not allocation attempts."""
    import ctypes as C
    lib = _lib.load()
    motor.cargar_productos(np.array([100, 200], np.int64))
    one32 = (C.c_int32 * 4)(5, 0, 1, 0)
    out = (C.c_int64 * 4)()
    ctx = motor._ctx
    rc = lib.acmpc_aggregate(ctx, one32, one32, one32, 1, one32, 1, 5, (1 << 30) + 1, out, out, out)
    assert rc == _lib.E_INVALID and "customers" in lib.acmpc_last_error(ctx).decode()
    rc = lib.acmpc_aggregate(ctx, one32, one32, one32, 1, one32, 1 << 31, 5, 1, out, out, out)
    assert rc == _lib.E_INVALID
    price = (C.c_int64 * 1)(1)
    assert lib.acmpc_set_products(ctx, price, (1 << 28) + 1) == _lib.E_INVALID
    assert lib.acmpc_set_product_codes(ctx, C.create_string_buffer(32), (1 << 21) + 1) == _lib.E_INVALID
    tot, pq, ct = motor.agregar([5, 6], [0, 1], [1, 1], [0, 0], 5, 1)      # the context still works
    assert tot.tolist() == [100, 200] and pq.tolist() == [1, 1] and ct.tolist() == [300]


def test_page_locked_columns_and_results(motor, oracle):
    """Columns filled in place in page-locked arrays, results received in page-locked arrays: same sums."""
    rng = np.random.default_rng(77)
    m, c, p = 3000, 40, 25
    lens = rng.integers(0, 6, m)
    n = int(lens.sum())
    price = rng.integers(1, 90000, p).astype(np.int64)
    motor.cargar_productos(price)
    oid = motor.reservar_host(n, np.int32)
    pidx = motor.reservar_host(n, np.int32)
    qty = motor.reservar_host(n, np.int32)
    cust = motor.reservar_host(m, np.int32)
    out = (motor.reservar_host(m, np.int64), motor.reservar_host(p, np.int64), motor.reservar_host(c, np.int64))
    assert all(a.ctypes.data % 16 == 0 for a in (oid, pidx, qty, cust) + out)
    oid[:] = np.repeat(np.arange(m, dtype=np.int32) + 5, lens)
    pidx[:] = rng.integers(-1, p + 1, n)
    qty[:] = rng.integers(0, 4, n)
    cust[:] = rng.integers(0, c, m)
    tot, pq, ct = motor.agregar(oid, pidx, qty, cust, 5, c, salida=out)
    assert tot is out[0] and pq is out[1] and ct is out[2]
    etot, epq, ect, flags = oracle.aggregate(oid, pidx, qty, cust, 5, c, price)
    assert flags == 0
    np.testing.assert_array_equal(tot, etot)
    np.testing.assert_array_equal(pq, epq)
    np.testing.assert_array_equal(ct, ect)
    motor.liberar_host(oid)
    with pytest.raises(ErrorLote):
        motor.liberar_host(oid)                         # already released
    with pytest.raises(ErrorLote):
        motor.liberar_host(np.zeros(4, np.int32))       # not a block of the context
    with pytest.raises(ErrorLote):
        motor.agregar(pidx, pidx, qty, cust, 5, c, salida=(out[0][:-1], out[1], out[2]))
    empty = motor.reservar_host(0, np.int64)            # a zero-length request is still a valid block
    assert empty.size == 0


@pytest.mark.parametrize("p", [14_000, 20_000, 29_000, 31_000])
@pytest.mark.parametrize("order", [False, True, "auto"])
def test_product_tables_around_the_lds_limits(motor, oracle, p, order):
    """Prices and histogram in LDS, the histogram alone (prices gathered from global memory), neither."""
    rng = np.random.default_rng(p)
    m, c = 60_000, 5_000
    lens = rng.integers(0, 9, m)
    oid = np.repeat(np.arange(m, dtype=np.int32) + 5, lens)
    n = oid.size
    pidx = rng.integers(-1, p + 1, n).astype(np.int32)
    qty = rng.integers(0, 5, n).astype(np.int32)
    cust = rng.integers(0, c, m).astype(np.int32)
    price = rng.integers(1, 3_000_000_000, p).astype(np.int64)          # some beyond int32
    if order is not False:
        perm = rng.permutation(n)
        oid, pidx, qty = oid[perm], pidx[perm], qty[perm]
    motor.orden_lineas(order)
    try:
        motor.cargar_productos(price)
        got = motor.agregar(oid, pidx, qty, cust, 5, c)
    finally:
        motor.orden_lineas(False)
    etot, epq, ect, flags = oracle.aggregate(oid, pidx, qty, cust, 5, c, price)
    assert flags & ~2 == 0
    for g, e in zip(got, (etot, epq, ect)):
        np.testing.assert_array_equal(g, e)


# ---- device-resident path, generator parity, properties at full size ------------------------------

def _device_run(motor, ds):
    import torch
    dev = ds.order_id.device
    tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
    pc = torch.empty(motor.n_products + ds.n_customers, dtype=torch.int64, device=dev)
    pq, ct = pc[:motor.n_products], pc[motor.n_products:]
    motor.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, ds.n_customers,
                         tot, pq, ct)
    assert motor.estado() == 0
    return tot, pq, ct


@pytest.mark.parametrize("zipf,qm", [(0, 0), (1, 1)])
def test_device_generator_and_device_path(motor, oracle, zipf, qm):
    n, p, c = 2_000_003, 10_000, 100_000
    ds = sintetico.generar(motor, n, p, c, zipf=bool(zipf), qty_mode=qm)
    d = oracle.synth(n, p, c, zipf=zipf, qty_mode=qm)
    np.testing.assert_array_equal(ds.order_id.cpu().numpy(), d["order_id"])
    np.testing.assert_array_equal(ds.product_idx.cpu().numpy(), d["product_idx"])
    np.testing.assert_array_equal(ds.qty.cpu().numpy(), d["qty"])
    np.testing.assert_array_equal(ds.order_customer.cpu().numpy(), d["order_customer"])
    tot, pq, ct = _device_run(motor, ds)
    etot, epq, ect, _ = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"], 5, c, ds.price_cents)
    np.testing.assert_array_equal(tot.cpu().numpy(), etot)
    np.testing.assert_array_equal(pq.cpu().numpy(), epq)
    np.testing.assert_array_equal(ct.cpu().numpy(), ect)
    # a second call on the same context must not see leftovers of the first
    tot2, pq2, ct2 = _device_run(motor, ds)
    assert bool((tot2 == tot).all()) and bool((pq2 == pq).all()) and bool((ct2 == ct).all())


def test_shards_merge_to_the_single_device_result(motor, oracle):
    """world=3 shards run one after the other on one GPU; summed vectors equal the unsharded run."""
    n, p, c = 1_500_000, 10_000, 100_000
    whole = sintetico.generar(motor, n, p, c, zipf=True, qty_mode=1)
    tot, pq, ct = _device_run(motor, whole)
    tot, pq, ct = tot.cpu().numpy(), pq.cpu().numpy(), ct.cpu().numpy()
    parts, spq, sct = [], np.zeros(p, np.int64), np.zeros(c, np.int64)
    for g in range(3):
        sh = sintetico.generar(motor, n, p, c, zipf=True, qty_mode=1, rank=g, world=3)
        t, a, b = _device_run(motor, sh)
        parts.append(t.cpu().numpy())
        spq += a.cpu().numpy()
        sct += b.cpu().numpy()
    np.testing.assert_array_equal(np.concatenate(parts), tot)
    np.testing.assert_array_equal(spq, pq)
    np.testing.assert_array_equal(sct, ct)


@pytest.mark.parametrize("zipf", [0, 1])
def test_full_size_100m_lines(motor, oracle, zipf):
    """The benchmark's primary configuration, 100 M lines on one GPU: size-independent invariants, then the
    C oracle itself on the same columns (it needs ~0.2 s for 100M lines)."""
    import torch
    n, p, c = 100_000_000, 10_000, 100_000
    ds = sintetico.generar(motor, n, p, c, zipf=bool(zipf), qty_mode=0)
    tot, pq, ct = _device_run(motor, ds)
    price = torch.from_numpy(ds.price_cents).to(tot.device)
    assert int(pq.sum()) == n                                            # qty == 1 on every line
    assert int(tot.sum()) == int(ct.sum()) == int((pq * price).sum())    # money is conserved three ways
    assert int(tot.min()) >= 1 and int(tot.max()) <= 7 * 500_000        # 1..7 lines of at most 5,000.00
    oid, pidx, qty, cust = (t.cpu().numpy() for t in (ds.order_id, ds.product_idx, ds.qty, ds.order_customer))
    etot, epq, ect, flags = oracle.aggregate(oid, pidx, qty, cust, 5, c, ds.price_cents)
    assert flags == 0
    np.testing.assert_array_equal(tot.cpu().numpy(), etot)
    np.testing.assert_array_equal(pq.cpu().numpy(), epq)
    np.testing.assert_array_equal(ct.cpu().numpy(), ect)


# What the eight shard tests below leave for the merge test after them: per popularity, rank -> the shard's
# [product_qty | customer_total] vector after the exchange step, the oracle's vector, and its bookkeeping.
_SHARDS_1G = {0: {}, 1: {}}


@pytest.mark.parametrize("rank", range(8))
@pytest.mark.parametrize("zipf", [0, 1])
def test_every_shard_of_the_1g_line_configurations(motor, oracle, zipf, rank):
    """The two 8-GPU configurations (1 G lines over 8 GPUs, uniform and Zipf products) as far as one GPU can run
    them: all eight shards one after the other, each at its true shape (about 125 M lines whose global line
    indices and order ids lie far from zero: shard 7 starts at line 875 M, order id about 219 M).  The device
    generator must produce exactly the oracle's slice and the engine exactly the oracle's sums on it; the
    shard's [product_qty | customer_total] vector goes through the exchange step of the C ABI
    (acmpc_merge_device: RCCL, a world of one here) before it is compared."""
    import torch
    from ac_mpc_amd import reparto
    n_total, p, c, world = 1_000_000_000, 10_000, 100_000, 8
    ds = sintetico.generar(motor, n_total, p, c, zipf=bool(zipf), qty_mode=0, rank=rank, world=world)
    m_total = ds.n_orders_total
    lo, hi = (m_total * rank) // world, (m_total * (rank + 1)) // world
    d = oracle.synth(n_total, p, c, zipf=zipf, qty_mode=0, order_lo=lo, order_hi=hi)
    assert d["n_orders_total"] == m_total and d["order_id_base"] == ds.order_id_base == 5 + lo
    assert 124_000_000 < ds.n_lines < 126_000_000 and ds.n_orders == hi - lo
    for name, t in (("order_id", ds.order_id), ("product_idx", ds.product_idx), ("qty", ds.qty),
                    ("order_customer", ds.order_customer)):
        assert np.array_equal(t.cpu().numpy(), d[name]), name
    dev = ds.order_id.device
    tot = torch.full((ds.n_orders,), -7, dtype=torch.int64, device=dev)
    vec = torch.full((p + c,), -7, dtype=torch.int64, device=dev)
    com = reparto.ComunicadorC(motor, 0, 1)
    try:
        motor.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, c, tot,
                             vec[:p], vec[p:])
        com.fusionar(vec)                                                 # the exchange step, in stream order
        assert motor.estado() == 0
    finally:
        com.cerrar()
    etot, epq, ect, flags = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                             d["order_id_base"], c, ds.price_cents, threads=8)
    assert flags == 0
    got_tot, got_vec = tot.cpu().numpy(), vec.cpu().numpy()
    np.testing.assert_array_equal(got_tot, etot)                          # totals stay with their shard
    np.testing.assert_array_equal(got_vec[:p], epq)
    np.testing.assert_array_equal(got_vec[p:], ect)
    if zipf:
        assert int(epq[0]) > 10 * int(epq[p // 2])                        # the hot product is hot in this shard
    _SHARDS_1G[zipf][rank] = dict(vec=got_vec, expect=np.concatenate([epq, ect]), sum_totals=int(got_tot.sum()),
                                  line_lo=d["line_lo"], n_lines=ds.n_lines, order_lo=lo, n_orders=ds.n_orders,
                                  m_total=m_total)


@pytest.mark.parametrize("zipf", [0, 1])
def test_merge_of_all_eight_shards_of_the_1g_line_configurations(oracle, zipf):
    """The eight vectors summed on the host, which is what the all-reduce of eight ranks computes: equal to the sum
    of the oracle's slices, every one of the 10^9 lines counted once, the money conserved; the shards tile the set."""
    got = _SHARDS_1G[zipf]
    if sorted(got) != list(range(8)):
        pytest.skip("needs the eight shard tests before it in the same run")
    n_total, p = 1_000_000_000, 10_000
    lines, orders = 0, 0
    for rank in range(8):
        assert got[rank]["line_lo"] == lines and got[rank]["order_lo"] == orders
        lines += got[rank]["n_lines"]
        orders += got[rank]["n_orders"]
    assert lines == n_total and orders == got[0]["m_total"]
    merged = sum(got[r]["vec"] for r in range(8))
    np.testing.assert_array_equal(merged, sum(got[r]["expect"] for r in range(8)))
    assert int(merged[:p].sum()) == n_total                               # qty == 1 on every one of the 10^9 lines
    sum_totals = sum(got[r]["sum_totals"] for r in range(8))
    assert int(merged[p:].sum()) == sum_totals == int((merged[:p] * oracle.synth_prices(p)).sum())
    got.clear()


def test_first_call_of_a_context_on_a_side_stream(oracle):
    """The call that allocates (and zeroes) the workspace runs on a stream of its own, which does not order
    against the null stream: the zeroing has to be queued on the call's stream too.  Grouped and any-order rows."""
    import torch
    from ac_mpc_amd import MotorLote
    m = MotorLote(0)
    try:
        n, p, c = 3_000_001, 2_000, 30_000
        ds = sintetico.generar(m, n, p, c, zipf=True, qty_mode=1)
        d = oracle.synth(n, p, c, zipf=1, qty_mode=1)
        etot, epq, ect, flags = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                                 d["order_id_base"], c, ds.price_cents)
        assert flags == 0
        dev = ds.order_id.device
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        for modo in (False, True):
            m.orden_lineas(modo)
            tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
            pq = torch.empty(p, dtype=torch.int64, device=dev)
            ct = torch.empty(c, dtype=torch.int64, device=dev)
            with torch.cuda.stream(side):
                m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, c, tot, pq, ct,
                                 stream=side.cuda_stream)
            assert m.estado(side.cuda_stream) == 0
            np.testing.assert_array_equal(tot.cpu().numpy(), etot)
            np.testing.assert_array_equal(pq.cpu().numpy(), epq)
            np.testing.assert_array_equal(ct.cpu().numpy(), ect)
    finally:
        m.orden_lineas(False)
        m.cerrar()


# ---- rows not grouped by order (concurrent checkouts interleave their detalle inserts) ------------

def _check_any_order(motor, oracle, oid, pidx, qty, cust, base, c_n, price):
    etot, epq, ect, flags = oracle.aggregate(oid, pidx, qty, cust, base, c_n, price)
    assert flags & ~2 == 0                                  # the oracle only notes that ids are unsorted
    for modo in (True, "auto"):
        motor.orden_lineas(modo)
        try:
            motor.cargar_productos(price)
            tot, pq, ct = motor.agregar(oid, pidx, qty, cust, base, c_n)
        finally:
            motor.orden_lineas(False)
        np.testing.assert_array_equal(tot, etot)
        np.testing.assert_array_equal(pq, epq)
        np.testing.assert_array_equal(ct, ect)


@pytest.mark.parametrize("n,p,c,mode", [(1000, 100, 50, "shuffle"), (300_003, 10_000, 100_000, "shuffle"),
                                         (1_000_000, 10_000, 100_000, "interleave"), (2_000_000, 10_000, 100_000, "sorted"),
                                         (700_001, 60_000, 3_000_000, "shuffle"),
                                         (20_000_003, 10_000, 100_000, "shuffle"), (20_000_003, 10_000, 100_000, "blocks")])
def test_any_line_order(motor, oracle, n, p, c, mode):
    """shuffle: every row somewhere else (the order-restoring path: at 20 M lines 306 buckets in 5 coarse bins);
    interleave / sorted: near-grouped rows (the device picks the atomic kernel); blocks: runs of 3000 rows in a
    random order (grouped inside a block, far from grouped overall)."""
    d = oracle.synth(n, p, c, zipf=1, qty_mode=1)
    oid, pidx, qty = d["order_id"], d["product_idx"], d["qty"]
    rng = np.random.default_rng(23)
    if mode == "shuffle":
        perm = rng.permutation(n)
    elif mode == "interleave":
        # two checkouts at a time insert alternately: swap neighbouring blocks of rows
        perm = np.arange(n)
        k = (n // 2) * 2
        perm[:k] = perm[:k].reshape(-1, 2)[:, ::-1].reshape(-1)
    elif mode == "blocks":
        k = n // 3000
        perm = np.concatenate([(rng.permutation(k)[:, None] * 3000 + np.arange(3000)[None, :]).reshape(-1),
                               np.arange(k * 3000, n)])
    else:
        perm = np.arange(n)
    _check_any_order(motor, oracle, oid[perm], pidx[perm], qty[perm], d["order_customer"], 5, c, oracle.synth_prices(p))


@pytest.mark.parametrize("shift", [15, 16])
@pytest.mark.parametrize("n", [70_001, 3_000_017])
def test_any_line_order_with_larger_buckets(monkeypatch, oracle, shift, n):
    """Beyond 2^25 orders the buckets of the order-restoring path hold 32768 or 65536 orders and the last pass
    reads every bucket with two or four workgroups, each keeping its own 16384 orders.  The developer switch puts
    small inputs through the same code: last buckets that are partly filled, parts without any order."""
    from ac_mpc_amd import MotorLote, _lib
    monkeypatch.setenv("ACMPC_PART_SHIFT", str(shift))
    m = MotorLote(0)
    try:
        p, c = 3_000, 40_000
        d = oracle.synth(n, p, c, zipf=1, qty_mode=1)
        perm = np.random.default_rng(shift).permutation(n)
        _check_any_order(m, oracle, d["order_id"][perm], d["product_idx"][perm], d["qty"][perm], d["order_customer"], 5, c,
                         oracle.synth_prices(p))
        m.orden_lineas(True)
        m.agregar(d["order_id"][perm], d["product_idx"][perm], d["qty"][perm], d["order_customer"], 5, c)
        assert m.camino_lineas() == _lib.PATH_MOVED
    finally:
        m.cerrar()


def test_any_line_order_beyond_2_to_the_25_orders(motor, oracle):
    """36 M orders (144 M lines) in no order at all, device resident: more than the 2^25 orders that buckets of
    16384 reach, so the path runs with buckets of 32768 orders; all three outputs equal the oracle's."""
    import torch
    from ac_mpc_amd import _lib
    n, p, c = 144_000_000, 10_000, 100_000
    ds = sintetico.generar(motor, n, p, c, zipf=True, qty_mode=0)
    assert ds.n_orders > (1 << 25)
    dev = ds.order_id.device
    oid_h, pidx_h, qty_h = (t.cpu().numpy() for t in (ds.order_id, ds.product_idx, ds.qty))
    etot, epq, ect, flags = oracle.aggregate(oid_h, pidx_h, qty_h, ds.order_customer.cpu().numpy(), ds.order_id_base, c,
                                             ds.price_cents, threads=8)
    assert flags == 0
    g = torch.Generator(device=dev); g.manual_seed(41)
    perm = torch.randperm(n, device=dev, generator=g)
    oid, pidx, qty = ds.order_id[perm].contiguous(), ds.product_idx[perm].contiguous(), ds.qty[perm].contiguous()
    del perm
    tot = torch.full((ds.n_orders,), -7, dtype=torch.int64, device=dev)
    pq = torch.full((p,), -7, dtype=torch.int64, device=dev)
    ct = torch.full((c,), -7, dtype=torch.int64, device=dev)
    motor.orden_lineas(True)
    try:
        motor.agregar_device(oid, pidx, qty, ds.order_customer, ds.order_id_base, c, tot, pq, ct)
        assert motor.estado() == 0
        assert motor.camino_lineas() == _lib.PATH_MOVED
    finally:
        motor.orden_lineas(False)
    np.testing.assert_array_equal(tot.cpu().numpy(), etot)
    np.testing.assert_array_equal(pq.cpu().numpy(), epq)
    np.testing.assert_array_equal(ct.cpu().numpy(), ect)


def test_any_line_order_rough_values(motor, oracle):
    v = np.load(os.path.join(GOLDEN, "synth_5k_rough.npz"))
    rng = np.random.default_rng(29)
    perm = rng.permutation(v["order_id"].size)
    _check_any_order(motor, oracle, v["order_id"][perm], v["product_idx"][perm], v["qty"][perm], v["order_customer"],
                     int(v["order_id_base"]), int(v["n_customers"]), v["price_cents"])


@pytest.mark.parametrize("what", ["one_wide", "negative_qty", "all_wide"])
def test_any_line_order_values_that_do_not_fit_a_record(motor, oracle, what):
    """The order-restoring path carries a line's value in 39 bits; one line whose price * qty is 2^39 or more, or
    negative (a return), sends the order totals through the atomic kernel while the stock deltas stay with the
    partition pass.  Both modes, shuffled rows."""
    n, p, c = 2_000_003, 2_000, 30_000
    d = oracle.synth(n, p, c, zipf=1, qty_mode=1)
    oid, pidx, qty = d["order_id"].copy(), d["product_idx"].copy(), d["qty"].copy()
    price = oracle.synth_prices(p).copy()
    if what == "one_wide":
        price[7] = 2_000_000_000
        k = int(np.flatnonzero(pidx == 7)[0])
        qty[k] = 1_000                                    # 2e12 > 2^39
    elif what == "negative_qty":
        qty[1_234_567] = -3
    else:
        price[:] = 2_100_000_000
        qty[:] = 1_000_000 + (qty % 7)                   # every value about 2e15
    perm = np.random.default_rng(37).permutation(n)
    _check_any_order(motor, oracle, oid[perm], pidx[perm], qty[perm], d["order_customer"], 5, c, price)


@pytest.mark.parametrize("align", [1, 4, 256, 4096, 2048 * 4096, "tail", "first"])
def test_any_order_one_descent_where_the_counting_pass_changes_hands(motor, oracle, align):
    """Rows whose ids never go down are handed to the grouped kernel; a single swapped pair must be seen wherever it
    lies: inside a lane's four ids, between lanes, between waves, between workgroups, between the sweeps of the
    grid, among the last lines (the count is not a multiple of 4) and at the very start."""
    n = 9_000_003
    d = oracle.synth(n, 10_000, 100_000, qty_mode=1)
    oid, pidx, qty = d["order_id"].copy(), d["product_idx"].copy(), d["qty"].copy()
    if align == "tail":
        oid[n - 1] = oid[n - 2] - 1                          # the last row belongs to the order before
    elif align == "first":
        k = 0
        while oid[k] == oid[k + 1]:
            k += 1
        for col in (oid, pidx, qty):                          # the first order's rows go behind the second's first row
            col[0], col[k + 1] = col[k + 1], col[0]
    else:
        k = align * max(1, 1_234_567 // align) - 1          # rows k, k + 1 lie either side of a multiple of `align`
        while oid[k] == oid[k + 1]:
            k += align
        for col in (oid, pidx, qty):
            col[k], col[k + 1] = col[k + 1], col[k]
    assert np.any(np.diff(oid) < 0)
    _check_any_order(motor, oracle, oid, pidx, qty, d["order_customer"], 5, 100_000, oracle.synth_prices(10_000))


def test_grouped_mode_still_rejects_shuffled_rows(motor, oracle):
    d = oracle.synth(50_000, 100, 50)
    perm = np.random.default_rng(31).permutation(50_000)
    motor.cargar_productos(oracle.synth_prices(100))
    with pytest.raises(ErrorLote) as e:
        motor.agregar(d["order_id"][perm], d["product_idx"][perm], d["qty"][perm], d["order_customer"], 5, 50)
    assert e.value.code == _lib.E_DATA


def test_auto_order_one_descent_in_a_large_input(motor, oracle):
    """A single pair of swapped rows, far from any tile start, sends the totals through the fallback;
    an id outside the order range is still reported."""
    d = oracle.synth(3_000_000, 10_000, 100_000, qty_mode=1)
    oid, pidx, qty = d["order_id"].copy(), d["product_idx"].copy(), d["qty"].copy()
    k = 1_234_567
    while oid[k] == oid[k + 1]:
        k += 1
    for col in (oid, pidx, qty):
        col[k], col[k + 1] = col[k + 1], col[k]
    price = oracle.synth_prices(10_000)
    etot, epq, ect, flags = oracle.aggregate(oid, pidx, qty, d["order_customer"], 5, 100_000, price)
    assert flags == 2
    motor.cargar_productos(price)
    motor.orden_lineas("auto")
    try:
        tot, pq, ct = motor.agregar(oid, pidx, qty, d["order_customer"], 5, 100_000)
        np.testing.assert_array_equal(tot, etot)
        np.testing.assert_array_equal(pq, epq)
        np.testing.assert_array_equal(ct, ect)
        oid[77] = 4                                           # below the base
        with pytest.raises(ErrorLote) as e:
            motor.agregar(oid, pidx, qty, d["order_customer"], 5, 100_000)
        assert e.value.code == _lib.E_DATA
    finally:
        motor.orden_lineas(False)
