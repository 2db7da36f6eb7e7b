"""The shard cuts of the library itself (`acmpc_shard_plan`, `acmpc_shard_partition`: host only, no GPU) against
the Python twin `reparto.cortes`, and on the inputs the twin does not cover: empty shards, orders without lines
at a cut, ids outside the order range, more devices than orders, columns that are not sorted.  What is under
test is what `acmpc_aggregate_sharded` uses to size and place every shard."""
import ctypes as C

import numpy as np
import pytest

from ac_mpc_amd import _lib, reparto
from ac_mpc_amd.lote import ErrorLote

WORLDS = (1, 2, 3, 8)


def _tiles(cs, n_orders, n_lines):
    assert cs[0].order_lo == 0 and cs[0].line_lo == 0
    assert cs[-1].order_hi == n_orders and cs[-1].line_hi == n_lines
    for a, b in zip(cs, cs[1:]):
        assert a.order_hi == b.order_lo and a.line_hi == b.line_lo
        assert a.line_lo <= a.line_hi


@pytest.mark.parametrize("world", WORLDS)
def test_grouped_cuts_equal_the_python_twin(oracle, world):
    d = oracle.synth(20_000, 100, 50)
    m = d["n_orders_total"]
    got = reparto.cortes_c(d["order_id"], 5, m, world)
    assert got == reparto.cortes(d["order_id"], 5, m, world)
    _tiles(got, m, d["order_id"].size)


@pytest.mark.parametrize("world", WORLDS)
def test_orders_without_lines_at_a_cut_and_odd_sizes(world):
    rng = np.random.default_rng(world)
    m = 1001                                             # odd, not a multiple of any world size here
    lens = rng.integers(0, 5, m)
    for g in range(1, world):                            # the orders either side of every cut have no lines
        o = (m * g) // world
        lens[max(0, o - 2):o + 2] = 0
    oid = np.repeat(np.arange(m, dtype=np.int32) + 7, lens)
    got = reparto.cortes_c(oid, 7, m, world)
    assert got == reparto.cortes(oid, 7, m, world)
    _tiles(got, m, oid.size)
    for c in got:
        ids = oid[c.line_lo:c.line_hi]
        assert ids.size == 0 or (ids[0] >= 7 + c.order_lo and ids[-1] < 7 + c.order_hi)


def test_no_lines_and_no_orders():
    for world in WORLDS:
        cs = reparto.cortes_c(np.zeros(0, np.int32), 5, 0, world)
        assert all(c.n_orders == 0 and c.n_lines == 0 for c in cs)
        cs = reparto.cortes_c(np.zeros(0, np.int32), 5, 10, world)
        _tiles(cs, 10, 0)
    with pytest.raises(ErrorLote) as e:                  # lines without any order: as acmpc_aggregate
        reparto.cortes_c(np.array([5], np.int32), 5, 0, 2)
    assert e.value.code == _lib.E_INVALID


@pytest.mark.parametrize("world", (2, 3, 8))
def test_more_devices_than_orders(world):
    m = world - 1                                        # at least one shard owns no order
    oid = np.repeat(np.arange(m, dtype=np.int32) + 5, 3)
    oid = np.concatenate([np.array([1, 2], np.int32), oid])        # and two ids below the base lead the column
    cs = reparto.cortes_c(oid, 5, m, world)
    _tiles(cs, m, oid.size)
    for c in cs:
        if c.n_orders == 0:
            assert c.n_lines == 0                        # a shard without orders gets no lines
    first = next(c for c in cs if c.n_orders > 0)
    assert first.line_lo == 0 and first.n_lines == 2 + 3            # the ids below the base go to the first owner


@pytest.mark.parametrize("world", (2, 3, 8))
def test_ids_outside_the_order_range_stay_with_the_end_shards(world):
    m = 64
    oid = np.concatenate([np.array([-3, 0, 4], np.int32), np.repeat(np.arange(m, dtype=np.int32) + 5, 2),
                          np.array([5 + m, 5 + m + 9], np.int32)])
    cs = reparto.cortes_c(oid, 5, m, world)
    _tiles(cs, m, oid.size)
    assert cs[0].n_lines == 3 + 2 * cs[0].n_orders
    assert cs[-1].n_lines == 2 + 2 * cs[-1].n_orders


@pytest.mark.parametrize("world", (2, 3, 8))
def test_grouped_cuts_of_an_unsorted_column_never_descend(world):
    """Under the grouped promise a column that is not sorted is a contract violation that the device pass
    reports; the cuts must still give every line to exactly one shard (sizes that are not negative)."""
    rng = np.random.default_rng(11 + world)
    m = 500
    oid = np.repeat(np.arange(m, dtype=np.int32) + 5, rng.integers(1, 5, m))
    for trial in range(20):
        bad = oid.copy()
        if trial % 2:
            rng.shuffle(bad)                             # no order at all
        else:
            i, j = rng.integers(0, bad.size, 2)          # one swapped pair
            bad[i], bad[j] = bad[j], bad[i]
        cs = reparto.cortes_c(bad, 5, m, world)
        _tiles(cs, m, bad.size)
        # some shard then holds a descent or an id outside its own range: that is what the device reports
        if not np.all(bad[:-1] <= bad[1:]):
            seen = False
            for c in cs:
                ids = bad[c.line_lo:c.line_hi].astype(np.int64)
                if ids.size and (np.any(ids[:-1] > ids[1:]) or ids.min() < 5 + c.order_lo or ids.max() >= 5 + c.order_hi):
                    seen = True
            assert seen


@pytest.mark.parametrize("world", WORLDS)
def test_any_order_plan_and_partition(world):
    rng = np.random.default_rng(3 * world)
    m = 700
    lens = rng.integers(0, 6, m)
    oid = np.repeat(np.arange(m, dtype=np.int32) + 5, lens)
    n = oid.size
    pidx = rng.integers(-1, 50, n).astype(np.int32)
    qty = rng.integers(1, 4, n).astype(np.int32)
    perm = rng.permutation(n)
    oid_s, pidx_s, qty_s = oid[perm], pidx[perm], qty[perm]
    # stray ids: below the base and past the last order
    oid_s = np.concatenate([oid_s, np.array([2, 5 + m + 1], np.int32)])
    pidx_s = np.concatenate([pidx_s, np.array([1, 2], np.int32)])
    qty_s = np.concatenate([qty_s, np.array([1, 1], np.int32)])
    cs = reparto.cortes_c(oid_s, 5, m, world, _lib.ORDER_ANY)
    _tiles(cs, m, oid_s.size)
    o_id, o_p, o_q, l_cut = reparto.particion_c(oid_s, pidx_s, qty_s, 5, m, world)
    assert [int(x) for x in l_cut] == [c.line_lo for c in cs] + [oid_s.size]
    idx = oid_s.astype(np.int64) - 5
    for g, c in enumerate(cs):
        mine = (idx >= c.order_lo) & (idx < c.order_hi)
        if g == 0:
            mine |= idx < 0
        if g == world - 1:
            mine |= idx >= m
        # a stable partition: the shard's rows in the order in which they stood
        np.testing.assert_array_equal(o_id[c.line_lo:c.line_hi], oid_s[mine])
        np.testing.assert_array_equal(o_p[c.line_lo:c.line_hi], pidx_s[mine])
        np.testing.assert_array_equal(o_q[c.line_lo:c.line_hi], qty_s[mine])


def test_any_order_on_a_sorted_column_cuts_where_the_rows_lie(oracle):
    d = oracle.synth(5_000, 100, 50)
    m = d["n_orders_total"]
    for world in WORLDS:
        assert reparto.cortes_c(d["order_id"], 5, m, world, _lib.ORDER_AUTO) == reparto.cortes(d["order_id"], 5, m, world)


def test_bad_arguments():
    lib = _lib.load()
    o = (C.c_int64 * 3)()
    l = (C.c_int64 * 3)()
    ids = np.array([5, 6], np.int32)
    assert lib.acmpc_shard_plan(ids.ctypes.data, 2, 2, 5, 0, 0, o, l) == _lib.E_INVALID         # no device
    assert lib.acmpc_shard_plan(ids.ctypes.data, 2, 2, 5, 65, 0, o, l) == _lib.E_INVALID        # too many
    assert lib.acmpc_shard_plan(None, 2, 2, 5, 2, 0, o, l) == _lib.E_INVALID                    # null column
    assert lib.acmpc_shard_plan(ids.ctypes.data, 2, 2, 5, 2, 7, o, l) == _lib.E_INVALID         # unknown line order
    assert lib.acmpc_shard_plan(ids.ctypes.data, 2, 2, 5, 2, 0, None, l) == _lib.E_INVALID
    assert b"shard plan" in lib.acmpc_last_error(None)
