"""The multi-GPU route behind the C ABI on the one GPU a test box has: a world of one rank.

RCCL refuses two ranks on one device, so what can run here is every entry point with n_devices = 1 /
world = 1: the all-reduce really goes through RCCL (a sum over one rank), and the sharded host entry
point must equal acmpc_aggregate bit for bit.  The cut logic for more shards is host-only code of the
library and is covered on the CPU (tests/test_shard_plan.py, tests/test_reparto.py) and by the shard-sum
tests in test_gpu_parity.py.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def motor1():
    from ac_mpc_amd import MotorLote
    m = MotorLote(0)
    yield m
    m.cerrar()


def test_sharded_host_entry_equals_plain_entry(motor1, oracle):
    from ac_mpc_amd import reparto
    p, c = 10_000, 100_000
    d = oracle.synth(1_000_003, p, c, zipf=1, qty_mode=1)
    price = oracle.synth_prices(p)
    motor1.cargar_productos(price)
    coms = reparto.comunicadores_locales([motor1])
    try:
        assert coms[0].world == 1
        got = reparto.agregar_repartido([motor1], coms, d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                        d["order_id_base"], c)
        plain = motor1.agregar(d["order_id"], d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"], c)
        etot, epq, ect, flags = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                                 d["order_id_base"], c, price)
        assert flags == 0
        for g, q, e in zip(got, plain, (etot, epq, ect)):
            assert np.array_equal(g, q) and np.array_equal(g, e)
        # empty input and a broken contract behave like acmpc_aggregate
        z = np.zeros(0, np.int32)
        tot, pq, ct = reparto.agregar_repartido([motor1], coms, z, z, z, np.array([0, 1], np.int32), 5, c)
        assert tot.tolist() == [0, 0] and not pq.any() and not ct.any()
        from ac_mpc_amd import ErrorLote, _lib
        bad = d["order_id"].copy()
        bad[1000], bad[2000] = bad[2000], bad[1000]
        with pytest.raises(ErrorLote) as err:
            reparto.agregar_repartido([motor1], coms, bad, d["product_idx"], d["qty"], d["order_customer"], d["order_id_base"], c)
        assert err.value.code == _lib.E_DATA
    finally:
        for k in coms:
            k.cerrar()


@pytest.mark.parametrize("modo", [True, "auto"])
def test_sharded_entry_honours_the_line_order_option(motor1, oracle, modo):
    """Rows in any order through the sharded entry point: same results as the oracle (ACMPC_ORDER_ANY and
    ACMPC_ORDER_AUTO), while under the grouped promise the same rows are a contract violation reported as
    ACMPC_E_DATA — never a size or memory error."""
    from ac_mpc_amd import ErrorLote, _lib, reparto
    p, c = 2_000, 5_000
    d = oracle.synth(300_007, p, c, zipf=1, qty_mode=1)
    price = oracle.synth_prices(p)
    motor1.cargar_productos(price)
    rng = np.random.default_rng(5)
    perm = rng.permutation(d["order_id"].size)
    oid, pidx, qty = d["order_id"][perm], d["product_idx"][perm], d["qty"][perm]
    etot, epq, ect, flags = oracle.aggregate(d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                             d["order_id_base"], c, price)
    assert flags == 0
    coms = reparto.comunicadores_locales([motor1])
    try:
        with pytest.raises(ErrorLote) as err:            # grouped promise, shuffled rows
            reparto.agregar_repartido([motor1], coms, oid, pidx, qty, d["order_customer"], d["order_id_base"], c)
        assert err.value.code == _lib.E_DATA
        motor1.orden_lineas(modo)
        got = reparto.agregar_repartido([motor1], coms, oid, pidx, qty, d["order_customer"], d["order_id_base"], c)
        for g, e in zip(got, (etot, epq, ect)):
            assert np.array_equal(g, e)
        # ... and grouped rows under the same option
        got = reparto.agregar_repartido([motor1], coms, d["order_id"], d["product_idx"], d["qty"], d["order_customer"],
                                        d["order_id_base"], c)
        for g, e in zip(got, (etot, epq, ect)):
            assert np.array_equal(g, e)
    finally:
        motor1.orden_lineas(False)
        for k in coms:
            k.cerrar()


def test_rank_communicator_from_an_id_and_the_overlapped_exchange(motor1):
    import torch
    from ac_mpc_amd import reparto
    ident = reparto.nuevo_id()
    assert len(ident) == reparto.ID_BYTES
    com = reparto.ComunicadorC(motor1, 0, 1, ident)
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev); g.manual_seed(3)
        a = torch.randint(-2**62, 2**62, (110_000,), dtype=torch.int64, device=dev, generator=g)
        want = a.clone()
        com.fusionar(a)                                   # the sum over one rank
        torch.cuda.synchronize()
        assert torch.equal(a, want)
        bufs = [a.clone(), a.clone()]
        turnos = reparto.FusionEnCursoC(bufs, com)
        for step in range(5):
            b = turnos.siguiente()
            b.copy_(want + step)
            turnos.lanzar()
        last = turnos.esperar()
        torch.cuda.synchronize()
        assert torch.equal(last, want + 4)
    finally:
        com.cerrar()


def test_bad_arguments_fail_cleanly(motor1):
    from ac_mpc_amd import _lib
    lib = motor1._lib
    h = C.c_void_p()
    assert lib.acmpc_comm_init_rank(C.byref(h), motor1._ctx, None, 0, 1) == _lib.E_INVALID
    assert lib.acmpc_comm_init_rank(C.byref(h), motor1._ctx, b"\0" * 128, 2, 2) == _lib.E_INVALID
    assert lib.acmpc_merge_device(motor1._ctx, None, None, 4, None) == _lib.E_INVALID
    ctxs = (C.c_void_p * 2)(motor1._ctx, motor1._ctx)
    out = (C.c_void_p * 2)()
    assert lib.acmpc_comm_init_all(out, ctxs, 2) == _lib.E_INVALID      # two contexts on one device
    assert b"share device" in lib.acmpc_last_error(motor1._ctx)
    lib.acmpc_comm_destroy(None)
