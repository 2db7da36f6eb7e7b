"""The allocation-free entry point (acmpc_aggregate_device_noalloc) against acmpc_aggregate_device.

Through the C ABI, bit for bit, on two calls in a row of which the second has the smaller input (the
fixed workspace keeps the first call's contents), and its limits are refused with a clear message.
Collected last on purpose: a failure here must not hide the tests before it.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(motor, ds, n_customers):
    """(outputs of acmpc_aggregate_device, outputs of acmpc_aggregate_device_noalloc), each poisoned first."""
    import torch
    dev = ds.order_id.device
    lib, ctx = motor._lib, motor._ctx
    outs = []
    for fn in (lib.acmpc_aggregate_device, lib.acmpc_aggregate_device_noalloc):
        tot = torch.full((ds.n_orders,), -7, dtype=torch.int64, device=dev)
        pq = torch.full((motor.n_products,), -7, dtype=torch.int64, device=dev)
        ct = torch.full((n_customers,), -7, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = fn(ctx, ds.order_id.data_ptr(), ds.product_idx.data_ptr(), ds.qty.data_ptr(), ds.n_lines,
                ds.order_customer.data_ptr(), ds.n_orders, ds.order_id_base, n_customers,
                tot.data_ptr(), pq.data_ptr(), ct.data_ptr(), stream)
        assert rc == 0, lib.acmpc_last_error(ctx)
        assert motor.estado() == 0
        outs.append((tot.cpu().numpy(), pq.cpu().numpy(), ct.cpu().numpy()))
    return outs


@pytest.mark.parametrize("fuse", [1, 2])
def test_two_calls_in_a_row_second_smaller(monkeypatch, oracle, fuse):
    from ac_mpc_amd import MotorLote, sintetico
    monkeypatch.setenv("ACMPC_FUSE", str(fuse))       # 2: the three-launch sequence also on small inputs
    m = MotorLote(0)
    try:
        p, c = 10_000, 100_000
        for n_lines, zipf in ((20_000_003, 1), (300_007, 0)):
            ds = sintetico.generar(m, n_lines, p, c, zipf=bool(zipf), qty_mode=1)
            (tot, pq, ct), (ntot, npq, nct) = _both(m, ds, c)
            assert np.array_equal(ntot, tot) and np.array_equal(npq, pq) and np.array_equal(nct, ct)
            # and both equal the oracle (the smaller case in full, the larger through its sums)
            if n_lines < 1_000_000:
                etot, epq, ect, flags = oracle.aggregate(ds.order_id.cpu().numpy(), ds.product_idx.cpu().numpy(),
                                                         ds.qty.cpu().numpy(), ds.order_customer.cpu().numpy(),
                                                         ds.order_id_base, c, ds.price_cents)
                assert flags == 0
                assert np.array_equal(ntot, etot) and np.array_equal(npq, epq) and np.array_equal(nct, ect)
            assert int(ntot.sum()) == int(nct.sum())
    finally:
        m.cerrar()


def test_a_warm_call_leaves_free_device_memory_unchanged(oracle):
    """Nothing is allocated or freed: the device's free memory is the same before and after a call on a
    context that is warm (hipMemGetInfo through torch, which itself allocates nothing for the query)."""
    import torch
    from ac_mpc_amd import MotorLote, sintetico
    m = MotorLote(0)
    try:
        p, c = 10_000, 100_000
        ds = sintetico.generar(m, 4_000_001, p, c, zipf=True, qty_mode=1)
        dev = ds.order_id.device
        tot = torch.zeros(ds.n_orders, dtype=torch.int64, device=dev)
        pq = torch.zeros(p, dtype=torch.int64, device=dev)
        ct = torch.zeros(c, dtype=torch.int64, device=dev)
        run = lambda: m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, c,
                                       tot, pq, ct, sin_reservas=True)
        run()
        assert m.estado() == 0
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info(dev)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info(dev)
        assert free_after == free_before
        assert m.estado() == 0
        # the allocating entry point on a fresh, larger problem does move it (the check can see an allocation)
        m2 = MotorLote(0)
        try:
            m2.cargar_productos(ds.price_cents)
            m2.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, c, tot, pq, ct)
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info(dev)[0] < free_after
        finally:
            m2.cerrar()
    finally:
        m.cerrar()


def test_limits_are_refused_with_a_message(oracle):
    import torch
    from ac_mpc_amd import MotorLote, _lib, sintetico
    m = MotorLote(0)
    try:
        ds = sintetico.generar(m, 5_000, 100, 200_000)          # more customers than the fixed workspace holds
        dev = ds.order_id.device
        tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
        pq = torch.empty(100, dtype=torch.int64, device=dev)
        ct = torch.empty(200_000, dtype=torch.int64, device=dev)
        rc = m._lib.acmpc_aggregate_device_noalloc(m._ctx, ds.order_id.data_ptr(), ds.product_idx.data_ptr(),
                                                   ds.qty.data_ptr(), ds.n_lines, ds.order_customer.data_ptr(), ds.n_orders,
                                                   ds.order_id_base, 200_000, tot.data_ptr(), pq.data_ptr(), ct.data_ptr(), None)
        assert rc == _lib.E_INVALID and b"131072 customers" in m._lib.acmpc_last_error(m._ctx)
        # more products than the fixed workspace holds
        ds = sintetico.generar(m, 5_000, 20_000, 100)
        pq = torch.empty(20_000, dtype=torch.int64, device=dev)
        ct = torch.empty(100, dtype=torch.int64, device=dev)
        tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
        rc = m._lib.acmpc_aggregate_device_noalloc(m._ctx, ds.order_id.data_ptr(), ds.product_idx.data_ptr(),
                                                   ds.qty.data_ptr(), ds.n_lines, ds.order_customer.data_ptr(), ds.n_orders,
                                                   ds.order_id_base, 100, tot.data_ptr(), pq.data_ptr(), ct.data_ptr(), None)
        assert rc == _lib.E_INVALID and b"16380 products" in m._lib.acmpc_last_error(m._ctx)
        # a product table within that limit but too large for LDS (prices beyond 32 bits take 12 bytes a product)
        import numpy as np
        ds = sintetico.generar(m, 5_000, 14_000, 100)
        m.cargar_productos(np.full(14_000, 2**33, np.int64))
        pq = torch.empty(14_000, dtype=torch.int64, device=dev)
        tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
        rc = m._lib.acmpc_aggregate_device_noalloc(m._ctx, ds.order_id.data_ptr(), ds.product_idx.data_ptr(),
                                                   ds.qty.data_ptr(), ds.n_lines, ds.order_customer.data_ptr(), ds.n_orders,
                                                   ds.order_id_base, 100, tot.data_ptr(), pq.data_ptr(), ct.data_ptr(), None)
        assert rc == _lib.E_INVALID and b"fits LDS" in m._lib.acmpc_last_error(m._ctx)
        # rows in any order are not served
        m.orden_lineas(True)
        rc = m._lib.acmpc_aggregate_device_noalloc(m._ctx, ds.order_id.data_ptr(), ds.product_idx.data_ptr(),
                                                   ds.qty.data_ptr(), ds.n_lines, ds.order_customer.data_ptr(), ds.n_orders,
                                                   ds.order_id_base, 100, tot.data_ptr(), pq.data_ptr(), ct.data_ptr(), None)
        assert rc == _lib.E_INVALID and b"grouped path" in m._lib.acmpc_last_error(m._ctx)
    finally:
        m.cerrar()


def test_captured_in_a_graph_and_replayed(oracle):
    """What the entry point is for: a caller under HIP graph capture, where nothing may be allocated.  The
    replayed graph writes the same three outputs as the call issued directly, also after the inputs changed."""
    import torch
    from ac_mpc_amd import MotorLote, sintetico
    m = MotorLote(0)
    try:
        p, c, n = 10_000, 100_000, 2_000_003
        ds = sintetico.generar(m, n, p, c, zipf=True, qty_mode=1)
        dev = ds.order_id.device
        tot = torch.zeros(ds.n_orders, dtype=torch.int64, device=dev)
        pq = torch.zeros(p, dtype=torch.int64, device=dev)
        ct = torch.zeros(c, dtype=torch.int64, device=dev)
        m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, c, tot, pq, ct,
                         sin_reservas=True)                 # (first call outside the capture: the library is warm)
        assert m.estado() == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=torch.cuda.Stream(device=dev)):
            m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, c, tot, pq, ct,
                             stream=torch.cuda.current_stream(dev).cuda_stream, sin_reservas=True)
        for qty_shift in (0, 1):
            if qty_shift:
                ds.qty.add_(1)                              # same buffers, new contents
            tot.fill_(-7); pq.fill_(-7); ct.fill_(-7)
            g.replay()
            torch.cuda.synchronize()
            assert m.estado() == 0
            etot, epq, ect, flags = oracle.aggregate(ds.order_id.cpu().numpy(), ds.product_idx.cpu().numpy(),
                                                     ds.qty.cpu().numpy(), ds.order_customer.cpu().numpy(),
                                                     ds.order_id_base, c, ds.price_cents)
            assert flags == 0
            assert np.array_equal(tot.cpu().numpy(), etot) and np.array_equal(pq.cpu().numpy(), epq)
            assert np.array_equal(ct.cpu().numpy(), ect)
    finally:
        m.cerrar()
