"""ctypes binding of oracle/liboracle.so — the CPU checker.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def _pack_codes(codes) -> np.ndarray:
    """The code list as the C oracle takes it: 32 bytes per code, NUL padded (varchar(30), store.sql:110).
    Packed here, not with the product's helper, so that the checker shares no code with what it checks."""
    out = np.zeros((len(codes), 32), np.uint8)
    for i, code in enumerate(codes):
        raw = code if isinstance(code, (bytes, bytearray)) else str(code).encode("latin-1")
        if len(raw) > 30:
            raise ValueError(f"code {i} is longer than 30 bytes")
        out[i, :len(raw)] = np.frombuffer(bytes(raw), np.uint8)
    return out


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_LIB):
            subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True)
        lib = C.CDLL(ORACLE_LIB)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        lib.oracle_aggregate.argtypes = [vp, vp, vp, i64, vp, i64, i32, i32, vp, i32, vp, vp, vp]
        lib.oracle_aggregate.restype = C.c_int
        lib.oracle_aggregate_mt.argtypes = [vp, vp, vp, i64, vp, i64, i32, i32, vp, i32, vp, vp, vp, i32]
        lib.oracle_aggregate_mt.restype = C.c_int
        lib.oracle_aggregate_lifecycle.argtypes = [vp, vp, vp, i64, vp, i64, i32, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp]
        lib.oracle_aggregate_lifecycle.restype = C.c_int
        lib.oracle_hash.argtypes = [C.c_uint64, C.c_uint64]
        lib.oracle_hash.restype = C.c_uint64
        lib.oracle_synth_n_orders.argtypes = [i64]
        lib.oracle_synth_n_orders.restype = i64
        lib.oracle_synth_prices.argtypes = [i32, vp]
        lib.oracle_synth_prices.restype = None
        lib.oracle_synth_count_lines.argtypes = [i64, i64, i64, C.POINTER(i64)]
        lib.oracle_synth_count_lines.restype = i64
        lib.oracle_synth_fill.argtypes = [i64, i32, i32, i32, i32, i32, i64, i64, vp, vp, vp, vp]
        lib.oracle_synth_fill.restype = None
        lib.oracle_ingest_detalle.argtypes = [vp, i64, vp, i32, vp, vp, vp, i64, C.POINTER(i64)]
        lib.oracle_ingest_detalle.restype = i64
        lib.oracle_emit_totals.argtypes = [vp, i64, i32, C.c_int, vp, i64]
        lib.oracle_emit_totals.restype = i64
        lib.oracle_emit_stock.argtypes = [vp, i32, vp, C.c_int, vp, i64]
        lib.oracle_emit_stock.restype = i64
        self.lib = lib

    def aggregate(self, order_id, product_idx, qty, order_customer, order_id_base, n_customers, price_cents, threads=1,
                  out=None):
        """`out`: optional preallocated (order_total, product_qty, customer_total) int64 arrays (benchmarks reuse
        them so that page faults of fresh arrays stay out of the timed passes)."""
        oid = np.ascontiguousarray(order_id, np.int32)
        pidx = np.ascontiguousarray(product_idx, np.int32)
        q = np.ascontiguousarray(qty, np.int32)
        cust = np.ascontiguousarray(order_customer, np.int32)
        price = np.ascontiguousarray(price_cents, np.int64)
        if out is None:
            out = (np.empty(cust.size, np.int64), np.empty(price.size, np.int64), np.empty(int(n_customers), np.int64))
        tot, pq, ct = out
        assert tot.size == cust.size and pq.size == price.size and ct.size == int(n_customers)
        common = (oid.ctypes.data, pidx.ctypes.data, q.ctypes.data, oid.size, cust.ctypes.data, cust.size,
                  int(order_id_base), int(n_customers), price.ctypes.data, price.size, tot.ctypes.data, pq.ctypes.data,
                  ct.ctypes.data)
        if threads > 1:
            flags = self.lib.oracle_aggregate_mt(*common, int(threads))
        else:
            flags = self.lib.oracle_aggregate(*common)
        return tot, pq, ct, flags

    def aggregate_lifecycle(self, order_id, product_idx, qty, order_customer, order_id_base, n_customers, price_cents,
                            line_price=None, order_state=None, order_descuento=None, descuento_mode=0):
        oid = np.ascontiguousarray(order_id, np.int32)
        pidx = np.ascontiguousarray(product_idx, np.int32)
        q = np.ascontiguousarray(qty, np.int32)
        cust = np.ascontiguousarray(order_customer, np.int32)
        price = np.ascontiguousarray(price_cents, np.int64)
        lp = None if line_price is None else np.ascontiguousarray(line_price, np.int64)
        st = None if order_state is None else np.ascontiguousarray(order_state, np.uint8)
        de = None if order_descuento is None else np.ascontiguousarray(order_descuento, np.int32)
        tot = np.empty(cust.size, np.int64)
        pq = np.empty(price.size, np.int64)
        ct = np.empty(int(n_customers), np.int64)
        ptr = lambda a: None if a is None else a.ctypes.data
        flags = self.lib.oracle_aggregate_lifecycle(oid.ctypes.data, pidx.ctypes.data, q.ctypes.data, oid.size,
                                                    cust.ctypes.data, cust.size, int(order_id_base), int(n_customers),
                                                    price.ctypes.data, price.size, ptr(lp), ptr(st), ptr(de),
                                                    int(descuento_mode), tot.ctypes.data, pq.ctypes.data, ct.ctypes.data)
        return tot, pq, ct, flags

    def synth_prices(self, n_products):
        out = np.empty(n_products, np.int64)
        self.lib.oracle_synth_prices(n_products, out.ctypes.data)
        return out

    def synth(self, n_lines_total, n_products, n_customers, zipf=0, qty_mode=0, order_lo=0, order_hi=None,
              order_id_base=5):
        m_total = int(self.lib.oracle_synth_n_orders(n_lines_total))
        if order_hi is None:
            order_hi = m_total
        line_lo = C.c_int64(0)
        n = int(self.lib.oracle_synth_count_lines(n_lines_total, order_lo, order_hi, C.byref(line_lo)))
        m = order_hi - order_lo
        oid = np.empty(n, np.int32)
        pidx = np.empty(n, np.int32)
        qty = np.empty(n, np.int32)
        cust = np.empty(m, np.int32)
        self.lib.oracle_synth_fill(n_lines_total, n_products, n_customers, order_id_base, zipf, qty_mode,
                                   order_lo, order_hi, oid.ctypes.data, pidx.ctypes.data, qty.ctypes.data,
                                   cust.ctypes.data)
        return dict(order_id=oid, product_idx=pidx, qty=qty, order_customer=cust, n_orders_total=m_total,
                    line_lo=int(line_lo.value), order_id_base=order_id_base + order_lo)

    def ingest(self, text: bytes, codes, capacity=None):
        """Parse `detalle` text with the C oracle; returns (order_id, product_idx, qty, n_rows, n_malformed)."""
        packed = _pack_codes(codes)
        if capacity is None:
            capacity = text.count(b"\n") + 1
        oid = np.empty(capacity, np.int32)
        pidx = np.empty(capacity, np.int32)
        qty = np.empty(capacity, np.int32)
        bad = C.c_int64(0)
        n = int(self.lib.oracle_ingest_detalle(text, len(text), packed.ctypes.data, len(codes), oid.ctypes.data,
                                               pidx.ctypes.data, qty.ctypes.data, capacity, C.byref(bad)))
        k = min(n, capacity)
        return oid[:k], pidx[:k], qty[:k], n, int(bad.value)

    def emit_totals(self, total_cents, order_id_base, style, capacity=None) -> bytes:
        """Audit rows of the per-order totals as the C oracle writes them."""
        v = np.ascontiguousarray(total_cents, np.int64)
        cap = v.size * 90 + 1 if capacity is None else capacity
        buf = C.create_string_buffer(max(cap, 1))
        n = int(self.lib.oracle_emit_totals(v.ctypes.data, v.size, int(order_id_base), style, buf, cap))
        return buf.raw[:min(n, cap)]

    def emit_stock(self, product_qty, codes, style) -> bytes:
        v = np.ascontiguousarray(product_qty, np.int64)
        packed = _pack_codes(codes)
        cap = v.size * 110 + 1
        buf = C.create_string_buffer(cap)
        n = int(self.lib.oracle_emit_stock(v.ctypes.data, v.size, packed.ctypes.data, style, buf, cap))
        return buf.raw[:n]
