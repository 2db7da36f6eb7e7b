"""Developer tool: the host-pointer ingest and emit entry points (text and columns cross PCIe) on 20 M rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ac_mpc_amd import MotorLote, emision, ingesta
from tests import ingest_cases as ic
from tests.oracle_binding import Oracle
orc = Oracle()
m = MotorLote(0)
p = 10_000
codes = [ic.code_of(i) for i in range(p)]
ingesta.cargar_codigos(m, codes)
d = orc.synth(1_000_000, p, 100_000)
block = ic.render(d["order_id"].tolist(), d["product_idx"].tolist(), d["qty"].tolist(), "dump")
text = block * 20
n_rows = 20_000_000
for _ in range(2):
    t0 = time.perf_counter()
    oid, pidx, qty = ingesta.leer_detalle(m, text, n_rows)
    dt = time.perf_counter() - t0
assert oid.size == n_rows
print(f"ingest, host pointers: {n_rows} rows, {len(text)/1e9:.2f} GB text in {dt*1e3:.1f} ms = {n_rows/dt/1e9:.2f} G rows/s, {len(text)/dt/1e9:.1f} GB/s of text (PCIe included)")
ltext = m.reservar_host(len(text), np.uint8)
ltext[:] = np.frombuffer(text, np.uint8)
lcols = tuple(m.reservar_host(n_rows, np.int32) for _ in range(3))
for _ in range(2):
    t0 = time.perf_counter()
    oid2, pidx2, qty2 = ingesta.leer_detalle(m, ltext, n_rows, salida=lcols)
    dt = time.perf_counter() - t0
assert np.array_equal(oid2, oid) and np.array_equal(pidx2, pidx) and np.array_equal(qty2, qty)
print(f"ingest, page-locked text and columns: {n_rows} rows in {dt*1e3:.1f} ms = {n_rows/dt/1e9:.2f} G rows/s, {len(text)/dt/1e9:.1f} GB/s of text (PCIe included)")
tot = np.random.default_rng(1).integers(100, 3_500_000, size=5_000_000, dtype=np.int64)
for _ in range(2):
    t0 = time.perf_counter()
    out = emision.totales(m, tot, 5)
    dt = time.perf_counter() - t0
print(f"emit, host pointers: {tot.size} rows -> {len(out)/1e9:.2f} GB text in {dt*1e3:.1f} ms = {tot.size/dt/1e9:.2f} G rows/s, {len(out)/dt/1e9:.1f} GB/s of text (PCIe included)")
ltot = m.reservar_host(tot.size, np.int64)
ltot[:] = tot
lout = m.reservar_host(tot.size * emision.MAX_FILA_TOTALES, np.uint8)
for _ in range(2):
    t0 = time.perf_counter()
    out2 = emision.totales(m, ltot, 5, salida=lout)
    dt = time.perf_counter() - t0
assert out2.tobytes() == out
print(f"emit, page-locked totals and text: {tot.size} rows in {dt*1e3:.1f} ms = {tot.size/dt/1e9:.2f} G rows/s, {len(out)/dt/1e9:.1f} GB/s of text (PCIe included)")
