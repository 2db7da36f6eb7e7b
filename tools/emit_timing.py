"""Developer tool: throughput of the device text emit on config-3 sized results already in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ac_mpc_amd import MotorLote, emision, ingesta
from tests import ingest_cases as ic
from tests.oracle_binding import Oracle
orc = Oracle()
m = MotorLote(0)
dev = torch.device("cuda", 0)
M, P = 25_000_000, 10_000
ingesta.cargar_codigos(m, [ic.code_of(i) for i in range(P)])
g = torch.Generator(device=dev); g.manual_seed(3)
tot = torch.randint(100, 3_500_000, (M,), dtype=torch.int64, device=dev, generator=g)   # 1..7 lines of <= 5,000.00
dq = torch.randint(9_000, 11_000, (P,), dtype=torch.int64, device=dev, generator=g)
text = torch.empty(M * 83, dtype=torch.uint8, device=dev)
nb = torch.zeros(1, dtype=torch.int64, device=dev)
for style, name in ((0, "sql"), (1, "csv")):
    for kind in ("totals", "stock"):
        run = (lambda: emision.totales_device(m, tot, 5, style, text, nb)) if kind == "totals" else (lambda: emision.stock_device(m, dq, style, text, nb))
        for _ in range(2):
            run()
        assert m.estado() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        n = int(nb.item())
        rows = M if kind == "totals" else P
        # algorithmic bytes: one int64 read per row, the text written once
        alg = 8 * rows + n
        print(f"{kind:6s} {name}: {rows} rows -> {n/1e6:.1f} MB ({n/rows:.1f} B/row): {ms:.3f} ms  {rows/ms/1e6:.2f} G rows/s  "
              f"algorithmic {alg/ms/1e6:.0f} GB/s = {alg/ms/1e6/8000*100:.1f}% of 8 TB/s")
    sample = tot[:2_000_000].cpu().numpy()
    t0 = time.perf_counter(); out = orc.emit_totals(sample, 5, style); dt = time.perf_counter() - t0
    print(f"       C oracle, 1 thread, totals {name}: {sample.size/dt/1e6:.1f} M rows/s, {len(out)/dt/1e9:.3f} GB/s")
