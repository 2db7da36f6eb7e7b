"""Developer tool: throughput of the device text ingest on ~100 M rows of dump-style text already in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ac_mpc_amd import MotorLote, ingesta
from tests import ingest_cases as ic
from tests.oracle_binding import Oracle
reps_block = int(sys.argv[1]) if len(sys.argv) > 1 else 100
orc = Oracle()
m = MotorLote(0)
p = 10_000
codes = [ic.code_of(i) for i in range(p)]
ingesta.cargar_codigos(m, codes)
d = orc.synth(1_000_000, p, 100_000)
for style in os.environ.get("ACMPC_ING_STYLES", "dump,csv,statements").split(","):
    block = ic.render(d["order_id"].tolist(), d["product_idx"].tolist(), d["qty"].tolist(), style)
    dev = torch.device("cuda", 0)
    tb = torch.frombuffer(bytearray(block), dtype=torch.uint8).to(dev)
    text = tb.repeat(reps_block)
    n_rows = 1_000_000 * reps_block
    oid, pidx, qty = (torch.empty(n_rows, dtype=torch.int32, device=dev) for _ in range(3))
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(2):
        ingesta.leer_detalle_device(m, text, oid, pidx, qty, cnt)
    ablated = bool(int(os.environ.get("ACMPC_ING_ABLATE", "0")))
    try:
        m.estado()
    except Exception:
        if not ablated:          # with parts of the kernel switched off the flags mean nothing
            raise
    assert int(cnt.item()) == n_rows or ablated, (int(cnt.item()), n_rows)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ingesta.leer_detalle_device(m, text, oid, pidx, qty, cnt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nb = text.numel()
    # algorithmic bytes: the text is read once, three int32 columns are written once
    alg = nb + 12 * n_rows
    print(f"{style:5s} {n_rows} rows, {nb/1e9:.2f} GB text ({nb/n_rows:.1f} B/row): {ms:.3f} ms  "
          f"{n_rows/ms/1e6:.1f} G rows/s  {nb/ms/1e6:.0f} GB/s of text  algorithmic {alg/ms/1e6:.0f} GB/s = {alg/ms/1e6/8000*100:.1f}% of 8 TB/s")
    if ablated:
        continue
    t0 = time.perf_counter(); orc.ingest(block, codes); dt = time.perf_counter() - t0
    print(f"      C oracle, 1 thread: {1e6/dt/1e6:.2f} M rows/s, {len(block)/dt/1e9:.3f} GB/s")
    # spot check
    eo, ep, eq, n, bad = orc.ingest(block, codes)
    assert np.array_equal(oid[:1_000_000].cpu().numpy(), eo) and np.array_equal(pidx[-1_000_000:].cpu().numpy(), ep)
