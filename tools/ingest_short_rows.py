import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ac_mpc_amd import MotorLote, ingesta
m = MotorLote(0)
p = 100
codes = [str(i) for i in range(p)]
ingesta.cargar_codigos(m, codes)
rng = np.random.default_rng(1)
for lo, hi, label in ((5, 100, "ids below 100"), (5, 100000, "ids below 1e5"), (10_000_000, 99_999_999, "8-digit ids")):
    n = 1_000_000
    oid = np.sort(rng.integers(lo, hi, n)); pidx = rng.integers(0, p, n); qty = rng.integers(1, 9, n)
    block = "".join(f"{a},{b},{c}\n" for a, b, c in zip(oid.tolist(), pidx.tolist(), qty.tolist())).encode()
    dev = torch.device("cuda", 0)
    reps = 50
    text = torch.frombuffer(bytearray(block), dtype=torch.uint8).to(dev).repeat(reps)
    rows = n * reps
    cols = [torch.empty(rows, dtype=torch.int32, device=dev) for _ in range(3)]
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(2):
        ingesta.leer_detalle_device(m, text, *cols, cnt)
    m.estado()
    assert int(cnt.item()) == rows
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ingesta.leer_detalle_device(m, text, *cols, cnt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label}: {len(block)/n:.1f} B/row, {rows} rows, {text.numel()/1e9:.2f} GB: {ms:.3f} ms, {rows/ms/1e6:.1f} G rows/s, {text.numel()/ms/1e6:.0f} GB/s", flush=True)
