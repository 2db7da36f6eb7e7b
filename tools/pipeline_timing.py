"""Developer tool: the whole device-resident job — `detalle` text in HBM -> columns -> aggregates -> the
`update venta ...` statements in HBM — on config-3 sized data, with the time of each stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ac_mpc_amd import MotorLote, emision, ingesta, sintetico
from tests import ingest_cases as ic
m = MotorLote(0)
dev = torch.device("cuda", 0)
n, p, c = 100_000_000, 10_000, 100_000
ds = sintetico.generar(m, n, p, c)
m.cargar_productos(ds.price_cents)
codes = [ic.code_of(i) for i in range(p)]
ingesta.cargar_codigos(m, codes)
# text of the same rows: rendered on the host in blocks of 1 M rows (the renderer is Python)
parts = []
oid_h, pidx_h, qty_h = ds.order_id.cpu().numpy(), ds.product_idx.cpu().numpy(), ds.qty.cpu().numpy()
for s in range(0, n, 10_000_000):
    e = min(n, s + 10_000_000)
    block = np.char.add(np.char.add(np.char.add(oid_h[s:e].astype(str), ","), np.array(codes)[pidx_h[s:e]]), np.char.add(",", qty_h[s:e].astype(str)))
    parts.append(("\n".join(block.tolist()) + "\n").encode())
text_h = b"".join(parts)
text = torch.frombuffer(bytearray(text_h + b"\0" * 16), dtype=torch.uint8).to(dev)[: len(text_h)]
oid, pidx, qty = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
mo = ds.n_orders
tot = torch.empty(mo, dtype=torch.int64, device=dev)
pq = torch.empty(p, dtype=torch.int64, device=dev)
ct = torch.empty(c, dtype=torch.int64, device=dev)
out = torch.empty(mo * 83, dtype=torch.uint8, device=dev)
nb = torch.zeros(1, dtype=torch.int64, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for rep in range(3):
    ev[0].record()
    ingesta.leer_detalle_device(m, text, oid, pidx, qty, cnt)
    ev[1].record()
    m.agregar_device(oid, pidx, qty, ds.order_customer, ds.order_id_base, c, tot, pq, ct)
    ev[2].record()
    emision.totales_device(m, tot, ds.order_id_base, emision.SQL, out, nb)
    ev[3].record()
    torch.cuda.synchronize()
assert m.estado() == 0 and int(cnt.item()) == n
assert torch.equal(oid, ds.order_id) and torch.equal(pidx, ds.product_idx) and torch.equal(qty, ds.qty)
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
print(f"{n} rows, {len(text_h)/1e9:.2f} GB of text in, {int(nb.item())/1e9:.2f} GB of statements out")
print(f"ingest {t[0]:.3f} ms   aggregate {t[1]:.3f} ms   emit {t[2]:.3f} ms   total {sum(t):.3f} ms  = {n/sum(t)/1e6:.1f} G rows/s end to end on the device")
