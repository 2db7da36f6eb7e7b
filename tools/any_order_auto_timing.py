"""Developer tool: cost of the automatic any-order fallback on grouped config-3 data (it should be three
near-empty launches) and on the same data with one swapped pair of rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ac_mpc_amd import MotorLote, sintetico
m = MotorLote(0)
dev = torch.device("cuda", 0)
n, p, c = 100_000_000, 10_000, 100_000
data = sintetico.generar(m, n, p, c)
m.cargar_productos(data.price_cents)
mo = data.order_customer.numel()
tot = torch.empty(mo, dtype=torch.int64, device=dev)
pq = torch.empty(p, dtype=torch.int64, device=dev)
ct = torch.empty(c, dtype=torch.int64, device=dev)
def run(label, modo, oid):
    m.orden_lineas(modo)
    for _ in range(3):
        m.agregar_device(oid, data.product_idx, data.qty, data.order_customer, data.order_id_base, c, tot, pq, ct)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.agregar_device(oid, data.product_idx, data.qty, data.order_customer, data.order_id_base, c, tot, pq, ct)
    e1.record(); torch.cuda.synchronize()
    try:
        m.estado()
        ok = "ok"
    except Exception as ex:
        ok = "reported: " + str(ex)[:60]
    print(f"{label:34s} {e0.elapsed_time(e1)/20:.3f} ms/step  {ok}")
run("grouped rows, GROUPED", False, data.order_id)
run("grouped rows, AUTO", "auto", data.order_id)
oid = data.order_id.clone()
k = 50_000_001
oid[k], oid[k + 40] = oid[k + 40].item(), oid[k].item()
run("one swapped pair, AUTO", "auto", oid)
run("one swapped pair, ANY", True, oid)
run("one swapped pair, GROUPED", False, oid)
