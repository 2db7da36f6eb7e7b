"""Developer tool: time the any-order path on config-3 sized data (grouped, pair-interleaved, shuffled rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ac_mpc_amd import MotorLote, sintetico
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
m = MotorLote(0)
ds = sintetico.generar(m, n)
dev = ds.order_id.device
tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
pq = torch.empty(10_000, dtype=torch.int64, device=dev)
ct = torch.empty(100_000, dtype=torch.int64, device=dev)
def run(oid, pidx, qty, label, any_order):
    m.orden_lineas(any_order)
    for _ in range(2):
        m.agregar_device(oid, pidx, qty, ds.order_customer, ds.order_id_base, 100_000, tot, pq, ct)
    m.estado()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.agregar_device(oid, pidx, qty, ds.order_customer, ds.order_id_base, 100_000, tot, pq, ct)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label:34s} {ms:8.3f} ms/step  {n/ms/1e6:8.1f} G lines/s  sum={int(tot.sum())}")
run(ds.order_id, ds.product_idx, ds.qty, "grouped rows, grouped path", False)
run(ds.order_id, ds.product_idx, ds.qty, "grouped rows, any-order path", True)
k = (n // 8) * 8
def inter(t):
    u = t.clone(); u[:k] = t[:k].view(-1, 2, 4).flip(1).reshape(-1); return u
run(inter(ds.order_id), inter(ds.product_idx), inter(ds.qty), "4-row blocks swapped, any-order", True)
perm = torch.randperm(n, device=dev)
so, sp, sq = ds.order_id[perm], ds.product_idx[perm], ds.qty[perm]
run(so, sp, sq, "fully shuffled rows, any-order", True)
# the same shuffled rows with every lifecycle column (line prices, order states, discounts in per cent)
g = torch.Generator(device=dev); g.manual_seed(11)
lp = torch.randint(1, 500_000, (n,), dtype=torch.int64, device=dev, generator=g)
stt = (torch.randint(0, 10, (ds.n_orders,), device=dev, generator=g) != 0).to(torch.uint8)
de = torch.randint(0, 101, (ds.n_orders,), dtype=torch.int32, device=dev, generator=g)
def run_life(label, **cols):
    m.orden_lineas(True)
    call = lambda: m.agregar_device(so, sp, sq, ds.order_customer, ds.order_id_base, 100_000, tot, pq, ct,
                                    descuento_porcentaje=True, **cols)
    for _ in range(2):
        call()
    m.estado()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:34s} {e0.elapsed_time(e1) / 5:8.3f} ms/step  path {m.camino_lineas()}")
run_life("shuffled + line prices", precio_linea=lp)
run_life("shuffled + order states", estado_pedido=stt)
run_life("shuffled + discounts", descuento=de)
run_life("shuffled + all three columns", precio_linea=lp, estado_pedido=stt, descuento=de)
