"""Developer tool: step time of the aggregation on config-3 data with and without the lifecycle columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ac_mpc_amd import MotorLote, sintetico, _lib
m = MotorLote(0)
dev = torch.device("cuda", 0)
n, p, c = 100_000_000, 10_000, 100_000
data = sintetico.generar(m, n, p, c)
m.cargar_productos(data.price_cents)
mo = data.order_customer.numel()
g = torch.Generator(device=dev); g.manual_seed(11)
lp = torch.randint(1, 500_000, (n,), dtype=torch.int64, device=dev, generator=g)
stt = (torch.randint(0, 10, (mo,), device=dev, generator=g) != 0).to(torch.uint8)
de = torch.randint(0, 101, (mo,), dtype=torch.int32, device=dev, generator=g)
tot = torch.empty(mo, dtype=torch.int64, device=dev)
pq = torch.empty(p, dtype=torch.int64, device=dev)
ct = torch.empty(c, dtype=torch.int64, device=dev)
m.perfilar(True)
for name, kw in (("plain", {}), ("line prices", dict(precio_linea=lp)), ("order state", dict(estado_pedido=stt)),
                 ("descuento %", dict(descuento=de, descuento_porcentaje=True)),
                 ("all three", dict(precio_linea=lp, estado_pedido=stt, descuento=de, descuento_porcentaje=True))):
    acc = [0.0] * 4
    reps = 10
    for i in range(3 + reps):
        m.agregar_device(data.order_id, data.product_idx, data.qty, data.order_customer, data.order_id_base, c, tot, pq, ct, **kw)
        ms = list(m.ultimos_ms().values())
        if i >= 3:
            acc = [a + b for a, b in zip(acc, ms)]
    assert m.estado() == 0
    ms = [a / reps for a in acc]
    # algorithmic bytes: 12 B/line (+8 with line prices), 12 B/order (+1 state, +4 descuento, +16 when the totals are rewritten)
    alg = 12 * n + 12 * mo + 16 * p + 8 * c + (8 * n if "precio_linea" in kw else 0) + (2 * mo if "estado_pedido" in kw else 0) + (20 * mo if "descuento" in kw else 0)
    print(f"{name:12s} lines {ms[0]:.3f}  fixup(+adjust) {ms[1]:.3f}  customers {ms[2]:.3f}  merge {ms[3]:.3f}  step {sum(ms):.3f} ms  "
          f"{n/sum(ms)/1e6:.0f} G lines/s  algorithmic {alg/sum(ms)/1e6:.0f} GB/s")
