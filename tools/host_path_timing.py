"""Developer tool: time the host-pointer entry point (PCIe included) on config 3 sizes, from ordinary
(pageable) numpy arrays and from page-locked arrays of the context (acmpc_host_alloc)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ac_mpc_amd import MotorLote, sintetico
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
m = MotorLote(0)
ds = sintetico.generar(m, n)
cols = [t.cpu().numpy() for t in (ds.order_id, ds.product_idx, ds.qty, ds.order_customer)]


def timed(label, cols, salida):
    m.agregar(*cols, ds.order_id_base, ds.n_customers, salida=salida)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        m.agregar(*cols, ds.order_id_base, ds.n_customers, salida=salida)
    dt = (time.perf_counter() - t0) / reps
    print(f"host-pointer acmpc_aggregate, {label}: {n} lines in {dt*1e3:.1f} ms = {n/dt/1e9:.2f} G lines/s "
          f"({(12*n + 4*ds.n_orders)/dt/1e9:.1f} GB/s in + {(8*ds.n_orders)/dt/1e9:.1f} GB/s out over PCIe)", flush=True)


timed("pageable host memory", cols, None)
locked = []
for c in cols:
    a = m.reservar_host(c.size, c.dtype)
    a[:] = c
    locked.append(a)
out = (m.reservar_host(ds.n_orders, np.int64), m.reservar_host(m.n_products, np.int64),
       m.reservar_host(ds.n_customers, np.int64))
timed("page-locked host memory", locked, out)
timed("page-locked columns, pageable results", locked, None)
m.cerrar()
