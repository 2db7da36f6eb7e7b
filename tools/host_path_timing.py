"""Developer tool: time the host-pointer entry point (PCIe included) on config 3 sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ac_mpc_amd import MotorLote, sintetico
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
m = MotorLote(0)
ds = sintetico.generar(m, n)
oid, pidx, qty, cust = (t.cpu().numpy() for t in (ds.order_id, ds.product_idx, ds.qty, ds.order_customer))
m.agregar(oid, pidx, qty, cust, ds.order_id_base, ds.n_customers)
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    m.agregar(oid, pidx, qty, cust, ds.order_id_base, ds.n_customers)
dt = (time.perf_counter() - t0) / reps
print(f"host-pointer acmpc_aggregate: {n} lines in {dt*1e3:.1f} ms = {n/dt/1e9:.2f} G lines/s "
      f"({(12*n + 4*ds.n_orders)/dt/1e9:.1f} GB/s over PCIe in, pageable host memory)")
