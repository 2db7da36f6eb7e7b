"""Developer tool: host-side enqueue time of one aggregation call against the wall time per call."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ac_mpc_amd import MotorLote, sintetico
m = MotorLote(0); dev = torch.device("cuda", 0)
for n in (1_000_000, 100_000_000):
    ds = sintetico.generar(m, n, 10_000, 100_000); m.cargar_productos(ds.price_cents)
    tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev); pq = torch.empty(10_000, dtype=torch.int64, device=dev); ct = torch.empty(100_000, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20): m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, 100_000, tot, pq, ct, stream=st)
    torch.cuda.synchronize()
    reps = 500 if n < 10_000_000 else 50
    t0 = time.perf_counter()
    for _ in range(reps): m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, 100_000, tot, pq, ct, stream=st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e6*(t1-t0)/reps:.1f} us per call, wall incl. GPU {1e6*(t2-t0)/reps:.1f} us per call")
