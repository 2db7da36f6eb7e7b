"""Developer tool: the allocation-free entry point captured in a HIP graph (torch.cuda.CUDAGraph) and replayed,
against the same call issued eagerly; the 1 M-line configuration, where the step is bound by launch latency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ac_mpc_amd import MotorLote, sintetico
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = MotorLote(0)
ds = sintetico.generar(m, n, 10_000, 100_000)
dev = ds.order_id.device
tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
pq = torch.empty(10_000, dtype=torch.int64, device=dev)
ct = torch.empty(100_000, dtype=torch.int64, device=dev)
def call(stream=None, noalloc=True):
    m.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, 100_000, tot, pq, ct,
                     stream=stream, sin_reservas=noalloc)
def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
call(); assert m.estado() == 0
ref = (tot.clone(), pq.clone(), ct.clone())
print(f"{n} lines, eager, growing workspace : {timed(lambda: call(noalloc=False)):.4f} ms/step")
print(f"{n} lines, eager, no allocation     : {timed(call):.4f} ms/step")
s = torch.cuda.Stream(device=dev)
g = torch.cuda.CUDAGraph()
tot.zero_(); pq.zero_(); ct.zero_()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    call(stream=torch.cuda.current_stream(dev).cuda_stream)
g.replay(); torch.cuda.synchronize()
assert m.estado() == 0
assert torch.equal(tot, ref[0]) and torch.equal(pq, ref[1]) and torch.equal(ct, ref[2]), "graph replay differs"
print(f"{n} lines, graph replay             : {timed(g.replay):.4f} ms/step")
