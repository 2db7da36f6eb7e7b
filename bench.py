#!/usr/bin/env python3
"""Benchmark of the checkout-aggregation hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (starts the N ranks itself, as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the whole hot path (order-line kernel, customer kernel with the fix-up of the
orders cut by wave boundaries riding along, merge; below 4 M orders the fix-up is a launch of its own; for
N > 1 the single all-reduce of the stock-delta and customer vectors) over one batch of synthetic
store.sql-schema order lines that already sits in HBM.  Every rank owns a contiguous order-id range of
one global data set of N x lines-per-gpu lines (weak scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E vendor peak (8 TB/s)
SUSTAINED_HBM_GBS = 6300.0     # what the part's HBM sustains on plain streams (vendor guide; about 79 % of the peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)     # about 0.15 s of GPU time: long enough to be seen by samplers
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--lines-per-gpu", type=int, default=100_000_000)
    ap.add_argument("--products", type=int, default=10_000)
    ap.add_argument("--customers", type=int, default=100_000)
    ap.add_argument("--zipf", type=int, default=0, help="1: Zipf(s=1) product popularity")
    ap.add_argument("--shard-of", type=str, default="",
                    help="R/W with --gpus 1: this GPU holds shard R of W of the set of W x lines-per-gpu lines, at its "
                         "true global offsets (the per-GPU shape of the sharded 1 B-line configurations); no exchange")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def host_threads() -> int:
    """Threads for the CPU baseline: the CPUs this process may run on, capped by the box's CPU share
    (its *_NUM_THREADS setting) when one is set; never more than the machine has."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        allowed = min(allowed, int(share))
    return max(1, min(allowed, os.cpu_count() or 1))


def copy_rate_gbs(dev, n_bytes: int = 1 << 30) -> float:
    """What the part delivers, in this run, to a plain device-to-device copy of `n_bytes` (torch's copy
    kernel): GB/s over the bytes read plus the bytes written, best of a few repetitions.  The yardstick
    beside the vendor peak; never used as `peak`."""
    import torch
    src = torch.empty(n_bytes // 8, dtype=torch.int64, device=dev)
    dst = torch.empty_like(src)
    src.random_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(8):
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del src, dst
    return 2.0 * n_bytes / (best * 1e-3) / 1e9


def kernel_source_digest() -> str:
    """sha256 over the device and host sources of libacmpc.so: a traffic record collected with other
    kernels than the ones that run now must not be quoted."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ac_mpc_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def comunicador_capi(motor, rank: int, world: int, dist, timeout_s: float = 120.0):
    """The library's own RCCL communicator for this rank, or None (on every rank alike) when some rank
    cannot load RCCL or does not get through ncclCommInitRank in time; the caller then falls back to
    torch.distributed for the one all-reduce."""
    import threading
    import torch
    from ac_mpc_amd import reparto
    dev = torch.device("cuda", motor.device)

    def everyone(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    ident, ok = None, True
    try:
        ident = reparto.nuevo_id()            # every rank: proves that RCCL loads here; only rank 0's id is used
    except Exception as e:                    # noqa: BLE001
        print(f"[bench] rank {rank}: {e}", file=sys.stderr, flush=True)
        ok = False
    if not everyone(ok):
        return None
    box = [ident]
    dist.broadcast_object_list(box, src=0)
    got = {}

    def join():
        try:
            got["com"] = reparto.ComunicadorC(motor, rank, world, box[0])
        except Exception as e:                # noqa: BLE001
            got["err"] = e

    th = threading.Thread(target=join, daemon=True)
    th.start()
    th.join(timeout_s)
    ok = "com" in got
    if not ok:
        print(f"[bench] rank {rank}: acmpc_comm_init_rank {'timed out' if th.is_alive() else got.get('err')}",
              file=sys.stderr, flush=True)
    if not everyone(ok):
        return None
    return got["com"]


def cpu_baseline(ds, seconds: float):
    """Time the C oracle (a port: the reference's PHP+MySQL path cannot run here) on a bounded prefix of
    the same columns: a few passes on one thread, then about `seconds` of wall time on all the host
    threads this process may use (orders cut into one contiguous range per thread)."""
    import numpy as np
    from tests.oracle_binding import Oracle
    orc = Oracle()
    n_s = min(ds.n_lines, 25_000_000)
    oid = ds.order_id[:n_s].cpu().numpy()
    # cut the sample on an order boundary
    last = int(oid[-1])
    if n_s < ds.n_lines:
        n_s = int(np.searchsorted(oid, last, side="left"))
        oid = oid[:n_s]
    m_s = int(oid[-1]) - ds.order_id_base + 1
    pidx = ds.product_idx[:n_s].cpu().numpy()
    qty = ds.qty[:n_s].cpu().numpy()
    cust = ds.order_customer[:m_s].cpu().numpy()
    args = (oid, pidx, qty, cust, ds.order_id_base, ds.n_customers, ds.price_cents)
    out = (np.empty(m_s, np.int64), np.empty(len(ds.price_cents), np.int64), np.empty(ds.n_customers, np.int64))

    def rate(threads, budget):
        orc.aggregate(*args, threads=threads, out=out)     # warm
        t0 = time.perf_counter()
        reps = 0
        while True:
            orc.aggregate(*args, threads=threads, out=out)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget or reps >= 20000:
                return n_s * reps / el, reps, el

    one, _, _ = rate(1, min(2.0, seconds))
    cores = host_threads()
    if cores == 1:
        many, reps, el = one, 0, 0.0
    else:
        many, reps, el = rate(cores, seconds)
    return {"value": max(many, one), "unit": "order lines/s", "cores": cores if many >= one else 1,
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"first {n_s} lines ({m_s} orders) of the same columns; {reps} passes of the C oracle on "
                      f"{cores} threads in {el:.1f} s; one thread: {one / 1e9:.2f} G lines/s"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` started plainly: run the N ranks (one process per GPU) under
    torch.distributed.run as a CHILD process and hand back its exit code.  Rank 0's JSON line reaches the
    caller through the inherited stdout.  Nothing in this process has touched the GPU (no HIP call, no torch
    import), and it is never replaced by another program."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process",
          file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit(f"[bench] rank {rank} of {world}: no GPU visible to this process; bench.py measures the HIP "
                         "path only (there is no CPU fallback)")
    from ac_mpc_amd import MotorLote, reparto, sintetico
    # Rehearsal on a box with fewer GPUs than ranks (never used by the driver): ACMPC_BENCH_BACKEND=gloo
    # puts every rank on ACMPC_BENCH_DEVICE and moves the one all-reduce to the CPU.
    backend = os.environ.get("ACMPC_BENCH_BACKEND", "nccl")
    if "ACMPC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ACMPC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    motor = MotorLote(local_rank)
    n_total = args.lines_per_gpu * world
    gen_rank, gen_world = rank, world
    if args.shard_of:
        if world != 1:
            raise SystemExit("--shard-of is for --gpus 1")
        gen_rank, gen_world = (int(x) for x in args.shard_of.split("/"))
        n_total = args.lines_per_gpu * gen_world
    ds = sintetico.generar(motor, n_total, args.products, args.customers, zipf=bool(args.zipf), qty_mode=0,
                           rank=gen_rank, world=gen_world)
    dev = ds.order_id.device
    P, Cn = args.products, args.customers
    tot = torch.empty(ds.n_orders, dtype=torch.int64, device=dev)
    # [product_qty | customer_total_cents] of a step.  With more than one rank the all-reduce of step i
    # runs on RCCL's stream while the kernels of step i + 1 run on this one, so two buffers take turns
    # (ACMPC_BENCH_OVERLAP=0: one buffer, the all-reduce waited for inside the step).
    overlap = world > 1 and backend == "nccl" and os.environ.get("ACMPC_BENCH_OVERLAP", "1") != "0"
    bufs = [torch.empty(P + Cn, dtype=torch.int64, device=dev) for _ in range(2 if overlap else 1)]
    # The exchange is issued by the library itself (ncclAllReduce behind the C ABI, the route a PHP host
    # has); ACMPC_BENCH_MERGE=torch, or a rank that cannot join, puts it back in torch.distributed.
    com = None
    if world > 1 and backend == "nccl" and os.environ.get("ACMPC_BENCH_MERGE", "capi") == "capi":
        com = comunicador_capi(motor, rank, world, dist)
    merge_path = "none (one GPU)" if world == 1 else \
        "ncclAllReduce inside libacmpc.so (acmpc_merge_device)" if com else f"torch.distributed all_reduce ({backend})"
    if overlap:
        turnos = reparto.FusionEnCursoC(bufs, com) if com else reparto.FusionEnCurso(bufs)
    else:
        turnos = None
    merged = bufs[0]

    def step():
        buf = turnos.siguiente() if overlap else merged
        motor.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, Cn, tot,
                             buf[:P], buf[P:])
        if overlap:
            turnos.lanzar()
        elif world > 1:
            if com:
                com.fusionar(merged)
            elif backend == "nccl":
                reparto.fusionar(merged)
            else:
                host = merged.cpu()
                reparto.fusionar(host)
                merged.copy_(host)

    def fence():
        if overlap:
            turnos.esperar()                 # every step's exchange is complete inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    assert motor.estado() == 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert motor.estado() == 0

    # sanity on the result of the last step (cheap, outside the timed region): every line counted once
    last = turnos.esperar() if overlap else merged
    pq, ct = last[:P], last[P:]
    total_qty = int(pq.sum().item())
    lines_per_step = ds.n_lines if args.shard_of else n_total        # what all ranks process in one step
    orders_per_step = ds.n_orders if args.shard_of else ds.n_orders_total
    assert total_qty == lines_per_step, (total_qty, lines_per_step)

    # roofline of the dominant kernel (the order-line pass), from HIP events recorded by the library on
    # the stream the kernels run on
    motor.perfilar(True)
    k_ms = {"lines": 0.0, "fixup": 0.0, "customers": 0.0, "custmerge": 0.0}
    reps = max(5, min(args.steps, 50))
    for _ in range(reps):
        motor.agregar_device(ds.order_id, ds.product_idx, ds.qty, ds.order_customer, ds.order_id_base, Cn, tot, pq, ct)
        ms = motor.ultimos_ms()
        for k in k_ms:
            k_ms[k] += ms[k] / reps
    motor.perfilar(False)
    # algorithmic bytes of one k_lines launch: three int32 columns per line, one int64 total per order,
    # one int64 price read and one stock-delta partial written per product
    b_lines = 12 * ds.n_lines + 8 * ds.n_orders + 16 * P
    achieved = b_lines / (k_ms["lines"] * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters: collected by separate rocprofv3 --pmc passes
    # (tools/profile_gpu.sh) and kept in profiles/pmc_traffic.json with a digest of the sources they were
    # collected with; a record of other kernels than the ones that ran just now is not quoted.
    traffic, traffic_note = None, "no PMC record for this workload"
    step_traffic = step_traffic_lower = None    # HBM bytes of all kernels of one step, same counters (a bracket)
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"lines={ds.n_lines},zipf={args.zipf}")
            if rec:
                if rec.get("source_digest") == kernel_source_digest():
                    traffic, traffic_note = rec.get("hbm_bytes_per_launch"), rec.get("source", "")
                    step_traffic = rec.get("step_hbm_bytes")
                    step_traffic_lower = rec.get("step_hbm_bytes_lower", step_traffic)
                else:
                    traffic_note = (f"PMC record is of sources {rec.get('source_digest')}, these are "
                                    f"{kernel_source_digest()}: not quoted")
        except Exception as e:                    # noqa: BLE001
            traffic_note = f"unreadable PMC record: {e}"
    copy_gbs = copy_rate_gbs(dev) if rank == 0 else 0.0
    b_step = 12 * ds.n_lines + 12 * ds.n_orders + 16 * P + 8 * Cn       # whole step, this GPU
    step_ms_device = sum(k_ms.values())

    if rank == 0:
        wg, wv = motor.geometria_lineas()
        out = {
            "metric": "order lines aggregated/sec (store.sql schema)",
            "value": lines_per_step * args.steps / elapsed,
            "unit": "order lines/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": (f"shard {gen_rank} of {gen_world} of a {n_total}-line set at its global offsets, on one GPU "
                             f"(per-GPU shape of BASELINE configs[{4 if args.zipf else 3}]), no exchange" if args.shard_of else
                             "BASELINE configs[2]: 100M order lines, per-order totals + per-product stock decrement "
                             "(+ per-customer totals), single GPU" if world == 1 and args.lines_per_gpu == 100_000_000 else
                             f"{n_total} order lines sharded by order-id range over {world} GPU(s), one all-reduce"),
                "lines_per_gpu": args.lines_per_gpu, "lines_total": n_total, "orders_total": ds.n_orders_total,
                "lines_this_gpu": ds.n_lines, "orders_this_gpu": ds.n_orders,
                "products": P, "customers": Cn, "product_popularity": "zipf(s=1)" if args.zipf else "uniform",
                "sharding": f"order-id range x{world}" if world > 1 else "none",
                "exchange": merge_path,
                "orders_per_sec": orders_per_step * args.steps / elapsed,
                "lines_kernel_geometry": f"{wg} workgroups x {wv} waves",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_note,
                "kernel": "k_lines", "kernel_ms": k_ms["lines"], "algorithmic_bytes_per_launch": b_lines,
                "other_kernels_ms": {k: v for k, v in k_ms.items() if k != "lines"},
                # the whole step (every kernel of it) against the same peak: 12 B per line, 12 B per order,
                # 16 B per product, 8 B per customer, over the wall time of a step on this GPU
                "whole_step_algorithmic_bytes": b_step,
                "whole_step_algorithmic_GBs": b_step / (elapsed / args.steps) / 1e9,
                "whole_step_frac": b_step / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS,
                "whole_step_device_ms": step_ms_device,
                # HBM bytes all kernels of a step really move (FETCH_SIZE / WRITE_SIZE counters, same record and
                # correction as `traffic`), over the same wall time: how busy the memory is, re-reads included
                # (an upper bound: FETCH_SIZE doubled in every kernel, which is calibrated for the streaming loads
                # of k_lines only; `_lower` doubles it only where the raw count is below the kernel's unavoidable bytes)
                "whole_step_hbm_bytes_counters": step_traffic,
                "whole_step_hbm_bytes_counters_lower": step_traffic_lower,
                "whole_step_hbm_frac_counters": (step_traffic / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS
                                                 if step_traffic and world == 1 else None),
                "whole_step_hbm_frac_counters_lower": (step_traffic_lower / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS
                                                       if step_traffic_lower and world == 1 else None),
                # against what HBM sustains rather than its peak: the dominant kernel on its counter traffic, the whole
                # step on its counter traffic (lower bracket) and on its algorithmic bytes
                "sustained_GBs": SUSTAINED_HBM_GBS,
                "frac_of_sustained": (traffic / (k_ms["lines"] * 1e-3) / 1e9 / SUSTAINED_HBM_GBS if traffic else None),
                "whole_step_frac_of_sustained": (step_traffic_lower / (elapsed / args.steps) / 1e9 / SUSTAINED_HBM_GBS
                                                 if step_traffic_lower and world == 1 else None),
                "whole_step_algorithmic_frac_of_sustained": b_step / (elapsed / args.steps) / 1e9 / SUSTAINED_HBM_GBS,
                # measured in this run: a plain device-to-device copy (read + written bytes per second)
                "device_copy_GBs": copy_gbs,
                "frac_of_device_copy": achieved / copy_gbs if copy_gbs else None,
                "launches_per_step": 3 if ds.n_orders >= 4_000_000 else 4,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ds, args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)

    if com:
        com.cerrar()
    motor.cerrar()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
