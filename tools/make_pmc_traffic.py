"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh into profiles/pmc_traffic.json.

    python tools/make_pmc_traffic.py gpurun_out/prof_TAG TAG

One entry per workload "lines=<n>,zipf=<z>" for the dominant kernel (k_lines): HBM bytes per launch =
2 x FETCH_SIZE (gfx950 reports half the bytes of 16-byte-per-lane streaming reads) + WRITE_SIZE, both in
KB in the CSVs.  The entry carries a digest of the library sources it was collected with; bench.py does not
quote a record whose digest differs from the sources it runs.

The factor 2 is calibrated for 16-byte-per-lane streaming loads only.  It is applied per kernel, and only where
the kernel's raw count lies below what the kernel cannot avoid fetching (k_lines: 12 B per line; the customer
pass: 12 B per order; the tail: the customer slabs it merges); both figures are recorded side by side, and the
step total is kept as a bracket: raw counts where the factor is not validated (lower) and doubled counts
everywhere (upper).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_digest  # noqa: E402


def averages(directory):
    acc = {}
    # (results of several runs may lie side by side in a directory that was fetched more than once: the newest counts)
    files = sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "acmpc::k_" in k and "synth" not in k:
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def orders_of(out_dir, name, n_lines):
    """Orders of the workload: the tool prints "N=... M=..." at the top of its log; mean 4 lines per order otherwise."""
    import re
    z = name.split("zipf=")[1]
    for log in glob.glob(os.path.join(out_dir, f"pmc_{z}_*.log")):
        m = re.search(r"N=(\d+) M=(\d+)", open(log).read())
        if m and int(m.group(1)) == n_lines:
            return int(m.group(2))
    return n_lines // 4


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    digest = kernel_source_digest()
    raw, rec = {}, {}
    for wl in sorted(glob.glob(os.path.join(out_dir, "pmc_*_FETCH_SIZE"))):
        name = os.path.basename(wl)[4:-len("_FETCH_SIZE")]           # e.g. lines=100000000,zipf=0
        fetch = averages(wl)
        write = averages(wl.replace("FETCH_SIZE", "WRITE_SIZE"))
        raw[name] = {k: {"FETCH_SIZE": fetch[k].get("FETCH_SIZE"), "WRITE_SIZE": write.get(k, {}).get("WRITE_SIZE")} for k in fetch}
        lines = [k for k in fetch if "k_lines<" in k]
        if not lines:
            continue
        k = lines[0]
        f_kb, w_kb = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
        # every kernel of one step (the generator is not part of it): raw and doubled fetch counts, and the
        # bytes the kernel cannot avoid fetching, which decide whether the doubled count is the one to use
        n_lines = int(name.split(",")[0].split("=")[1])
        n_orders = orders_of(out_dir, name, n_lines)
        # (the tail merges the partial sums that the customer pass wrote: it reads at least what that pass wrote)
        slabs = sum((write.get(kk, {}).get("WRITE_SIZE") or 0.0) * 1024 for kk in fetch if "k_customers" in kk)
        floor = {"k_lines<": 12.0 * n_lines, "k_customers": 12.0 * n_orders, "k_tail": slabs}
        step, detail = {}, {}
        lower = upper = 0.0
        for kk in fetch:
            if "k_synth" in kk:
                continue
            raw_b = fetch[kk]["FETCH_SIZE"] * 1024
            wr_b = (write.get(kk, {}).get("WRITE_SIZE") or 0.0) * 1024
            need = next((v for pat, v in floor.items() if pat in kk), 0.0)
            validated = need > 0 and raw_b < need <= 2 * raw_b * 1.02
            used = 2 * raw_b if validated else raw_b
            short = kk.split("acmpc::")[-1]
            step[short] = 2 * raw_b + wr_b
            detail[short] = {"fetch_raw": raw_b, "fetch_doubled": 2 * raw_b, "fetch_floor_algorithmic": need,
                             "factor_2_validated": validated, "write": wr_b}
            lower += used + wr_b
            upper += 2 * raw_b + wr_b
        rec[name] = {
            "step_hbm_bytes": upper,
            "step_hbm_bytes_lower": lower,
            "step_hbm_bytes_note": "upper: FETCH_SIZE doubled in every kernel; lower: doubled only where the raw count "
                                   "lies below the kernel's unavoidable fetch (see step_hbm_detail)",
            "step_hbm_bytes_by_kernel": step,
            "step_hbm_detail": detail,
            "hbm_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024,
            "fetch_bytes_corrected": 2 * f_kb * 1024,
            "write_bytes": w_kb * 1024,
            "kernel": k,
            "source_digest": digest,
            "source": f"profiles/{tag}_pmc_fetch_write_raw.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over "
                      f"tools/acmpc_check; FETCH_SIZE doubled per the gfx950 correction for 16-byte-per-lane streaming loads, "
                      f"WRITE_SIZE as counted)",
        }
    json.dump({"about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, values in KB, averages per launch",
               "source_digest": digest, "workloads": raw},
              open(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write_raw.json"), "w"), indent=1)
    json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
