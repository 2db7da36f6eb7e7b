"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh into profiles/pmc_traffic.json.

    python tools/make_pmc_traffic.py gpurun_out/prof_TAG TAG

One entry per workload "lines=<n>,zipf=<z>" for the dominant kernel (k_lines): HBM bytes per launch =
2 x FETCH_SIZE (gfx950 reports half the bytes of 16-byte-per-lane streaming reads) + WRITE_SIZE, both in
KB in the CSVs.  The entry carries a digest of the library sources it was collected with; bench.py does not
quote a record whose digest differs from the sources it runs.
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
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "acmpc::k_" in k and "synth" not in k:
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


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
        # every kernel of one step (the generator is not part of it), same correction
        step = {kk: 2 * fetch[kk]["FETCH_SIZE"] * 1024 + (write.get(kk, {}).get("WRITE_SIZE") or 0.0) * 1024
                for kk in fetch if "k_synth" not in kk}
        rec[name] = {
            "step_hbm_bytes": sum(step.values()),
            "step_hbm_bytes_by_kernel": {kk.split("acmpc::")[-1]: v for kk, v in step.items()},
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
