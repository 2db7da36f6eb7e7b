#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_life; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/p1" -- python3 "$REPO/tools/lifecycle_timing.py" > "$OUT/p1.log" 2>&1 || { echo failed; tail -3 "$OUT/p1.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_lines" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, len(next(iter(d.values()))))
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
