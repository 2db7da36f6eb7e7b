#!/bin/bash
# Per-kernel durations of the any-order timing tool (rocprofv3 kernel trace), engine kernels only.
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_any; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -- python3 "$REPO/tools/any_order_timing.py" > "$OUT/run.txt" 2>&1 || { tail -5 "$OUT/run.txt"; exit 1; }
cat "$OUT/run.txt" | grep "ms/step"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "acmpc::" in r["Name"] and "synth" not in r["Name"]:
        print("%-60s calls %5s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
