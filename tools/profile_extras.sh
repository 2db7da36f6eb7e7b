#!/bin/bash
# rocprofv3 --kernel-trace --stats of the ingest, emit and lifecycle timing tools (run through gpurun from
# the repo root).  Outputs land under gpurun_out/prof_extras_$TAG; the kernel_stats files worth keeping
# are copied to profiles/ by hand.
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_extras_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for tool in ingest_timing emit_timing lifecycle_timing any_order_timing; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tool" -- \
    python3 "$REPO/tools/$tool.py" > "$OUT/$tool.out" 2> "$OUT/$tool.err" || { echo "$tool failed"; tail -5 "$OUT/$tool.err"; exit 1; }
  f=$(find "$OUT/$tool" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$OUT/${tool}_kernel_stats.csv"
  grep -v amdgpu.ids "$OUT/$tool.out"
done
