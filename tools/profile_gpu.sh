#!/bin/bash
# Profile the benchmark on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of bench.py            -> per-kernel average durations
#   2. separate --pmc passes over the short C++ tool           -> HBM read / write bytes per launch
# Outputs land under gpurun_out/prof_$TAG; the summaries worth keeping are copied to profiles/ by hand.
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
  python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" || { echo "trace run failed"; tail -5 "$OUT/trace.err"; exit 1; }

for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -- \
    "$REPO/tools/acmpc_check" 100000000 10000 100000 0 0 5 0 > "$OUT/pmc_$ctr.log" 2>&1 || { echo "pmc $ctr failed"; tail -5 "$OUT/pmc_$ctr.log"; exit 1; }
done
find "$OUT" -name "*.csv" | head -40
