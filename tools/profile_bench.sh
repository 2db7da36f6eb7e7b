#!/bin/bash
# Profiles of the benchmark on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of bench.py: the default workload, the Zipf workload, and shard 7 of 8 of
#      the 1 G-line Zipf set (the per-GPU shape of the 8-GPU configurations)
#   2. separate --pmc FETCH_SIZE / WRITE_SIZE passes over the short C++ tool, uniform and Zipf
# Outputs land under gpurun_out/prof_$TAG; tools/make_pmc_traffic.py and a few cp's bring the summaries to profiles/.
set -u
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_trace() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- \
    python3 "$REPO/bench.py" --steps 100 --warmup 5 --no-cpu-baseline "$@" > "$OUT/bench_$name.json" 2> "$OUT/trace_$name.err" || { echo "trace $name failed"; tail -5 "$OUT/trace_$name.err"; exit 1; }
  cp "$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv"
}
run_trace bench
run_trace bench_zipf --zipf 1
run_trace shard7of8_zipf --lines-per-gpu 125000000 --zipf 1 --shard-of 7/8
for z in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_lines=100000000,zipf=${z}_$ctr" -- \
      "$REPO/tools/acmpc_check" 100000000 10000 100000 $z 0 5 0 > "$OUT/pmc_${z}_$ctr.log" 2>&1 || { echo "pmc $ctr zipf=$z failed"; tail -5 "$OUT/pmc_${z}_$ctr.log"; exit 1; }
  done
done
# un-profiled reference runs of the same commands (a profiled run clocks lower)
timeout -k 10 300 python3 "$REPO/bench.py" > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || exit 1
timeout -k 10 300 python3 "$REPO/bench.py" --zipf 1 --no-cpu-baseline > "$OUT/bench_plain_zipf.json" 2>> "$OUT/bench_plain.err" || exit 1
timeout -k 10 300 python3 "$REPO/bench.py" --lines-per-gpu 125000000 --zipf 1 --shard-of 7/8 --no-cpu-baseline > "$OUT/bench_plain_shard7of8_zipf.json" 2>> "$OUT/bench_plain.err" || exit 1
timeout -k 10 300 python3 "$REPO/bench.py" --lines-per-gpu 125000000 --shard-of 0/8 --no-cpu-baseline > "$OUT/bench_plain_shard0of8.json" 2>> "$OUT/bench_plain.err" || exit 1
echo done; ls "$OUT"
