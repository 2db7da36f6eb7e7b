#!/bin/bash
out=gpurun_out/r3/exp_e.txt
mkdir -p gpurun_out/r3
: > $out
for ab in 79 95 111 127; do
  echo "== ablate $ab" >> $out
  ACMPC_DEV_ABLATE=1 ACMPC_ING_ABLATE=$ab timeout -k 10 200 python3 tools/ingest_timing.py >> $out 2>&1 || exit 1
done
