#!/bin/bash
# developer experiment: what the statement-style ingest time is made of (parts of the kernel switched off)
out=gpurun_out/r3/exp_d.txt
mkdir -p gpurun_out/r3
: > $out
for ab in 0 64 68 71 79; do
  echo "== ablate $ab" >> $out
  ACMPC_DEV_ABLATE=1 ACMPC_ING_ABLATE=$ab timeout -k 10 200 python3 tools/ingest_timing.py >> $out 2>&1 || exit 1
done
