#!/bin/bash
# Developer tool: parse-kernel time with parts switched off (results are then wrong; timing only).
for a in 0 1 33 2; do
  echo "ACMPC_ING_ABLATE=$a"
  ACMPC_DEV_ABLATE=1 ACMPC_ING_ABLATE=$a timeout -k 10 200 python tools/ingest_timing.py 2>&1 | grep "rows," | cut -c1-90
done
