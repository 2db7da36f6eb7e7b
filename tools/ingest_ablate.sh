#!/bin/bash
# Developer tool: parse-kernel time with parts switched off (results are then wrong; timing only).
# 1: no field conversion / dictionary, 2: no dictionary, 4: no automaton, 8: no column stores, 64: no look-back
# (without the look-back the parts add up; with it, a chunk that parses less only waits longer)
for a in ${ABLATIONS:-0 64}; do
  echo "ACMPC_ING_ABLATE=$a"
  ACMPC_DEV_ABLATE=1 ACMPC_ING_ABLATE=$a timeout -k 10 200 python tools/ingest_timing.py 2>&1 | grep "rows," | cut -c1-90 || exit 1
done
