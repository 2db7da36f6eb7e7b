#!/bin/bash
# Developer sweep: tiles per wave of the line pass (the byte distance between the runs of neighbouring waves).
for t in ${TPW_LIST:-0 97 104 112 128}; do
  echo "K1_TPW=$t"
  ACMPC_K1_TPW=$t timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 5 0 2>&1 | grep "whole call\|kernels (ms)\|PASS\|FAIL" || exit 1
done
