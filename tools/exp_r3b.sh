#!/bin/bash
out=gpurun_out/r3/exp_b.txt
mkdir -p gpurun_out/r3
: > $out
export LD_LIBRARY_PATH=tools/_exp/e1
export ACMPC_DEV_ABLATE=1
for n in 100000000 72000000; do
for ab in 0 4 8 12 16 20 28; do
  echo "== N=$n K2_ABLATE=$ab fuse0" >> $out
  ACMPC_FUSE=0 ACMPC_K2_ABLATE=$ab timeout -k 10 120 ./tools/acmpc_check $n 10000 100000 0 0 20 0 >> $out 2>&1 || exit 1
done
done
for ab in 0 4 8 12; do
  echo "== N=100M K2_ABLATE=$ab fused" >> $out
  ACMPC_K2_ABLATE=$ab timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 20 0 >> $out 2>&1 || exit 1
done
