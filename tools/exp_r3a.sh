#!/bin/bash
# developer experiment: clean per-kernel times (four-launch sequence) at several sizes, burst periods
out=gpurun_out/r3/exp_a.txt
mkdir -p gpurun_out/r3
: > $out
for n in 100000000 72000000 60000000 50000000 33333333; do
  echo "== FUSE=0 N=$n" >> $out
  ACMPC_FUSE=0 timeout -k 10 120 ./tools/acmpc_check $n 10000 100000 0 0 20 0 >> $out 2>&1 || exit 1
done
for b in 0 1 3 4; do
  echo "== BURST=$b" >> $out
  ACMPC_K1_BURST=$b timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 20 0 >> $out 2>&1 || exit 1
done
echo "== LPL=8" >> $out
ACMPC_K1_LPL=8 timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 20 0 >> $out 2>&1 || exit 1
