#!/bin/bash
# developer experiment: 32-bit staging slots in k_lines — longer output bursts with the full product table
out=gpurun_out/r3/exp_e2.txt
mkdir -p gpurun_out/r3
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 20 0 >> $out 2>&1 || exit 1; }
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=10
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=11
run ACMPC_K1_STAGE=1152 ACMPC_K1_BURST=12
run ACMPC_K1_STAGE=1152 ACMPC_K1_BURST=13
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=9
