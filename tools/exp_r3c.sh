#!/bin/bash
# developer experiment: larger output bursts in k_lines (a small product table leaves the LDS for them)
out=gpurun_out/r3/exp_c.txt
mkdir -p gpurun_out/r3
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 ./tools/acmpc_check 100000000 2000 100000 0 0 20 0 >> $out 2>&1 || exit 1; }
run ACMPC_K1_STAGE=512 ACMPC_K1_BURST=2
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=2
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=4
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=6
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=8
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=10
run ACMPC_K1_STAGE=1152 ACMPC_K1_BURST=12
run ACMPC_K1_STAGE=1024 ACMPC_K1_BURST=0
