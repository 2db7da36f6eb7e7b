#!/bin/bash
# developer experiment: output bursts, CU sweeps, overlap probe
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/exp_r2a.txt
: > $L
run() { echo "== $*" >> $L; timeout -k 10 120 "$@" >> $L 2>&1 || { echo "FAILED rc=$?" >> $L; exit 1; }; }
run ./tools/acmpc_check 100000000 10000 100000 0 0 10 0
for b in 0 1 3 4 6; do echo "K1_BURST=$b" >> $L; ACMPC_K1_BURST=$b timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 10 0 2>&1 | grep -E "kernels|whole" >> $L || exit 1; done
for c in 240 224 208 192 160; do echo "K1_CUS=$c" >> $L; ACMPC_K1_CUS=$c timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 10 0 2>&1 | grep -E "kernels" >> $L || exit 1; done
for c in 192 128 96 64 48; do echo "K2_CUS=$c" >> $L; ACMPC_K2_CUS=$c timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 10 0 2>&1 | grep -E "kernels" >> $L || exit 1; done
run ./tools/overlap_probe 100000000
echo done >> $L
