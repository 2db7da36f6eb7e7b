#!/bin/bash
set -o pipefail
L=gpurun_out/exp_r2b.txt
: > $L
for ab in 0 1 2 3 4; do echo "K2_ABLATE=$ab" >> $L; ACMPC_DEV_ABLATE=1 ACMPC_K2_ABLATE=$ab timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 10 0 2>&1 | grep -E "kernels|PASS|FAIL" >> $L || exit 1; done
for t in 512; do echo "K2_THREADS=$t" >> $L; ACMPC_K2_THREADS=$t timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 10 0 2>&1 | grep -E "kernels|PASS|FAIL" >> $L || exit 1; done
echo done >> $L
