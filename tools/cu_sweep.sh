for c in 256 224 192 160 128 96; do echo "K1_CUS=$c"; ACMPC_K1_CUS=$c timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 5 0 2>&1 | grep -i "k_lines\|lines " | tail -2; done
for c in 256 192 144 96 72 48; do echo "K2_CUS=$c"; ACMPC_K2_CUS=$c timeout -k 10 120 ./tools/acmpc_check 100000000 10000 100000 0 0 5 0 2>&1 | grep -i "cust" | tail -2; done
