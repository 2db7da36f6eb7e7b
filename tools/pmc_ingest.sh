#!/bin/bash
# SQ counter passes over the ingest timing tool (one rocprofv3 run per pass; <= 8 SQ counters each).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_ingest_${1:-x}; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/ingest_timing.py" 20 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "acmpc::k_ingest" in k:
            acc[k.split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
