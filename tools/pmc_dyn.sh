#!/bin/bash
# PMC passes for the dynamic-atmosphere step (config-4 shape): bash tools/pmc_dyn.sh [kernel substring]
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-k_extrude16_split}
OUT=$R/gpurun_out/r2_pmc_dyn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/dyn_loop.py 1024 20 > $OUT/p$i.log 2>&1 || echo fail $i
done
python3 - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:45s} {sum(v)/len(v):16.1f}  ({len(v)} dispatches)")
PY
