#!/bin/bash
# Memory-path counters of the fused kernel (latency and stall view), one rocprofv3 --pmc pass per set with --kernel-trace only.
# (No TA_* counters: that set made rocprofv3 abort and sit silent for minutes on this pool.)
# usage (on the GPU box, from the repo root): bash tools/pmc_memory.sh <outdir> [step_loop args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/m$i -- python3 $R/tools/step_loop.py --steps 5 "$@" > $OUT/m$i.log 2>&1 || echo "pmc set $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:45s} {sum(v)/len(v):16.1f}  ({len(v)} dispatches)")
PY
