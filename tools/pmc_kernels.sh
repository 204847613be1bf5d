#!/bin/bash
# PMC counters of every library kernel of one workload, one rocprofv3 --pmc pass per counter set with --kernel-trace only (the pool refuses
# --pmc together with other trace domains).  usage (on the GPU box, from the repo root):
#   bash tools/pmc_kernels.sh <outdir under the repo> <python script + args ...>
# then here:  python3 tools/summarize_pmc.py <outdir> --tag r03_reset --kernels k_screen2_rows k_screen2_cols k_pack_tiles ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || echo "pmc set $i failed"
  echo "pmc set $i done"
done
