#!/bin/bash
# Round profile collection on a 1-GPU MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Writes raw rocprofv3 output under gpurun_out/<tag>_* and the tracked summaries profiles/<tag>_config2.md (bench line, kernel stats,
# PMC tables of the fused kernel, HBM traffic) and profiles/<tag>_next_rows.md (kernel stats of bench.py --config 3/4/5, the K4 and
# Shack-Hartmann loops).  Counters are collected in their own passes with --kernel-trace only.
# gpurun only brings gpurun_out/ back: run the collection on the box, then `bash tools/collect_profiles.sh r02 summarize` here.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
if [ "$2" != "summarize" ]; then
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (un-profiled), then the same command under rocprofv3 --stats
python3 $R/bench.py > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -- python3 $R/bench.py --no-cpu-baseline --no-parity > $OUT/${TAG}_prof_c2.log 2>&1
# 2. PMC passes for the fused kernel (config-2 step loop)
bash $R/tools/pmc_collect.sh gpurun_out/${TAG}_pmc > $OUT/${TAG}_pmc.log 2>&1
bash $R/tools/pmc_memory.sh gpurun_out/${TAG}_pmc_mem > $OUT/${TAG}_pmc_mem.txt 2>&1
cd /tmp
# 3. the other configs: bench line + kernel stats
for c in 3 4 5; do
  python3 $R/bench.py --config $c --no-cpu-baseline > $OUT/${TAG}_bench_c$c.json 2> $OUT/${TAG}_bench_c$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c$c -- python3 $R/bench.py --config $c --no-cpu-baseline --no-parity > $OUT/${TAG}_prof_c$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_k4 -- python3 $R/tools/k4_loop.py > $OUT/${TAG}_prof_k4.log 2>&1
python3 $R/tools/sh_loop2.py 1024 256 64 single > $OUT/${TAG}_sh_loop.log 2>&1
python3 $R/tools/k4_loop.py > $OUT/${TAG}_k4_loop.log 2>&1
python3 $R/tools/dyn_loop.py 1024 200 > $OUT/${TAG}_dyn_loop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_sh -- python3 $R/tools/sh_loop2.py 1024 256 64 single > $OUT/${TAG}_prof_sh.log 2>&1
python3 $R/tools/single_env_latency.py > $OUT/${TAG}_single.log 2>&1
# 4. config-2 shape at B = 4096 (screens + tables 0.9 GB: four times the 256 MiB Infinity Cache): bench line, kernel stats, PMC of the fused kernel
python3 $R/bench.py --batch 4096 --no-cpu-baseline --no-parity > $OUT/${TAG}_bench_c2_b4096.json 2> $OUT/${TAG}_bench_c2_b4096.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2_b4096 -- python3 $R/bench.py --batch 4096 --no-cpu-baseline --no-parity > $OUT/${TAG}_prof_c2_b4096.log 2>&1
bash $R/tools/pmc_collect.sh gpurun_out/${TAG}_pmc_b4096 --B 4096 > $OUT/${TAG}_pmc_b4096.log 2>&1
# 5. PMC of the reset kernels (semi_dynamic reset of 4096 envs: two-band synthesis + packing) and of the other rows' dominant kernels
bash $R/tools/pmc_kernels.sh gpurun_out/${TAG}_pmc_reset $R/tools/reset_loop.py --B 4096 --episodes 1 > $OUT/${TAG}_pmc_reset.log 2>&1
bash $R/tools/pmc_kernels.sh gpurun_out/${TAG}_pmc_sh $R/tools/sh_loop2.py 1024 512 20 single > $OUT/${TAG}_pmc_sh.log 2>&1
# 6. the dynamic atmosphere's int8 extrusion kernels (config-4 shape without the policy kernel): kernel trace + PMC
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_dyn -- python3 $R/tools/dyn_loop.py 1024 200 > $OUT/${TAG}_prof_dyn.log 2>&1
bash $R/tools/pmc_kernels.sh gpurun_out/${TAG}_pmc_x8 $R/tools/dyn_loop.py 1024 200 > $OUT/${TAG}_pmc_x8.log 2>&1
fi
cd $R
# 4. summaries (from the raw files under gpurun_out/)
python3 tools/summarize_prof.py --stats gpurun_out/${TAG}_prof_c2 --pmc gpurun_out/${TAG}_pmc --pmc-mem gpurun_out/${TAG}_pmc_mem --tag ${TAG}_config2 \
        --bench-json gpurun_out/${TAG}_bench_c2.json > /dev/null
python3 tools/summarize_prof.py --stats gpurun_out/${TAG}_prof_c2_b4096 --pmc gpurun_out/${TAG}_pmc_b4096 --tag ${TAG}_config2_b4096 \
        --bench-json gpurun_out/${TAG}_bench_c2_b4096.json --traffic-json ${TAG}_traffic_b4096.json > /dev/null
{
  echo "# ${TAG}_reset_pmc — counters of the semi_dynamic reset kernels (tools/reset_loop.py --B 4096 --episodes 1: launches of 4096 envs)"; echo
  python3 tools/summarize_pmc.py gpurun_out/${TAG}_pmc_reset --tag ${TAG}_reset --kernels k_screen2_rows k_screen2_cols k_screen_means k_pack_tiles
} > profiles/${TAG}_reset_pmc.md
{
  echo "# ${TAG}_sh_pmc — counters of the Shack-Hartmann kernels at config 5's pupil (tools/sh_loop2.py 1024 512 20 single: launches of 1024 envs, N = 512)"; echo
  python3 tools/summarize_pmc.py gpurun_out/${TAG}_pmc_sh --tag ${TAG}_sh --kernels k_phase_mfma k_sh_rows_sep k_sh_cols_sep
} > profiles/${TAG}_sh_pmc.md
{
  echo "# ${TAG}_x8_pmc — the dynamic atmosphere's int8 extrusion kernels (tools/dyn_loop.py 1024 200: B = 1024, N = 256, v = 10 m/s, launches of 1024 envs)"; echo
  python3 tools/summarize_stats.py gpurun_out/${TAG}_prof_dyn "kernels of the dynamic step loop" "rocprofv3 --kernel-trace --stats -- python3 tools/dyn_loop.py 1024 200"
  echo '```'; grep -h "us per step" gpurun_out/${TAG}_dyn_loop.log; echo '```'; echo
  python3 tools/summarize_pmc.py gpurun_out/${TAG}_pmc_x8 --tag ${TAG}_x8 --kernels k_x8_product k_x8_prepare k_x8_plan k_fused_tab
} > profiles/${TAG}_x8_pmc.md
python3 - <<PY
import json
d = json.load(open("profiles/${TAG}_reset_pmc.json"))
d["envs_per_launch"] = 4096
d["source"] = "${TAG}_reset_pmc"
json.dump(d, open("profiles/reset_pmc_latest.json", "w"), indent=1)
d = json.load(open("profiles/${TAG}_sh_pmc.json"))
d["envs_per_launch"] = 1024
d["n_pupil"] = 512
d["source"] = "${TAG}_sh_pmc"
json.dump(d, open("profiles/sh_pmc_latest.json", "w"), indent=1)
PY
{
  echo "# ${TAG}_next_rows — bench lines and per-kernel time of the other BASELINE configs and of the SURVEY 8(f) rows (rocprofv3 --kernel-trace --stats, 1x MI355X)"
  echo
  echo "Bench lines are from un-profiled runs of the same commands; kernel tables from the profiled run."
  echo
  for c in 3 4 5; do
    echo "## bench.py --config $c"; echo; echo '```json'; tail -1 gpurun_out/${TAG}_bench_c$c.json; echo '```'; echo
    python3 tools/summarize_stats.py gpurun_out/${TAG}_prof_c$c "kernels of config $c" "rocprofv3 --kernel-trace --stats -- python3 bench.py --config $c --no-cpu-baseline --no-parity"
  done
  python3 tools/summarize_stats.py gpurun_out/${TAG}_prof_k4 "K4 batched focal fields (aog_focal_images, B=1024, N=256)" "rocprofv3 --kernel-trace --stats -- python3 tools/k4_loop.py"
  echo '```'; grep aog_focal_images gpurun_out/${TAG}_prof_k4.log; echo '```'; echo
  python3 tools/summarize_stats.py gpurun_out/${TAG}_prof_sh "Shack-Hartmann loop (SH_step + step, B=1024, N=256, complex64 transforms)" "rocprofv3 --kernel-trace --stats -- python3 tools/sh_loop2.py 1024 256 64 single"
  echo "Un-profiled runs of the same loops (the profiler's per-launch overhead inflates the loop times above):"; echo
  echo '```'; grep -h "per SH_step" gpurun_out/${TAG}_sh_loop.log; grep -h aog_focal_images gpurun_out/${TAG}_k4_loop.log; grep -h "us per step" gpurun_out/${TAG}_dyn_loop.log; echo '```'; echo
  echo "## single-env drop-in latency (tools/single_env_latency.py)"; echo; echo '```'; cat gpurun_out/${TAG}_single.log | grep AOEnv; echo '```'
  if [ -f gpurun_out/${TAG}_share6.json ]; then
    echo; echo "## multi-rank rehearsals on ONE card (AOG_BENCH_SHARE_GPU=1: every rank on card 0, gloo; the pool allows six GPU processes)"; echo
    echo "\`AOG_BENCH_SHARE_GPU=1 python bench.py --gpus 6 --batch 256 --steps 40 --warmup 10 --no-cpu-baseline\` (rc 0; the ranks share the card, so the value says nothing about scaling):"; echo
    echo '```json'; tail -1 gpurun_out/${TAG}_share6.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('metric', 'value', 'n_gpus', 'steps', 'ms_per_step', 'scaling', 'timing')}))"; echo '```'
  fi
  if [ -f gpurun_out/${TAG}_share4_c4.json ]; then
    echo; echo "\`AOG_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --config 4 --batch 256 --steps 60 --warmup 30 --no-cpu-baseline\` (dynamic atmosphere: int8 extrusion with its side streams in four processes on one card, rc 0):"; echo
    echo '```json'; tail -1 gpurun_out/${TAG}_share4_c4.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('metric', 'value', 'n_gpus', 'steps', 'ms_per_step', 'scaling', 'timing')}))"; echo '```'
  fi
} > profiles/${TAG}_next_rows.md
echo done
