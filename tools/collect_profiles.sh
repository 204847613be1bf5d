#!/bin/bash
# Round profile collection on a 1-GPU MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Writes raw rocprofv3 output under gpurun_out/<tag>_* and the tracked summaries profiles/<tag>_config2.md (bench line, kernel stats,
# PMC tables of the fused kernel, HBM traffic) and profiles/<tag>_next_rows.md (kernel stats of bench.py --config 3/4/5, the K4 and
# Shack-Hartmann loops).  Counters are collected in their own passes with --kernel-trace only.
# gpurun only brings gpurun_out/ back: run the collection on the box, then `bash tools/collect_profiles.sh r02 summarize` here.
TAG=${1:-r02}
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
fi
cd $R
# 4. summaries (from the raw files under gpurun_out/)
python3 tools/summarize_prof.py --stats gpurun_out/${TAG}_prof_c2 --pmc gpurun_out/${TAG}_pmc --pmc-mem gpurun_out/${TAG}_pmc_mem --tag ${TAG}_config2 \
        --bench-json gpurun_out/${TAG}_bench_c2.json > /dev/null
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
} > profiles/${TAG}_next_rows.md
echo done
