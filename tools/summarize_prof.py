#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

  python tools/summarize_prof.py --stats gpurun_out/prof_dir --pmc gpurun_out/pmc_dir --tag r01_c2 [--bench-json line.json]

* --stats: a `rocprofv3 --kernel-trace --stats --output-format csv` directory (kernel_stats.csv)
* --pmc:   the directory written by tools/pmc_collect.sh (one sub-directory per --pmc pass)
HBM traffic per launch of the fused kernel follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in
KiB, collected in separate passes; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read,
so it is doubled: traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import argparse, collections, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("--stats"); ap.add_argument("--pmc"); ap.add_argument("--pmc-mem"); ap.add_argument("--tag", required=True)
ap.add_argument("--bench-json"); ap.add_argument("--out", default="profiles")
ap.add_argument("--traffic-json", default="traffic_latest.json", help="name of the JSON the fused kernel's traffic / instruction counts go to ('' = none)")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)
lines = [f"# {args.tag}", ""]
if args.bench_json and os.path.exists(args.bench_json):
    lines += ["bench.py line:", "", "```json", open(args.bench_json).read().strip(), "```", ""]
if args.stats:
    f = glob.glob(os.path.join(args.stats, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        lines += [f"## rocprofv3 --kernel-trace --stats ({os.path.basename(f[0])})", "",
                  "| kernel | calls | avg ns | min ns | max ns | % of GPU time |", "|---|---|---|---|---|---|"]
        for r in rows:
            n = r["Name"]
            if "aog::" in n or "3aog" in n or float(r["Percentage"]) > 3:
                lines.append(f"| `{n[:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
        lines.append("")
traffic = None
extra = {}
if args.pmc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dirs = [args.pmc] + ([args.pmc_mem] if args.pmc_mem else [])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["## PMC (separate `--pmc` passes with `--kernel-trace` only; mean per dispatch)", ""]
    for k, v in agg.items():
        if "k_fused" not in k and "k_prologue" not in k and "k_epilogue" not in k:   # the per-step kernels; set-up kernels are in the stats table
            continue
        lines.append(f"### `{k[:100]}`")
        lines.append("")
        lines.append("| counter | mean per dispatch | dispatches |")
        lines.append("|---|---|---|")
        for c in sorted(v):
            lines.append(f"| {c} | {sum(v[c]) / len(v[c]):.6g} | {len(v[c])} |")
        lines.append("")
        if "fused" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
            write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            traffic = (2 * fetch + write) * 1024
            mean = lambda c: sum(v[c]) / len(v[c]) if c in v else None
            extra = {"valu_insts_per_launch": mean("SQ_INSTS_VALU"), "mfma_insts_per_launch": mean("SQ_INSTS_MFMA"),
                     "trans_insts_per_launch": mean("SQ_INSTS_VALU_TRANS_F32"), "l1_to_l2_read_requests_per_launch": mean("TCP_TCC_READ_REQ_sum")}
            lines.append(f"HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = **{traffic / 1e6:.1f} MB** "
                         f"(FETCH_SIZE {fetch:.0f} KiB doubled per the gfx950 correction, WRITE_SIZE {write:.0f} KiB)")
            lines.append("")
            if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
                wc = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"])
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                    if c in v:
                        lines.append(f"- {c} / SQ_WAVE_CYCLES = {sum(v[c]) / len(v[c]) / wc:.3f}")
                lines.append("")
open(os.path.join(args.out, args.tag + ".md"), "w").write("\n".join(lines) + "\n")
if traffic is not None and args.traffic_json:
    json.dump({"hbm_bytes_per_launch": traffic, "source": args.tag, **{k: x for k, x in extra.items() if x is not None}},
              open(os.path.join(args.out, args.traffic_json), "w"))
print("\n".join(lines[:60]))
