#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_kernels.sh, as a markdown section (stdout) and a small JSON
(profiles/<tag>_pmc.json) that bench.py cites for the vector-issue model of kernels whose bound is instruction issue.

  python tools/summarize_pmc.py gpurun_out/r03_pmc_reset --tag r03_reset --kernels k_screen2_rows k_screen2_cols k_pack_tiles k_screen_means

HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are KiB, collected in separate passes; on gfx950
FETCH_SIZE reports half the bytes of a wide coalesced streaming read, so it is doubled (stated per kernel; narrow loads are uncalibrated).
Issue cycles per launch (the model bench.py uses): 8 per transcendental and per matrix instruction, 4 per other vector instruction."""
import argparse, collections, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("dir"); ap.add_argument("--tag", required=True); ap.add_argument("--kernels", nargs="+", required=True)
ap.add_argument("--out", default="profiles")
a = ap.parse_args()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in a.kernels:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
js = {}
print(f"## PMC per kernel ({a.tag}; separate `--pmc` passes with `--kernel-trace` only; mean per dispatch)\n")
for k in a.kernels:
    v = agg.get(k)
    if not v:
        continue
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    print(f"### `{k}`\n\n| counter | mean per dispatch | dispatches |\n|---|---|---|")
    for c in sorted(mean):
        print(f"| {c} | {mean[c]:.6g} | {len(v[c])} |")
    print()
    e = {"dispatches": max(len(x) for x in v.values())}
    if "SQ_INSTS_VALU" in mean:
        valu, trans, mfma = mean["SQ_INSTS_VALU"], mean.get("SQ_INSTS_VALU_TRANS_F32", 0.0), mean.get("SQ_INSTS_MFMA", 0.0)
        e.update(valu_insts=valu, trans_insts=trans, mfma_insts=mfma, issue_cycles=8 * trans + 8 * mfma + 4 * (valu - trans - mfma))
        print(f"issue cycles per launch = 8 x {trans:.4g} (transcendental) + 8 x {mfma:.4g} (matrix) + 4 x {valu - trans - mfma:.4g} (other vector) = "
              f"**{e['issue_cycles']:.4g}** SIMD-cycles\n")
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        e["hbm_bytes"] = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        print(f"HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = **{e['hbm_bytes'] / 1e6:.1f} MB** (FETCH_SIZE {mean['FETCH_SIZE']:.0f} KiB "
              f"doubled per the gfx950 correction, WRITE_SIZE {mean['WRITE_SIZE']:.0f} KiB)\n")
    if "SQ_WAVE_CYCLES" in mean:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in mean:
                print(f"- {c} / SQ_WAVE_CYCLES = {mean[c] / mean['SQ_WAVE_CYCLES']:.3f}")
        print()
    js[k] = e
os.makedirs(a.out, exist_ok=True)
json.dump(js, open(os.path.join(a.out, a.tag + "_pmc.json"), "w"), indent=1)
