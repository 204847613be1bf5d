"""Per-kernel mean durations from a rocprofv3 kernel trace, the int8 extrusion kernels split by phase (alternate launches)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = collections.defaultdict(list)
count = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:40]
    if "x8_pr" in n:
        count[n] += 1
        n += " phase %d" % ((count[n] - 1) % 2)
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = v[skip:] if len(v) > 2 * skip else v
    if len(v) > 5:
        print("%-44s n %5d  mean %8.1f us  min %8.1f  max %8.1f" % (n, len(v), sum(v) / len(v), min(v), max(v)))
