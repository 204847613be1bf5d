"""Top kernels of a rocprofv3 --stats run as a markdown table (for profiles/)."""
import csv, glob, sys
d, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
import os
f = max(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
print(f"### {title}\n\n`{cmd}`\n\n| kernel | calls | avg µs | % of GPU time |\n|---|---|---|---|")
for r in rows[:10]:
    if float(r["Percentage"]) < 0.5: break
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
print()
