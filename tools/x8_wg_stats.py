"""Per-workgroup records of k_x8_product (AOG_DEV build, AOG_X8_DEV=1024 AOG_X8_DEV_DUMP=file): steps, class, phase, XCC, hw id, cycles per
double step, start and loop-end ticks (10 ns)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=int)
steps, k, ph, xcc, hw, cyc, t0, t1 = d.T
for p in (0, 1):
    s = d[ph == p]
    if not len(s):
        continue
    base = s[:, 6].min()
    dur = s[:, 7] - s[:, 6]
    print(f"phase {p}: {len(s)} workgroups, starts {s[:,6].min()-base}..{s[:,6].max()-base}, last loop end {s[:,7].max()-base} ticks; "
          f"cycles per double step min {s[:,5].min()} median {int(np.median(s[:,5]))} max {s[:,5].max()}; clock {np.median(s[:,5] * ((s[:,0]+3)//4*2) / (dur*10.0)):.2f} GHz")
    for x in range(8):
        q = s[s[:, 3] == x]
        if len(q):
            print(f"   xcc {x}: n {len(q):3d} cyc mean {q[:,5].mean():6.0f} max {q[:,5].max():5d}  steps sum {q[:,0].sum():5d}  late starts {np.sum(q[:,6] > base + 300):3d}  loop end max {q[:,7].max()-base}")
    o = np.argsort(-s[:, 5])[:8]
    print("   slowest (steps, k, xcc, cyc, start, end):", [(int(a), int(b), int(c), int(e), int(f - base), int(g - base)) for a, b, c, e, f, g in s[o][:, [0, 1, 3, 5, 6, 7]]])
    o = np.argsort(-s[:, 7])[:8]
    print("   last to end (steps, k, xcc, cyc, start, end):", [(int(a), int(b), int(c), int(e), int(f - base), int(g - base)) for a, b, c, e, f, g in s[o][:, [0, 1, 3, 5, 6, 7]]])
