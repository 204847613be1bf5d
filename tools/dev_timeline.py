"""AOG_DEV build only: per-wave timeline of one fused-kernel launch (config-2 shape).  Run with AOG_DEV_TIMELINE=1."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv, _lib
B, N, A, o = 1024, 256, 64, int(os.environ.get("O", 2))
dev = torch.device("cuda:0")
g = torch.Generator(dev).manual_seed(1)
scr = torch.nn.functional.interpolate(torch.randn(B, 1, 16, 16, device=dev, generator=g), size=(N, N), mode="bicubic").squeeze(1) * 2e-6
env = BatchedAOEnv(B, dev, num_pupil_pixels=N, act_dim=A, obs_dim=o, timesteps_per_episode=10**6, screens=scr, verbose=False)
a = torch.randn(B, A, device=dev, generator=g) * 0.7071
env.reset()
for _ in range(300): env.step(a)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib = _lib.load()
nw = 8192
buf = np.zeros((nw, 8), dtype=np.int64)
for rep in range(3):
    env.step(a)
    torch.cuda.synchronize()
    lib.aog_dev_read_timeline.argtypes = [C.c_void_p, C.c_size_t]
    assert lib.aog_dev_read_timeline(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    r = buf[buf[:, 5] > 0].astype(np.float64) * 0.01   # us
    t0 = r[:, 0].min()
    q = lambda x: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
    print(f"waves {len(r)}  tiles/wave {buf[buf[:,5]>0][:,5].min()}..{buf[:,5].max()}")
    print("  entry (after first wave)      ", q(r[:, 0] - t0))
    print("  setup: entry -> loop start    ", q(r[:, 1] - r[:, 0]))
    print("  first stage                   ", q(r[:, 2] - r[:, 1]))
    print("  remaining stages              ", q(r[:, 3] - r[:, 2]))
    print("  per stage (remaining)         ", q((r[:, 3] - r[:, 2]) / np.maximum(buf[buf[:, 5] > 0][:, 5] - 1, 1)))
    print("  stores                        ", q(r[:, 4] - r[:, 3]))
    print("  wave life                     ", q(r[:, 4] - r[:, 0]))
    print("  exit (after first wave entry) ", q(r[:, 4] - t0))
# groupings of the last launch (pair mapping: see fused_wg_map)
idx = np.flatnonzero(buf[:, 5] > 0)
W = int(os.environ.get("WAVES", 8))
wg = idx // W
xcd, j = wg & 7, wg >> 3
half, cu = ((idx % W) >> 2) if W == 8 else ((j >> 5) & 1), j & 31
def grp(name, key):
    print("by", name)
    for k in np.unique(key):
        m = key == k
        print(f"   {k:3d}: n {m.sum():4d}  setup {np.mean(r[m,1]-r[m,0]):5.1f}  stage {np.mean((r[m,3]-r[m,2])/np.maximum(buf[idx[m],5]-1,1)):5.2f}  exit mean {np.mean(r[m,4]-t0):5.1f} max {np.max(r[m,4]-t0):5.1f}")
grp("xcd", xcd); grp("half (light sub-chunk / second workgroup of the CU)", half); grp("wave in workgroup", idx % W)
