"""Developer aid: per-wave timeline of the fused kernel.  Runs a few steps with the AOG_ABLATE=6 diagnostic build (each wave
records wall_clock64 ticks, 10 ns, into the partial-sum buffer) and summarises when waves start, how long their phases take and
when they end.  Usage: AOG_ABLATE=6 python tools/fused_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv, _lib
assert os.environ.get("AOG_ABLATE") == "6"
B, N, A = 1024, 256, 64
dev = torch.device("cuda:0")
scr = torch.zeros(B, N, N, device=dev)
env = BatchedAOEnv(B, dev, num_pupil_pixels=N, act_dim=A, obs_dim=2, act_type="num_actuators", timesteps_per_episode=1000000, screens=scr, verbose=False)
a = torch.zeros(B, A, device=dev)
env.reset()
for _ in range(5): env.step(a)
n_wg = 512
rec = np.zeros((n_wg * 4, 8), dtype=np.int64)
_lib.check(env.lib.aog_debug_read_partials(env._handle, rec.ctypes.data_as(C.c_void_p), rec.nbytes))
wg = np.arange(n_wg * 4) // 4
keep = rec[:, 5] > 0
rec, wg = rec[keep], wg[keep]
print('waves recorded:', len(rec))
t0 = rec[:, 5].min()
us = lambda x: (x - t0) / 100.0
print("kernel entry (first instruction) spread: %.1f us" % us(rec[:, 5].max()))
print("tables staged + first loads issued (enter): med %.1f max %.1f us" % (np.median(us(rec[:, 0])), us(rec[:, 0].max())))
print("first tile ready: med %.1f max %.1f us" % (np.median(us(rec[:, 6])), us(rec[:, 6].max())))
print("loop: load-issue segments total med %.1f us" % np.median((rec[:, 3] - rec[:, 6] - rec[:, 1] - rec[:, 2]) / 100))
print("loop end: min %.1f med %.1f max %.1f us" % (us(rec[:, 3].min()), np.median(us(rec[:, 3])), us(rec[:, 3].max())))
print("wave end: min %.1f med %.1f max %.1f us" % (us(rec[:, 4].min()), np.median(us(rec[:, 4])), us(rec[:, 4].max())))
print("per wave: slot0 total med %.1f  matrix+rest total med %.1f  first tile (enter->loop) med %.1f us" % (
    np.median(rec[:, 1]) / 100, np.median(rec[:, 2]) / 100, np.median((rec[:, 3] - rec[:, 0] - rec[:, 1] - rec[:, 2]) / 100)))
for x in range(8):
    m = wg % 8 == x
    print("xcd %d: end med %.1f max %.1f" % (x, np.median(us(rec[m, 4])), us(rec[m, 4].max())))
order = np.argsort(rec[:, 4])
print("slowest waves (wg, end us):", [(int(wg[i]), round(float(us(rec[i, 4])), 1)) for i in order[-8:]])
