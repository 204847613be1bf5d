"""Host vs wall time per step of a 20-step burst right after a long asynchronous run (why bench.py fences after its spin-up).  Developer probe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
B = 1024
env = BatchedAOEnv(B, "cuda:0", act_dim=64, obs_dim=2, num_pupil_pixels=256, seed=3, screen_oversampling=4, timesteps_per_episode=10**7, verbose=False)
a = torch.randn(B, 64, device="cuda") * 0.7
env.reset()
prof = len(sys.argv) > 2
if prof: env.profile(True, every=8)
def timed(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): env.step(a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for spin in (300, 3000, 10000):
    for _ in range(spin): env.step(a)
    if prof: env.profile_read()
    h, w = timed(20)
    print(f"after {spin} more steps: host {h:.1f} us/step, wall {w:.1f} us/step (20 steps)", end="")
    h, w = timed(20)
    print(f" | again: host {h:.1f}, wall {w:.1f}")
