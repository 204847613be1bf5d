"""Latency of the single-env drop-in (what the reference's main.py loop sees): us per AOEnv.step at the reference's N = 240."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adaptive_optics_gym_amd.envs import AOEnv

for kw in (dict(act_type="num_actuators", act_dim=64, obs_dim=2), dict(act_type="zernike", act_dim=6, obs_dim=5, rew_type="smf_ssim")):
    env = AOEnv(num_pupil_pixels=240, timesteps_per_episode=30, screens=np.random.RandomState(0).randn(1, 240, 240) * 1e-6,
                rng=np.random.RandomState(1), verbose=False, **kw)
    env.reset()
    a = np.random.RandomState(2).randn(kw["act_dim"]).astype(np.float32)
    for _ in range(200):
        env.step(a)
    n, t0 = 2000, time.perf_counter()
    for i in range(n):
        _, _, d, _, _ = env.step(a)
        if d:
            env.reset()
    dt = time.perf_counter() - t0
    print(f"AOEnv.step N=240 {kw}: {dt / n * 1e6:.1f} us per step incl. a reset every 30 steps ({n / dt:.0f} steps/s)", flush=True)
    env.close()
