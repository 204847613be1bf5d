import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv
def run(N, q, full, B=3):
    if full: os.environ["AOG_SCREENS_FULLFFT"] = "1"
    else: os.environ.pop("AOG_SCREENS_FULLFFT", None)
    env = BatchedAOEnv(B, "cuda:0", atm_type="semi_dynamic", atm_fried=0.15, num_pupil_pixels=N, act_dim=6, act_type="zernike", obs_dim=2,
                       timesteps_per_episode=5, seed=7, screen_oversampling=q, verbose=False)
    env.reset()
    ps = np.stack([env.phase_screen(i).cpu().numpy() for i in range(B)])
    env.close()
    return ps
for N, q in ((60, 8), (120, 4), (240, 16), (480, 2), (256, 16), (64, 4)):
    a = run(N, q, True); b = run(N, q, False)
    print(N, q, "rms full %.4e  max|diff|/rms %.3e" % (a.std(), np.abs(a - b).max() / a.std()), flush=True)
