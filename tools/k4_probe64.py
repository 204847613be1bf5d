"""Structure of the K4 error of a fast-precision handle against the float64 validation handle (same library, same inputs)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv
from helpers import smooth_screens, actions_for
for N, B, A, act_type, do_step in [(64, 37, 16, "num_actuators", True), (64, 37, 16, "num_actuators", False), (64, 37, 6, "zernike", True)]:
    scr = smooth_screens(B, N, 80 + N)
    a = actions_for(B, A, 5)
    kw = dict(act_type=act_type, act_dim=A, obs_dim=2, timesteps_per_episode=5)
    F = {}
    for prec in ("fp64", "fast"):
        env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, verbose=False, precision=prec, **kw)
        env.reset()
        if do_step: env.step(torch.from_numpy(a).cuda())
        F[prec] = np.stack([env.focal_image(b).cpu().numpy().astype(np.complex128) for b in (0, B // 2, B - 1)])
        env.close()
    for i in range(3):
        f0, f1 = F["fp64"][i], F["fast"][i]
        c = np.vdot(f0, f1) / np.vdot(f0, f0)
        res = f1 - c * f0
        P0 = np.abs(f0) ** 2; tol = 1e-5 * np.maximum(P0, 1e-3 * P0.max())
        ratio = np.abs(np.abs(f1) ** 2 - P0) / tol
        ratio_c = np.abs(np.abs(f1 / c) ** 2 - P0) / tol
        apk = np.abs(f0).max()
        print(f"{act_type} step={do_step} N={N} env {i}: c-1 = {c - 1:.3e}, |res| rms/apk = {np.sqrt(np.mean(np.abs(res)**2))/apk:.3e}, max/apk = {np.abs(res).max()/apk:.3e}, "
              f"worst ratio {ratio.max():.3f} (after removing c: {ratio_c.max():.3f}), pixels > 0.5: {(ratio > 0.5).sum()}", flush=True)
