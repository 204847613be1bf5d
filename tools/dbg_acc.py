import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from adaptive_optics_gym_amd import BatchedAOEnv
from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch
def run(N, A, o, r0, act_type, B=48):
    dev = torch.device("cuda:0"); g = torch.Generator(dev).manual_seed(77)
    scr = screens_torch(B, N, 0.5 / N, cn_squared_from_fried_parameter(r0, 2.2e-6), 10.0, dev, g, oversampling=4)
    a = torch.randn((B, A), device=dev, generator=g) * 0.7071
    kw = dict(act_dim=A, obs_dim=o, rew_type="smf_ssim", act_type=act_type, atm_fried=r0, timesteps_per_episode=3, num_pupil_pixels=N, verbose=False)
    ref = BatchedAOEnv(B, dev, screens=scr, precision="fp64", **kw); env = BatchedAOEnv(B, dev, screens=scr, kernel="mfma", **kw)
    ref.reset(); env.reset()
    r = ref.step(a)[4]["obs_raw"].double().cpu().numpy(); o_ = env.step(a)[4]["obs_raw"].double().cpu().numpy()
    peak = r.max(axis=1, keepdims=True); err = np.abs(o_ - r); tol = 1e-5 * np.maximum(np.abs(r), 1e-3 * peak)
    print(f"N={N} A={A} o={o} r0={r0} {act_type} B={B}: chunks={env.info.pixel_chunks} max err/tol {np.max(err/tol):.2f}  median err/peak {np.median(err/peak):.2e}  frac>tol {np.mean(err>tol):.3f}", flush=True)
    ref.close(); env.close()
run(256, 64, 5, 0.15, "num_actuators", 1024)
run(256, 20, 5, 0.15, "zernike", 1024)
run(512, 20, 5, 0.15, "zernike", 512)
run(256, 64, 3, 0.15, "num_actuators", 1024)
run(256, 64, 5, 0.4, "num_actuators", 1024)
