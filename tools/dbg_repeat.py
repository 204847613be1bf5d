"""Run-to-run repeatability and parity of the fast fused kernel at config-2 size (a data race shows up as differing repeats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from adaptive_optics_gym_amd import BatchedAOEnv
from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch
N, B, A = 256, 1024, 64
o = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0"); g = torch.Generator(dev).manual_seed(1234)
scr = screens_torch(B, N, 0.5 / N, cn_squared_from_fried_parameter(0.2, 2.2e-6), 10.0, dev, g, oversampling=4)
kw = dict(act_dim=A, obs_dim=o, atm_fried=0.2, timesteps_per_episode=2, num_pupil_pixels=N, verbose=False, rew_type="strehl_ratio" if o == 2 else "smf_ssim")
ref = BatchedAOEnv(B, dev, screens=scr, precision="fp64", **kw); ref.reset()
r = ref.last_obs_raw.double().cpu().numpy()
env = BatchedAOEnv(B, dev, screens=scr, kernel="mfma", **kw)
peak = r.max(axis=1, keepdims=True); tol = 1e-5 * np.maximum(np.abs(r), 1e-3 * peak)
outs = []
for rep in range(8):
    env.reset(); outs.append(env.last_obs_raw.double().cpu().numpy())
print("o=%d: max err/tol vs float64 kernel %.2f" % (o, np.max(np.abs(outs[0] - r) / tol)))
print("repeat differences:", [float(np.abs(x - outs[0]).max() / peak.max()) for x in outs[1:]])
