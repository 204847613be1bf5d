"""Worst error of the batched K4 focal fields against the oracle's propagator_fiber, in units of the test tolerance
(tests/test_gpu_parity.py::_assert_power_image_close), per pupil size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv
from oracle.ao_env_oracle import AOEnvOracle
from helpers import smooth_screens, actions_for

for N, B, A, act_type in [(64, 37, 16, "num_actuators"), (128, 5, 6, "zernike"), (240, 3, 64, "num_actuators"), (256, 70, 64, "num_actuators")]:
    scr = smooth_screens(B, N, 80 + N)
    a = actions_for(B, A, 5)
    kw = dict(act_type=act_type, act_dim=A, obs_dim=2, timesteps_per_episode=5)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, verbose=False, **kw)
    env.reset()
    env.step(torch.from_numpy(a).cuda())
    F_all = env.focal_images()
    area = env.tables.focal_pixel_area
    worst = 0.0
    for b in sorted({0, B // 2, B - 1}):
        ref = AOEnvOracle(num_pupil_pixels=N, screen=scr[b].ravel(), verbose=False, **kw)
        ref.reset(); ref.step(a[b])
        P = np.abs(F_all[b].cpu().numpy().astype(np.complex128)) ** 2 * area
        R = ref.wf_wfs_after_foc.power.reshape(128, 128)
        tol = 1e-5 * np.maximum(R, 1e-3 * R.max())
        worst = max(worst, float(np.max(np.abs(P - R) / tol)))
    print(f"N={N}: worst error {worst:.2f} x tolerance", flush=True)
    env.close()
