"""Does the fused DYN kernel's cost come from the misalignment of its ring loads?  Same dynamic workload with every wind along +y (origins stay
at ox = 0: every 128-byte tile row is line-aligned) against the usual random directions.  Prints us per step; kernel times via rocprofv3."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv, _lib
B = 1024
mode = sys.argv[1] if len(sys.argv) > 1 else "random"
env = BatchedAOEnv(B, "cuda:0", atm_type="dynamic", atm_vel=10, atm_fried=0.15, act_dim=64, obs_dim=2, num_pupil_pixels=256,
                   timesteps_per_episode=10**6, seed=1234, screen_oversampling=4, verbose=False)
if mode != "random":
    v = np.zeros((B, 2)); v[:, 1 if mode == "y" else 0] = 10.0
    env.velocity_vectors = v
    t = torch.from_numpy(v).cuda()
    _lib.check(env.lib.aog_set_wind(env._handle, C.c_void_p(t.data_ptr()), 10.0, env._stream()))
a = torch.randn(B, 64, device="cuda")
env.reset()
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.5:
    env.step(a); torch.cuda.synchronize()
if os.environ.get("AOG_X8_DEV"):
    for _ in range(5):
        env.step(a); env.device_status()
    sys.exit(0)
n, t0 = 200, time.perf_counter()
for _ in range(n): env.step(a)
torch.cuda.synchronize()
print(f"wind {mode}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per step, status {env.device_status()}")
