"""Dynamic-atmosphere step loop (config-4 shape, device Philox): us per step and kernel breakdown hints."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedAOEnv(B, "cuda:0", atm_type="dynamic", atm_vel=10, atm_fried=0.15, act_dim=64, obs_dim=2, num_pupil_pixels=256,
                   timesteps_per_episode=10**6, seed=1234, screen_oversampling=4, verbose=False)
a = torch.randn(B, 64, device="cuda")
env.reset()
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.5:   # the device needs a few hundred ms of load to reach its clocks
    env.step(a); torch.cuda.synchronize()
env.device_status()   # (AOG_EXTRUDE_TIMING=1: switches the phase clocks of the extrusion kernel on)
torch.cuda.synchronize()
n, t0 = (200 if len(sys.argv) < 3 else int(sys.argv[2])), time.perf_counter()
for _ in range(n): env.step(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"dynamic B={B} ring_direct={env.info.reserved}: {dt*1e6:.1f} us per step  {B/dt/1e6:.2f} M env-steps/s  status {env.device_status()}")
