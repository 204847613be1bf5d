"""SH_step + step loop timing (device photon noise): B, N from argv."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
B, N, A = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "single"
env = BatchedAOEnv(B, "cuda:0", act_type="zernike", act_dim=A, obs_dim=5, rew_type="smf_ssim", timesteps_per_episode=10**6, num_pupil_pixels=N,
                   SH_operation=True, seed=3, screen_oversampling=4, sh_fft_precision=prec, verbose=False)
env.reset()
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.5:   # the device needs a few hundred ms of load to reach its clocks (some processes measured 5 ms per iteration right after start)
    a, _ = env.SH_step(); env.step(a)
    torch.cuda.synchronize()
torch.cuda.synchronize()
n, t0 = 20, time.perf_counter()
for _ in range(n):
    a, _ = env.SH_step(); env.step(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B} N={N} A={A} {prec}: {dt*1e3:.2f} ms per SH_step+step  {B/dt/1e3:.1f} k env-steps/s")
