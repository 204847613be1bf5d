"""Timing of the batched K4 focal fields (aog_focal_images) at config-2 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
B, N, A = 1024, 256, 64
env = BatchedAOEnv(B, "cuda:0", act_dim=A, obs_dim=2, num_pupil_pixels=N, seed=3, screen_oversampling=4, verbose=False)
env.reset(); env.step(torch.randn(B, A, device="cuda"))
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.5:   # the device needs a few hundred ms of load to reach its clocks
    F = env.focal_images(); torch.cuda.synchronize()
torch.cuda.synchronize()
n, t0 = 5, time.perf_counter()
for _ in range(n): F = env.focal_images()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = B * 8 * (128 * N * N + 128 * N * 128)
print(f"aog_focal_images B={B} N={N}: {dt*1e3:.2f} ms per call ({dt/B*1e6:.2f} us per env), {flops/dt/1e12:.1f} TFLOP/s of nominal complex-GEMM flops (3 split-f16 matrix instructions per real product)")
