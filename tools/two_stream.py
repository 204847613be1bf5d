"""Experiment: one batch of 1024 envs vs two half batches stepped on two streams (start-up / drain / small kernels of one half
overlapping the other half's fused kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
dev = torch.device("cuda:0")
N, A, B = 256, 64, 1024
g = torch.Generator(dev).manual_seed(1)
scr = torch.nn.functional.interpolate(torch.randn(B, 1, 16, 16, device=dev, generator=g), size=(N, N), mode="bicubic").squeeze(1) * 2e-6
a = torch.randn(B, A, device=dev, generator=g) * 0.7071
def make(lo, hi): return BatchedAOEnv(hi - lo, dev, num_pupil_pixels=N, act_dim=A, obs_dim=2, act_type="num_actuators", timesteps_per_episode=10 ** 6, screens=scr[lo:hi], verbose=False)
def timeit(fn, n=400):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
one = make(0, B); one.reset()
t1 = timeit(lambda: one.step(a))
print(f"one batch of {B}: {t1*1e6:.1f} us/step -> {B/t1/1e6:.2f} M env-steps/s", flush=True)
one.close()
for parts in (2, 4):
    envs = [make(i * B // parts, (i + 1) * B // parts) for i in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    acts = [a[i * B // parts:(i + 1) * B // parts].contiguous() for i in range(parts)]
    for e in envs: e.reset()
    torch.cuda.synchronize()
    def step_all():
        for e, s, x in zip(envs, streams, acts):
            with torch.cuda.stream(s):
                e.step(x)
    t2 = timeit(step_all)
    print(f"{parts} x {B//parts} on {parts} streams: {t2*1e6:.1f} us/step -> {B/t2/1e6:.2f} M env-steps/s", flush=True)
    for e in envs: e.close()
