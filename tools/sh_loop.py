"""Timing of the Shack-Hartmann baseline loop (SH_step + step) on the device."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--N", type=int, default=240); ap.add_argument("--A", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(dev).manual_seed(1)
scr = torch.nn.functional.interpolate(torch.randn(args.B, 1, 16, 16, device=dev, generator=g), size=(args.N, args.N), mode="bicubic").squeeze(1) * 1.5e-6
t0 = time.perf_counter()
env = BatchedAOEnv(args.B, dev, num_pupil_pixels=args.N, act_dim=args.A, obs_dim=2, SH_operation=True, timesteps_per_episode=10 ** 6,
                   screens=scr, verbose=False)
print(f"init (host calibration incl.) {time.perf_counter() - t0:.1f} s, n_sub={env.sh.n_sub}", flush=True)
env.reset()
s0 = None
for _ in range(3):
    a, _ = env.SH_step(); info = env.step(a)[4]
    s0 = info["strehl"].mean().item() if s0 is None else s0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    a, _ = env.SH_step(); info = env.step(a)[4]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"SH loop B={args.B} N={args.N} A={args.A}: {args.B * args.steps / dt / 1e3:.1f} k env-steps/s ({dt / args.steps * 1e3:.2f} ms per SH_step+step), "
      f"mean Strehl {s0:.3f} -> {info['strehl'].mean().item():.3f}", flush=True)
