"""Time semi_dynamic resets (device screen synthesis, K8) and a C3-shaped episode loop."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=256); ap.add_argument("--A", type=int, default=64)
ap.add_argument("--o", type=int, default=5); ap.add_argument("--T", type=int, default=20); ap.add_argument("--episodes", type=int, default=3)
ap.add_argument("--oversampling", type=int, default=16); ap.add_argument("--method", default="twoband")
args = ap.parse_args()
dev = torch.device("cuda:0")
env = BatchedAOEnv(args.B, dev, atm_type="semi_dynamic", atm_fried=0.15, num_pupil_pixels=args.N, act_dim=args.A, obs_dim=args.o,
                   act_type="num_actuators", timesteps_per_episode=args.T, screen_oversampling=args.oversampling, screen_method=args.method,
                   verbose=False)
a = torch.randn(args.B, args.A, device=dev) * 0.7071
env.reset(); torch.cuda.synchronize()
t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
t0 = time.perf_counter()
for ep in range(args.episodes):
    env.reset()
    for _ in range(args.T): env.step(a)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={args.B} N={args.N} o={args.o} oversampling={args.oversampling} {args.method}: reset {t_reset*1e3:.1f} ms ({t_reset/args.B*1e6:.2f} us/env), "
      f"{args.episodes} episodes x {args.T} steps: {args.B*args.T*args.episodes/dt/1e6:.3f} M env-steps/s", flush=True)
