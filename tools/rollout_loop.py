"""Throughput of the batched rollout (policy query + env step + bookkeeping), config-2 shaped."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
from adaptive_optics_gym_amd.rollout import make_actor, rollout
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=256); ap.add_argument("--A", type=int, default=64)
ap.add_argument("--o", type=int, default=2); ap.add_argument("--T", type=int, default=30); ap.add_argument("--episodes", type=int, default=10)
ap.add_argument("--hidden", type=int, default=150); ap.add_argument("--actor", default="auto")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(dev).manual_seed(1)
scr = torch.nn.functional.interpolate(torch.randn(args.B, 1, 16, 16, device=dev, generator=g), size=(args.N, args.N), mode="bicubic").squeeze(1) * 2e-6
env = BatchedAOEnv(args.B, dev, num_pupil_pixels=args.N, act_dim=args.A, obs_dim=args.o, act_type="num_actuators", timesteps_per_episode=args.T,
                   screens=scr, verbose=False)
actor = make_actor(args.o ** 2, args.A, args.hidden, device=dev)
kw = {} if args.actor == "auto" else {"actor_impl": args.actor}
rollout(env, actor, 1, **kw); torch.cuda.synchronize()
t0 = time.perf_counter()
out = rollout(env, actor, args.episodes, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"rollout B={args.B} T={args.T} episodes={args.episodes} actor={args.actor}: {args.B*args.T*args.episodes/dt/1e6:.3f} M env-steps/s "
      f"({dt/(args.T*args.episodes)*1e6:.1f} us per step), avg_ep_rew {out['avg_ep_rew']:.3f}", flush=True)
