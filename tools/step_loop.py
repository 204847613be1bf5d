"""Minimal driver for profiling: config-2 shaped batch, a few steps, nothing else on the GPU after setup."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=256)
ap.add_argument("--A", type=int, default=64); ap.add_argument("--o", type=int, default=2)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--kernel", default="auto")
ap.add_argument("--chunks", type=int, default=0); ap.add_argument("--atm", default="quasi_static"); ap.add_argument("--vel", type=float, default=0.0); ap.add_argument("--act_type", default="num_actuators"); ap.add_argument("--graph", action="store_true", help="replay one captured step (torch.cuda.CUDAGraph) instead of calling step()"); ap.add_argument("--wind_dir", type=float, default=None, help="degrees; same for every env (default: random per env)")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(dev).manual_seed(1)
# cheap synthetic screens (smooth random, a few rad rms): the fused kernel's cost does not depend on their spectrum
scr = torch.nn.functional.interpolate(torch.randn(args.B, 1, 16, 16, device=dev, generator=g), size=(args.N, args.N),
                                      mode="bicubic").squeeze(1) * 2e-6
env = BatchedAOEnv(args.B, dev, atm_type=args.atm, atm_vel=args.vel, num_pupil_pixels=args.N, act_dim=args.A, obs_dim=args.o, act_type=args.act_type,
                   timesteps_per_episode=1000000, kernel=args.kernel, pixel_chunks=args.chunks, screens=scr, verbose=False)
if args.wind_dir is not None and args.atm == "dynamic":
    import ctypes as C, numpy as np
    from adaptive_optics_gym_amd import _lib
    th = np.deg2rad(args.wind_dir)
    env.velocity_vectors = float(env.velocity) * np.tile([[np.cos(th), np.sin(th)]], (args.B, 1))
    v = torch.from_numpy(np.ascontiguousarray(env.velocity_vectors)).to(dev)
    _lib.check(env.lib.aog_set_wind(env._handle, C.c_void_p(v.data_ptr()), float(np.abs(env.velocity_vectors).max()), env._stream()))
    torch.cuda.synchronize()
a = torch.randn(args.B, args.A, device=dev, generator=g) * 0.7071
env.reset()
for _ in range(20): env.step(a)
torch.cuda.synchronize()
env.device_status()
env.profile(True, every=8)   # one block of 8 launches in 8: the event records hold the stream ~6 us per timed launch
step = lambda: env.step(a)
if args.graph:
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        env.step(a)
    step = g_.replay
    for _ in range(5): step()
    torch.cuda.synchronize()
blocks = []
t0 = time.perf_counter()
done = 0
while done < args.steps:
    nb = min(50, args.steps - done)
    tb = time.perf_counter()
    for _ in range(nb): step()
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - tb) / nb)
    done += nb
dt = time.perf_counter() - t0
print('blocks (us/step, in order): ' + ' '.join(f'{b*1e6:.0f}' for b in blocks))
blocks.sort()
print(f"per-step wall over blocks of <=50: min {blocks[0]*1e6:.1f} us  median {blocks[len(blocks)//2]*1e6:.1f} us  max {blocks[-1]*1e6:.1f} us")
ms, n = env.profile_read()
env.device_status()
print(f"B={args.B} N={args.N} A={args.A} o={args.o} kernel={env.info.kernel} chunks={env.info.pixel_chunks} "
      f"wall {args.B*args.steps/dt/1e6:.3f} Msteps/s  fused {ms*1e3:.1f} us -> {args.B/ms/1e3:.3f} Msteps/s", flush=True)
