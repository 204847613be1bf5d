"""Host-side cost per call (the GPU queue is drained first, so this is pure CPU submit time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_optics_gym_amd import BatchedAOEnv
from adaptive_optics_gym_amd.rollout import DeviceActor, make_actor
dev = torch.device("cuda:0")
B, N, A = 64, 64, 64     # tiny GPU work: the CPU is the bottleneck by construction
env = BatchedAOEnv(B, dev, num_pupil_pixels=N, act_dim=A, obs_dim=2, timesteps_per_episode=10 ** 6, screens=torch.zeros(B, N, N, device=dev), verbose=False)
actor = make_actor(4, A, 150, device=dev)
da = DeviceActor(actor)
obs, _ = env.reset()
a = torch.zeros(B, A, device=dev)
bufs = (torch.empty(B, A, device=dev), torch.empty(B, device=dev), torch.empty(B, A, device=dev))
so = (torch.empty(B, 4, dtype=torch.float16, device=dev), torch.empty(B, device=dev), torch.empty(B, dtype=torch.bool, device=dev))
def bench(name, f, n=3000):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: submit {1e6*(t1-t0)/n:.1f} us/call, incl. drain {1e6*(t2-t0)/n:.1f} us/call", flush=True)
bench("env.step", lambda: env.step(a))
bench("env.step(out=)", lambda: env.step(a, out=so))
bench("env.step(next_actions=)", lambda: env.step(a, next_actions=a))
env.step(a, next_actions=None)
env.persistent_outputs(True)
bench("env.step, persistent outputs", lambda: env.step(a))
bench("env.step(next_actions=), persistent outputs", lambda: env.step(a, next_actions=a))
env.step(a, next_actions=None)
env.persistent_outputs(False)
bench("DeviceActor", lambda: da(obs, 0.5, out=bufs))
bench("torch.empty x3", lambda: (torch.empty(B, 4, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)))
big = torch.empty(300, B, 4, device=dev)
bench("slice x3", lambda: (big[5], big[6], big[7]))
