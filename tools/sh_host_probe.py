"""SH_step + step loop with HIP-event timing per call (is a slow process slow everywhere?).  Developer probe."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv, _lib
B, N, A = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 64
env = BatchedAOEnv(B, "cuda:0", act_type="zernike", act_dim=A, obs_dim=5, rew_type="smf_ssim", timesteps_per_episode=10**6, num_pupil_pixels=N,
                   SH_operation=True, seed=3, screen_oversampling=4, verbose=False)
env.reset()
for _ in range(3):
    a, _ = env.SH_step(); env.step(a)
torch.cuda.synchronize()
action = torch.empty((B, A), dtype=torch.float64, device="cuda")
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(20)]
t0 = time.perf_counter()
for it in range(20):
    ev[it][0].record()
    _lib.check(env.lib.aog_sh_image(env._handle, None, env._stream()))
    ev[it][1].record()
    _lib.check(env.lib.aog_sh_update(env._handle, None, C.c_void_p(action.data_ptr()), env._stream()))
    ev[it][2].record()
    env.step(action)
    ev[it][3].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
seg = np.array([[ev[i][k].elapsed_time(ev[i][k + 1]) for k in range(3)] for i in range(20)])
gap = np.array([ev[i][3].elapsed_time(ev[i + 1][0]) for i in range(19)])
print(f"   gap between iterations: median {np.median(gap):.3f} max {gap.max():.3f} ms")
for name, fn in (("sh_image only", lambda: _lib.check(env.lib.aog_sh_image(env._handle, None, env._stream()))),
                 ("step only", lambda: env.step(action)),
                 ("sh_image+sh_update", lambda: (_lib.check(env.lib.aog_sh_image(env._handle, None, env._stream())), _lib.check(env.lib.aog_sh_update(env._handle, None, C.c_void_p(action.data_ptr()), env._stream()))))):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    print(f"   {name}: {(time.perf_counter() - t1) / 20 * 1e3:.2f} ms/iter")
print(f"N={N}: wall {wall:.2f} ms/iter | sh_image median {np.median(seg[:,0]):.2f} (min {seg[:,0].min():.2f} max {seg[:,0].max():.2f}) | sh_update {np.median(seg[:,1]):.3f} | step {np.median(seg[:,2]):.3f}")
