"""Where the time of bench.py's timed region goes beyond the three per-step kernels (developer tool): per-segment wall time of the
same loop with (a) event timing of every launch, (b) none, (c) one block of 32 in 8, each with the episode returns summed by torch
(`returns += reward`, one more launch per step) or inside the epilogue (`EpisodeReturnGatherer.attach`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from adaptive_optics_gym_amd import BatchedAOEnv
from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch
from adaptive_optics_gym_amd.params import OpticalParams
from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer
w = bench.WORKLOAD; B = w["batch_per_gpu"]; device = torch.device("cuda", 0)
p = OpticalParams(num_pupil_pixels=w["n_pupil"])
gen = torch.Generator(device).manual_seed(1234)
screens = screens_torch(B, p.num_pupil_pixels, p.pupil_pixel, cn_squared_from_fried_parameter(w["atm_fried"], p.wavelength_sci), p.outer_scale, device, gen, oversampling=16)
env = BatchedAOEnv(B, device, atm_type=w["atm_type"], atm_fried=w["atm_fried"], act_type=w["act_type"], act_dim=w["act_dim"], obs_dim=w["obs_dim"], rew_type=w["rew_type"],
                   timesteps_per_episode=w["timesteps_per_episode"], num_pupil_pixels=w["n_pupil"], screens=screens, verbose=False)
T = w["timesteps_per_episode"]
actions = torch.randn((T, B, w["act_dim"]), device=device) * 0.7071
gather = EpisodeReturnGatherer(B, device, False)
def run(n):
    t = 0; env.reset(); gather.start_episode()
    for i in range(n):
        _, rew, _, _, _ = env.step(actions[t]); gather.add(rew); t += 1
        if t == T:
            gather.finish_episode(); env.reset(); gather.start_episode(); t = 0
for attach in (False, True, False, True):
    gather.attach(env) if attach else gather.detach()
    for every in (1, 0, 8):
        run(30); torch.cuda.synchronize()
        if every: env.profile(True, every=every)
        res = []
        for n in (300, 1500):
            t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / n * 1e6)
        if every: env.profile_read(); env.profile(False)
        print(f"returns in {'epilogue' if attach else 'torch   '} | events every {every}: {res[0]:.1f} us/step over 300, {res[1]:.1f} over 1500", flush=True)
