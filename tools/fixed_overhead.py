import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
def run(n, marks=None):
    t = 0; env.reset(); gather.start_episode()
    for i in range(n):
        _, rew, _, _, _ = env.step(actions[t]); gather.add(rew); t += 1
        if marks is not None and i in marks:
            torch.cuda.synchronize(); marks[i] = time.perf_counter()
        if t == T:
            gather.finish_episode(); env.reset(); gather.start_episode(); t = 0
for mode in ("profile_on_before_timed", "profile_off", "profile_on_during_warmup"):
    if mode == "profile_on_during_warmup": env.profile(True)
    run(30); torch.cuda.synchronize()
    if mode == "profile_on_before_timed": env.profile(True)
    marks = {9: 0, 29: 0, 59: 0, 119: 0, 299: 0}
    t0 = time.perf_counter(); run(300, marks); torch.cuda.synchronize(); t1 = time.perf_counter()
    ks = sorted(marks); prev_t, prev_k = t0, -1; seg = []
    for k in ks:
        seg.append(f"steps {prev_k+1}-{k}: {(marks[k]-prev_t)/(k-prev_k)*1e6:.1f} us/step"); prev_t, prev_k = marks[k], k
    print(mode, f"total {(t1-t0)*1e3:.2f} ms |", " | ".join(seg), flush=True)
    env.profile_read(); env.profile(False)
