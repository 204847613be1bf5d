"""First-light probe for the GPU box: parity of every kernel variant vs the oracle + raw timings."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from adaptive_optics_gym_amd import BatchedAOEnv
from helpers import smooth_screens, actions_for, run_oracle

def parity(N, B, A, o, act_type, rew, T=2):
    scr = smooth_screens(B, N, 3)
    acts = np.stack([actions_for(B, A, s) for s in range(T)])
    kw = dict(act_type=act_type, act_dim=A, obs_dim=o, rew_type=rew, timesteps_per_episode=5)
    ref = run_oracle(scr, acts, **kw)
    for prec, kern, sc in [("fp64","auto","poly"),("fast","valu","poly"),("fast","valu","hw"),("fast","mfma","poly"),("fast","mfma","hw")]:
        os.environ["AOG_SINCOS"] = sc
        try:
            env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, precision=prec, kernel=kern, verbose=False, **kw)
            env.reset()
            e0 = np.max(np.abs(env.last_obs_raw.cpu().numpy()/ref["obs0"]-1))
            errs = dict(obs0=e0, obs=0, strehl=0, power=0, reward=0)
            for t in range(T):
                obs, r, d, tr, info = env.step(torch.from_numpy(acts[t]).cuda())
                errs["obs"] = max(errs["obs"], np.max(np.abs(info["obs_raw"].cpu().numpy()/ref["obs_raw"][t]-1)))
                errs["power"] = max(errs["power"], np.max(np.abs(info["power"].cpu().numpy()/ref["power"][t]-1)))
                if rew == "strehl_ratio":
                    errs["strehl"] = max(errs["strehl"], np.max(np.abs(info["strehl"].cpu().numpy()/ref["strehl"][t]-1)))
                errs["reward"] = max(errs["reward"], np.max(np.abs(r.cpu().numpy()-ref["reward"][t])))
            print(f"N={N} B={B} A={A} o={o} {act_type} {rew} | {prec:4s} {kern:4s} {sc:4s} | " + " ".join(f"{k}={v:.2e}" for k,v in errs.items()), flush=True)
            env.close()
        except Exception as ex:
            print(f"N={N} {prec} {kern} {sc} FAILED: {type(ex).__name__}: {ex}", flush=True)

def timing(B, N, A, o, kern, sc, steps=20, chunks=0):
    os.environ["AOG_SINCOS"] = sc
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, act_dim=A, obs_dim=o, atm_fried=0.2, timesteps_per_episode=30,
                       kernel=kern, screen_oversampling=2, pixel_chunks=chunks, verbose=False)
    a = torch.randn(B, A, device="cuda") * 0.7071
    env.reset()
    for _ in range(3): env.step(a)
    torch.cuda.synchronize()
    env.profile(True)
    t0 = time.time()
    for _ in range(steps): env.step(a)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ms, n = env.profile_read()
    print(f"timing B={B} N={N} A={A} o={o} {kern} {sc} chunks={env.info.pixel_chunks}: wall {B*steps/dt/1e6:.3f} M steps/s, fused kernel {ms*1e3:.1f} us x{n} -> {B/ms/1e3:.3f} M steps/s kernel-only", flush=True)
    env.close()

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    parity(64, 5, 64, 2, "num_actuators", "strehl_ratio")
    parity(64, 3, 6, 5, "zernike", "smf_ssim")
    parity(128, 2, 6, 2, "zernike", "strehl_ratio")
    for kern in ("valu", "mfma"):
        for sc in ("poly", "hw"):
            timing(1024, 256, 64, 2, kern, sc)
    timing(4096, 256, 64, 5, "mfma", "poly", steps=5)
    timing(4096, 256, 64, 5, "valu", "poly", steps=5)
    timing(1, 240, 64, 2, "mfma", "poly", steps=50)
