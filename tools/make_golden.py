"""Generate tests/golden/*.npz.

    python tools/make_golden.py                    # from the CPU oracle (what is committed today)
    python tools/make_golden.py --from-reference   # from the REFERENCE ITSELF: tests/golden/ref_*.npz (needs hcipy; see below)

Default mode — SELF-GENERATED, NOT HCIPy-DERIVED: hcipy cannot be imported in the build image and the reference holds no
fixtures, so these vectors pin the HIP path (and future oracle edits) to the oracle's current restatement only.

``--from-reference`` is the pin that is one command away: in a build container where ``hcipy==0.5.1`` (requirements.txt:1),
``gymnasium`` and ``scikit-image`` ARE importable it runs the reference's own ``AOEnv`` (imported from /root/reference, never copied,
never shipped to the GPU box) on four workloads at the reference's 240-pixel pupil with ``np.random.seed(s)`` as SURVEY.md section 5
prescribes, and stores inputs (the layer's achromatic screen, the actions) and outputs (raw and float16 observations, reward, done,
fiber power).  ``tests/test_golden_oracle.py`` replays such files through the oracle at north_star's 1e-5 and
``tests/test_gpu_parity.py`` through the HIP path; both skip with a stated reason while no ``ref_*.npz`` exists.  Today the imports
fail with ModuleNotFoundError (an absent package, not a refusal), the script says so and exits 2, and parity stays "unpinned".
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import actions_for, run_oracle, smooth_screens  # noqa: E402

CASES = {
    "disk16_o2_strehl": dict(N=32, B=3, T=5, kw=dict(act_type="num_actuators", act_dim=16, obs_dim=2, rew_type="strehl_ratio",
                                                     timesteps_per_episode=3)),
    "zern6_o5_ssim_thr": dict(N=32, B=2, T=4, kw=dict(act_type="zernike", act_dim=6, obs_dim=5, rew_type="smf_ssim",
                                                      rew_threshold=0.05, timesteps_per_episode=4)),
    "disk64_o2_keepdm": dict(N=48, B=2, T=4, kw=dict(act_type="num_actuators", act_dim=64, obs_dim=2, rew_type="strehl_ratio",
                                                     timesteps_per_episode=2, flat_mirror_start_per_episode=False)),
    "zern20_o3_strehl": dict(N=40, B=2, T=3, kw=dict(act_type="zernike", act_dim=20, obs_dim=3, rew_type="strehl_ratio",
                                                     timesteps_per_episode=10)),
}


REF_CASES = {   # the reference hard-codes a 240-pixel pupil (AO_env.py:216); kwargs are AOEnv.__init__'s (AO_env.py:17-29)
    "ref_disk64_o2_strehl": dict(seed=11, T=6, kw=dict(atm_type="quasi_static", act_type="num_actuators", act_dim=64, obs_dim=2,
                                                       rew_type="strehl_ratio", timesteps_per_episode=3)),
    "ref_zern6_o5_ssim_thr": dict(seed=12, T=4, kw=dict(atm_type="quasi_static", act_type="zernike", act_dim=6, obs_dim=5, rew_type="smf_ssim",
                                                        rew_threshold=0.05, timesteps_per_episode=4)),
    "ref_disk16_o2_keepdm": dict(seed=13, T=4, kw=dict(atm_type="quasi_static", act_type="num_actuators", act_dim=16, obs_dim=2,
                                                       rew_type="strehl_ratio", timesteps_per_episode=2, flat_mirror_start_per_episode=False)),
    "ref_zern20_o3_strehl": dict(seed=14, T=3, kw=dict(atm_type="quasi_static", act_type="zernike", act_dim=20, obs_dim=3,
                                                       rew_type="strehl_ratio", timesteps_per_episode=10, atm_fried=0.25)),
}


def from_reference():
    """Run the reference's AOEnv (needs its third-party stack) and write tests/golden/ref_*.npz.  Returns a process exit code."""
    missing = []
    for mod in ("hcipy", "gymnasium", "skimage"):
        try:
            __import__(mod)
        except Exception as exc:   # ModuleNotFoundError in this image
            missing.append(f"{mod}: {type(exc).__name__}: {exc}")
    if missing:
        print("make_golden --from-reference: the reference's dependencies are not importable here, nothing written:\n  " + "\n  ".join(missing))
        return 2
    ref_root = os.environ.get("AOG_REFERENCE_ROOT", "/root/reference")
    sys.path.insert(0, ref_root)
    from gym_AO.envs.AO_env import AOEnv   # the reference's own class, imported where it lies

    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, c in REF_CASES.items():
        np.random.seed(c["seed"])                     # the reference draws everything from numpy's global stream
        env = AOEnv(**c["kw"])
        screen = np.array(env.layer._achromatic_screen, dtype=np.float64)     # hcipy InfiniteAtmosphericLayer state (phase * lambda)
        acts = np.stack([actions_for(1, c["kw"]["act_dim"], 100 * c["seed"] + t)[0] for t in range(c["T"])])
        obs0, _ = env.reset()
        rec = dict(obs0_raw=np.array(env.wf_wfs_after_foc_subsample.power, dtype=np.float64), obs0=np.array(obs0))
        keys = ("obs_raw", "obs", "reward", "done", "power")
        seq = {k: [] for k in keys}
        for t in range(c["T"]):
            o, r, d, _, info = env.step(acts[t])
            seq["obs_raw"].append(np.array(env.wf_wfs_after_foc_subsample.power, dtype=np.float64))
            seq["obs"].append(np.array(o))
            seq["reward"].append(float(r))
            seq["done"].append(bool(d))
            seq["power"].append(float(info["power"]))
            if d:
                env.reset()
        import hcipy

        np.savez_compressed(os.path.join(out_dir, name + ".npz"), screen=screen, actions=acts, kw=np.array(repr(c["kw"])),
                            source=np.array(f"hcipy {getattr(hcipy, '__version__', '?')} through the reference's AOEnv, np.random.seed({c['seed']})"),
                            **rec, **{"exp_" + k: np.array(v) for k, v in seq.items()})
        print(name, "written (hcipy-derived)")
    return 0


def main():
    if "--from-reference" in sys.argv:
        raise SystemExit(from_reference())
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, c in CASES.items():
        scr = smooth_screens(c["B"], c["N"], seed=hash(name) % 1000 if False else sum(map(ord, name)))
        acts = np.stack([actions_for(c["B"], c["kw"]["act_dim"], s) for s in range(c["T"])])
        ref = run_oracle(scr, acts, **c["kw"])
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), screens=scr, actions=acts,
                            kw=np.array(repr(c["kw"])), **{"exp_" + k: v for k, v in ref.items()})
        print(name, {k: v.shape for k, v in ref.items()})


if __name__ == "__main__":
    main()
