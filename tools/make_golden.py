"""Generate tests/golden/*.npz from the CPU oracle.

SELF-GENERATED, NOT HCIPy-DERIVED: hcipy cannot be imported in the build image and the reference holds no
fixtures, so these vectors pin the HIP path (and future oracle edits) to the oracle's current restatement only.
Run:  python tools/make_golden.py
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


def main():
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
