"""``AOEnv`` — drop-in for the reference's ``gym_AO.envs.AOEnv`` (``/root/reference/gym_AO/envs/AO_env.py:16``):
same constructor keywords and defaults (AO_env.py:17-29), same ``reset()``/``step()`` tuples (AO_env.py:103,153),
numpy in / numpy out, one environment.  It is the B = 1 view of ``BatchedAOEnv``: all arithmetic runs in the HIP
library; there is no CPU path.

The single-env wrapper draws its phase screen from the process-global legacy ``np.random`` stream in hcipy's
order (wind direction, two stencil draws, 2 x (16 N)^2 normals), so ``np.random.seed(s)`` before construction
plays the same role as in the reference.
"""
from __future__ import annotations

import numpy as np

from ..batched_env import BatchedAOEnv
from ..spaces import env_base


class AOEnv(env_base()):
    metadata = {"render_modes": []}

    def __init__(self, atm_type="quasi_static", atm_vel=0, atm_fried=0.15, act_type="num_actuators", act_dim=64,
                 obs_dim=2, rew_type="strehl_ratio", rew_threshold=None, timesteps_per_episode=20,
                 flat_mirror_start_per_episode=True, SH_operation=False, *, num_pupil_pixels=240, device=None,
                 screens=None, precision="fast", kernel="auto", rng=np.random, verbose=True):
        super().__init__()
        self._env = BatchedAOEnv(1, device, atm_type, atm_vel, atm_fried, act_type, act_dim, obs_dim, rew_type,
                                 rew_threshold, timesteps_per_episode, flat_mirror_start_per_episode, SH_operation,
                                 num_pupil_pixels=num_pupil_pixels, screen_source="numpy", screens=screens,
                                 precision=precision, kernel=kernel, rng=rng, verbose=verbose)
        e = self._env
        self.atm_type, self.rew_type, self.act_type = e.atm_type, e.rew_type, e.act_type
        self.flat_mirror_start_per_episode = e.flat_mirror_start_per_episode
        self.rew_threshold, self.SH_operation = e.rew_threshold, e.SH_operation
        self.observation_space, self.action_space = e.observation_space, e.action_space
        self.num_modes, self.max_steps = e.num_modes, e.max_steps
        self.num_focal_pixels_fiber_subsample = e.obs_dim
        self.wavelength_wfs, self.wavelength_sci, self.delta_t = e.wavelength_wfs, e.wavelength_sci, e.delta_t
        self.velocity, self.fried_parameter = e.velocity, e.fried_parameter
        self.timestep = 0          # AO_env.py:70 — monotone across episodes
        self.episode_no = 0        # AO_env.py:71
        self.timestep_render = 0
        # one pinned staging buffer each way: a step is ONE host-to-device copy (the action), the library's launches, ONE
        # device-to-host copy of the packed outputs and ONE stream synchronisation
        import torch

        self._torch = torch
        lay = e.persistent_outputs(True)
        self._nb32, self._nb16 = lay["float32_bytes"], lay["float16_bytes"]
        self._host_out = torch.empty((self._nb32 + self._nb16 + lay["uint8_bytes"],), dtype=torch.uint8).pin_memory()
        self._host_act = torch.empty((1, e.num_modes), dtype=torch.float32).pin_memory()
        self._dev_act = torch.empty((1, e.num_modes), dtype=torch.float32, device=e.device)

    def reset(self, seed=None, options=None):
        obs, info = self._env.reset()
        self.timestep_render = 0
        self.last_obs_raw = self._env.last_obs_raw[0].cpu().numpy()
        return obs[0].cpu().numpy(), {}

    def step(self, action):
        torch = self._torch
        e = self._env
        self._host_act.numpy()[...] = np.asarray(action, dtype=np.float32).reshape(1, self.num_modes)
        self._dev_act.copy_(self._host_act, non_blocking=True)
        e.step(self._dev_act)
        self._host_out.copy_(e._pack, non_blocking=True)
        torch.cuda.current_stream(e.device).synchronize()            # the step's only synchronisation
        self.timestep += 1
        self.timestep_render += 1
        n = e.obs_dim ** 2
        host = self._host_out.numpy()
        f32 = host[:self._nb32].view(np.float32)                       # obs_raw [n] | reward | power | strehl
        obs = host[self._nb32:self._nb32 + self._nb16].view(np.float16).copy()
        d = bool(host[self._nb32 + self._nb16])
        if d:
            self.episode_no += 1
        self.last_obs_raw = f32[:n].copy()
        self.last_strehl = float(f32[n + 2])
        return obs, float(f32[n]), d, False, {"power": float(f32[n + 1])}

    def render_data(self):
        """The three images the reference's render() draws (AO_env.py:156-194), as numpy arrays: atmospheric phase screen OPD in
        micrometres [N, N], sensing-arm focal-plane power [128, 128], photodetector (observation) power [o, o]."""
        e = self._env
        opd = e.phase_screen(0).cpu().numpy().astype(np.float64) * (e.wavelength_wfs / (2 * np.pi)) * 1e6
        focal = (e.focal_image(0).abs() ** 2).cpu().numpy().astype(np.float64) * e.tables.focal_pixel_area
        obs = np.asarray(self.last_obs_raw, dtype=np.float64).reshape(e.obs_dim, e.obs_dim)
        return {"phase_screen_opd": opd, "focal_power": focal, "obs_power": obs}

    def render(self, close=False):
        """AO_env.py:156-194 with matplotlib when it is importable; always returns ``render_data()``."""
        data = self.render_data()
        try:
            import matplotlib.pyplot as plt
        except Exception:
            return data
        plt.suptitle("episode %d - timestep %d / %d" % (self.episode_no + 1, self.timestep_render + 1, self.max_steps))
        plt.subplots_adjust(wspace=1, hspace=1)
        for pos, key, title, kw in ((1, "phase_screen_opd", "Atmospheric phase screen $ [\\mu m]$", dict(vmin=-6, vmax=6, cmap="RdBu")),
                                    (3, "focal_power", "Wavefront power on focal plane", dict(vmin=0)),
                                    (4, "obs_power", "Wavefront power on photodetector", dict(vmin=0))):
            plt.subplot(2, 2, pos)
            plt.title(title)
            plt.imshow(data[key], origin="lower", **kw)
            plt.colorbar()
        plt.show(block=False)
        plt.pause(0.05)
        plt.clf()
        return data

    def SH_step(self):
        """AO_env.py:254-290: (actuators [A] float64, torch.tensor([1]))."""
        action, log_action = self._env.SH_step()
        return action[0].cpu().numpy(), log_action

    def close(self):
        self._env.close()
