"""Shared test inputs: seeded smooth screens and actions, and the oracle driven on them."""
import numpy as np
from scipy.ndimage import gaussian_filter


def smooth_screens(B, N, seed, amp=2.5e-5, sigma_frac=0.05):
    """Achromatic screens (phase * lambda) with a few radians rms at the sensing wavelength."""
    rng = np.random.RandomState(seed)
    out = np.empty((B, N, N))
    for b in range(B):
        s = gaussian_filter(rng.randn(N, N), max(1.0, sigma_frac * N))
        out[b] = s / s.std() * amp * 0.12
    return out


def actions_for(B, A, seed):
    return np.random.RandomState(seed + 1000).randn(B, A).astype(np.float32)


def run_oracle(screens, actions_seq, **kw):
    """actions_seq: [T, B, A].  Returns dict of arrays [T, B, ...] from the CPU oracle."""
    from oracle.ao_env_oracle import AOEnvOracle

    T, B = actions_seq.shape[0], screens.shape[0]
    N = screens.shape[1]
    res = {k: [] for k in ("obs0", "obs_raw", "obs", "reward", "done", "power", "strehl")}
    per_env = []
    for b in range(B):
        env = AOEnvOracle(num_pupil_pixels=N, screen=screens[b].ravel(), verbose=False, **kw)
        o0, _ = env.reset()
        rec = dict(obs0=env.last_obs_raw.copy(), obs_raw=[], obs=[], reward=[], done=[], power=[], strehl=[])
        for t in range(T):
            o, r, d, _, info = env.step(actions_seq[t, b])
            rec["obs_raw"].append(env.last_obs_raw.copy())
            rec["obs"].append(o)
            rec["reward"].append(r)
            rec["done"].append(d)
            rec["power"].append(info["power"])
            rec["strehl"].append(getattr(env, "last_strehl", np.nan))
            if d:
                env.reset()
        per_env.append(rec)
    out = {"obs0": np.stack([p["obs0"] for p in per_env])}
    for k in ("obs_raw", "obs", "reward", "done", "power", "strehl"):
        out[k] = np.stack([np.stack(p[k]) for p in per_env], axis=1)  # [T, B, ...]
    return out


class ScriptedRNG:
    """Stands in for the numpy legacy generator the oracle's InfiniteAtmosphericLayer draws from (hcipy's order: ``rand()`` wind
    direction, ``geometric(0.5, n)`` twice for the stencils, then ``normal(0, 1, N)`` per extrusion), replaying values a device
    run used: its wind draw, the batch's shared stencil draws and the normals handed to ``set_extrusion_noise``."""

    def __init__(self, wind_u, geometric_draws, normals=()):
        self._u = float(wind_u)
        self._geo = [np.asarray(g) for g in geometric_draws]
        self.normals = list(normals)     # rows are consumed front to back; the test appends each step's rows before stepping the oracle

    def rand(self):
        return self._u

    def geometric(self, p, n):
        g = self._geo.pop(0)
        assert p == 0.5 and len(g) == n
        return g

    def normal(self, loc, scale, size):
        row = np.asarray(self.normals.pop(0), dtype=np.float64)
        assert loc == 0 and scale == 1 and row.shape == (size,)
        return row

    def randn(self, *a):
        raise AssertionError("the scripted stream holds no screen normals: pass the oracle its initial screen")


def device_mode_stencil_draws(seed, total_envs, n):
    """The two ``geometric(0.5, n)`` stencil draws of a ``BatchedAOEnv(screen_source='device', seed=seed, total_envs=...)``: the stencils
    come from a stream of their own, RandomState([seed, 0x57E9C11]) — independent of ``total_envs`` (wind directions are
    RandomState(seed).rand(total_envs)), so instances holding different slices of one batch share the AR matrices."""
    r = np.random.RandomState([int(seed) & 0xFFFFFFFF, 0x57E9C11])
    return [r.geometric(0.5, n), r.geometric(0.5, n)]
