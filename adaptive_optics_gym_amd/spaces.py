"""Observation/action spaces.  ``gymnasium`` is used when importable (the reference's callers assert
``type(space) == gym.spaces.Box``, algorithm.py:32-35); otherwise a duck-typed Box with the attributes the
callers read (``shape``, ``low``, ``high``, ``dtype``, ``sample``, ``contains``)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
except Exception:  # gymnasium is not installed in the build image
    _gym = None
    _spaces = None


class Box:
    """Minimal stand-in for ``gymnasium.spaces.Box`` (AO_env.py:45-46)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng()

    def sample(self):
        return self._rng.uniform(self.low.astype(np.float64), self.high.astype(np.float64)).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def __repr__(self):
        return f"Box({self.low.flat[0]}, {self.high.flat[0]}, {self.shape}, {self.dtype})"


def make_box(low, high, shape, dtype):
    if _spaces is not None:
        return _spaces.Box(low=low, high=high, shape=shape, dtype=dtype)
    return Box(low, high, shape, dtype)


def env_base():
    """Base class for the single-env wrapper: ``gymnasium.Env`` when available."""
    return _gym.Env if _gym is not None else object
