"""TEST STAND-IN for the ``gymnasium`` package (not installable in the build image): the handful of names the reference's callers
touch — ``gym.Env``, ``gym.spaces.Box``, ``gymnasium.envs.registration.register`` / ``registry`` and ``gym.make`` (main.py:280-292,
gym_AO/__init__.py:9-12, algorithm.py:32-35) — with gymnasium 0.29's call signatures.  ``make`` imports the registered entry point
and instantiates it with the keyword arguments; the real package additionally wraps the env in PassiveEnvChecker / OrderEnforcing,
which only validate calls.  Only tests put this directory on sys.path."""
import importlib

from . import spaces  # noqa: F401
from .envs.registration import register, registry  # noqa: F401


class Env:
    metadata = {"render_modes": []}
    observation_space = None
    action_space = None

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


def make(id, **kwargs):
    spec = registry[id]
    mod, cls = spec.entry_point.split(":")
    return getattr(importlib.import_module(mod), cls)(**{**spec.kwargs, **kwargs})
