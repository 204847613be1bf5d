"""MI355X-native implementation of the AOEnv.reset()/step() hot path of payamparvizi/adaptive_optics_gym.

    from adaptive_optics_gym_amd import BatchedAOEnv      # B envs on one GPU, torch tensors
    from adaptive_optics_gym_amd.envs import AOEnv        # the reference's single-env gym API
    import gym_AO                                         # registers 'AO-v0' like the reference (needs gymnasium)
"""
from .batched_env import BatchedAOEnv  # noqa: F401
from .params import OpticalParams  # noqa: F401

__all__ = ["BatchedAOEnv", "OpticalParams", "register"]


def register():
    """gym_AO/__init__.py:9-12 — register 'AO-v0' with gymnasium when it is importable."""
    try:
        from gymnasium.envs.registration import register as _register, registry
    except Exception:
        return False
    if "AO-v0" not in registry:
        _register(id="AO-v0", entry_point="adaptive_optics_gym_amd.envs:AOEnv")
    return True
