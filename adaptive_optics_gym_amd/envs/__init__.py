from .AO_env import AOEnv  # noqa: F401  (mirrors gym_AO/envs/__init__.py:7)
