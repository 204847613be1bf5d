from adaptive_optics_gym_amd.envs import AOEnv  # noqa: F401
