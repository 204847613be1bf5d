"""Drop-in shim with the reference's package name: ``import gym_AO`` registers 'AO-v0' (gym_AO/__init__.py:9-12
in the reference) backed by the MI355X implementation, so the reference's ``main.py`` runs unchanged."""
from adaptive_optics_gym_amd import register as _register

registered = _register()
