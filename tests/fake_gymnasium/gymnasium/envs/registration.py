"""register / registry of the gymnasium stand-in (see gymnasium/__init__.py here)."""
from dataclasses import dataclass, field


@dataclass
class EnvSpec:
    id: str
    entry_point: str
    kwargs: dict = field(default_factory=dict)


registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = EnvSpec(id=id, entry_point=entry_point, kwargs=kwargs.get("kwargs", {}) or {})
