"""Dummy-v0 (envpool2/dummy/__init__.py:15-30)."""
from .._loader import load_library

load_library()  # fail loudly before the pybind module tries to resolve its symbols

from ..python.api import py_env, raw_env  # noqa: E402
from .dummy_envpool import _DummyEnvPool, _DummyEnvSpec  # noqa: E402

DummyEnvSpec, DummyDMEnvPool, DummyGymEnvPool, DummyGymnasiumEnvPool = py_env(_DummyEnvSpec, _DummyEnvPool)
DummyRawEnvPool = raw_env(_DummyEnvPool)

__all__ = ["DummyEnvSpec", "DummyDMEnvPool", "DummyGymEnvPool", "DummyGymnasiumEnvPool", "DummyRawEnvPool"]
