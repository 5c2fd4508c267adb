"""Registers every env of this package (envpool2/entry.py:16-25). Registration itself is
pure Python; the native modules are imported on the first `make`."""
from .registration import register

register(task_id="Dummy-v0", import_path="envpool_amd.dummy", spec_cls="DummyEnvSpec",
         dm_cls="DummyDMEnvPool", gym_cls="DummyGymEnvPool", gymnasium_cls="DummyGymnasiumEnvPool",
         raw_cls="DummyRawEnvPool")
