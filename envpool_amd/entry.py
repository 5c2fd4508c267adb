"""Registers every env of this package (envpool2/entry.py:16-25). Registration itself is
pure Python; the native modules are imported on the first `make`."""
from .registration import register

register(task_id="Dummy-v0", import_path="envpool_amd.dummy", spec_cls="DummyEnvSpec",
         dm_cls="DummyDMEnvPool", gym_cls="DummyGymEnvPool", gymnasium_cls="DummyGymnasiumEnvPool",
         raw_cls="DummyRawEnvPool")

# The spec surface and the GPU observation encoder exist; the pool classes explain that the duel
# engine is not part of this library (envpool2/entry.py registers "YGOPro-v0" the same way).
register(task_id="YGOPro-v0", import_path="envpool_amd.ygopro", spec_cls="YGOProEnvSpec",
         dm_cls="YGOProDMEnvPool", gym_cls="YGOProGymEnvPool", gymnasium_cls="YGOProGymnasiumEnvPool",
         raw_cls="YGOProRawEnvPool")
