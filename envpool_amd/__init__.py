"""envpool_amd — MI355X-native drop-in for envpool2's data-parallel hot path.

Public surface mirrors envpool2/__init__.py:16-36 (make / make_gym / make_gymnasium /
make_dm / make_spec / register / list_all_envs). Native code is loaded lazily, on the
first `make`, and fails loudly when the HIP library is missing.
"""
from .registration import (  # noqa: F401
  list_all_envs,
  make,
  make_dm,
  make_gym,
  make_gymnasium,
  make_raw,
  make_spec,
  register,
)
from . import entry  # noqa: F401  (registers Dummy-v0 / YGOPro-v0)

__version__ = "0.1.0"
