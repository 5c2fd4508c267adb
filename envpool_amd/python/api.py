"""py_env: wrap a pybind (spec, pool) pair into user-facing classes (envpool2/python/api.py:25-39)."""
from typing import Tuple

from .pool import DMFrontEnd, EnvPoolFrontEnd, GymFrontEnd, GymnasiumFrontEnd, make_pool_class
from .spec import make_spec_class


def py_env(envspec: type, envpool: type) -> Tuple[type, type, type, type]:
  spec_name = envspec.__name__[1:]  # drop the "_" pybind prefix
  pool_name = envpool.__name__[1:]
  return (
    make_spec_class(spec_name, envspec),
    make_pool_class(pool_name.replace("EnvPool", "DMEnvPool"), envpool, DMFrontEnd),
    make_pool_class(pool_name.replace("EnvPool", "GymEnvPool"), envpool, GymFrontEnd),
    make_pool_class(pool_name.replace("EnvPool", "GymnasiumEnvPool"), envpool, GymnasiumFrontEnd),
  )


def raw_env(envpool: type) -> type:
  """A fourth flavour with no gym/gymnasium/dm_env dependency at all."""
  return make_pool_class(envpool.__name__[1:].replace("EnvPool", "RawEnvPool"), envpool, EnvPoolFrontEnd)
