"""Shared helpers of the parity tests."""
import numpy as np

FIXED_KEYS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11]  # every _state_keys index except obs:dyn (9)
KEY_NAMES = ["info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount",
             "step_type", "trunc", "obs:raw", "obs:dyn", "info:players.done", "info:players.id"]


def dummy_action(env_ids, players_env_id=None, value=1.5):
  env_ids = np.asarray(env_ids, dtype=np.int32)
  pe = env_ids if players_env_id is None else np.asarray(players_env_id, dtype=np.int32)
  la = np.full((env_ids.shape[0], 6), value, dtype=np.float64)
  z = np.zeros(pe.shape[0], dtype=np.int32)
  return [env_ids, pe, la, z, z]


def assert_batches_equal(got, want, ctx=""):
  """Bit-exact comparison of two batches in _state_keys order."""
  for k in FIXED_KEYS:
    g, w = np.asarray(got[k]), np.asarray(want[k])
    assert g.dtype == w.dtype, f"{ctx} key {KEY_NAMES[k]}: dtype {g.dtype} != {w.dtype}"
    assert g.shape == w.shape, f"{ctx} key {KEY_NAMES[k]}: shape {g.shape} != {w.shape}"
    # compare raw bytes: +0.0 vs -0.0 must not pass
    assert g.tobytes() == w.tobytes(), f"{ctx} key {KEY_NAMES[k]} differs:\n got {g}\nwant {w}"
