"""Locates and loads the native pieces of envpool_amd.

`libenvpool_amd.so` (HIP kernels + C ABI) links against `libamdhip64.so.7`. PyTorch
ships its own copy of that runtime under torch/lib with the same SONAME; two HIP
runtimes in one process do not share streams or allocations, so when torch is
installed its copy is pre-loaded (by path, RTLD_GLOBAL) and the dynamic linker
resolves our NEEDED entry to it — whichever of torch / envpool_amd is imported first.
"""
import ctypes
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ENVPOOL_AMD_LIB points diagnostics at an alternative build of the same library (e.g. the
# -DYGO_STAMPS one); the product default is the in-tree library.
LIB_PATH = os.environ.get("ENVPOOL_AMD_LIB") or os.path.join(_HERE, "libenvpool_amd.so")
_lib = None


def _preload_hip_runtime() -> None:
  spec = importlib.util.find_spec("torch")
  if spec is not None and spec.origin:
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
      ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
      return
  for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
    if os.path.exists(cand):
      ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
      return


def load_library() -> ctypes.CDLL:
  """Load libenvpool_amd.so; fails loudly (no CPU fallback exists)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
        "envpool_amd has no CPU fallback."
      )
    _preload_hip_runtime()
    _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
  return _lib
