"""Zero-copy hand-over of device-resident StateBuffer views to PyTorch.

`_recv_device()` of the pybind pools returns (ptr, shape, typestr) descriptors; torch-ROCm
consumes them through the `__cuda_array_interface__` protocol. The view aliases the ring slot:
it stays valid until ring_depth-1 further batches have been received.
"""


class _DeviceArray:
  def __init__(self, ptr: int, shape, typestr: str):
    self.__cuda_array_interface__ = {
      "shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False),
      "version": 2, "strides": None,
    }


def as_torch(desc, device=None):
  """(ptr, shape, typestr) -> torch tensor aliasing the device memory."""
  import torch
  ptr, shape, typestr = desc
  if any(int(s) == 0 for s in shape) or ptr == 0:
    dt = {"<i4": torch.int32, "<f4": torch.float32, "<f8": torch.float64, "|b1": torch.bool, "|u1": torch.uint8}[typestr]
    return torch.empty(tuple(int(s) for s in shape), dtype=dt, device=device)
  return torch.as_tensor(_DeviceArray(ptr, shape, typestr), device=device)
