// py_common.h — shared pieces of the pybind11 modules: the Python-visible surface that
// envpool2/core/py_envpool.h builds with templates (PyEnvSpec :136-169, PyEnvPool :204-252,
// REGISTER :268-287) is rebuilt here at run time on top of the C ABI.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "envpool_amd.h"

namespace py = pybind11;

namespace epy {

// C status -> the Python exception type the reference would raise.
inline void check(int rc) {
  if (rc == EP_OK) return;
  std::string msg = ep_last_error();
  if (rc == EP_ERR_INVALID) throw std::invalid_argument(msg);  // -> ValueError (env_spec.h:74-79)
  throw std::runtime_error(msg);
}

template <typename T>
inline py::tuple default_bounds() {  // spec.h:61-62: numeric_limits<D>::min()/max()
  return py::make_tuple(std::numeric_limits<T>::min(), std::numeric_limits<T>::max());
}
inline py::tuple empty_elementwise() { return py::make_tuple(py::list(), py::list()); }

// One entry of _state_spec / _action_spec: (dtype, shape, bounds, elementwise_bounds)
// (py_envpool.h:102-108).
template <typename T>
inline py::tuple spec_entry(std::vector<int> shape, py::object bounds = py::none()) {
  if (bounds.is_none()) bounds = default_bounds<T>();
  return py::make_tuple(py::dtype::of<T>(), shape, bounds, empty_elementwise());
}
// Container entry: shape is (outer_shape, inner_shape) (py_envpool.h:117-125).
template <typename T>
inline py::tuple container_entry(std::vector<int> outer, std::vector<int> inner) {
  return py::make_tuple(py::dtype::of<T>(), py::make_tuple(outer, inner), default_bounds<T>(),
                        empty_elementwise());
}

// Keeps the host block of one received batch alive until the last numpy view of it dies
// (the reference: capsule holding the buffer's shared_ptr, py_envpool.h:42-46).
struct BatchHolder {
  ep_batch batch;
  ~BatchHolder() { ep_batch_release(&batch); }
};

inline py::dtype np_dtype(int ep_dtype) {
  switch (ep_dtype) {
    case EP_I32: return py::dtype::of<int32_t>();
    case EP_F32: return py::dtype::of<float>();
    case EP_F64: return py::dtype::of<double>();
    case EP_BOOL: return py::dtype::of<bool>();
    case EP_U8: return py::dtype::of<uint8_t>();
    default: throw std::runtime_error("no numpy dtype for this ep_dtype");
  }
}

inline py::array view_of(const std::shared_ptr<BatchHolder>& h, const ep_array& a) {
  auto* keep = new std::shared_ptr<BatchHolder>(h);
  py::capsule cap(keep, [](void* p) { delete reinterpret_cast<std::shared_ptr<BatchHolder>*>(p); });
  std::vector<py::ssize_t> shape(a.shape, a.shape + a.ndim);
  return py::array(np_dtype(a.dtype), shape, a.data, cap);
}

// (ptr, shape, typestr) descriptor of a device array, consumed by envpool_amd.interop.
inline py::tuple device_desc(const ep_array& a) {
  const char* typestr = "<i4";
  switch (a.dtype) {
    case EP_I32: typestr = "<i4"; break;
    case EP_F32: typestr = "<f4"; break;
    case EP_F64: typestr = "<f8"; break;
    case EP_BOOL: typestr = "|b1"; break;
    case EP_U8: typestr = "|u1"; break;
    default: break;
  }
  py::tuple shape(a.ndim);
  for (int i = 0; i < a.ndim; ++i) shape[i] = a.shape[i];
  return py::make_tuple(reinterpret_cast<uintptr_t>(a.data), shape, std::string(typestr));
}

}  // namespace epy
