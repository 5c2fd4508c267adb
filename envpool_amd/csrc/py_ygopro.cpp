// py_ygopro.cpp — pybind11 module `ygopro_envpool`: the `_YGOProEnvSpec` / `_YGOProEnvPool`
// classes of envpool2/ygopro/ygopro.cpp:4-10 (REGISTER macro, envpool2/core/py_envpool.h:268-287)
// over the C ABI in include/envpool_amd_ygo.h. The duel engine is not part of this library: the
// pool takes it as two native callbacks (addresses), see ep_ygo_engine. No HIP here: host glue only.
#include <cstring>

#include "envpool_amd_ygo.h"
#include "py_common.h"

namespace {

// common config (env_spec.h:25-30) ++ YGOProEnvFns::DefaultConfig (ygopro.h:1179-1186)
const std::vector<std::string> kConfigKeys = {
    "num_envs", "batch_size", "num_threads", "max_num_players", "thread_affinity_offset", "base_path", "seed",
    "gym_reset_return_info", "max_episode_steps", "deck1", "deck2", "player", "play_mode", "verbose",
    "max_options", "max_cards", "n_history_actions", "max_multi_select"};
// common state keys (env_spec.h:36-42) then StateSpec (ygopro.h:1188-1201)
const std::vector<std::string> kStateKeys = {
    "info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount", "step_type", "trunc",
    "obs:cards_", "obs:global_", "obs:actions_", "obs:h_actions_", "info:num_options", "info:to_play",
    "info:is_selfplay", "info:win_reason"};
const std::vector<std::string> kActionKeys = {"env_id", "players.env_id", "action"};

py::tuple default_config_values() {
  return py::make_tuple(1, 0, 0, 1, -1, std::string("envpool2"), 42, false, std::numeric_limits<int>::max(),
                        std::string("OldSchool"), std::string("OldSchool"), -1, std::string("bot"), false, 16, 75, 16, 5);
}

struct YgoSpec {
  int num_envs = 1, batch_size = 0, num_threads = 0, thread_affinity_offset = -1;
  ep_ygo_config cfg{};
  py::tuple config_values, state_spec, action_spec;
  int n_action_feats = 19;

  explicit YgoSpec(const py::tuple& conf) {
    if (conf.size() != kConfigKeys.size())
      throw std::invalid_argument("_YGOProEnvSpec expects " + std::to_string(kConfigKeys.size()) + " config values, got " +
                                  std::to_string(conf.size()));
    num_envs = conf[0].cast<int>();
    batch_size = conf[1].cast<int>();
    num_threads = conf[2].cast<int>();
    thread_affinity_offset = conf[4].cast<int>();
    if (batch_size > num_envs)  // env_spec.h:74-79
      throw std::invalid_argument("It is required that batch_size <= num_envs, got num_envs = " + std::to_string(num_envs) +
                                  ", batch_size = " + std::to_string(batch_size));
    if (batch_size == 0) batch_size = num_envs;
    cfg.max_episode_steps = conf[8].cast<int>();
    cfg.max_options = conf[14].cast<int>();
    cfg.max_cards = conf[15].cast<int>();
    cfg.n_history_actions = conf[16].cast<int>();
    cfg.max_multi_select = conf[17].cast<int>();
    if (cfg.max_options < 1) throw std::invalid_argument("max_options must be >= 1");
    if (cfg.max_cards < 1) throw std::invalid_argument("max_cards must be >= 1");
    if (cfg.n_history_actions < 1) throw std::invalid_argument("n_history_actions must be >= 1");
    if (cfg.max_multi_select < 1) throw std::invalid_argument("max_multi_select must be >= 1");
    py::list vals;
    for (size_t i = 0; i < conf.size(); ++i) vals.append(conf[i]);
    vals[1] = py::int_(batch_size);
    config_values = py::tuple(vals);
    n_action_feats = 9 + 2 * cfg.max_multi_select;  // ygopro.h:1189
    using epy::spec_entry;
    const int nf = n_action_feats;
    auto u8 = py::make_tuple(0, 255);
    state_spec = py::make_tuple(
        spec_entry<int>({}), spec_entry<int>({-1}), spec_entry<int>({}), spec_entry<bool>({}), spec_entry<float>({-1}),
        spec_entry<float>({-1}, py::make_tuple(0.0f, 1.0f)), spec_entry<int>({}), spec_entry<bool>({}),
        spec_entry<uint8_t>({cfg.max_cards * 2, 39}, u8), spec_entry<uint8_t>({8}, u8),
        spec_entry<uint8_t>({cfg.max_options, nf}, u8), spec_entry<uint8_t>({cfg.n_history_actions, nf}, u8),
        spec_entry<int>({}, py::make_tuple(0, cfg.max_options - 1)), spec_entry<int>({}, py::make_tuple(0, 1)),
        spec_entry<int>({}, py::make_tuple(0, 1)), spec_entry<int>({}, py::make_tuple(-1, 1)));
    // env_spec.h:33-34 ++ ygopro.h:1205-1206
    action_spec = py::make_tuple(spec_entry<int>({}), spec_entry<int>({-1}),
                                 spec_entry<int>({}, py::make_tuple(0, cfg.max_options - 1)));
  }
};

struct YgoPool {
  YgoSpec spec;
  ep_ygo_pool* pool = nullptr;
  py::object keep;  // whatever owns the engine callbacks and the card table

  // engine = (ctx, reset_fn, step_fn) as integer addresses of native code with the ep_ygo_engine
  // signatures; card_table = buffer of n * sizeof(ygo_card_data) bytes (cards_ / card_ids_ after the
  // reference's init_module, ygopro.h:1030-1064)
  YgoPool(const YgoSpec& s, const py::tuple& engine, const py::buffer& card_table, int device_id, int env_id_offset,
          py::object owner)
      : spec(s), keep(std::move(owner)) {
    if (engine.size() != 3) throw std::invalid_argument("engine must be (ctx, reset_fn, step_fn)");
    ep_ygo_engine e{};
    e.ctx = reinterpret_cast<void*>(engine[0].cast<uintptr_t>());
    e.reset = reinterpret_cast<int (*)(void*, int32_t, ygo_raw_state*)>(engine[1].cast<uintptr_t>());
    e.step = reinterpret_cast<int (*)(void*, int32_t, int32_t, ygo_raw_state*)>(engine[2].cast<uintptr_t>());
    py::buffer_info info = card_table.request();
    const size_t bytes = (size_t)info.size * (size_t)info.itemsize;
    if (bytes % sizeof(ygo_card_data) != 0) throw std::invalid_argument("card_table is not a whole number of ygo_card_data records");
    ep_ygo_config cfg = spec.cfg;
    cfg.device_id = device_id;
    cfg.ring_depth = 0;
    cfg.max_batch = spec.batch_size;
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_ygo_pool_create(&cfg, &e, spec.num_envs, spec.batch_size, spec.num_threads, spec.thread_affinity_offset,
                              static_cast<const ygo_card_data*>(info.ptr), (int32_t)(bytes / sizeof(ygo_card_data)), &pool);
      if (rc == EP_OK && env_id_offset != 0) rc = ep_ygo_pool_set_env_id_offset(pool, env_id_offset);
    }
    epy::check(rc);
  }
  ~YgoPool() { close(); }
  YgoPool(const YgoPool&) = delete;

  void close() {
    if (pool) {
      py::gil_scoped_release rel;  // workers may be inside Python callbacks that need the GIL to finish
      ep_ygo_pool_destroy(pool);
      pool = nullptr;
    }
  }

  // PyEnvPool::PySend (py_envpool.h:219-225): _action_keys order, force-cast, C-contiguous
  void send(const std::vector<py::array>& action) {
    if (action.size() != 3) throw std::invalid_argument("YGOPro expects 3 action arrays (env_id, players.env_id, action)");
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> ids(action[0]);
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> act(action[2]);
    if (ids.size() != act.size()) throw std::invalid_argument("env_id and action differ in length");
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_ygo_pool_send(pool, (int32_t)ids.size(), ids.data(), act.data());
    }
    epy::check(rc);
  }

  void reset(const py::array& env_ids) {  // py_envpool.h:246-251
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> ids(env_ids);
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_ygo_pool_reset(pool, (int32_t)ids.size(), ids.data());
    }
    epy::check(rc);
  }

  // PyEnvPool::PyRecv (py_envpool.h:230-241): numpy views of the pinned host block, no copy;
  // key 16 (aux:h_card_ids) is additive and returned separately
  py::object recv_all_impl(bool poll) {
    auto holder = std::make_shared<epy::BatchHolder>();
    int rc;
    {
      py::gil_scoped_release rel;
      if (poll) (void)ep_ygo_pool_set_recv_timeout(pool, 0);
      rc = ep_ygo_pool_recv(pool, EP_MEM_HOST, &holder->batch);
      if (poll) (void)ep_ygo_pool_set_recv_timeout(pool, timeout_ms);
    }
    if (poll && rc == EP_ERR_WOULD_BLOCK) return py::none();
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    py::list ret;
    for (int k = 0; k < holder->batch.num_keys; ++k) ret.append(view_of_any(holder, holder->batch.arrays[k]));
    return py::make_tuple(ret, batch_with_error ? std::string(ep_last_error()) : std::string());
  }
  py::object recv_all() { return recv_all_impl(false); }
  // Additive: recv that never waits for a batch that is not complete yet (None then): what a
  // front-end over several pools polls with (python/sharded.py).
  py::object try_recv_all() { return recv_all_impl(true); }

  py::object recv_device_impl(bool poll) {
    ep_batch b;
    int rc;
    {
      py::gil_scoped_release rel;
      if (poll) (void)ep_ygo_pool_set_recv_timeout(pool, 0);
      rc = ep_ygo_pool_recv(pool, EP_MEM_DEVICE, &b);
      if (poll) (void)ep_ygo_pool_set_recv_timeout(pool, timeout_ms);
    }
    if (poll && rc == EP_ERR_WOULD_BLOCK) return py::none();
    if (rc != EP_OK && rc != EP_ERR_ACTION_CHECK) epy::check(rc);
    py::list out;
    for (int k = 0; k < b.num_keys; ++k) out.append(desc_any(b.arrays[k]));
    return py::make_tuple(out, b.num_rows, b.num_rows);
  }
  py::object recv_device() { return recv_device_impl(false); }
  py::object try_recv_device() { return recv_device_impl(true); }

  // Blocking recv of device views for a front-end that keeps its views per output slot of the
  // encoder (python/streamstep.py): (first address, rows); descriptors on request (_last_descs_all).
  py::tuple recv_device_slot() {
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_ygo_pool_recv(pool, EP_MEM_DEVICE, &last_batch);
    }
    if (rc != EP_OK && rc != EP_ERR_ACTION_CHECK) epy::check(rc);
    return py::make_tuple(reinterpret_cast<uintptr_t>(last_batch.arrays[0].data), last_batch.num_rows);
  }
  py::tuple last_descs_all() {
    py::list out;
    for (int k = 0; k < last_batch.num_keys; ++k) out.append(desc_any(last_batch.arrays[k]));
    return py::make_tuple(out, last_batch.num_rows, last_batch.num_rows);
  }
  ep_batch last_batch{};

  // info:env_id of the batch received last, from host memory (a copy: the library's array only lives
  // until the next recv)
  py::array_t<int32_t> last_env_ids() {
    const int32_t* ids = nullptr;
    int32_t n = 0;
    epy::check(ep_ygo_pool_last_env_ids(pool, &ids, &n));
    py::array_t<int32_t> out(n);
    std::memcpy(out.mutable_data(), ids, sizeof(int32_t) * (size_t)n);
    return out;
  }

  int64_t timeout_ms = -1;
  void set_recv_timeout(int64_t ms) { epy::check(ep_ygo_pool_set_recv_timeout(pool, ms)); timeout_ms = ms; }
  int num_threads() { return ep_ygo_pool_num_threads(pool); }
  uintptr_t handle() { return reinterpret_cast<uintptr_t>(pool); }

 private:
  static py::array view_of_any(const std::shared_ptr<epy::BatchHolder>& h, const ep_array& a) {
    if (a.dtype == EP_U16) {  // aux:h_card_ids
      auto* keep = new std::shared_ptr<epy::BatchHolder>(h);
      py::capsule cap(keep, [](void* p) { delete reinterpret_cast<std::shared_ptr<epy::BatchHolder>*>(p); });
      std::vector<py::ssize_t> shape(a.shape, a.shape + a.ndim);
      return py::array(py::dtype::of<uint16_t>(), shape, a.data, cap);
    }
    return epy::view_of(h, a);
  }
  static py::tuple desc_any(const ep_array& a) {
    if (a.dtype == EP_U16) {
      py::tuple shape(a.ndim);
      for (int i = 0; i < a.ndim; ++i) shape[i] = a.shape[i];
      return py::make_tuple(reinterpret_cast<uintptr_t>(a.data), shape, std::string("<u2"));
    }
    return epy::device_desc(a);
  }
};

}  // namespace

PYBIND11_MODULE(ygopro_envpool, m) {
  py::object abc_meta = py::module_::import("abc").attr("ABCMeta");  // py_envpool.h:261
  py::class_<YgoSpec>(m, "_YGOProEnvSpec", py::metaclass(abc_meta))
      .def(py::init<const py::tuple&>())
      .def_readonly("_config_values", &YgoSpec::config_values)
      .def_readonly("_state_spec", &YgoSpec::state_spec)
      .def_readonly("_action_spec", &YgoSpec::action_spec)
      .def_readonly("n_action_feats", &YgoSpec::n_action_feats)
      .def_property_readonly_static("_state_keys", [](py::object) { return kStateKeys; })
      .def_property_readonly_static("_action_keys", [](py::object) { return kActionKeys; })
      .def_property_readonly_static("_config_keys", [](py::object) { return kConfigKeys; })
      .def_property_readonly_static("_default_config_values", [](py::object) { return default_config_values(); });
  py::class_<YgoPool>(m, "_YGOProNativePool", py::metaclass(abc_meta))
      .def(py::init<const YgoSpec&, const py::tuple&, const py::buffer&, int, int, py::object>(), py::arg("spec"),
           py::arg("engine"), py::arg("card_table"), py::arg("device_id") = 0, py::arg("env_id_offset") = 0,
           py::arg("owner") = py::none())
      .def_readonly("_spec", &YgoPool::spec)
      .def("_send", &YgoPool::send)
      .def("_reset", &YgoPool::reset)
      .def("_recv_all", &YgoPool::recv_all)
      .def("_recv_device_all", &YgoPool::recv_device)
      .def("_try_recv_all", &YgoPool::try_recv_all)
      .def("_try_recv_device_all", &YgoPool::try_recv_device)
      .def("_recv_device_slot", &YgoPool::recv_device_slot)
      .def("_last_descs_all", &YgoPool::last_descs_all)
      .def("_last_env_ids", &YgoPool::last_env_ids)
      .def("_set_recv_timeout", &YgoPool::set_recv_timeout)
      .def("_num_threads", &YgoPool::num_threads)
      .def("_handle", &YgoPool::handle)
      .def("close", &YgoPool::close)
      .def_property_readonly_static("_state_keys", [](py::object) { return kStateKeys; })
      .def_property_readonly_static("_action_keys", [](py::object) { return kActionKeys; });
}
