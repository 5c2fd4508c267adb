// py_dummy.cpp — pybind11 module `dummy_envpool`: the drop-in `_DummyEnvSpec` /
// `_DummyEnvPool` classes of envpool2/dummy/dummy_envpool.cpp:25-31 (REGISTER macro,
// envpool2/core/py_envpool.h:268-287), implemented over the C ABI in
// include/envpool_amd.h. No HIP here: this file is host glue only.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "py_common.h"

namespace {

// Config keys in the reference's order (env_spec.h:25-30 ++ dummy_envpool.h:47-49); the
// three build-only keys come last so positional 11-tuples of the reference still work.
const std::vector<std::string> kConfigKeys = {
    "num_envs", "batch_size", "num_threads", "max_num_players", "thread_affinity_offset",
    "base_path", "seed", "gym_reset_return_info", "max_episode_steps", "state_num", "action_num",
    "device_id", "env_id_offset", "ring_depth"};
const std::vector<std::string> kStateKeys = {
    "info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount",
    "step_type", "trunc", "obs:raw", "obs:dyn", "info:players.done", "info:players.id"};
const std::vector<std::string> kActionKeys = {"env_id", "players.env_id", "list_action",
                                              "players.action", "players.id"};

py::tuple default_config_values() {
  return py::make_tuple(1, 0, 0, 1, -1, std::string("envpool2"), 42, false,
                        std::numeric_limits<int>::max(), 10, 6, 0, 0, 0);
}

struct DummySpec {
  ep_dummy_config cfg{};
  std::string base_path;
  bool gym_reset_return_info = false;
  py::tuple config_values, state_spec, action_spec;

  explicit DummySpec(const py::tuple& conf) {
    py::tuple def = default_config_values();
    const size_t n = conf.size();
    if (n != 11 && n != kConfigKeys.size())
      throw std::invalid_argument("_DummyEnvSpec expects a config tuple of 11 (reference) or 14 values");
    auto get = [&](size_t i) -> py::object { return i < n ? py::object(conf[i]) : py::object(def[i]); };
    cfg.num_envs = get(0).cast<int>();
    cfg.batch_size = get(1).cast<int>();
    cfg.num_threads = get(2).cast<int>();
    cfg.max_num_players = get(3).cast<int>();
    cfg.thread_affinity_offset = get(4).cast<int>();
    base_path = get(5).cast<std::string>();
    cfg.seed = get(6).cast<int>();
    gym_reset_return_info = get(7).cast<bool>();
    cfg.max_episode_steps = get(8).cast<int>();
    cfg.state_num = get(9).cast<int>();
    cfg.action_num = get(10).cast<int>();
    cfg.device_id = get(11).cast<int>();
    cfg.env_id_offset = get(12).cast<int>();
    cfg.ring_depth = get(13).cast<int>();
    // env_spec.h:74-82
    if (cfg.batch_size > cfg.num_envs)
      throw std::invalid_argument("It is required that batch_size <= num_envs, got num_envs = " +
                                  std::to_string(cfg.num_envs) +
                                  ", batch_size = " + std::to_string(cfg.batch_size));
    if (cfg.batch_size == 0) cfg.batch_size = cfg.num_envs;
    config_values = py::make_tuple(cfg.num_envs, cfg.batch_size, cfg.num_threads, cfg.max_num_players,
                                   cfg.thread_affinity_offset, base_path, cfg.seed, gym_reset_return_info,
                                   cfg.max_episode_steps, cfg.state_num, cfg.action_num, cfg.device_id,
                                   cfg.env_id_offset, cfg.ring_depth);
    using epy::spec_entry;
    const int S = cfg.state_num;
    // env_spec.h:36-42 ++ dummy_envpool.h:76-84
    state_spec = py::make_tuple(
        spec_entry<int>({}), spec_entry<int>({-1}), spec_entry<int>({}), spec_entry<bool>({}),
        spec_entry<float>({-1}), spec_entry<float>({-1}, py::make_tuple(0.0f, 1.0f)),
        spec_entry<int>({}), spec_entry<bool>({}), spec_entry<int>({-1, S}),
        epy::container_entry<int>({-1}, {-1, S}), spec_entry<bool>({-1}),
        spec_entry<int>({-1}, py::make_tuple(0, cfg.max_num_players)));
    // env_spec.h:33-34 ++ dummy_envpool.h:98-103
    action_spec = py::make_tuple(spec_entry<int>({}), spec_entry<int>({-1}), spec_entry<double>({6}),
                                 spec_entry<int>({-1}), spec_entry<int>({-1}));
  }
};

// Above this many envs obs:dyn is handed out lazily (it is O(N^2) bytes: building the
// exact object array costs 4 us per step at 32 envs, 12 us at 64, 127 us at 256 and 2 ms at 1024, against 20-30 us
// for everything else a step does). EP_DUMMY_EAGER_DYN_MAX moves the limit.
inline int eager_dyn_max_env_id() {
  static const int v = [] {
    const char* e = std::getenv("EP_DUMMY_EAGER_DYN_MAX");
    return e ? std::atoi(e) : 32;
  }();
  return v;
}

// ep_host_blocks behind two Python objects: the set, and one block out of it. A block goes back
// to the set when the last numpy array over it (and the HostBlock object itself) has died.
struct HostBlocksPy {
  ep_host_blocks* hb = nullptr;
  size_t bytes = 0;
  explicit HostBlocksPy(size_t n) : bytes(n) { epy::check(ep_host_blocks_create(n, &hb)); }
  ~HostBlocksPy() { ep_host_blocks_destroy(hb); }
  HostBlocksPy(const HostBlocksPy&) = delete;
};
struct HostBlockPy {
  std::shared_ptr<HostBlocksPy> home;
  void* ptr = nullptr;
  ~HostBlockPy() {
    if (ptr) ep_host_blocks_give(home->hb, ptr);
  }
};
inline std::shared_ptr<HostBlockPy> take_host_block(const std::shared_ptr<HostBlocksPy>& home) {
  auto b = std::make_shared<HostBlockPy>();
  b->home = home;
  epy::check(ep_host_blocks_take(home->hb, &b->ptr));
  return b;
}
// numpy array over `count` elements of the block at `offset`; the array keeps the block out of the set
inline py::array host_block_array(const std::shared_ptr<HostBlockPy>& b, size_t offset, const std::vector<py::ssize_t>& shape,
                                  const std::string& dtype) {
  py::dtype dt(dtype);
  size_t n = (size_t)dt.itemsize();
  for (auto d : shape) {
    if (d < 0) throw std::invalid_argument("negative dimension");
    n *= (size_t)d;
  }
  if (offset > b->home->bytes || n > b->home->bytes - offset) throw std::invalid_argument("array does not fit into the block");
  auto* keep = new std::shared_ptr<HostBlockPy>(b);
  py::capsule cap(keep, [](void* p) { delete reinterpret_cast<std::shared_ptr<HostBlockPy>*>(p); });
  return py::array(dt, shape, (char*)b->ptr + offset, cap);
}

// The sharded front-end's stitch (python/sharded.py): K batches become one array per key. For the
// orders that cannot be written in place by the shards (async parts, permuted ids, YGOPro pools) this
// is a copy - 28 MB per batch of 4096 YGOPro rows, 1 ms through one core into freshly mapped pages,
// 70 % of that front-end's recv + send. Here the copy goes into a recycled block (pages already
// there) in chunks over a few threads that are kept for the purpose.
struct CopyJob {
  char* dst;
  const char* src;
  size_t n;
};
class CopyCrew {
 public:
  explicit CopyCrew(int helpers) {
    for (int i = 0; i < helpers; ++i) crew_.emplace_back([this] { loop(); });
  }
  ~CopyCrew() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& t : crew_) t.join();
  }
  void run(const std::vector<CopyJob>& jobs) {  // one caller at a time (run_mu_)
    std::lock_guard<std::mutex> one(run_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      jobs_ = &jobs;
      next_.store(0);
      busy_ = (int)crew_.size();
      ++gen_;
    }
    cv_.notify_all();
    work(jobs);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return busy_ == 0; });
    jobs_ = nullptr;
  }

 private:
  void work(const std::vector<CopyJob>& jobs) {
    for (size_t i; (i = next_.fetch_add(1)) < jobs.size();) memcpy(jobs[i].dst, jobs[i].src, jobs[i].n);
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::vector<CopyJob>* jobs;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
        jobs = jobs_;
      }
      work(*jobs);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --busy_;
      }
      done_cv_.notify_one();
    }
  }
  std::vector<std::thread> crew_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  const std::vector<CopyJob>* jobs_ = nullptr;
  std::atomic<size_t> next_{0};
  int busy_ = 0;
  uint64_t gen_ = 0;
  bool quit_ = false;
};
inline CopyCrew& copy_crew() {
  static CopyCrew crew([] {
    int n = 5;  // + the calling thread
    if (const char* e = getenv("EP_STITCH_THREADS")) n = std::max(0, atoi(e) - 1);
    return std::min(n, 31);
  }());
  return crew;
}

// parts[j][k]: key k of part j (None: a key without a fixed-shape array). Key k's parts go one after
// the other into the block from offsets[k]; returns the rows of each key.
inline std::vector<py::ssize_t> stitch_parts(const std::shared_ptr<HostBlockPy>& block, const std::vector<size_t>& offsets,
                                             const std::vector<std::vector<py::object>>& parts) {
  constexpr size_t kChunk = 256 << 10;
  std::vector<CopyJob> jobs;
  std::vector<py::ssize_t> rows(offsets.size(), 0);
  std::vector<py::array> keep;  // (the arrays must outlive the copies; contiguous copies of the odd one included)
  size_t total = 0;
  for (size_t k = 0; k < offsets.size(); ++k) {
    size_t at = offsets[k];
    for (const auto& part : parts) {
      if (part.size() != offsets.size()) throw std::invalid_argument("every part needs one entry per key");
      if (part[k].is_none()) continue;
      py::array a = py::array::ensure(part[k], py::array::c_style);
      if (!a) throw std::invalid_argument("a part is not an array");
      const size_t n = (size_t)a.nbytes();
      if (at > block->home->bytes || n > block->home->bytes - at) throw std::invalid_argument("the parts do not fit into the block");
      for (size_t o = 0; o < n; o += kChunk) jobs.push_back({(char*)block->ptr + at + o, (const char*)a.data() + o, std::min(kChunk, n - o)});
      at += n;
      total += n;
      rows[k] += a.ndim() ? a.shape(0) : 1;
      keep.push_back(std::move(a));
    }
  }
  {
    py::gil_scoped_release rel;
    if (total < (1 << 20)) {
      for (const auto& j : jobs) memcpy(j.dst, j.src, j.n);
    } else {
      copy_crew().run(jobs);
    }
  }
  return rows;
}

struct DummyPool {
  DummySpec spec;
  ep_pool* pool = nullptr;

  explicit DummyPool(const DummySpec& s) : spec(s) {
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_dummy_pool_create(&spec.cfg, &pool);
    }
    epy::check(rc);
  }
  ~DummyPool() { close(); }
  // Additive (gym's convention; the reference's pool has no close()): releases the device memory
  // now instead of when the object dies. Every later call is refused (null pool -> ValueError).
  void close() {
    if (pool) {
      py::gil_scoped_release rel;
      ep_pool_destroy(pool);
      pool = nullptr;
    }
  }
  DummyPool(const DummyPool&) = delete;

  // PyEnvPool::PySend (py_envpool.h:219-225): force-cast to the spec dtype, C-contiguous.
  void send(const std::vector<py::array>& action) {
    if (action.size() != 5) throw std::invalid_argument("Dummy expects 5 action arrays");
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> a0(action[0]);
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> a1(action[1]);
    py::array_t<double, py::array::c_style | py::array::forcecast> a2(action[2]);
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> a3(action[3]);
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> a4(action[4]);
    if (a2.ndim() != 2 || a2.shape(1) != 6) throw std::invalid_argument("list_action must have shape (n, 6)");
    const void* ptrs[5] = {a0.data(), a1.data(), a2.data(), a3.data(), a4.data()};
    int64_t dims[5] = {a0.ndim() ? a0.shape(0) : 1, a1.ndim() ? a1.shape(0) : 1, a2.shape(0),
                       a3.ndim() ? a3.shape(0) : 1, a4.ndim() ? a4.shape(0) : 1};
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_send(pool, 5, ptrs, dims, EP_MEM_HOST);
    }
    epy::check(rc);
  }

  // PyEnvPool::PyReset (py_envpool.h:246-251)
  void reset(const py::array& env_ids) {
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> ids(env_ids);
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_reset(pool, ids.data(), (int32_t)ids.size(), EP_MEM_HOST);
    }
    epy::check(rc);
  }

  py::object make_dyn(const py::array& players_env_id) {
    const int S = spec.cfg.state_num;
    const auto n = players_env_id.shape(0);
    const int32_t* ids = static_cast<const int32_t*>(players_env_id.data());
    if (spec.cfg.env_id_offset + spec.cfg.num_envs <= eager_dyn_max_env_id()) {
      // exact object array of arrays (py_envpool.h:50-77); row r = full((id+1, S), id)
      // (dummy_envpool.h:154-162, 205-208)
      py::array out(py::dtype("object"), std::vector<py::ssize_t>{n});
      auto** cells = static_cast<PyObject**>(out.mutable_data());
      for (py::ssize_t r = 0; r < n; ++r) {
        py::array_t<int32_t> inner({(py::ssize_t)ids[r] + 1, (py::ssize_t)S});
        std::fill(inner.mutable_data(), inner.mutable_data() + inner.size(), ids[r]);
        Py_XDECREF(cells[r]);
        cells[r] = inner.release().ptr();
      }
      return std::move(out);
    }
    py::object lazy = py::module_::import("envpool_amd.python.lazy").attr("LazyDyn");
    return lazy(players_env_id, S);
  }

  // PyEnvPool::PyRecv (py_envpool.h:230-241)
  std::vector<py::object> recv() { return recv_impl(false).cast<std::vector<py::object>>(); }
  // Additive: recv that never waits for rows that have not been sent yet (None then): what a
  // front-end over several pools polls with (python/sharded.py).
  py::object try_recv() { return recv_impl(true); }

  py::object recv_impl(bool poll) {
    auto holder = std::make_shared<epy::BatchHolder>();
    int rc;
    {
      py::gil_scoped_release rel;
      if (poll) (void)ep_pool_set_recv_timeout(pool, 0);
      rc = ep_pool_recv(pool, EP_MEM_HOST, &holder->batch);
      if (poll) (void)ep_pool_set_recv_timeout(pool, timeout_ms);
    }
    if (poll && rc == EP_ERR_WOULD_BLOCK) return py::none();
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    std::string deferred = batch_with_error ? ep_last_error() : "";
    std::vector<py::object> ret;
    ret.reserve(12);
    for (int k = 0; k < holder->batch.num_keys; ++k) {
      const ep_array& a = holder->batch.arrays[k];
      if (a.dtype == EP_CONTAINER_I32) ret.emplace_back(py::none());
      else ret.emplace_back(epy::view_of(holder, a));
    }
    ret[9] = make_dyn(py::array(ret[1]));
    if (!deferred.empty()) throw std::runtime_error(deferred);
    return py::cast(ret);
  }

  // Additive (ep_pool_host_target): the next batch this pool starts goes into `block`, slot key k
  // at offsets[k]; `_recv_target` then receives it (no arrays are made here: the caller has its own
  // over the block) and returns the number of rows.
  void host_target(const std::shared_ptr<HostBlockPy>& block, const std::vector<size_t>& offsets, int64_t row_offset) {
    if (offsets.size() != 11) throw std::invalid_argument("11 key offsets expected");
    epy::check(ep_pool_host_target(pool, block->home->hb, block->ptr, offsets.data(), row_offset));
  }
  void host_target_cancel() { epy::check(ep_pool_host_target(pool, nullptr, nullptr, nullptr, 0)); }
  int recv_target() {
    ep_batch b;
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_recv(pool, EP_MEM_HOST, &b);
    }
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    std::string deferred = batch_with_error ? ep_last_error() : "";
    const bool in_target = b.owner == nullptr;
    ep_batch_release(&b);
    if (!deferred.empty()) throw std::runtime_error(deferred);
    if (!in_target) throw std::runtime_error("_recv_target: the batch received was started before the target was set");
    return b.num_rows;
  }
  py::object make_dyn_of(const py::array& players_env_id) { return make_dyn(players_env_id); }

  // Additive: the poll of a front-end over several async pools (python/sharded.py). The next batch,
  // if one is complete, is copied to rows row_offset... of the key arrays in `block` (slot-order
  // offsets, as for _host_target) and its own block released at once; no numpy object is made per
  // part. Returns the number of rows, or None when no batch is complete.
  py::object try_recv_into(const std::shared_ptr<HostBlockPy>& block, const std::vector<size_t>& offsets, int64_t row_offset) {
    static const int kStateToSlot[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, -1, 9, 10};
    if (offsets.size() != 11 || row_offset < 0) throw std::invalid_argument("11 key offsets and a row offset >= 0 expected");
    if (spec.cfg.max_num_players != 1) throw std::invalid_argument("_try_recv_into needs max_num_players == 1");
    ep_batch b;
    int rc;
    std::string refused;
    {
      py::gil_scoped_release rel;
      (void)ep_pool_set_recv_timeout(pool, 0);
      rc = ep_pool_recv(pool, EP_MEM_HOST, &b);
      (void)ep_pool_set_recv_timeout(pool, timeout_ms);
      if (rc == EP_OK || rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV) {
        for (int k = 0; k < b.num_keys && refused.empty(); ++k) {
          const int sk = kStateToSlot[k];
          if (sk < 0) continue;
          const ep_array& a = b.arrays[k];
          size_t rowb = a.dtype == EP_BOOL || a.dtype == EP_U8 ? 1 : a.dtype == EP_F64 ? 8 : 4;
          for (int d = 1; d < a.ndim; ++d) rowb *= (size_t)a.shape[d];
          const size_t at = offsets[sk] + (size_t)row_offset * rowb, n = (size_t)a.shape[0] * rowb;
          if (at > block->home->bytes || n > block->home->bytes - at) refused = "_try_recv_into: the batch does not fit into the block";
          else if (n) memcpy((char*)block->ptr + at, a.data, n);
        }
      }
    }
    if (rc == EP_ERR_WOULD_BLOCK) return py::none();
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    std::string deferred = batch_with_error ? ep_last_error() : "";
    const int rows = b.num_rows;
    ep_batch_release(&b);
    if (!refused.empty()) throw std::invalid_argument(refused);
    if (!deferred.empty()) throw std::runtime_error(deferred);
    return py::int_(rows);
  }

  // Additive API: zero-copy device views, as (ptr, shape, typestr) descriptors in
  // _state_keys order (None for obs:dyn). See envpool_amd/interop.py.
  py::tuple recv_device() { return recv_device_impl(false).cast<py::tuple>(); }
  py::object try_recv_device() { return recv_device_impl(true); }

  py::object recv_device_impl(bool poll) {
    ep_batch b;
    int rc;
    {
      py::gil_scoped_release rel;
      if (poll) (void)ep_pool_set_recv_timeout(pool, 0);
      rc = ep_pool_recv(pool, EP_MEM_DEVICE, &b);
      if (poll) (void)ep_pool_set_recv_timeout(pool, timeout_ms);
    }
    if (poll && rc == EP_ERR_WOULD_BLOCK) return py::none();
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    std::string deferred = batch_with_error ? ep_last_error() : "";
    py::list out;
    for (int k = 0; k < b.num_keys; ++k) {
      if (b.arrays[k].dtype == EP_CONTAINER_I32) out.append(py::none());
      else out.append(epy::device_desc(b.arrays[k]));
    }
    if (!deferred.empty()) throw std::runtime_error(deferred);
    return py::make_tuple(out, b.num_rows, b.num_player_rows);
  }

  // Blocking recv of device views for a front-end that keeps its views per ring slot: (first
  // address, rows, address of the batch's `terminated` array or 0); descriptors via _last_descs.
  py::tuple recv_device_slot() {
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_recv(pool, EP_MEM_DEVICE, &last_batch);
    }
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    if (batch_with_error) throw std::runtime_error(ep_last_error());
    void* term = nullptr;
    if (ep_pool_terminated_of(pool, &last_batch, &term) != EP_OK) term = nullptr;
    return py::make_tuple(reinterpret_cast<uintptr_t>(last_batch.arrays[0].data), last_batch.num_rows, reinterpret_cast<uintptr_t>(term));
  }

  // Stream-ordered form (ep_pool_recv_on_stream): the same descriptors at once; `stream` (a
  // hipStream_t as an integer, 0 = the pool's own step stream, _stream()) is made to wait for
  // the kernels that write the batch instead of the host.
  py::tuple recv_on_stream(uintptr_t stream) {
    ep_batch b;
    epy::check(ep_pool_recv_on_stream(pool, reinterpret_cast<void*>(stream), &b));
    py::list out;
    for (int k = 0; k < b.num_keys; ++k) {
      if (b.arrays[k].dtype == EP_CONTAINER_I32) out.append(py::none());
      else out.append(epy::device_desc(b.arrays[k]));
    }
    return py::make_tuple(out, b.num_rows, b.num_player_rows);
  }

  void send_device(const std::vector<uintptr_t>& ptrs, const std::vector<int64_t>& dims) {
    if (ptrs.size() != 5 || dims.size() != 5) throw std::invalid_argument("need 5 pointers and 5 dims");
    const void* p[5];
    for (int i = 0; i < 5; ++i) p[i] = reinterpret_cast<const void*>(ptrs[i]);
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_send(pool, 5, p, dims.data(), EP_MEM_DEVICE);
    }
    epy::check(rc);
  }

  // ep_pool_step: one blocking step with device-resident actions (no kernel launch for sync pools
  // stepped as a whole); returns what _recv_device returns.
  py::tuple step_device(const std::vector<uintptr_t>& ptrs, const std::vector<int64_t>& dims) {
    if (ptrs.size() != 5 || dims.size() != 5) throw std::invalid_argument("need 5 pointers and 5 dims");
    const void* p[5];
    for (int i = 0; i < 5; ++i) p[i] = reinterpret_cast<const void*>(ptrs[i]);
    ep_batch b;
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_step(pool, 5, p, dims.data(), &b);
    }
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    std::string deferred = batch_with_error ? ep_last_error() : "";
    py::list out;
    for (int k = 0; k < b.num_keys; ++k) {
      if (b.arrays[k].dtype == EP_CONTAINER_I32) out.append(py::none());
      else out.append(epy::device_desc(b.arrays[k]));
    }
    if (!deferred.empty()) throw std::runtime_error(deferred);
    return py::make_tuple(out, b.num_rows, b.num_player_rows);
  }

  // The same for a front-end that keeps its views per ring slot: (first address, rows, address of
  // the batch's `terminated` array or 0), descriptors on request (_last_descs).
  py::tuple step_device_slot(const std::vector<uintptr_t>& ptrs, const std::vector<int64_t>& dims) {
    if (ptrs.size() != 5 || dims.size() != 5) throw std::invalid_argument("need 5 pointers and 5 dims");
    const void* p[5];
    for (int i = 0; i < 5; ++i) p[i] = reinterpret_cast<const void*>(ptrs[i]);
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_step(pool, 5, p, dims.data(), &last_batch);
    }
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    if (batch_with_error) throw std::runtime_error(ep_last_error());
    void* term = nullptr;
    if (ep_pool_terminated_of(pool, &last_batch, &term) != EP_OK) term = nullptr;
    return py::make_tuple(reinterpret_cast<uintptr_t>(last_batch.arrays[0].data), last_batch.num_rows, reinterpret_cast<uintptr_t>(term));
  }

  // ep_pool_step in two halves (ep_pool_step_begin / ep_pool_step_end): a front-end over several
  // pools rings every pool first and then waits for each.
  void step_begin(const std::vector<uintptr_t>& ptrs, const std::vector<int64_t>& dims) {
    if (ptrs.size() != 5 || dims.size() != 5) throw std::invalid_argument("need 5 pointers and 5 dims");
    const void* p[5];
    for (int i = 0; i < 5; ++i) p[i] = reinterpret_cast<const void*>(ptrs[i]);
    epy::check(ep_pool_step_begin(pool, 5, p, dims.data()));
  }
  py::tuple step_end() {
    ep_batch b;
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_step_end(pool, &b);
    }
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    std::string deferred = batch_with_error ? ep_last_error() : "";
    py::list out;
    for (int k = 0; k < b.num_keys; ++k) {
      if (b.arrays[k].dtype == EP_CONTAINER_I32) out.append(py::none());
      else out.append(epy::device_desc(b.arrays[k]));
    }
    if (!deferred.empty()) throw std::runtime_error(deferred);
    return py::make_tuple(out, b.num_rows, b.num_player_rows);
  }
  // The stream-ordered step in two halves (ep_pool_send_on_stream, ep_pool_recv_on_stream): the pool
  // waits for `stream`, `stream` for the rows, the host for nothing. A front-end over several pools
  // sends to all of them before it receives from any. The recv half returns (address of the batch's
  // first key, rows): the front-end keeps its views per ring slot and asks _last_descs() only for
  // a slot it has not seen yet.
  void send_on_stream(const std::vector<uintptr_t>& ptrs, const std::vector<int64_t>& dims, uintptr_t stream) {
    if (ptrs.size() != 5 || dims.size() != 5) throw std::invalid_argument("need 5 pointers and 5 dims");
    const void* p[5];
    for (int i = 0; i < 5; ++i) p[i] = reinterpret_cast<const void*>(ptrs[i]);
    epy::check(ep_pool_send_on_stream(pool, reinterpret_cast<void*>(stream), 5, p, dims.data()));
  }
  py::tuple recv_slot_on_stream(uintptr_t stream) {
    epy::check(ep_pool_recv_on_stream(pool, reinterpret_cast<void*>(stream), &last_batch));
    // third element: the batch's `terminated` array where its step kernels wrote one (ep_pool_terminated_of), else 0
    void* term = nullptr;
    if (ep_pool_terminated_of(pool, &last_batch, &term) != EP_OK) term = nullptr;
    return py::make_tuple(reinterpret_cast<uintptr_t>(last_batch.arrays[0].data), last_batch.num_rows, reinterpret_cast<uintptr_t>(term));
  }
  // ep_pool_recv_many_on_stream: the T batches of a rollout as (first address, rows) each; a
  // front-end asks _descs_of(i) only for ring slots it has not seen yet.
  py::list recv_slots_on_stream(uintptr_t stream, int n) {
    if (n < 0) throw std::invalid_argument("n < 0");
    last_many.resize((size_t)n);
    epy::check(ep_pool_recv_many_on_stream(pool, reinterpret_cast<void*>(stream), n, last_many.data()));
    py::list out;
    for (const ep_batch& b : last_many) out.append(py::make_tuple(reinterpret_cast<uintptr_t>(b.arrays[0].data), b.num_rows));
    return out;
  }
  py::tuple descs_of(int i) {
    if (i < 0 || (size_t)i >= last_many.size()) throw std::invalid_argument("no such batch");
    const ep_batch& b = last_many[(size_t)i];
    py::list out;
    for (int k = 0; k < b.num_keys; ++k) {
      if (b.arrays[k].dtype == EP_CONTAINER_I32) out.append(py::none());
      else out.append(epy::device_desc(b.arrays[k]));
    }
    return py::make_tuple(out, b.num_rows, b.num_player_rows);
  }
  py::tuple last_descs() {
    py::list out;
    for (int k = 0; k < last_batch.num_keys; ++k) {
      if (last_batch.arrays[k].dtype == EP_CONTAINER_I32) out.append(py::none());
      else out.append(epy::device_desc(last_batch.arrays[k]));
    }
    return py::make_tuple(out, last_batch.num_rows, last_batch.num_player_rows);
  }
  void poll_errors() { epy::check(ep_pool_poll_errors(pool)); }
  // ep_pool_set_stream: the pool adopts `stream` as its step stream (own = true: back to its own)
  void set_stream(uintptr_t stream, bool own) {
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_set_stream(pool, reinterpret_cast<void*>(stream), own ? 1 : 0);
    }
    epy::check(rc);
  }
  // ep_pool_step_end for a front-end that keeps its views per ring slot (see _step_device_slot)
  py::tuple step_end_slot() {
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_step_end(pool, &last_batch);
    }
    const bool batch_with_error = rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV;
    if (rc != EP_OK && !batch_with_error) epy::check(rc);
    if (batch_with_error) throw std::runtime_error(ep_last_error());
    void* term = nullptr;
    if (ep_pool_terminated_of(pool, &last_batch, &term) != EP_OK) term = nullptr;
    return py::make_tuple(reinterpret_cast<uintptr_t>(last_batch.arrays[0].data), last_batch.num_rows, reinterpret_cast<uintptr_t>(term));
  }
  py::tuple server_stats() {
    uint64_t a = 0, b = 0, c = 0;
    int64_t d = 0;
    epy::check(ep_pool_server_stats(pool, &a, &b, &c, &d));
    return py::make_tuple(a, b, c, d);
  }

  void rollout(int T, const std::vector<uintptr_t>& ptrs, const std::vector<int64_t>& dims,
               const std::vector<int64_t>& strides) {
    if (ptrs.size() != 5 || dims.size() != 5 || strides.size() != 5)
      throw std::invalid_argument("need 5 pointers, dims and strides");
    const void* p[5];
    for (int i = 0; i < 5; ++i) p[i] = reinterpret_cast<const void*>(ptrs[i]);
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_rollout(pool, T, 5, p, dims.data(), strides.data());
    }
    epy::check(rc);
  }

  py::tuple episode_stats() {
    const int n = ep_pool_num_envs(pool);
    py::array_t<float> ret(n);
    py::array_t<int32_t> len(n), cnt(n);
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_pool_episode_stats(pool, ret.mutable_data(), len.mutable_data(), cnt.mutable_data(), EP_MEM_HOST);
    }
    epy::check(rc);
    return py::make_tuple(ret, len, cnt);
  }

  py::bytes checkpoint_save() {
    std::string buf((size_t)ep_pool_checkpoint_bytes(pool), '\0');
    epy::check(ep_pool_checkpoint_save(pool, buf.data()));
    return py::bytes(buf);
  }
  void checkpoint_restore(const py::bytes& b) {
    std::string buf = b;
    if ((int64_t)buf.size() != ep_pool_checkpoint_bytes(pool))
      throw std::invalid_argument("checkpoint size does not match this pool");
    epy::check(ep_pool_checkpoint_restore(pool, buf.data()));
  }
  void sync() { py::gil_scoped_release rel; epy::check(ep_pool_sync(pool)); }
  int64_t timeout_ms = -1;
  ep_batch last_batch{};  // of the last _recv_slot_on_stream
  std::vector<ep_batch> last_many;  // of the last _recv_slots_on_stream
  void set_recv_timeout(int64_t ms) { epy::check(ep_pool_set_recv_timeout(pool, ms)); timeout_ms = ms; }
  uintptr_t stream() { return reinterpret_cast<uintptr_t>(ep_pool_stream(pool)); }
  uintptr_t handle() { return reinterpret_cast<uintptr_t>(pool); }
};

}  // namespace

// ep_terminated_many with its argument arrays built once: a front-end sees the same ring slots come
// round again, so "terminated of THESE batches" is a handful of fixed plans launched again and again.
struct TermPlan {
  std::vector<const void*> done, trunc;
  std::vector<void*> out;
  std::vector<int64_t> n;
  int device;
  TermPlan(const std::vector<uintptr_t>& d, const std::vector<uintptr_t>& t, const std::vector<uintptr_t>& o, const std::vector<int64_t>& nn, int dev)
      : n(nn), device(dev) {
    if (d.size() != t.size() || d.size() != o.size() || d.size() != nn.size()) throw std::invalid_argument("TermPlan: lists of different lengths");
    for (size_t i = 0; i < d.size(); ++i) {
      done.push_back(reinterpret_cast<const void*>(d[i]));
      trunc.push_back(reinterpret_cast<const void*>(t[i]));
      out.push_back(reinterpret_cast<void*>(o[i]));
    }
  }
  void launch(uintptr_t stream) const {
    epy::check(ep_terminated_many((int32_t)done.size(), done.data(), trunc.data(), out.data(), n.data(), device, reinterpret_cast<void*>(stream)));
  }
};

PYBIND11_MODULE(dummy_envpool, m) {
  py::object abc_meta = py::module_::import("abc").attr("ABCMeta");  // py_envpool.h:261
  py::class_<DummySpec>(m, "_DummyEnvSpec", py::metaclass(abc_meta))
      .def(py::init<const py::tuple&>())
      .def_readonly("_config_values", &DummySpec::config_values)
      .def_readonly("_state_spec", &DummySpec::state_spec)
      .def_readonly("_action_spec", &DummySpec::action_spec)
      .def_property_readonly_static("_state_keys", [](py::object) { return kStateKeys; })
      .def_property_readonly_static("_action_keys", [](py::object) { return kActionKeys; })
      .def_property_readonly_static("_config_keys", [](py::object) { return kConfigKeys; })
      .def_property_readonly_static("_default_config_values", [](py::object) { return default_config_values(); });
  py::class_<HostBlocksPy, std::shared_ptr<HostBlocksPy>>(m, "HostBlocks")
      .def(py::init<size_t>())
      .def_readonly("bytes", &HostBlocksPy::bytes)
      .def("take", [](const std::shared_ptr<HostBlocksPy>& self) { return take_host_block(self); });
  py::class_<HostBlockPy, std::shared_ptr<HostBlockPy>>(m, "HostBlock")
      .def_property_readonly("ptr", [](const HostBlockPy& b) { return reinterpret_cast<uintptr_t>(b.ptr); })
      .def("array", [](const std::shared_ptr<HostBlockPy>& self, size_t offset, const std::vector<py::ssize_t>& shape,
                       const std::string& dtype) { return host_block_array(self, offset, shape, dtype); });
  m.def("stitch_parts", &stitch_parts, py::arg("block"), py::arg("offsets"), py::arg("parts"));
  py::class_<DummyPool>(m, "_DummyEnvPool", py::metaclass(abc_meta))
      .def(py::init<const DummySpec&>())
      .def_readonly("_spec", &DummyPool::spec)
      .def("_recv", &DummyPool::recv)
      .def("_send", &DummyPool::send)
      .def("_reset", &DummyPool::reset)
      .def_property_readonly_static("_state_keys", [](py::object) { return kStateKeys; })
      .def_property_readonly_static("_action_keys", [](py::object) { return kActionKeys; })
      // additive (no reference counterpart)
      .def("_recv_device", &DummyPool::recv_device)
      .def("_host_target", &DummyPool::host_target, py::arg("block"), py::arg("offsets"), py::arg("row_offset") = 0)
      .def("_host_target_cancel", &DummyPool::host_target_cancel)
      .def("_recv_target", &DummyPool::recv_target)
      .def("_make_dyn", &DummyPool::make_dyn_of)
      .def("_try_recv_into", &DummyPool::try_recv_into, py::arg("block"), py::arg("offsets"), py::arg("row_offset"))
      .def("_recv_device_slot", &DummyPool::recv_device_slot)
      .def("_try_recv", &DummyPool::try_recv)
      .def("_try_recv_device", &DummyPool::try_recv_device)
      .def("_recv_on_stream", &DummyPool::recv_on_stream, py::arg("stream") = 0)
      .def("_send_device", &DummyPool::send_device)
      .def("_step_device", &DummyPool::step_device)
      .def("_step_device_slot", &DummyPool::step_device_slot)
      .def("_step_begin", &DummyPool::step_begin)
      .def("_step_end", &DummyPool::step_end)
      .def("_step_end_slot", &DummyPool::step_end_slot)
      .def("_server_stats", &DummyPool::server_stats)
      .def("_send_on_stream", &DummyPool::send_on_stream)
      .def("_recv_slot_on_stream", &DummyPool::recv_slot_on_stream)
      .def("_last_descs", &DummyPool::last_descs)
      .def("_recv_slots_on_stream", &DummyPool::recv_slots_on_stream)
      .def("_descs_of", &DummyPool::descs_of)
      .def("_poll_errors", &DummyPool::poll_errors)
      .def("_set_stream", &DummyPool::set_stream, py::arg("stream"), py::arg("own") = false)
      .def("_rollout", &DummyPool::rollout)
      .def("_episode_stats", &DummyPool::episode_stats)
      .def("_checkpoint_save", &DummyPool::checkpoint_save)
      .def("_checkpoint_restore", &DummyPool::checkpoint_restore)
      .def("_sync", &DummyPool::sync)
      .def("_set_recv_timeout", &DummyPool::set_recv_timeout)
      .def("_stream", &DummyPool::stream)
      .def("_handle", &DummyPool::handle)
      .def("close", &DummyPool::close);
  py::class_<TermPlan>(m, "TermPlan")
      .def(py::init<const std::vector<uintptr_t>&, const std::vector<uintptr_t>&, const std::vector<uintptr_t>&, const std::vector<int64_t>&, int>())
      .def("launch", &TermPlan::launch);
  m.def("device_count", []() { return ep_device_count(); });
  m.def("stream_synchronize", [](int device, uintptr_t stream) {
    int rc;
    {
      py::gil_scoped_release rel;
      rc = ep_stream_synchronize(device, reinterpret_cast<void*>(stream));
    }
    epy::check(rc);
  });
  // ep_terminated_many: done & ~trunc of k device arrays in one launch on `stream` of `device`
  m.def("terminated_many", [](const std::vector<uintptr_t>& done, const std::vector<uintptr_t>& trunc, const std::vector<uintptr_t>& out,
                              const std::vector<int64_t>& n, int device, uintptr_t stream) {
    if (done.size() != trunc.size() || done.size() != out.size() || done.size() != n.size())
      throw std::invalid_argument("terminated_many: lists of different lengths");
    epy::check(ep_terminated_many((int32_t)done.size(), reinterpret_cast<const void* const*>(done.data()),
                                  reinterpret_cast<const void* const*>(trunc.data()), reinterpret_cast<void* const*>(out.data()), n.data(), device,
                                  reinterpret_cast<void*>(stream)));
  });
}
