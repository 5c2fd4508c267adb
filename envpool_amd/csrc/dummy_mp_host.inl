// dummy_mp_host.inl — host side of the multi-player (ragged player dimension) path.
// Included by pool.hip after ep_pool is complete.

namespace epk_mp {

int enqueue_mp(ep_pool* p, const int32_t* d_env_id, const int32_t* d_penv, const double* d_la, int n,
               int np_rows, bool force_reset) {
  const uint64_t B = (uint64_t)p->B;
  const bool count = !force_reset && np_rows > 0;
  if (count) {
    hipLaunchKernelGGL(epk::mp_count_rows, dim3((np_rows + 255) / 256), dim3(256), 0, p->stream, d_penv,
                       np_rows, p->cfg.env_id_offset, p->N, p->d_mp_counts, 1);
    HIP_TRY(hipGetLastError());
  }
  int i0 = 0;
  while (i0 < n) {
    uint64_t pos = p->alloc_count;
    uint64_t bi = pos / B;
    int slot = (int)(bi % (uint64_t)p->Q);
    int room = (int)(B - pos % B);
    int cnt = std::min(room, n - i0);
    int row0 = (int)(pos % B);
    epk::MpStepArgs a;
    a.env_id = d_env_id + i0;
    a.list_action = d_la ? d_la + (size_t)i0 * 6 : nullptr;
    a.counts = p->d_mp_counts;
    a.env = p->d_env;
    a.stats = p->stats;
    a.slot = host_slot_ptrs(p, slot);
    a.err = p->d_err + slot;
    a.pc = pool_const(p);
    a.items = (epk::MpItem*)p->d_mp_items;
    a.n = cnt;
    a.row0 = row0;
    a.force_reset = force_reset ? 1 : 0;
    a.P = p->P;
    hipLaunchKernelGGL(epk::mp_step, dim3((cnt + 255) / 256), dim3(256), 0, p->stream, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(epk::mp_scan, dim3(1), dim3(1024), 0, p->stream, (const epk::MpItem*)p->d_mp_items, cnt,
                       p->d_mp_off, p->d_slot_players + slot, p->d_mp_base);
    HIP_TRY(hipGetLastError());
    epk::MpWriteArgs w;
    w.items = (const epk::MpItem*)p->d_mp_items;
    w.off = p->d_mp_off;
    w.seg_base = p->d_mp_base;
    w.slot = a.slot;
    w.n = cnt;
    w.S = p->S;
    hipLaunchKernelGGL(epk::mp_write_players, dim3((cnt + 255) / 256), dim3(256), 0, p->stream, w);
    HIP_TRY(hipGetLastError());
    SlotHost& sh = p->slots[slot];
    sh.alloc_count += cnt;
    sh.shared_off += cnt;
    sh.written.emplace_back(row0, row0 + cnt);
    sh.has_work = true;
    sh.launch_id = ++p->launch_seq;
    if (int rc2 = mark_launch(p, sh.launch_id)) return rc2;
    p->alloc_count += cnt;
    i0 += cnt;
  }
  if (count) {
    hipLaunchKernelGGL(epk::mp_count_rows, dim3((np_rows + 255) / 256), dim3(256), 0, p->stream, d_penv,
                       np_rows, p->cfg.env_id_offset, p->N, p->d_mp_counts, 0);
    HIP_TRY(hipGetLastError());
  }
  return EP_OK;
}

}  // namespace epk_mp
