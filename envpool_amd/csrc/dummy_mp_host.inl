// dummy_mp_host.inl — host side of the multi-player (ragged player dimension) path.
// Included by pool.hip after ep_pool is complete.

namespace epk_mp {

// Fills and launches mp_step_fused for `cnt` items. `sc` / `ticket_base` / epochs: see
// dummy_mp_kernels.hpp. Shared by the pool path and ep_dummy_send_noalloc.
int launch_fused(hipStream_t st, epk::MpFusedArgs& a, int cnt, int cu_count, bool force_ticket, bool force_generic = false) {
  a.n = cnt;
  a.num_tiles = (cnt + epk::kMpTile - 1) / epk::kMpTile;
  // a CU holds 8 workgroups of 256 threads; half of that is the bound for "all tiles at once"
  a.static_tiles = (a.num_tiles <= 4 * cu_count && !force_ticket) ? 1 : 0;
  if (a.fold_rows > 0 && !(a.static_tiles && a.sc.barrier)) return fail(EP_ERR_INVALID, "internal: folded histogram without static tiles");
  const bool dev = a.P > 1 && !a.force_reset && a.list_action && a.stamp && a.stats.ep_return && a.fold_rows == 0 && !force_generic;
  if (dev) hipLaunchKernelGGL(epk::mp_step_fused<true>, dim3(a.num_tiles), dim3(epk::kMpTile), 0, st, a);
  else hipLaunchKernelGGL(epk::mp_step_fused<false>, dim3(a.num_tiles), dim3(epk::kMpTile), 0, st, a);
  HIP_TRY(hipGetLastError());
  return EP_OK;
}

int enqueue_mp(ep_pool* p, const int32_t* d_env_id, const int32_t* d_penv, const double* d_la, int n,
               int np_rows, bool force_reset, const std::vector<int>* cuts, bool host_checked,
               bool host_mode, const uint32_t* la_ok) {
  const uint64_t B = (uint64_t)p->B;
  // one histogram per send (ParseAction scans the WHOLE players.env_id array for every env,
  // env.h:128-132), tagged with a per-pool send counter
  uint32_t count_epoch = (uint32_t)++p->mp_count_epoch;
  if (count_epoch == 0) {  // 2^32 sends: tags of the last lap must not look new again
    HIP_TRY(hipMemsetAsync(p->mp_sc.counts, 0, 8 * (size_t)p->N, p->stream));
    count_epoch = (uint32_t)++p->mp_count_epoch;
  }
  // The histogram as a phase of the fused kernel instead of a launch of its own (dummy_mp_kernels.hpp,
  // MpFusedArgs::fold_*): when the send is ONE launch (it fits the head batch and nothing cuts it) whose
  // workgroups are all resident at once - at most two per CU, a quarter of what a CU holds, so that
  // other kernels on the device (another pool's resident server, the caller's own) cannot keep one out.
  bool fold = false;
  if (p->mp_fold && !force_reset && np_rows > 0 && !p->mp_force_ticket) {
    const int room0 = (int)(B - p->alloc_count % B);
    bool cut_inside = false;
    if (cuts)
      for (int c : *cuts) cut_inside = cut_inside || (c > 0 && c < n);
    const int tiles = (n + epk::kMpTile - 1) / epk::kMpTile;
    fold = n <= room0 && !cut_inside && tiles <= 2 * p->cu_count && p->mp_sc.barrier != nullptr;
  }
  if (!force_reset && np_rows > 0 && !p->mp_skip_count && !fold) {
    hipLaunchKernelGGL(epk::mp_count_rows, dim3((np_rows + 255) / 256), dim3(256), 0, p->stream, d_penv,
                       np_rows, p->cfg.env_id_offset, p->N, p->mp_sc.counts, count_epoch);
    HIP_TRY(hipGetLastError());
  }
  int i0 = 0;
  size_t next_cut = 0;
  while (i0 < n) {
    uint64_t pos = p->alloc_count;
    uint64_t bi = pos / B;
    int slot = (int)(bi % (uint64_t)p->Q);
    int room = (int)(B - pos % B);
    int cnt = std::min(room, n - i0);
    if (cuts) {
      while (next_cut < cuts->size() && (*cuts)[next_cut] <= i0) ++next_cut;
      if (next_cut < cuts->size()) cnt = std::min(cnt, (*cuts)[next_cut] - i0);
    }
    int row0 = (int)(pos % B);  // rows follow the allocation order: order == -1 whenever P > 1 (async_envpool.h:69,95)
    SlotHost& sh = p->slots[slot];
    const int word = err_index(p, bi);
    // host mode (see enqueue_slices): the batch's rows go into a pinned host block
    if (host_mode && !sh.host_blk && sh.alloc_count == 0 && !sh.has_work) {
      void* blk = nullptr;
      if (int rc = epc::take_block(p->blocks, &blk)) return rc;
      sh.host_blk = blk;
    }
    char* out_base = nullptr;
    if (sh.host_blk) {
      void* dv = nullptr;
      if (hipHostGetDevicePointer(&dv, sh.host_blk, 0) != hipSuccess) return fail(EP_ERR_HIP, "pinned result block is not device-mappable");
      out_base = (char*)dv;
    }
    epk::MpFusedArgs a;
    a.env_id = d_env_id + i0;
    a.players_env_id = nullptr;
    a.list_action = d_la ? d_la + (size_t)i0 * 6 : nullptr;
    a.la_ok = (!d_la && !force_reset) ? la_ok : nullptr;
    a.la_bit0 = i0;
    a.sc = p->mp_sc;
    a.count_epoch = count_epoch;
    a.tile_epoch = next_stamp(p);
    a.ticket_base = p->mp_ticket_total;
    a.env = p->d_env;
    a.stats = p->stats;
    a.slot = out_base ? slot_ptrs_at(p, out_base) : host_slot_ptrs(p, slot);
    a.err = p->d_err + word;
    a.stamp = (!host_checked && !force_reset) ? p->d_stamp : nullptr;
    a.stamp_val = a.tile_epoch;
    a.pc = pool_const(p);
    a.slot_players = p->d_slot_players + slot;
    a.host_players = p->d_host_players + word;
    a.row0 = row0;
    a.force_reset = force_reset ? 1 : 0;
    a.P = p->P;
    a.first_segment = sh.alloc_count == 0 ? 1 : 0;
    a.fold_penv = fold ? d_penv : nullptr;
    a.fold_rows = fold ? np_rows : 0;
    a.fold_base = p->mp_barrier_total;
    {
      TimedScope ts(p);
      if (int rc = launch_fused(p->stream, a, cnt, p->cu_count, p->mp_force_ticket, p->force_generic)) return rc;
    }
    if (!a.static_tiles) p->mp_ticket_total += (uint32_t)a.num_tiles;
    if (fold) p->mp_barrier_total += (uint32_t)a.num_tiles;
    sh.alloc_count += cnt;
    sh.shared_off += cnt;
    sh.written.emplace_back(row0, row0 + cnt);
    sh.has_work = true;
    sh.mp_word = word;
    sh.launch_id = ++p->launch_seq;
    if (int rc2 = mark_launch(p, sh.launch_id)) return rc2;
    p->alloc_count += cnt;
    i0 += cnt;
  }
  return EP_OK;
}

}  // namespace epk_mp
