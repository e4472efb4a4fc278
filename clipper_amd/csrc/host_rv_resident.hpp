// host_rv_resident.hpp — planning and launch of the resident solver on a row view (k_rv_resident.hip.h)
// Part of clipper_hip.hip (one translation unit; included there, in order).
#pragma once

namespace {

void rvr_free(Ctx* h) {
  ViewResident& r = h->vres;
  if (r.host_plan) hipHostFree(r.host_plan);
  if (r.xb) hipFree(r.xb);
  if (r.ctl) hipFree(r.ctl);
  if (r.giveup_host) hipHostFree(r.giveup_host);
  if (r.backup) hipFree(r.backup);
  for (hipEvent_t e : r.ev)
    if (e) hipEventDestroy(e);
  const unsigned long long ep = r.epoch;
  const int tu = r.target_units, cd = r.cooldown, cn = r.cooldown_next, mu = r.max_units_device;
  r = ViewResident{};
  r.epoch = ep;
  r.target_units = tu;
  r.cooldown = cd;
  r.cooldown_next = cn;
  r.max_units_device = mu;
}

// Start of a solve: the give-up words of the previous one are gone, a back-off counts down.
void rvr_begin_solve(Ctx* h) {
  ViewResident& r = h->vres;
  r.launches_this_solve = 0;
  r.ev_n = 0;
  // (the launches' control words — error word, arrivals, unit 0's start — are one block per launch of a solve, zeroed by
  // the k_rv_resume of the view build that precedes every launch: no memset on any critical path)
  if (r.giveup_host) std::memset(r.giveup_host, 0, RVR_GIVEUP_SLOTS * 4 * sizeof(uint32_t));
  std::atomic_thread_fence(std::memory_order_seq_cst);
}

// End of a solve (every launch of it has retired: the solve's end was seen behind them in stream order): launches
// that gave up are taken out of `resident_launches`, counted, and the context backs off.
void rvr_end_solve(Ctx* h) {
  ViewResident& r = h->vres;
  if (r.cooldown > 0 && r.launches_this_solve == 0) --r.cooldown;  // (a solve that streamed its views)
  if (r.launches_this_solve == 0 || !r.giveup_host) return;
  std::atomic_thread_fence(std::memory_order_acquire);
  int gave_up = 0;
  volatile const uint32_t* w = r.giveup_host;
  for (int k = 0; k < std::min(r.launches_this_solve, RVR_GIVEUP_SLOTS); ++k) {
    if (w[4 * k] != 0u) {
      ++gave_up;
    } else {  // a launch that ran: its iterations and its duration by the device's wall clock (100 MHz)
      h->rv_stats.resident_iterations += static_cast<int64_t>(w[4 * k + 1]);
      h->rv_stats.resident_us += static_cast<double>(static_cast<uint64_t>(w[4 * k + 2]) | (static_cast<uint64_t>(w[4 * k + 3]) << 32)) * 1e-2;
    }
  }
  for (int k = 0; k < r.ev_n; ++k) {  // (profiling level 2; a launch that gave up is in the sum: it took the time)
    float ms = 0.f;
    // (the solve's end was SEEN from inside the launch: its end event may still be a few microseconds out)
    if (hipEventSynchronize(r.ev[2 * k + 1]) == hipSuccess && hipEventElapsedTime(&ms, r.ev[2 * k], r.ev[2 * k + 1]) == hipSuccess)
      h->rv_stats.resident_event_us += static_cast<double>(ms) * 1e3;
    else (void)hipGetLastError();
  }
  h->rv_stats.resident_entries = static_cast<int64_t>(r.plan_entries);
  h->rv_stats.resident_units = r.nunits;
  if (gave_up > 0) {
    h->rv_stats.resident_launches -= gave_up;
    h->rv_stats.resident_giveups += gave_up;
    r.cooldown = r.cooldown_next;
    r.cooldown_next = std::min(64, 2 * r.cooldown_next);
    if (rs_debug())
      std::fprintf(stderr, "[view-resident] %d launch(es) gave up (error %u): the next %d solve(s) stream their views\n", gave_up,
                   static_cast<unsigned>(w[0]), r.cooldown);
  } else {
    r.cooldown_next = 1;
  }
}

// What every rank of a sharded solve knows alike: the mode set through the API, the storage, the window — the
// GATE of a hand-over. What only this process knows — an environment switch, a back-off after a launch that gave up,
// a device that refuses the LDS — is `rvr_local_ok`: on one shard the two are simply and-ed; with column shards the
// local part is what a rank SAYS in the agreement round, never what decides whether it enters it
// (a rank that skipped a round its peers entered would leave them waiting in the exchange).
bool rvr_gate(const Ctx* h) { return h->rv_mode == 0 && csc_possible(h) && h->sh.size() == 1 && (h->V == 6 || h->V == 4); }
bool rvr_local_ok(const Ctx* h) {
  static const bool env_off = [] {
    const char* e = std::getenv("CLIPPER_HIP_VIEW_RESIDENT");
    return e && std::atoi(e) == 0;
  }();
  return !env_off && h->vres.cooldown == 0;  // (backing off after a launch that gave up: rvr_end_solve)
}
bool rvr_common(const Ctx* h) { return rvr_gate(h) && rvr_local_ok(h); }
// one device, one shard: the shard's view is the whole view
bool rvr_enabled(const Ctx* h) { return rvr_common(h) && csc_single(h); }
// One process per shard (column shards over RCCL or the caller's exchange): every rank keeps a REPLICA of a small
// view over all columns — scored from the points, which are replicated — and runs the launch redundantly: the same
// bits on every rank, no exchange for the iterations inside it (reference loop clipper.cpp:226-280).
bool rvr_replica_gate(const Ctx* h) { return rvr_gate(h) && h->multiproc && rect_fill_possible(h); }
bool rvr_replica_enabled(const Ctx* h) { return rvr_replica_gate(h) && rvr_local_ok(h); }

// could a view of this many rows go to the resident solver? (what rvr_plan checks before it looks at the directory)
bool rvr_candidate(const Ctx* h, int64_t nrows) { return rvr_enabled(h) && nrows >= 1 && nrows <= RVR_MAXROWS; }

// Called when a view has just been built and its directory (csc_hLq) is on the host: does it fit the
// resident solver? Lays out the units (host_plan.hpp) into mapped pinned memory.
int rvr_plan(Ctx* h, Shard& s, bool replica = false) {
  ViewResident& r = h->vres;
  r.ready = false;
  r.on_replica = replica;
  if (!(replica ? rvr_replica_enabled(h) && s.rv.full_valid : rvr_enabled(h)) || !s.rv.valid) return 0;
  const RowView& v = s.rv;
  const SliceStore& vst = replica ? v.full : v.st;
  if (v.nrows < 1 || v.nrows > RVR_MAXROWS) return 0;
  static_assert(sizeof(clipper_plan::ViewUnit) == sizeof(RvrUnit), "the planner's unit is the kernel's");
  const clipper_plan::ResidentConsts K{RVR_NT, RVR_NWV, RVR_TMAX, RVR_PMAX, 2, RS_LDS_MAX,
                                       RVR_RED_BYTES, RVR_TAB_BYTES, RS_SLICE_PAD, SL_SO};
  static thread_local clipper_plan::ViewResidentPlan plan;
  if (r.target_units == 0) {
    const char* e = std::getenv("CLIPPER_HIP_VIEW_RESIDENT_WGS");
    r.target_units = e ? std::max(1, std::atoi(e)) : -1;
  }
  // Co-residency is what the exchange relies on: ask the runtime once how many workgroups of the kernel (512 threads,
  // 159 KB of LDS) the device holds at a time — a device that gives a workgroup less LDS, or a partition with
  // fewer CUs than the properties say, gets fewer units (or none) instead of a launch that can only time out.
  if (r.max_units_device < 0) {
    int per_cu = 0;
    hipError_t e = hipErrorUnknown;
    dispatch_vt(h, [&](auto tag) {
      using VT = decltype(tag);
      auto kern = (h->V == 6) ? k_solve_view_resident<VT, 6> : k_solve_view_resident<VT, 4>;
      if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), s.device, static_cast<int>(RS_LDS_MAX)))
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), RVR_NT, RS_LDS_MAX);
    });
    if (e != hipSuccess) (void)hipGetLastError();
    r.max_units_device = (e == hipSuccess) ? per_cu * h->cus : 0;
    if (rs_debug()) std::fprintf(stderr, "[view-resident] the device holds %d workgroups of the kernel at a time\n", r.max_units_device);
  }
  if (r.max_units_device < 16) return 0;
  const int max_units = std::min({RVR_MAXUNITS, h->cus - 8, r.max_units_device - 8});
  // one workgroup per CU (its LDS is the unit's): about two thirds of the chip by default — more units
  // shorten the pass (the LDS gathers spread over more CUs), every unit adds a granule to everybody's sweep
  const int target = r.target_units > 0 ? r.target_units : std::max(8, (2 * h->cus) / 3);
  const uint32_t fixed = rvr_xt_bytes(h->V, static_cast<int>(v.nrows)) + RVR_RED_BYTES + RVR_TAB_BYTES;
  clipper_plan::plan_view_resident(h->csc_hLq, vst.s_ncg, vst.s_nchunks, static_cast<int>(h->esize()), target,
                                   max_units, fixed, K, plan);
  if (rs_debug())
    std::fprintf(stderr, "[view-resident] plan rows=%lld ncg=%d nchunks=%d entries=%llu -> units=%zu ok=%d\n",
                 static_cast<long long>(v.nrows), vst.s_ncg, vst.s_nchunks,
                 static_cast<unsigned long long>(plan.entries), plan.units.size(), plan.ok ? 1 : 0);
  if (!plan.ok) return 0;
  if (rs_debug() && std::atoi(std::getenv("CLIPPER_HIP_RESIDENT_DEBUG")) >= 2) {  // the units, one line each (tools/rvr_timeline.py --units)
    for (size_t ui = 0; ui < plan.units.size(); ++ui) {
      const clipper_plan::ViewUnit& U = plan.units[ui];
      uint64_t ent = 0;
      for (int c = 0; c < U.ncgs; ++c) ent += plan.ent[static_cast<size_t>(U.cg0 + c)];
      if (U.l1 - U.l0 < 64) ent = ent * static_cast<uint64_t>(U.l1 - U.l0) / 64;
      int wmax = 0, wsum = 0;
      for (int w = 0; w < RVR_NWV; ++w) {
        int st = 0;
        for (int j = 0; j < plan.npieces[ui * RVR_NWV + w]; ++j) {
          const uint32_t pc = plan.pieces[(ui * RVR_NWV + w) * RVR_PMAX + j];
          st += static_cast<int>((pc >> 16) & 255u) - static_cast<int>((pc >> 8) & 255u);
        }
        wmax = std::max(wmax, st);
        wsum += st;
      }
      std::fprintf(stderr, "[view-resident-unit] %zu cg0 %d ncgs %d lanes %d-%d pack %d entries %llu steps_max_wave %d steps_sum %d\n", ui, U.cg0, U.ncgs,
                   U.l0, U.l1, U.pack, static_cast<unsigned long long>(ent), wmax, wsum);
    }
  }
  HIPCHK(hipSetDevice(s.device));
  const size_t off_np = plan.units.size() * sizeof(RvrUnit);
  const size_t off_wc = off_np + static_cast<size_t>(round_up(static_cast<int64_t>(plan.npieces.size()), 16));
  const size_t off_pc = off_wc + static_cast<size_t>(round_up(static_cast<int64_t>(plan.wave_cg.size()), 16));
  const size_t bytes = off_pc + plan.pieces.size() * sizeof(uint32_t) + 64;
  if (bytes > r.host_plan_cap) {
    if (r.host_plan) hipHostFree(r.host_plan);
    r.host_plan = nullptr;
    r.host_plan_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r.host_plan), bytes + 4096, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.host_plan_dev), r.host_plan, 0));
    r.host_plan_cap = bytes + 4096;
  }
  std::memcpy(r.host_plan, plan.units.data(), plan.units.size() * sizeof(RvrUnit));
  std::memcpy(r.host_plan + off_np, plan.npieces.data(), plan.npieces.size());
  std::memcpy(r.host_plan + off_wc, plan.wave_cg.data(), plan.wave_cg.size());
  {  // the pieces: only the ones in use (a few per wave of the pmax slots: a tenth of the table, into pinned memory)
    uint32_t* dst = reinterpret_cast<uint32_t*>(r.host_plan + off_pc);
    const uint32_t* src = plan.pieces.data();
    for (size_t wv = 0; wv < plan.npieces.size(); ++wv)
      for (int j = 0; j < plan.npieces[wv]; ++j) dst[wv * RVR_PMAX + j] = src[wv * RVR_PMAX + j];
  }
  std::atomic_thread_fence(std::memory_order_seq_cst);
  r.off_np = off_np;
  r.off_wc = off_wc;
  r.off_pc = off_pc;
  const size_t xb_bytes = static_cast<size_t>(rvr_xb_granules(6)) * sizeof(unsigned long long);
  if (!r.xb) {
    HIPCHK(hipMalloc(&r.xb, xb_bytes));
    HIPCHK(hipMemsetAsync(r.xb, 0, xb_bytes, s.stream));  // no granule may carry a future epoch
    r.xb_bytes = xb_bytes;
  }
  if (!r.ctl) {
    HIPCHK(hipMalloc(&r.ctl, RVR_GIVEUP_SLOTS * 64));
    HIPCHK(hipMemsetAsync(r.ctl, 0, RVR_GIVEUP_SLOTS * 64, s.stream));
  }
  if (!r.giveup_host) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r.giveup_host), RVR_GIVEUP_SLOTS * 4 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.giveup_host_dev), r.giveup_host, 0));
    std::memset(r.giveup_host, 0, RVR_GIVEUP_SLOTS * 4 * sizeof(uint32_t));
  }
  r.nunits = static_cast<int>(plan.units.size());
  r.plan_entries = plan.entries;
  r.lds_slices = plan.lds_slices;
  r.ready = true;
  return 0;
}

template <typename VT, int V>
int rvr_launch_t(Ctx* h, Shard& s, const RvrArgs& a) {
  auto kern = k_solve_view_resident<VT, V>;
  if (!raise_dynamic_lds(reinterpret_cast<const void*>(kern), s.device, static_cast<int>(RS_LDS_MAX))) return 1;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(a.nunits)), dim3(RVR_NT), RS_LDS_MAX, s.stream, a);
  if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
    if (rs_debug()) std::fprintf(stderr, "[view-resident] launch failed: %s\n", hipGetErrorString(e));
    return 1;
  }
  return 0;
}

// Enqueues the launch behind the decide-only iteration that prepared the pass it starts from. The launch
// continues the solve from state copy h->par on the view in use and leaves a prepared pass (or the end of
// the solve) in the same copy; if it cannot run (no prepared pass, a time-out, a device that refuses the
// LDS) it changes nothing and the streaming launches queued behind it carry on. launched = false: nothing
// was enqueued.
int rvr_enqueue(Ctx* h, const SolverParams& prm, bool& launched) {
  launched = false;
  ViewResident& r = h->vres;
  if (!r.ready || !(r.on_replica ? rvr_replica_enabled(h) : rvr_enabled(h)) || prm.maxiniters < 1) return 0;
  Shard& s = h->sh[0];
  if (!s.rv.valid || (r.on_replica && !s.rv.full_valid)) return 0;
  // Every launch of a solve has its own control block and give-up slot: a launch without one could time out unseen
  // (with column shards the ranks would then disagree on what was committed). A solve builds at most
  // RV_MAX_BUILDS views and launches at most once per build, so this never binds; it is enforced all the same.
  static_assert(RV_MAX_BUILDS <= RVR_GIVEUP_SLOTS, "one give-up slot per view a solve can build");
  if (r.launches_this_solve >= RVR_GIVEUP_SLOTS) return 0;
  HIPCHK(hipSetDevice(s.device));
  RvrArgs a{};
  a.R = r.on_replica ? row_view_full(h, s) : row_view(h, s);
  a.units = reinterpret_cast<const RvrUnit*>(r.host_plan_dev);
  a.nunits = r.nunits;
  a.npieces = r.host_plan_dev + r.off_np;
  a.wave_cg = r.host_plan_dev + r.off_wc;
  a.pieces = reinterpret_cast<const uint32_t*>(r.host_plan_dev + r.off_pc);
  a.viewpos = s.rv.viewpos;
  a.st = s.st + h->par;
  a.shared = s.shared;
  a.host = h->mirror_dev;
  a.host_u = h->u_pinned_dev;
  a.marks = h->profiling ? s.marks : nullptr;
  a.kind = (a.marks && !h->multiproc) ? h->kind_dev : nullptr;  // (as solve_args: a multi-process solve copies the marks itself)
  a.marks_n = h->launch_counter;  // (profiling: launch index = the device's iteration count)
  a.prm = prm;
  a.m = h->m;
  a.mp = h->mp;
  a.pt = s.pt;
  a.xb = r.xb;
  // the granules carry the low 32 bits of the epoch: long before they wrap, start over on a clean buffer
  if ((r.epoch & 0xffffffffull) > 0xf0000000ull) {
    HIPCHK(hipMemsetAsync(r.xb, 0, r.xb_bytes, s.stream));
    r.epoch = (r.epoch & ~0xffffffffull) + (1ull << 32);
  }
  a.epoch0 = r.epoch;
  r.epoch += 1ull << 20;  // whatever this launch publishes (even if it gives up half-way) lies below the next one's
  const bool own_ctl = r.launches_this_solve < RVR_GIVEUP_SLOTS;  // (its block was zeroed by the build's k_rv_resume)
  a.ctl = r.ctl + (own_ctl ? 16 * r.launches_this_solve : 0);
  a.giveup_host = (r.giveup_host_dev && r.launches_this_solve < RVR_GIVEUP_SLOTS) ? r.giveup_host_dev + 4 * r.launches_this_solve : nullptr;
  a.lds_slices = r.lds_slices;
  // The longest a unit waits for the others' granules of ONE exchange, on the 100 MHz wall clock: 2 ms — a hundred
  // iterations' worth (an iteration is 14 us, the first exchange comes 25 us into the launch). A unit that is not
  // resident by then is behind another tenant's work; the streaming launches, which need no co-residency, take over
  // (round 4: 0.2 s — two hundred solves' worth of waiting before a 1.2 ms solve went on).
  a.timeout_ticks = 200000ll;
  // (the first launch of a context waits ten times as long — a cold code object, a clock that ramps up or a
  // profiler that serialises kernels must not look like another tenant and send the context into its back-off)
  if (r.epoch == 0) a.timeout_ticks = 2000000ll;
  if (const char* e = std::getenv("CLIPPER_HIP_VIEW_RESIDENT_TIMEOUT_TICKS")) a.timeout_ticks = std::atoll(e);  // (test knob)
  // (a launch owns 2^20 epochs, one per exchange: the budget stays below, so that nothing it publishes can carry a
  // tag of the next launch's range)
  a.max_exchanges = 1 << 18;
  if (const char* e = std::getenv("CLIPPER_HIP_VIEW_RESIDENT_MAX_EXCHANGES"))
    a.max_exchanges = std::min((1 << 20) - 1, std::max(0, std::atoi(e)));  // (test knob)
  a.rvp = h->rvp;
  a.rv_rows = static_cast<int>(s.rv.nrows);
  a.stamps = h->stamps_dev;
  if (!own_ctl) HIPCHK(hipMemsetAsync(r.ctl, 0, 64, s.stream));  // (more launches in one solve than blocks: block 0, cleared each time)
  const bool timed = h->profiling_level >= 2 && r.ev_n < 16;
  if (timed) {
    for (int k = 0; k < 2; ++k)
      if (!r.ev[2 * r.ev_n + k]) HIPCHK(hipEventCreate(&r.ev[2 * r.ev_n + k]));
    HIPCHK(hipEventRecord(r.ev[2 * r.ev_n], s.stream));
  }
  int lr = 1;
  dispatch_vt(h, [&](auto tag) {
    using VT = decltype(tag);
    lr = (h->V == 6) ? rvr_launch_t<VT, 6>(h, s, a) : rvr_launch_t<VT, 4>(h, s, a);
  });
  if (lr != 0) {
    r.ready = false;  // this device does not take the launch: the streaming launches keep the view
    return 0;
  }
  if (timed) {
    HIPCHK(hipEventRecord(r.ev[2 * r.ev_n + 1], s.stream));
    ++r.ev_n;
  }
  launched = true;
  r.launches_this_solve += 1;
  h->rv_stats.resident_launches += 1;  // (rvr_end_solve takes the ones that gave up out again)
  return 0;
}

// ---- column shards: the hand-over of a freshly built view to redundant resident launches -------------------------
// One flag per rank through the solver's own exchange (a one-slot block, the flag in its first element — as
// gather_slice_bytes): all = every rank said yes. Every rank calls this the same number of times (what leads here
// is a function of the solver state, which is bit-identical on all ranks).
int rvr_agree(Ctx* h, bool mine, bool& all) {
  Shard& s = h->sh[0];
  HIPCHK(hipSetDevice(s.device));
  HIPCHK(hipStreamSynchronize(s.stream));
  const double v = mine ? 1.0 : 0.0;
  HIPCHK(hipMemcpy(s.ab + static_cast<int64_t>(s.slot) * h->W, &v, sizeof(double), hipMemcpyHostToDevice));
  if (int rc = exchange(h, 1)) return rc;
  if (int rc = sync_all(h)) return rc;
  all = true;
  // (the first element of every rank's block, in one strided copy)
  std::vector<double> flags(static_cast<size_t>(h->world), 0.0);
  HIPCHK(hipMemcpy2D(flags.data(), sizeof(double), s.ab, static_cast<size_t>(h->W) * sizeof(double), sizeof(double),
                     static_cast<size_t>(h->world), hipMemcpyDeviceToHost));
  for (double x : flags) all = all && x == 1.0;
  return 0;
}

// The slices of M[view rows, ALL columns] on this rank (RowView::full): the rectangular fill from the replicated
// points with the column range widened to the whole matrix; the same row list as the shard's own view.
int rvr_build_replica(Ctx* h, Shard& s) {
  RowView& v = s.rv;
  v.full_valid = false;
  if (!v.valid) return 0;
  const int64_t wfull = round_up(h->m, 64);
  for (int attempt = 0;; ++attempt) {
    SliceOut O{};
    int rc;
    if ((rc = emit_prepare(h, s, v.full, v.nrows, O, wfull))) return rc;
    if ((rc = launch_rect(h, s, v.rowmap[v.cur], v.nrows, O, 0, wfull))) return rc;
    if ((rc = emit_enqueue(h, s, v.full))) return rc;
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipGetLastError());
    bool again = false;
    if ((rc = emit_check(h, s, v.full, false, again, false))) return rc;  // (no work list: nothing streams the replica)
    if (!again) break;
    if (attempt >= 3) return fail(CLIPPER_HIP_E_HIP, "row view replica: the build keeps overflowing");
  }
  v.full_valid = true;
  return 0;
}

// Called by every rank of a multi-process solve when a view has just been built (the streams are drained, the hold
// is about to be lifted). If the view is small enough, every rank builds the replica and plans the SAME units (the
// plan is a function of the replica's directory, which is a function of the matrix); the ranks agree that all of
// them can (one flag each through the exchange); a decide-only iteration (with its exchange, like any other) turns
// the held decision into a prepared pass; every rank runs the launch — redundantly, on the same bits: no exchange
// for the iterations inside it —; the ranks agree that every launch ran, and where one gave up (a unit that did
// not become resident in time on that GPU) the others put the state back as it was before theirs: all ranks go on
// alike, streaming. The iterations queued afterwards find the solve finished or carry on from the prepared pass
// the launch left, exactly as on one shard.
int rvr_replica_handover(Ctx* h, const SolverParams& prm) {
  ViewResident& r = h->vres;
  r.ready = false;
  if (!h->multiproc || h->sh.size() != 1) return 0;
  Shard& s = h->sh[0];
  // The GATE: only what is the same on every rank — mode, storage, window, the view's rows. Everything a single
  // rank may see differently (an environment switch, its back-off, a device that refuses the plan) goes into the
  // flag it brings to the agreement, not into whether it comes.
  if (!rvr_replica_gate(h) || !s.rv.valid || s.rv.nrows < 1 || s.rv.nrows > RVR_MAXROWS || prm.maxiniters < 1) return 0;
  int rc;
  const auto t0 = std::chrono::high_resolution_clock::now();
  bool mine = false;
  //  A LOCAL failure in front of the first agreement — the replica's build, the plan, the backup's
  // allocation — must not keep this rank out of a round its peers enter: it says "no" in the agreement and reports
  // its error afterwards. A solve never launches more often than it has give-up slots (rvr_enqueue).
  int local_rc = 0;
  if (rvr_local_ok(h) && r.launches_this_solve < RVR_GIVEUP_SLOTS) {
    local_rc = rvr_build_replica(h, s);
    if (!local_rc && s.rv.full_valid) {
      local_rc = rvr_plan(h, s, true);
      mine = !local_rc && r.ready;
    }
    if (mine && !r.backup && hipMalloc(reinterpret_cast<void**>(&r.backup), sizeof(SolverState) + sizeof(SolveShared)) != hipSuccess) {
      (void)hipGetLastError();
      r.backup = nullptr;
      local_rc = fail(CLIPPER_HIP_E_NOMEM, "resident solver on a view: no memory for the state's backup");
      mine = false;
    }
  }
  bool all = false;
  if ((rc = rvr_agree(h, mine, all))) return rc;
  if (local_rc) {
    r.ready = false;
    return local_rc;
  }
  if (rs_debug())
    std::fprintf(stderr, "[view-resident] replica: rows %lld, %.1f MB, this rank %s, all ranks %s\n", static_cast<long long>(s.rv.nrows),
                 static_cast<double>(s.rv.full.s_bytes) / 1e6, mine ? "ready" : "not ready", all ? "ready" : "not ready");
  if (!all) {
    r.ready = false;
    return 0;
  }
  h->decide_only = true;
  rc = enqueue_iteration(h, prm);
  h->decide_only = false;
  if (rc) return rc;
  HIPCHK(hipSetDevice(s.device));
  HIPCHK(hipMemcpyAsync(r.backup, s.st + h->par, sizeof(SolverState), hipMemcpyDeviceToDevice, s.stream));
  HIPCHK(hipMemcpyAsync(r.backup + sizeof(SolverState), s.shared, sizeof(SolveShared), hipMemcpyDeviceToDevice, s.stream));
  bool launched = false;
  const int slot = r.launches_this_solve;
  if ((rc = rvr_enqueue(h, prm, launched))) return rc;
  if (!launched) {  // (the device refused the launch: an attempt that gave up, for the books and the back-off)
    r.launches_this_solve += 1;
    h->rv_stats.resident_launches += 1;
    if (r.giveup_host && slot < RVR_GIVEUP_SLOTS) r.giveup_host[4 * slot] = RVR_ERR_STATE;
  }
  HIPCHK(hipStreamSynchronize(s.stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  const bool ran = launched && !(r.giveup_host && slot < RVR_GIVEUP_SLOTS && static_cast<volatile uint32_t*>(r.giveup_host)[4 * slot] != 0u);
  bool all_ran = false;
  if ((rc = rvr_agree(h, ran, all_ran))) return rc;
  if (!all_ran) {
    if (ran) {  // another rank's launch gave up: this rank's never happened
      HIPCHK(hipMemcpyAsync(s.st + h->par, r.backup, sizeof(SolverState), hipMemcpyDeviceToDevice, s.stream));
      HIPCHK(hipMemcpyAsync(s.shared, r.backup + sizeof(SolverState), sizeof(SolveShared), hipMemcpyDeviceToDevice, s.stream));
      if (r.giveup_host && slot < RVR_GIVEUP_SLOTS) r.giveup_host[4 * slot] = RVR_ERR_PEER;  // (counted as a give-up on every rank: they back off together)
    }
    r.ready = false;
  }
  h->rv_stats.build_ms += std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
  return 0;
}

}  // namespace
