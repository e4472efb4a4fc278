// host_solver.hpp — planning, allocation, kernel dispatch, the exchange and one solver iteration
// Part of clipper_hip.hip (one translation unit; included there, in order).
#pragma once

namespace {

using Ctx = clipper_hip_ctx;

void rowview_free(Shard& s);
void rowview_drop(Ctx* h);
void sub_free(Ctx* h);

int free_shard_buffers(Shard& s) {
  hipSetDevice(s.device);
  auto fr = [](auto*& p) {
    if (p) hipFree(p);
    p = nullptr;
  };
  fr(s.S);
  fr(s.Cs);
  fr(s.part);
  fr(s.u0);
  fr(s.pt);
  fr(s.cab);
  fr(s.X[0]);
  fr(s.X[1]);
  fr(s.ab);
  fr(s.scal);
  fr(s.st);
  fr(s.shared);
  fr(s.marks);
  fr(s.gOff);
  fr(s.gPre);
  fr(s.gvals);
  fr(s.grows);
  fr(s.cctl);
  s.gcap_units = s.gcap_groups = 0;
  fr(s.sSizes);
  fr(s.sLq);
  fr(s.sPre);
  fr(s.sBlk);
  fr(s.sdata);
  fr(s.swork);
  s.scap_slices = s.scap_bytes = s.scap_work = 0;
  rowview_free(s);
  s.part_tiles = 0;
  fr(s.P1);
  fr(s.P2);
  fr(s.P1f);
  fr(s.P2f);
  s.capPf = 0;
  fr(s.Adev);
  fr(s.dD1);
  fr(s.dD2);
  s.capP = s.capA = s.capD1 = s.capD2 = 0;
  s.bytes_S = 0;
  return 0;
}

// CLIPPER_HIP_STORE_F32_CSC / _F64_CSC (C == pattern(M) is checked per matrix): M lives in the
// slices only. On one unsharded device with fp32 values the fill kernel emits the groups the
// slices are packed from (csc_single: no dense store at any time); otherwise a dense store
// exists while the matrix is being built and is dropped once the slices are valid.
bool csc_possible(const Ctx* h) { return h->compressed; }
bool csc_single(const Ctx* h) { return csc_possible(h) && h->world == 1 && !h->multiproc; }

int plan_unr(const Ctx* h) {
  return gemv_unr(h->V, static_cast<int>(h->esize()), h->explicitC);
}

// largest row-tile count plan_tiles considers for this (m, W)
int64_t max_tiles(const Ctx* h) {
  const int64_t slots = static_cast<int64_t>(h->cus) * GEMV_WG_PER_CU;
  int64_t nt = std::max<int64_t>(16, ceil_div(slots, std::max(1, h->nstrips)) + 1);
  return nt;
}

// Row tiles per column strip. The grid (strips x tiles) runs in waves of `slots` co-resident
// workgroups (two 8-wave workgroups per CU); a grid a few percent OVER a whole number of waves
// costs a whole extra wave (measured: m = 30k, 118 strips: 5 tiles = 1.15 waves 742 us,
// 4 tiles = 0.92 waves 599 us, 13 tiles = 3.0 waves 611 us; m = 10k, 40 strips: 13 tiles =
// 1.016 waves 78 us, 12 tiles 80 us, 16 tiles = 1.25 waves 95 us). Pick the tile count whose
// last wave is fullest; more tiles cost partial sums, hence the small per-tile penalty.
void plan_tiles(Ctx* h) {
  const int unr = plan_unr(h);
  const int64_t chunk = static_cast<int64_t>(GEMV_NW) * unr;
  h->nstrips = static_cast<int>(ceil_div(h->W, 256));
  const double slots = static_cast<double>(h->cus) * GEMV_WG_PER_CU;
  int64_t nt_max = std::min<int64_t>(max_tiles(h), std::max<int64_t>(1, ceil_div(h->m, chunk)));
  // column shards: the slices are narrow — bound the tile count (k_reduce_pass adds them per
  // element) instead of chasing a full wave of tiny workgroups
  if (h->world > 1) nt_max = std::min<int64_t>(nt_max, 32);
  int64_t best = 1;
  double best_cost = 1e300;
  for (int64_t nt = 1; nt <= nt_max; ++nt) {
    const double w = static_cast<double>(h->nstrips) * static_cast<double>(nt) / slots;
    const double whole = std::floor(w), frac = w - whole;
    const double waves = whole + ((frac <= 0.03 && whole >= 1.0) ? frac : (frac > 0.0 ? 1.0 : 0.0));
    const double cost = waves / w + 0.003 * static_cast<double>(nt);
    if (cost < best_cost) {
      best_cost = cost;
      best = nt;
    }
  }
  int64_t rpt = round_up(ceil_div(h->m, best), chunk);
  h->rows_per_tile = static_cast<int>(rpt);
  h->ntiles = static_cast<int>(ceil_div(h->m, rpt));
}

// (re)allocate everything for an m x m problem
int ensure_problem(Ctx* h, int64_t m) {
  if (m <= 0) return fail(CLIPPER_HIP_E_INVALID, "m must be positive");
  const int64_t P = h->world;
  const int64_t W = round_up(ceil_div(m, P), 64);
  h->m = m;
  h->W = W;
  h->mp = P * W;
  const int64_t six_from = h->compressed ? WINDOW_MIN_M_CSC : WINDOW_MIN_M;
  const int V = h->V_forced ? h->V_forced : (m >= six_from ? 6 : (m >= WINDOW4_MIN_M ? 4 : 1));
  const bool same = (h->alloc_m == m && h->alloc_W == W && h->V == V);
  h->V = V;
  plan_tiles(h);
  if (same) return 0;
  sub_free(h);  // (the live sub-problem of another problem size: its context and lists go with the buffers)
  for (auto& s : h->sh) {
    free_shard_buffers(s);
    HIPCHK(hipSetDevice(s.device));
    const size_t bytesS = static_cast<size_t>(m) * static_cast<size_t>(W) * h->esize();
    s.bytes_S = bytesS;
    // CLIPPER_HIP_STORE_F32_CSC keeps M compressed: the dense store exists only while a path
    // that needs it is in use (ensure_dense)
    if (!h->compressed) HIPCHK(hipMalloc(&s.S, bytesS));
    const size_t nvec = static_cast<size_t>(P * W) * sizeof(double);
    const size_t V = static_cast<size_t>(h->V);
    HIPCHK(hipMalloc(&s.u0, nvec));
    const size_t NSLOT = static_cast<size_t>(nslot(h->V));
    HIPCHK(hipMalloc(&s.pt, 2 * V * 2 * nvec));
    HIPCHK(hipMalloc(&s.cab, 2 * nvec));
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipMalloc(&s.X[k], (V + 1) * VS * nvec));
      HIPCHK(hipMemsetAsync(s.X[k], 0, (V + 1) * VS * nvec, s.stream));
    }
    const size_t Q = static_cast<size_t>(tail_q(h->V));
    const size_t nwg = static_cast<size_t>(ceil_div(m, TAIL_THREADS));
    HIPCHK(hipMalloc(&s.scal, (nwg + ceil_div(nwg, SCAL_FOLD) + 1) * Q * sizeof(double)));
    HIPCHK(hipMalloc(&s.ab, NSLOT * nvec));
    HIPCHK(hipMemsetAsync(s.ab, 0, NSLOT * nvec, s.stream));
    s.part_tiles = static_cast<size_t>(max_tiles(h));
    HIPCHK(hipMalloc(&s.part, s.part_tiles * NSLOT * W * sizeof(double)));
    HIPCHK(hipMalloc(&s.st, 2 * sizeof(SolverState)));
    HIPCHK(hipMemsetAsync(s.st, 0, 2 * sizeof(SolverState), s.stream));
    HIPCHK(hipMalloc(&s.shared, sizeof(SolveShared)));
    HIPCHK(hipMalloc(&s.marks, KIND_CAP));
    HIPCHK(hipMemsetAsync(s.marks, 0, KIND_CAP, s.stream));
    HIPCHK(hipMemsetAsync(s.shared, 0, sizeof(SolveShared), s.stream));
  }
  h->alloc_m = m;
  h->alloc_W = W;
  h->has_matrix = false;
  h->csc_valid = false;
  h->explicitC = false;
  plan_tiles(h);
  h->u0_staged = false;
  h->staged_d = 0;
  return 0;
}

// ---- kernel dispatch over (storage type, explicit C, window size) -------------------------
template <typename T, bool HASC, int V>
void launch_pass_tv(Ctx* h, Shard& s, const SolveArgs& a) {
  constexpr int UNR = gemv_unr(V, sizeof(T), HASC);
  dim3 grid(h->nstrips, h->ntiles), block(GEMV_NW * 64);
  hipLaunchKernelGGL((k_gemv<T, HASC, V, GEMV_NW, UNR>), grid, block, 0, s.stream,
                     static_cast<const T*>(s.S), static_cast<const T*>(s.Cs), h->rows_per_tile, a);
}

template <typename T, bool HASC>
void launch_plain_t(Ctx* h, Shard& s, const double* X) {
  constexpr int UNR = gemv_unr(1, sizeof(T), HASC);
  dim3 grid(h->nstrips, h->ntiles), block(GEMV_NW * 64);
  hipLaunchKernelGGL((k_gemv_plain<T, HASC, GEMV_NW, UNR>), grid, block, 0, s.stream,
                     static_cast<const T*>(s.S), static_cast<const T*>(s.Cs), h->W, h->m,
                     h->rows_per_tile, X, s.part);
}

// calls f(type tag, HASC tag) for the context's storage type and constraint mode
template <typename F>
void dispatch_storage(Ctx* h, F&& f) {
  if (h->storage == CLIPPER_HIP_STORE_F64) {
    if (h->explicitC) f(double{}, std::true_type{});
    else f(double{}, std::false_type{});
  } else {
    if (h->explicitC) f(float{}, std::true_type{});
    else f(float{}, std::false_type{});
  }
}

// G of one solver iteration: decision + mat-vec of the pending window
template <int V>
void launch_pass(Ctx* h, Shard& s, const SolveArgs& a) {
  dispatch_storage(h, [&](auto t, auto c) {
    launch_pass_tv<decltype(t), decltype(c)::value, V>(h, s, a);
  });
}

SliceView slice_view(const Ctx* h, const Shard& s);
SliceView row_view(const Ctx* h, const Shard& s);

// the pair-mode mat-vec alone on table X (matvec API, micro-benchmark)
void launch_plain(Ctx* h, Shard& s, const double* X) {
  if (h->csc_valid) {  // on the slices: part[slot][2][W]
    const SliceView M = slice_view(h, s);
    dim3 grid(static_cast<unsigned>(s.s_nwork)), block(SL_NW * 64);
    if (h->storage == CLIPPER_HIP_STORE_F64)
      hipLaunchKernelGGL((k_gemv_slices_plain<double, 1>), grid, block, 0, s.stream, M, h->W, h->m, X, s.part);
    else
      hipLaunchKernelGGL((k_gemv_slices_plain<float, 1>), grid, block, 0, s.stream, M, h->W, h->m, X, s.part);
    return;
  }
  dispatch_storage(h, [&](auto t, auto c) {
    launch_plain_t<decltype(t), decltype(c)::value>(h, s, X);
  });
}

// G on the slices of M (one shard, C == pattern(M))
SliceView slice_view(const Ctx* h, const Shard& s);

template <int V>
void launch_pass_csc(Ctx* h, Shard& s, const SolveArgs& a) {
  const SliceView M = slice_view(h, s), R = row_view(h, s);
  // (the LAST workgroup records the decided state: on M it is the one with the least to stream)
  dim3 grid(static_cast<unsigned>(std::max(M.nwork, R.nwork))), block(SL_NW * 64);
  const SliceView* rdev = s.rv.desc;  // (read only while a.in_view says there is a view)
  if (h->storage == CLIPPER_HIP_STORE_F64)
    hipLaunchKernelGGL((k_gemv_slices<double, 1, V>), grid, block, 0, s.stream, M, rdev, a);
  else
    hipLaunchKernelGGL((k_gemv_slices<float, 1, V>), grid, block, 0, s.stream, M, rdev, a);
}

// calls f(integral_constant<V>) for the context's window size
template <typename F>
void dispatch_window(const Ctx* h, F&& f) {
  switch (h->V) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
#ifndef CLIPPER_NO_V8  /* (harness builds with narrower workgroups) */
    case 8: f(std::integral_constant<int, 8>{}); break;
#endif
    default: f(std::integral_constant<int, 6>{}); break;
  }
}

// plain reduction of `nslots` partial slots into this shard's block (matvec API)
void launch_reduce(Ctx* h, Shard& s, int nslots) {
  dim3 grid(static_cast<unsigned>(ceil_div(static_cast<int64_t>(nslots) * h->W, 256))), block(256);
  hipLaunchKernelGGL(k_reduce, grid, block, 0, s.stream, s.part,
                     h->csc_valid ? s.s_nslots : h->ntiles, nslots, h->W,
                     s.ab + static_cast<int64_t>(s.slot) * nslots * h->W);
}

// exchange of the per-shard blocks [nslots][W] so that every shard holds the gathered sums
int exchange(Ctx* h, int nslots) {
  if (h->world == 1 && !h->multiproc) return 0;
  const int64_t blk_elems = static_cast<int64_t>(nslots) * h->W;
  const size_t blk = static_cast<size_t>(blk_elems) * sizeof(double);
  if (h->multiproc && h->xchg_fn) {
    // through the caller: own block -> pinned host memory -> callback (e.g. a gloo all-gather) ->
    // every rank's block back to the device. One host round trip per pass: a portability and test
    // back-end (two processes on one GPU), not a fast path.
    Shard& s = h->sh[0];
    HIPCHK(hipSetDevice(s.device));
    const size_t need = static_cast<size_t>(h->world) * blk;
    if (need > h->xchg_cap) {
      if (h->xchg_send) hipHostFree(h->xchg_send);
      if (h->xchg_recv) hipHostFree(h->xchg_recv);
      h->xchg_send = h->xchg_recv = nullptr;
      h->xchg_cap = 0;
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->xchg_send), blk, hipHostMallocDefault));
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->xchg_recv), need, hipHostMallocDefault));
      h->xchg_cap = need;
    }
    HIPCHK(hipMemcpyAsync(h->xchg_send, s.ab + static_cast<int64_t>(s.slot) * blk_elems, blk,
                          hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    if (int rc = h->xchg_fn(h->xchg_user, h->xchg_send, h->xchg_recv, blk))
      return fail(CLIPPER_HIP_E_COMM, "the exchange callback returned %d", rc);
    HIPCHK(hipMemcpyAsync(s.ab, h->xchg_recv, need, hipMemcpyHostToDevice, s.stream));
    return 0;
  }
  if (h->multiproc) {
    if (!h->comm) return fail(CLIPPER_HIP_E_COMM, "clipper_hip_comm_init was not called");
    Shard& s = h->sh[0];
    ncclResult_t r = g_rccl.AllGather(s.ab + static_cast<int64_t>(s.slot) * blk_elems, s.ab,
                                      static_cast<size_t>(blk_elems), ncclDouble, h->comm,
                                      s.stream);
    if (r != ncclSuccess)
      return fail(CLIPPER_HIP_E_COMM, "ncclAllGather: %s",
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
    return 0;
  }
  // in-process group: device-to-device copies, ordered with events
  for (auto& p : h->sh) {
    HIPCHK(hipSetDevice(p.device));
    HIPCHK(hipEventRecord(p.ev_reduced, p.stream));
  }
  for (auto& q : h->sh) {
    HIPCHK(hipSetDevice(q.device));
    for (auto& p : h->sh) {
      if (p.slot == q.slot) continue;
      HIPCHK(hipStreamWaitEvent(q.stream, p.ev_reduced, 0));
      const int64_t off = static_cast<int64_t>(p.slot) * blk_elems;
      if (p.device == q.device) {
        HIPCHK(hipMemcpyAsync(q.ab + off, p.ab + off, blk, hipMemcpyDeviceToDevice, q.stream));
      } else {
        HIPCHK(hipMemcpyPeerAsync(q.ab + off, q.device, p.ab + off, p.device, blk, q.stream));
      }
    }
    HIPCHK(hipEventRecord(q.ev_copied, q.stream));
  }
  // a producer may not overwrite its block (next reduce) before every consumer copied it
  for (auto& p : h->sh) {
    HIPCHK(hipSetDevice(p.device));
    for (auto& q : h->sh) {
      if (p.slot == q.slot) continue;
      HIPCHK(hipStreamWaitEvent(p.stream, q.ev_copied, 0));
    }
  }
  return 0;
}

// arguments of the launches of ONE solver iteration: starts from state copy / table set `par`,
// records what it decided in state copy `par ^ 1` and writes the windows of every outcome to
// table set `par ^ 1`
SolveArgs solve_args(Ctx* h, Shard& s, const SolverParams& prm, int par) {
  SolveArgs a;
  a.st_cur = s.st + par;
  a.st_next = s.st + (par ^ 1);
  a.shared = s.shared;
  a.host = (&s == &h->sh[0]) ? h->mirror_dev : nullptr;
  a.prm = prm;
  a.m = h->m;
  a.W = h->W;
  a.mp = h->mp;
  a.u0 = s.u0;
  a.pt = s.pt;
  a.cab = s.cab;
  a.Xin = s.X[par];
  a.Xout = s.X[par ^ 1];
  a.ab = s.ab;
  a.part = s.part;
  a.ntiles = h->csc_valid ? s.s_nslots : h->ntiles;
  a.slot = s.slot;
  a.scal = s.scal;
  a.nwg = static_cast<int>(ceil_div(h->m, TAIL_THREADS));
  a.scal_in = s.scal;
  a.nwg_in = a.nwg;
  if (a.nwg > SCAL_FOLD_MIN) {  // folded copy behind the partials themselves
    a.scal_in = s.scal + static_cast<int64_t>(a.nwg) * tail_q(h->V);
    a.nwg_in = static_cast<int>(ceil_div(a.nwg, SCAL_FOLD));
  }
  a.marks = (h->profiling && &s == &h->sh[0]) ? s.marks : nullptr;
  a.kind = (a.marks && !h->multiproc) ? h->kind_dev : nullptr;
  a.host_u = (!h->multiproc && &s == &h->sh[0]) ? h->u_pinned_dev : nullptr;
  a.stamps = (&s == &h->sh[0]) ? h->stamps_dev : nullptr;
  a.stamps_wide = h->stamps_rows > 4096 ? 1 : 0;
  // the row view, while one is in use (one shard)
  const bool view = h->csc_valid && s.rv.valid;
  a.in_view = view ? s.rv.in_view[s.rv.cur] : nullptr;
  a.rv_nslots = view ? s.rv.st.s_nslots : 0;
  a.rv_nwork = view ? s.rv.st.s_nwork : 0;
  a.rv_fresh = (view && h->rv_fresh) ? 1 : 0;
  a.rv_rows = view ? static_cast<int>(s.rv.nrows) : 0;
  a.rvp = h->rvp;
  a.decide_only = h->decide_only ? 1 : 0;
  // the window in use (SolverState::weff): the pass on the slices of one shard may multiply candidate 0 alone
  a.adaptive_window = (h->csc_valid && h->world == 1 && !h->multiproc && h->sh.size() == 1 && h->adaptive_window) ? 1 : 0;
  // the live sub-problem (host_subproblem.hpp)
  a.sub_state = 0;
  a.sub_ncol = 0.0;
  a.colmap = nullptr;
  if (h->parent != nullptr) {  // these launches run ON a sub-problem: they report where the parent's would
    Ctx* p = h->parent;
    a.host = p->mirror_dev;
    a.host_u = p->u_pinned_dev;
    a.marks = p->profiling ? p->sh[0].marks : nullptr;
    a.kind = a.marks ? p->kind_dev : nullptr;
    a.stamps = p->stamps_dev;
    a.stamps_wide = p->stamps_rows > 4096 ? 1 : 0;
    a.sub_state = 2;
    a.sub_ncol = p->sub.ncol;
    a.colmap = p->sub.colmap;
    // (test knob, CLIPPER_HIP_SUB_TEST_LEAVE = k: the k-th launch after every hand-over is told that a column outside
    // holds 1e300 entries — its decision, if it plans a window, hands the solve back)
    static const int test_leave = std::getenv("CLIPPER_HIP_SUB_TEST_LEAVE") ? std::atoi(std::getenv("CLIPPER_HIP_SUB_TEST_LEAVE")) : 0;
    if (test_leave > 0 && p->sub.launches_since_entry >= test_leave) a.sub_ncol = 1e300;
  } else if (h->sub.ready && !h->sub.active && !h->decide_only) {
    a.sub_state = 1;
    a.sub_ncol = h->sub.ncol;
  }
  return a;
}

// One full solver iteration:
//   one shard : k_gemv[_csc] (decision + pass) -> k_tail<V, true> (adds the tile partials itself)
//   sharded   : k_gemv[_csc] -> k_reduce_pass (tile partials -> own block) -> exchange -> k_tail<V, false>
template <int V>
int enqueue_iteration_v(Ctx* h, const SolverParams& prm) {
  const int par = h->par;
  h->par ^= 1;
  const bool sharded = !(h->world == 1 && !h->multiproc);
  // timing events cost ~5-10 us of stream time each: sample every 8th launch only
  Shard& s0 = h->sh[0];
  constexpr int every = PROFILE_EVERY;
  // (the launches of a live sub-problem are sampled into the PARENT's record: one series of launch indices per solve)
  Ctx* hp = h->parent != nullptr ? h->parent : h;
  // iterations 4, 11, then every `every`-th: short solves (20 iterations) still get samples, and
  // one of them is a pass (3 and 9 both hit transitions at cfg4)
  const bool prof = hp->profiling && (hp->launch_counter % every == 4 || hp->launch_counter == 11) &&
                    hp->ev_used < MAX_EVENT_PAIRS;
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    const SolveArgs a = solve_args(h, s, prm, par);
    if (prof && &s == &s0) HIPCHK(hipEventRecord(hp->ev_pairs[2 * hp->ev_used], s.stream));
    if (h->csc_valid) launch_pass_csc<V>(h, s, a);
    else launch_pass<V>(h, s, a);
    if (prof && &s == &s0) {
      HIPCHK(hipEventRecord(hp->ev_pairs[2 * hp->ev_used + 1], s.stream));
      hp->ev_launch_index[hp->ev_used] = hp->launch_counter;
      ++hp->ev_used;
    }
  }
  ++hp->launch_counter;
  if (h->parent != nullptr) ++hp->sub.launches_since_entry;
  const bool fresh_used = h->rv_fresh;
  const int xk = (prof && sharded) ? h->ev_used - 1 : -1;  // this iteration's exchange is timed too
  if (xk >= 0) HIPCHK(hipEventRecord(h->ev_xchg[2 * xk], s0.stream));
  if (sharded) {
    for (auto& s : h->sh) {  // the tile partials of the pass -> this shard's block of `ab`
      HIPCHK(hipSetDevice(s.device));
      const SolveArgs a = solve_args(h, s, prm, par);
      const int64_t n = static_cast<int64_t>(nslot(V)) * h->W;
      hipLaunchKernelGGL(k_reduce_pass, dim3(static_cast<unsigned>(ceil_div(n, 256))), dim3(256), 0,
                         s.stream, a, nslot(V));
    }
    int rc = exchange(h, nslot(V));
    if (rc) return rc;
    if (xk >= 0) {
      HIPCHK(hipSetDevice(s0.device));
      HIPCHK(hipEventRecord(h->ev_xchg[2 * xk + 1], s0.stream));
      h->ev_xchg_used[static_cast<size_t>(xk)] = 1;
    }
  }
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    const SolveArgs a = solve_args(h, s, prm, par);
    dim3 grid(static_cast<unsigned>(a.nwg), V);
    // the pass on the slices builds its window from the point slot: no candidate tables to write
    const bool tables = !h->csc_valid;
    if (sharded) {
      if (tables) hipLaunchKernelGGL((k_tail<V, false, true>), grid, dim3(TAIL_THREADS), 0, s.stream, a);
      else hipLaunchKernelGGL((k_tail<V, false, false>), grid, dim3(TAIL_THREADS), 0, s.stream, a);
    } else {
      if (tables) hipLaunchKernelGGL((k_tail<V, true, true>), grid, dim3(TAIL_THREADS * TAIL_SPLIT), 0, s.stream, a);
      else hipLaunchKernelGGL((k_tail<V, true, false>), grid, dim3(TAIL_THREADS * TAIL_SPLIT), 0, s.stream, a);
    }
    if (a.nwg_in != a.nwg)
      hipLaunchKernelGGL(k_scal_fold, dim3(static_cast<unsigned>(a.nwg_in)), dim3(128), 0, s.stream,
                         a.scal, a.nwg, tail_q(V),
                         const_cast<double*>(a.scal_in), a.shared);
  }
  if (fresh_used) h->rv_fresh = false;  // only the first iteration after a build decides from the state it saw
  return 0;
}

int enqueue_iteration(Ctx* h, const SolverParams& prm) {
  int rc = 0;
  dispatch_window(h, [&](auto v) { rc = enqueue_iteration_v<decltype(v)::value>(h, prm); });
  return rc;
}

// plain pair-mode mat-vec of every local shard on table X[0] (matvec API)
int enqueue_gemv_plain(Ctx* h) {
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    launch_plain(h, s, s.X[0]);
  }
  return 0;
}

// raw (un-normalised) sums of the pair partials into every shard's gathered `ab`
int enqueue_reduce_exchange(Ctx* h) {
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    launch_reduce(h, s, 2);
  }
  return exchange(h, 2);
}

int sync_all(Ctx* h) {
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipStreamSynchronize(s.stream));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// utils::findIndicesOfkLargest (utils.cpp:33-55): min-heap of (value,index), strict '<'
// replacement, output descending. k is clamped to n (the reference pops an empty queue).
std::vector<int32_t> indices_of_k_largest(const std::vector<double>& x, int k) {
  using T = std::pair<double, int>;
  if (k < 1) return {};
  if (static_cast<size_t>(k) > x.size()) k = static_cast<int>(x.size());
  // u >= 0 and mostly 0 after the solve. With at least k positive entries the zeros can be left out of the
  // walk without changing what it keeps: a zero only ever sits in the queue until a positive entry replaces
  // it (strict '<'), and the queue holds positives only from the k-th positive entry on — in the walk over
  // the positives alone it is full at that same moment with the same content, and identical from there.
  std::vector<int> pos;
  bool nonneg = true;
  for (size_t i = 0; i < x.size(); ++i) {
    if (x[i] > 0.0) pos.push_back(static_cast<int>(i));
    else if (!(x[i] == 0.0)) nonneg = false;  // (negative or NaN: the plain walk)
  }
  const bool sparse_walk = nonneg && pos.size() >= static_cast<size_t>(k);
  const size_t n = sparse_walk ? pos.size() : x.size();
  std::priority_queue<T, std::vector<T>, std::greater<T>> q;
  for (size_t t = 0; t < n; ++t) {
    const size_t i = sparse_walk ? static_cast<size_t>(pos[t]) : t;
    if (q.size() < static_cast<size_t>(k)) {
      q.push({x[i], static_cast<int>(i)});
    } else if (q.top().first < x[i]) {
      q.pop();
      q.push({x[i], static_cast<int>(i)});
    }
  }
  std::vector<int32_t> out(static_cast<size_t>(k));
  for (int i = 0; i < k; ++i) {
    out[static_cast<size_t>(k - i - 1)] = q.top().second;
    q.pop();
  }
  return out;
}

Ctx* make_ctx(const int* devices, int nlocal, int storage, int world, int first_slot,
              bool multiproc) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fail(CLIPPER_HIP_E_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
    return nullptr;
  }
  if (storage != CLIPPER_HIP_STORE_F32 && storage != CLIPPER_HIP_STORE_F64 &&
      storage != CLIPPER_HIP_STORE_F32_CSC && storage != CLIPPER_HIP_STORE_F64_CSC) {
    fail(CLIPPER_HIP_E_INVALID, "storage must be CLIPPER_HIP_STORE_F32, _F64, _F32_CSC or _F64_CSC");
    return nullptr;
  }
  Ctx* h = new Ctx();
  h->compressed = (storage == CLIPPER_HIP_STORE_F32_CSC || storage == CLIPPER_HIP_STORE_F64_CSC);
  h->storage = (storage == CLIPPER_HIP_STORE_F32_CSC) ? CLIPPER_HIP_STORE_F32
               : (storage == CLIPPER_HIP_STORE_F64_CSC) ? CLIPPER_HIP_STORE_F64 : storage;
  h->world = world;
  h->multiproc = multiproc;
  h->sh.resize(static_cast<size_t>(nlocal));
  for (int p = 0; p < nlocal; ++p) {
    Shard& s = h->sh[static_cast<size_t>(p)];
    s.device = devices[p];
    s.slot = first_slot + p;
    if (s.device < 0 || s.device >= ndev) {
      fail(CLIPPER_HIP_E_INVALID, "device %d out of range (%d visible)", s.device, ndev);
      clipper_hip_destroy(h);
      return nullptr;
    }
    if (hipSetDevice(s.device) != hipSuccess ||
        hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev_reduced, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming) != hipSuccess) {
      fail(CLIPPER_HIP_E_HIP, "cannot create stream/events on device %d", s.device);
      clipper_hip_destroy(h);
      return nullptr;
    }
  }
  // peer access between distinct devices of an in-process group
  for (auto& a : h->sh)
    for (auto& b : h->sh)
      if (a.device != b.device) {
        hipSetDevice(a.device);
        int can = 0;
        hipDeviceCanAccessPeer(&can, a.device, b.device);
        if (can) {
          hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
          (void)hipGetLastError();
        }
      }
  hipSetDevice(h->sh[0].device);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, h->sh[0].device) == hipSuccess)
    h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipHostMalloc(reinterpret_cast<void**>(&h->host_state), 2 * sizeof(SolveShared),
                    hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_poll[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_poll[1], hipEventDisableTiming) != hipSuccess) {
    fail(CLIPPER_HIP_E_HIP, "cannot allocate pinned solver state");
    clipper_hip_destroy(h);
    return nullptr;
  }
  // progress record the deciding workgroup writes straight into host memory (coherent, mapped)
  if (hipHostMalloc(reinterpret_cast<void**>(&h->mirror), sizeof(HostMirror),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h->mirror_dev), h->mirror, 0) !=
          hipSuccess) {
    fail(CLIPPER_HIP_E_HIP, "cannot allocate the pinned progress record");
    clipper_hip_destroy(h);
    return nullptr;
  }
  std::memset(h->mirror, 0, sizeof(HostMirror));
  if (hipHostMalloc(reinterpret_cast<void**>(&h->kind), KIND_CAP,
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h->kind_dev), h->kind, 0) != hipSuccess) {
    fail(CLIPPER_HIP_E_HIP, "cannot allocate the pinned iteration marks");
    clipper_hip_destroy(h);
    return nullptr;
  }
  std::memset(h->kind, 0, KIND_CAP);
  if (const char* e = std::getenv("CLIPPER_HIP_STAMPS")) {  // measurement only; 2 = every workgroup of a pass
    h->stamps_rows = std::atoi(e) == 2 ? 16384 : 4096;
    const size_t bytes = static_cast<size_t>(h->stamps_rows) * 4 * sizeof(long long);
    if (hipMalloc(&h->stamps_dev, bytes) == hipSuccess) (void)hipMemset(h->stamps_dev, 0, bytes);
    else h->stamps_dev = nullptr;
  }
  // CLIPPER_HIP_WINDOW = 1 | 4 | 6 | 8: line-search candidates multiplied per pass over M
  if (const char* w = std::getenv("CLIPPER_HIP_WINDOW")) {
    const int v = std::atoi(w);
    if (v == 1 || v == 4 || v == 6 || v == 8) h->V_forced = v;
  }
  if (const char* e = std::getenv("CLIPPER_HIP_ADAPTIVE_WINDOW"))
    if (std::atoi(e) == 0) h->adaptive_window = false;
  // CLIPPER_HIP_RESIDENT = 0: never the resident solver
  if (const char* e = std::getenv("CLIPPER_HIP_RESIDENT"))
    if (std::atoi(e) == 0) h->resident_mode = 1;
  return h;
}

// uploads D (d x n, column-major) and gathers the per-association point table on device
template <typename T>
int ensure_cap(T*& p, size_t& cap, size_t bytes) {
  if (bytes <= cap && p) return 0;
  if (p) hipFree(p);
  p = nullptr;
  cap = 0;
  HIPCHK(hipMalloc(&p, bytes));
  cap = bytes;
  return 0;
}

int upload_points(Ctx* h, Shard& s, const double* D1, const double* D2, int d, int64_t n1,
                  int64_t n2, int64_t pstride) {
  const size_t b1 = static_cast<size_t>(d) * n1 * sizeof(double);
  const size_t b2 = static_cast<size_t>(d) * n2 * sizeof(double);
  const size_t bp = static_cast<size_t>(d) * pstride * sizeof(double);
  const size_t ba = static_cast<size_t>(2 * h->m) * sizeof(int32_t);
  int rc;
  if ((rc = ensure_cap(s.dD1, s.capD1, b1))) return rc;
  if ((rc = ensure_cap(s.dD2, s.capD2, b2))) return rc;
  size_t capP2 = s.capP;
  if ((rc = ensure_cap(s.P1, s.capP, bp))) return rc;
  if ((rc = ensure_cap(s.P2, capP2, bp))) return rc;
  if ((rc = ensure_cap(s.Adev, s.capA, ba))) return rc;
  size_t capPf2 = s.capPf;
  if ((rc = ensure_cap(s.P1f, s.capPf, bp / 2))) return rc;
  if ((rc = ensure_cap(s.P2f, capPf2, bp / 2))) return rc;
  HIPCHK(hipMemcpyAsync(s.dD1, D1, b1, hipMemcpyHostToDevice, s.stream));
  HIPCHK(hipMemcpyAsync(s.dD2, D2, b2, hipMemcpyHostToDevice, s.stream));
  HIPCHK(hipMemcpyAsync(s.Adev, h->A.data(), ba, hipMemcpyHostToDevice, s.stream));
  dim3 grid(static_cast<unsigned>(ceil_div(pstride, 256))), block(256);
  hipLaunchKernelGGL(k_gather_points, grid, block, 0, s.stream, s.dD1, d, s.Adev, h->m, pstride,
                     s.P1, s.P1f);
  hipLaunchKernelGGL(k_gather_points, grid, block, 0, s.stream, s.dD2, d, s.Adev + h->m, h->m,
                     pstride, s.P2, s.P2f);
  HIPCHK(hipStreamSynchronize(s.stream));
  return 0;
}

// common front part of both affinity entry points: A handling + allocation + point tables
int stage_inputs(Ctx* h, const double* D1, int d, int64_t n1, const double* D2, int64_t n2,
                 const int32_t* A, int64_t m_in) {
  if (!h || !D1 || !D2 || d < 1 || n1 < 1 || n2 < 1)
    return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  int64_t m = m_in;
  if (A == nullptr || m_in == 0) {  // clipper.cpp:24 -> utils::createAllToAll (utils.h:61-71)
    m = n1 * n2;
    h->A.assign(static_cast<size_t>(2 * m), 0);
    for (int64_t i = 0; i < n1; ++i)
      for (int64_t j = 0; j < n2; ++j) {
        h->A[static_cast<size_t>(j + i * n2)] = static_cast<int32_t>(i);
        h->A[static_cast<size_t>(m + j + i * n2)] = static_cast<int32_t>(j);
      }
  } else {
    h->A.assign(A, A + 2 * m);
  }
  for (int64_t r = 0; r < m; ++r) {
    const int32_t a0 = h->A[static_cast<size_t>(r)], a1 = h->A[static_cast<size_t>(m + r)];
    if (a0 < 0 || a0 >= n1 || a1 < 0 || a1 >= n2)
      return fail(CLIPPER_HIP_E_INVALID, "association %lld = (%d,%d) out of range",
                  static_cast<long long>(r), a0, a1);
  }
  h->nodes.clear();
  // The point tables are about to be replaced. A matrix that is still held was scored from the OLD
  // points: a row view of it must no longer be re-scored from what is staged (it would mix two point
  // sets, and the device-side coverage check looks at rows, not at values) — from here on it is
  // viewed through the filter of its own slices, until an affinity call scores the new points.
  h->fill_kind = 0;
  rowview_drop(h);
  int rc = ensure_problem(h, m);
  if (rc) return rc;
  const int64_t pstride = round_up(m, 64);
  // (the fill kernels address the gathered point tables with 32-bit byte offsets: k_affinity.hip.h, pt_at)
  if (static_cast<int64_t>(d) * pstride * static_cast<int64_t>(sizeof(double)) >= (int64_t(1) << 32))
    return fail(CLIPPER_HIP_E_SCOPE, "point tables of %d x %lld doubles exceed 4 GiB", d, static_cast<long long>(pstride));
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    rc = upload_points(h, s, D1, D2, d, n1, n2, pstride);
    if (rc) return rc;
  }
  h->staged_d = d;
  h->staged_pstride = pstride;
  double mx = 0.0;
  for (int64_t i = 0; i < static_cast<int64_t>(d) * n1; ++i) mx = std::max(mx, std::fabs(D1[i]));
  for (int64_t i = 0; i < static_cast<int64_t>(d) * n2; ++i) mx = std::max(mx, std::fabs(D2[i]));
  h->staged_maxabs = mx;
  const char* mode = std::getenv("CLIPPER_HIP_AFFINITY");
  h->plain_affinity = (mode && std::strcmp(mode, "plain") == 0);
  h->strip_affinity = (mode && std::strcmp(mode, "strip") == 0);
  return 0;
}

// Threshold of the conservative fp32 prefilter: eps + a bound on the fp32 evaluation error of
// | ||pr-pc|| - ||qr-qc|| | for coordinates of magnitude <= maxabs in dimension d
// (input rounding 2^-24 each, d+2 roundings in the norm, both norms, the subtraction:
// < 50 * 2^-24 * maxabs at d = 3; 128*(d+1) * 2^-24 leaves a 10x margin), rounded up.
float guarded_threshold(double eps, double maxabs, int d) {
  const double guard = std::ldexp(128.0 * (d + 1), -24) * maxabs;
  const double t = eps + guard;
  if (!(t < 3.0e38)) return std::numeric_limits<float>::infinity();
  return std::nextafter(static_cast<float>(t), std::numeric_limits<float>::infinity());
}

// E^2 for the square-root-free prefilter of k_affinity_sym, rounded up
float guarded_threshold_sq(float E) {
  if (!(E < 1.0e19f)) return std::numeric_limits<float>::infinity();
  const double e2 = static_cast<double>(E) * static_cast<double>(E);
  return std::nextafter(static_cast<float>(e2), std::numeric_limits<float>::infinity());
}

// the symmetric tile kernel: one shard; fp32 values (dense store or slices), or fp64 values in SLICES (its fp64
// image leaves no room for a second workgroup per CU, but the rectangular kernel scores every pair twice)
bool use_sym_fill(const Ctx* h) {
  return !h->plain_affinity && !h->strip_affinity && h->world == 1 && !h->multiproc &&
         (h->storage == CLIPPER_HIP_STORE_F32 || (h->storage == CLIPPER_HIP_STORE_F64 && h->compressed));
}

// Kernels that need more dynamic LDS than the 64 KiB a kernel gets by default: the attribute is
// per (function, device) — contexts on distinct devices each raise it, concurrent contexts do not
// race on the record. Returns false if the device refuses (the launch that follows then fails and is
// reported by the caller's launch check).
bool raise_dynamic_lds(const void* fn, int device, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> raised;
  std::lock_guard<std::mutex> lock(mu);
  const std::pair<const void*, int> key{fn, device};
  if (std::find(raised.begin(), raised.end(), key) != raised.end()) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  raised.push_back(key);
  return true;
}

// k_affinity_sym needs more dynamic LDS than the 64 KiB a kernel gets by default
template <typename VT, typename K>
void launch_sym(K kernel, dim3 grid, hipStream_t stream, VT* S, int64_t W, int64_t mm, int nT,
                const Shard& s, int64_t pstride, const int32_t* A0, const int32_t* A1,
                const EuclidParams& e, const PointNormalParams& n, float E2, const CscOut& O) {
  constexpr int L = at_sym_lds_bytes<VT>();
  raise_dynamic_lds(reinterpret_cast<const void*>(kernel), s.device, L);
  hipLaunchKernelGGL(kernel, grid, dim3(AT_WAVES * 64), L, stream, S, W, mm, nT, s.P1, s.P2,
                     s.P1f, s.P2f, pstride, A0, A1, e, n, E2, O);
}


}  // namespace
