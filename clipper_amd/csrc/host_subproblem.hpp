// host_subproblem.hpp — the LIVE SUB-PROBLEM (k_subproblem.hip.h): when it is prepared, the hand-over of a running
// solve to it and the way back.
// Part of clipper_hip.hip (one translation unit; included there, in order).
//
// The sub-problem is a child CONTEXT on the parent's device and stream: the same fill (k_affinity_sym on the gathered
// points of the associations of S), the same planner, the same launches (k_gemv_slices, k_tail) — only their
// arguments differ (solve_args: sub_state = 2, the parent's progress record and pinned u, the list of S for the final
// u). The solver's state machine does not know: it continues from a prepared pass (SolverState::resume) on other arrays.
#pragma once

namespace {

bool sub_enabled(const Ctx* h) {
  static const bool env_off = [] {
    const char* e = std::getenv("CLIPPER_HIP_SUBPROBLEM");
    return e && std::atoi(e) == 0;
  }();
  static const int64_t min_m = std::getenv("CLIPPER_HIP_SUBPROBLEM_MIN_M") ? std::atoll(std::getenv("CLIPPER_HIP_SUBPROBLEM_MIN_M")) : SUB_MIN_M;
  return !env_off && h->sub_mode != 1 && h->parent == nullptr && h->sh.size() == 1 && h->world == 1 && !h->multiproc &&
         h->csc_valid && !h->explicitC && h->m >= min_m && rect_fill_possible(h);
}

void sub_begin_solve(Ctx* h) {
  SubProblem& sp = h->sub;
  sp.ready = sp.active = false;
  sp.entries = 0;
  sp.sub_passes = sp.leaves = 0;
  sp.build_ms = 0.0;
}

void sub_free(Ctx* h) {
  SubProblem& sp = h->sub;
  for (Ctx** pc : {&sp.ctx, &sp.ctx_dense}) {
    if (*pc) {
      (*pc)->sh[0].stream = nullptr;  // (the parent's: not the child's to destroy)
      clipper_hip_destroy(*pc);
      *pc = nullptr;
    }
  }
  sp.use = nullptr;
  if (!h->sh.empty()) hipSetDevice(h->sh[0].device);
  auto fr = [](auto*& p) {
    if (p) hipFree(p);
    p = nullptr;
  };
  fr(sp.cnt);
  fr(sp.flags);
  fr(sp.colmap);
  fr(sp.pos);
  fr(sp.blk);
  fr(sp.nout_acc);
  if (sp.rec) hipHostFree(sp.rec);
  sp.rec = sp.rec_dev = nullptr;
  sp.cap = sp.cap_blk = 0;
  sp.ready = sp.active = false;
}

// the child's point tables and association pairs, gathered on the device from the parent's (stage_inputs without
// host data); everything else of the child as after clipper_hip_stage_inputs
int sub_stage(Ctx* h, Ctx* c, int64_t nS) {
  Shard& s = h->sh[0];
  Shard& cs = c->sh[0];
  c->V_forced = h->V;      // the state's point slots and partial scalars are laid out by the window size
  c->resident_mode = 1;    // (a solve that is handed over mid-way never starts on the resident solver)
  c->rv_mode = 1;          // no views inside it: its rows are the live rows
  c->nodes.clear();
  c->fill_kind = 0;
  rowview_drop(c);
  int rc = ensure_problem(c, nS);
  if (rc) return rc;
  const int d = h->staged_d;
  const int64_t qs = round_up(nS, 64);
  const size_t bp = static_cast<size_t>(d) * qs * sizeof(double);
  size_t capP2 = cs.capP;
  if ((rc = ensure_cap(cs.P1, cs.capP, bp))) return rc;
  if ((rc = ensure_cap(cs.P2, capP2, bp))) return rc;
  size_t capPf2 = cs.capPf;
  if ((rc = ensure_cap(cs.P1f, cs.capPf, bp / 2))) return rc;
  if ((rc = ensure_cap(cs.P2f, capPf2, bp / 2))) return rc;
  if ((rc = ensure_cap(cs.Adev, cs.capA, static_cast<size_t>(2 * nS) * sizeof(int32_t)))) return rc;
  dim3 grid(static_cast<unsigned>(ceil_div(qs, 256))), block(256);
  hipLaunchKernelGGL(k_sub_gather_points, grid, block, 0, s.stream, s.P1, s.P1f, d, h->staged_pstride, s.Adev, h->m,
                     h->sub.colmap, nS, qs, cs.P1, cs.P1f, cs.Adev, 0);
  hipLaunchKernelGGL(k_sub_gather_points, grid, block, 0, s.stream, s.P2, s.P2f, d, h->staged_pstride, s.Adev, h->m,
                     h->sub.colmap, nS, qs, cs.P2, cs.P2f, cs.Adev, 1);
  c->staged_d = d;
  c->staged_pstride = qs;
  c->staged_maxabs = h->staged_maxabs;  // (the prefilter's guard: the same threshold, the same scores)
  c->plain_affinity = h->plain_affinity;
  c->strip_affinity = h->strip_affinity;
  return 0;
}

// Called when a row view has just been built (the stream is idle but for the lift of the hold): select S, and if it
// is small enough build its problem. Synchronous — a few launches, one wait for the selection, the fill's own wait.
template <int V>
int sub_prepare_v(Ctx* h) {
  SubProblem& sp = h->sub;
  sp.ready = false;
  Shard& s = h->sh[0];
  RowView& v = s.rv;
  if (!sub_enabled(h) || !v.valid || h->vres.ready || sp.entries >= SUB_MAX_ENTRIES) return 0;
  const auto t0 = std::chrono::high_resolution_clock::now();
  HIPCHK(hipSetDevice(s.device));
  const int64_t m = h->m, mp = h->mp;
  const int nblk = static_cast<int>(ceil_div(m, RV_BLK));
  int rc;
  if (static_cast<size_t>(mp) > sp.cap) {
    for (void** p : {reinterpret_cast<void**>(&sp.cnt), reinterpret_cast<void**>(&sp.flags), reinterpret_cast<void**>(&sp.colmap),
                     reinterpret_cast<void**>(&sp.pos)}) {
      if (*p) hipFree(*p);
      *p = nullptr;
    }
    sp.cap = 0;
    HIPCHK(hipMalloc(&sp.cnt, static_cast<size_t>(mp) * sizeof(int32_t)));
    HIPCHK(hipMalloc(&sp.flags, static_cast<size_t>(mp)));
    HIPCHK(hipMalloc(&sp.colmap, static_cast<size_t>(mp) * sizeof(int32_t)));
    HIPCHK(hipMalloc(&sp.pos, static_cast<size_t>(mp) * sizeof(int32_t)));
    sp.cap = static_cast<size_t>(mp);
  }
  if ((rc = rv_grow(sp.blk, sp.cap_blk, static_cast<size_t>(nblk) + 2))) return rc;
  if (!sp.nout_acc) HIPCHK(hipMalloc(&sp.nout_acc, 8 * sizeof(int32_t)));
  if (!sp.rec) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sp.rec), 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&sp.rec_dev), sp.rec, 0));
  }
  sp.rec->nS = -1;
  sp.rec->ncol = -1;
  sp.rec->n0 = -1;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const SliceView RV = row_view(h, s);
  hipLaunchKernelGGL(k_sub_begin, dim3(1), dim3(64), 0, s.stream, sp.nout_acc);
  hipLaunchKernelGGL(k_sub_colcount, dim3(static_cast<unsigned>(ceil_div(RV.ncg, 4))), dim3(256), 0, s.stream, RV, sp.cnt, mp);
  hipLaunchKernelGGL((k_sub_flags<V>), dim3(static_cast<unsigned>(nblk)), dim3(256), 0, s.stream, s.st + h->par, sp.cnt,
                     v.in_view[v.cur], m, mp, sp.flags, sp.blk, sp.nout_acc);
  hipLaunchKernelGGL(k_rv_scan, dim3(1), dim3(1024), 0, s.stream, sp.blk, nblk, &sp.rec_dev->nS);
  hipLaunchKernelGGL(k_rv_scatter, dim3(static_cast<unsigned>(nblk)), dim3(256), 0, s.stream, sp.flags, m, sp.blk, sp.colmap,
                     static_cast<int64_t>(sp.cap), sp.pos, mp);
  hipLaunchKernelGGL(k_sub_publish, dim3(1), dim3(64), 0, s.stream, sp.nout_acc, sp.rec_dev);
  HIPCHK(hipStreamSynchronize(s.stream));
  HIPCHK(hipGetLastError());
  std::atomic_thread_fence(std::memory_order_acquire);
  const int64_t nS = sp.rec->nS;
  static const bool host_timing = std::getenv("CLIPPER_HIP_HOST_TIMING") != nullptr;
  if (host_timing)
    std::fprintf(stderr, "[sub] view of %lld rows: S = %lld associations, N0 = %d, the largest count outside %d\n",
                 static_cast<long long>(v.nrows), static_cast<long long>(nS), sp.rec->n0, sp.rec->ncol);
  // worth a problem of its own: not much more than the view's rows, and far fewer than the full problem's columns
  if (nS < 64 || sp.rec->ncol < 0 || nS > 2 * v.nrows + 1024 || 3 * nS > m) return 0;
  // the child context(s): on the parent's device and stream (their fills, plans and launches are ordered with the
  // parent's by construction), filled with the invariant the parent's matrix was scored with
  auto build_child = [&](Ctx*& c, int storage) -> int {
    if (!c) {
      c = make_ctx(&s.device, 1, storage, 1, 0, false);
      if (!c) return CLIPPER_HIP_E_HIP;
      hipStreamDestroy(c->sh[0].stream);
      c->sh[0].stream = s.stream;
      c->parent = h;
    }
    if (int r2 = sub_stage(h, c, nS)) return r2;
    if (h->fill_kind == 1)
      return clipper_hip_affinity_euclidean_staged(c, h->fill_e.sigma, h->fill_e.epsilon, h->fill_e.mindist, h->fill_e.affinityeps);
    return clipper_hip_affinity_pointnormal_staged(c, h->fill_n.sigp, h->fill_n.epsp, h->fill_n.sign, h->fill_n.epsn,
                                                   h->fill_n.affinityeps);
  };
  const int storage = h->compressed ? (h->storage == CLIPPER_HIP_STORE_F64 ? CLIPPER_HIP_STORE_F64_CSC : CLIPPER_HIP_STORE_F32_CSC)
                                    : h->storage;
  // Mostly non-zero (the inliers of a registration problem are consistent with each other: M[S,S] IS the dense block the
  // slices' work list cuts by step range)? Then a dense fp32 store holds it in 4 bytes per element instead of 5.3 per
  // stored entry and its pass is k_gemv (wave-uniform multipliers, no gathers; the window from candidate tables:
  // k_sub_enter writes the pending one): 37 -> 30 us per pass at m = 100k, 236 -> 182 us at 300k (0.71 of the HBM peak).
  // How dense it will be is known before it is built: the selection summed the counts of the columns it took.
  static const bool dense_off = std::getenv("CLIPPER_HIP_SUB_DENSE") && std::atoi(std::getenv("CLIPPER_HIP_SUB_DENSE")) == 0;
  const double in_entries = static_cast<double>((static_cast<unsigned long long>(sp.rec->entries_hi) << 32) | sp.rec->entries_lo);
  const double density = in_entries / (static_cast<double>(nS) * static_cast<double>(std::max<int64_t>(1, v.nrows)));
  const bool dense = !dense_off && h->sub_mode != 2 && h->storage == CLIPPER_HIP_STORE_F32 && density >= 0.5 && nS >= 1024;
  Ctx* c = nullptr;
  if (dense) {
    rc = build_child(sp.ctx_dense, CLIPPER_HIP_STORE_F32);
    if (rc == 0 && sp.ctx_dense->has_matrix && !sp.ctx_dense->csc_valid) c = sp.ctx_dense;
    else if (rc != 0 && rc != CLIPPER_HIP_E_NOMEM) return rc;
    else (void)hipGetLastError();
  }
  if (c == nullptr) {
    if ((rc = build_child(sp.ctx, storage))) return rc;
    if (!sp.ctx->csc_valid) return 0;  // (a fill route without slices: not taken)
    c = sp.ctx;
  }
  sp.use = c;
  HIPCHK(hipSetDevice(s.device));
  sp.nS = nS;
  // N of the bound: the counts are over the VIEW's rows; on the sub-problem a row of S outside the view may come back
  // to life (it is inside S: fine) and then adds at most one entry to any column — |S \ R| on top, once and for all
  sp.ncol = static_cast<double>(std::max(1, sp.rec->ncol)) + static_cast<double>(std::max<int64_t>(0, nS - v.nrows));
  sp.ready = true;
  sp.build_ms += std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
  if (host_timing)
    std::fprintf(stderr, "[sub] ready: %lld associations, density %.2f -> %s (%.1f MB), prepared in %.2f ms\n",
                 static_cast<long long>(nS), density, c->csc_valid ? "slices" : "a dense fp32 store",
                 (c->csc_valid ? static_cast<double>(c->sh[0].s_bytes) : algorithmic_gemv_bytes(c)) * 1e-6, sp.build_ms);
  return 0;
}

int sub_prepare(Ctx* h) {
  int rc = 0;
  dispatch_window(h, [&](auto v) { rc = sub_prepare_v<decltype(v)::value>(h); });
  return rc;
}

int enqueue_iteration(Ctx* h, const SolverParams& prm);

// hold = 2: the decision on the full problem found that no column outside S can come back to life. The hold is lifted,
// a decide-only iteration turns the held decision into a prepared pass, the point and the state move over.
int sub_enter(Ctx* h, const SolverParams& prm) {
  SubProblem& sp = h->sub;
  if (!sp.ready || sp.active || !sp.use) return fail(CLIPPER_HIP_E_INTERNAL, "sub-problem: a hand-over nobody prepared");
  Shard& s = h->sh[0];
  Ctx* c = sp.use;
  Shard& cs = c->sh[0];
  HIPCHK(hipSetDevice(s.device));
  hipLaunchKernelGGL(k_sub_resume, dim3(1), dim3(64), 0, s.stream, s.st + h->par, s.shared);
  h->decide_only = true;  // (solve_args: a decide-only launch does not ask for the hand-over again)
  int rc = enqueue_iteration(h, prm);
  h->decide_only = false;
  if (rc) return rc;
  // the final u is written through the list of S: the rest is zero
  std::memset(h->u_pinned, 0, static_cast<size_t>(h->m) * sizeof(double));
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipLaunchKernelGGL(k_sub_enter, dim3(static_cast<unsigned>(ceil_div(c->mp, 256))), dim3(256), 0, s.stream, s.st + h->par, s.pt,
                     s.cab, h->mp, h->V, sp.colmap, sp.nS, cs.st, cs.shared, cs.pt, cs.cab, c->mp,
                     c->csc_valid ? static_cast<double*>(nullptr) : cs.X[0], prm.beta);
  c->par = 0;
  c->decide_only = false;
  c->rv_fresh = false;
  sp.active = true;
  sp.entries += 1;
  sp.launches_since_entry = 0;
  sp.passes_at_entry = h->mirror->n_passes;
  h->rv_stats.sub_entries += 1;
  return 0;
}

// hold = 3: the decision on the sub-problem left its pass prepared — a column outside S could come back to life under
// one of the pending candidates. The point and the state go back; the full problem's launches run that pass.
int sub_leave(Ctx* h) {
  SubProblem& sp = h->sub;
  if (!sp.active || !sp.use) return fail(CLIPPER_HIP_E_INTERNAL, "sub-problem: nothing to leave");
  Shard& s = h->sh[0];
  Ctx* c = sp.use;
  Shard& cs = c->sh[0];
  HIPCHK(hipSetDevice(s.device));
  hipLaunchKernelGGL(k_sub_leave, dim3(static_cast<unsigned>(ceil_div(h->mp, 256))), dim3(256), 0, s.stream, cs.st + c->par, cs.pt,
                     cs.cab, c->mp, h->V, sp.pos, s.rv.in_view[s.rv.cur], h->m, s.pt, s.cab, h->mp, sp.nout_acc);
  hipLaunchKernelGGL(k_sub_leave_state, dim3(1), dim3(256), 0, s.stream, cs.st + c->par, s.st + h->par, s.shared, sp.nout_acc);
  sp.active = false;
  sp.leaves += 1;
  if (sp.entries >= SUB_MAX_ENTRIES) sp.ready = false;
  return 0;
}

}  // namespace
