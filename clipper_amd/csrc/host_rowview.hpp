// host_rowview.hpp — the rectangular fill's launcher, and the ROW VIEW of the solver: when it pays to
// build one, how it is built between two iterations, what the launches are told about it.
// Part of clipper_hip.hip (one translation unit; included there, in order).
//
// Why: row r of every line-search candidate max(u + alpha g, 0) (clipper.cpp:235-236) is exactly zero
// unless u[r] > 0 or g[r] > 0 ("live", k_solver.hip.h), and the reference's iteration drives the
// outliers' rows out of the live set within a few steps (m = 10k, 95 % outliers: 36 % live during
// the first outer iteration, 5 % from the second on — 52 of 66 trials). A pass then needs
// M[live rows, :] only. The view is the slices of exactly that sub-matrix, written by the same tile
// kernel that serves column shards and fp64 values (k_affinity_rect); the device decides per pass
// whether the view may stand in for M (no live row outside it), so a view that is out of date costs
// time, never correctness.
#pragma once

namespace {

// ---- k_affinity_rect for the invariant the current matrix was scored with -------------------------
template <typename K>
void launch_rect_kernel(K kernel, int lds_bytes, dim3 grid, hipStream_t stream, const RectGeom& G,
                        const Shard& s, int64_t pstride, const int32_t* A0, const int32_t* A1,
                        const EuclidParams& e, const PointNormalParams& n, float E2, const SliceOut& O) {
  raise_dynamic_lds(reinterpret_cast<const void*>(kernel), s.device, lds_bytes);
  hipLaunchKernelGGL(kernel, grid, dim3(AT_WAVES * 64), lds_bytes, stream, G, s.P1, s.P2, s.P1f, s.P2f,
                     pstride, A0, A1, e, n, E2, O);
}

bool rect_fill_possible(const Ctx* h) {
  return (h->fill_kind == 1 && (h->staged_d == 2 || h->staged_d == 3)) ||
         (h->fill_kind == 2 && h->staged_d == 6);
}

// The slices of M[rows, this shard's columns] into the store O describes. rowmap == null: all rows.
// (col0, wcols) >= 0: other columns than the shard's own — [col0, col0 + wcols) — for the replica of a row view
// over ALL columns that every rank of a sharded solve keeps for the resident solver (the points are replicated).
int launch_rect(Ctx* h, Shard& s, const int32_t* rowmap, int64_t nrows, const SliceOut& O, int64_t col0, int64_t wcols) {
  if (!rect_fill_possible(h)) return fail(CLIPPER_HIP_E_STATE, "no built-in invariant is staged");
  const int64_t W = wcols > 0 ? wcols : h->W;
  RectGeom G;
  G.m = h->m;
  G.nrows = nrows;
  G.rowmap = rowmap;
  G.col0 = col0 >= 0 ? col0 : static_cast<int64_t>(s.slot) * h->W;
  G.ncols = std::max<int64_t>(0, std::min<int64_t>(W, h->m - G.col0));
  const int64_t nTr = ceil_div(nrows, AT);
  const int32_t* A0 = s.Adev;
  const int32_t* A1 = s.Adev + h->m;
  const int64_t ps = h->staged_pstride;
  // a thin view (fewer 128-wide tiles than two per CU): 64-wide tiles — see k_affinity_rect
  const bool thin = nTr * ceil_div(W, 128) <= 2 * static_cast<int64_t>(h->cus);
  dispatch_vt(h, [&](auto t) {
    using VT = decltype(t);
    auto run = [&](auto twc) {
      constexpr int TW = decltype(twc)::value;
      G.nTc = static_cast<int>(ceil_div(W, TW));
      const int64_t ntiles = nTr * G.nTc;
      constexpr int64_t PER_LAUNCH = int64_t(1) << 22;  // x 512 threads < 2^32 work-items per dispatch
      constexpr int L = rect_lds_bytes<VT, TW>();
      for (int64_t t0 = 0; t0 < ntiles; t0 += PER_LAUNCH) {
        G.tile0 = t0;
        const dim3 grid(static_cast<unsigned>(std::min<int64_t>(PER_LAUNCH, ntiles - t0)));
        if (h->fill_kind == 2)
          launch_rect_kernel(k_affinity_rect<3, true, VT, TW>, L, grid, s.stream, G, s, ps, A0, A1, h->fill_e, h->fill_n, h->fill_E2, O);
        else if (h->staged_d == 3)
          launch_rect_kernel(k_affinity_rect<3, false, VT, TW>, L, grid, s.stream, G, s, ps, A0, A1, h->fill_e, h->fill_n, h->fill_E2, O);
        else
          launch_rect_kernel(k_affinity_rect<2, false, VT, TW>, L, grid, s.stream, G, s, ps, A0, A1, h->fill_e, h->fill_n, h->fill_E2, O);
      }
    };
    if constexpr (rect_tw<VT>() == 64) {
      run(std::integral_constant<int, 64>{});
    } else {
      if (thin) run(std::integral_constant<int, 64>{});
      else run(std::integral_constant<int, 128>{});
    }
  });
  HIPCHK(hipGetLastError());
  return 0;
}

// The slices of M[rows, this shard's columns] filtered out of the shard's own slices of M
// (k_slice_filter_rows): for any matrix that lives in slices, whatever it was built from.
int launch_filter(Ctx* h, Shard& s, const int32_t* rowmap, const int32_t* viewpos, int64_t nrows, const SliceOut& O) {
  FilterGeom G;
  G.nrows = nrows;
  G.rowmap = rowmap;
  G.viewpos = viewpos;
  const int64_t nTr = ceil_div(nrows, AT);
  const SliceView M = slice_view(h, s);
  dispatch_vt(h, [&](auto t) {
    using VT = decltype(t);
    G.nTc = static_cast<int>(ceil_div(h->W, FILT_TW));
    const int64_t ntiles = nTr * G.nTc;
    constexpr int64_t PER_LAUNCH = int64_t(1) << 22;  // x 512 threads < 2^32 work-items per dispatch
    constexpr int L = filt_lds_bytes<VT>();
    raise_dynamic_lds(reinterpret_cast<const void*>(k_slice_filter_rows<VT>), s.device, L);
    for (int64_t t0 = 0; t0 < ntiles; t0 += PER_LAUNCH) {
      G.tile0 = t0;
      const dim3 grid(static_cast<unsigned>(std::min<int64_t>(PER_LAUNCH, ntiles - t0)));
      hipLaunchKernelGGL((k_slice_filter_rows<VT>), grid, dim3(AT_WAVES * 64), L, s.stream, M, G, O);
    }
  });
  HIPCHK(hipGetLastError());
  return 0;
}

// How a view is built. Scored again from the staged points (k_affinity_rect) where the matrix was scored
// from points: its cost falls with the rows, the filter always reads all of M (profiles/r03_view_by_filter.txt:
// the small views late in a solve are 2 x cheaper scored again, the large first one about the same).
// Filtered out of M's own slices (k_slice_filter_rows) where there are no points — matrices handed over
// with setMatrixData / setSparseMatrixData, custom invariants — which had no views before.
// CLIPPER_HIP_RV_BUILD = filter: always the filter; rectfill: the rectangular fill under the filter's cost
// model (the two builds must then give the same solve bit for bit, tests/test_gpu_rowview.py).
int rowview_build_env() {
  static const int mode = [] {
    const char* e = std::getenv("CLIPPER_HIP_RV_BUILD");
    if (!e) return 0;
    const std::string v(e);
    return v == "filter" ? 1 : (v == "rectfill" ? 2 : 0);
  }();
  return mode;
}
bool rowview_fill_by_filter(const Ctx* h) { return rowview_build_env() == 1 || !rect_fill_possible(h); }
bool rowview_cost_of_filter(const Ctx* h) { return rowview_build_env() != 0 || !rect_fill_possible(h); }

// ---- the row view -----------------------------------------------------------------------------------

int rvr_plan(Ctx* h, Shard& s, bool replica);  // host_rv_resident.hpp: does the view just built fit the resident solver?
// the control block of the resident launch that may follow the build in progress (null: none allocated yet, or the
// solve has used up its blocks — the launch then clears block 0 itself)
uint32_t* rvr_ctl_block(Ctx* h) {
  ViewResident& r = h->vres;
  return (r.ctl != nullptr && r.launches_this_solve < RVR_GIVEUP_SLOTS) ? r.ctl + 16 * r.launches_this_solve : nullptr;
}
bool rvr_candidate(const Ctx* h, int64_t nrows);
int rowview_put_descriptor(Ctx* h, Shard& s);

void rowview_drop(Ctx* h) {
  for (auto& s : h->sh) s.rv.valid = s.rv.full_valid = false;
  h->vres.ready = false;
}

void rowview_free(Shard& s) {
  auto fr = [](auto*& p) {
    if (p) hipFree(p);
    p = nullptr;
  };
  RowView& v = s.rv;
  fr(v.st.sSizes);
  fr(v.st.sLq);
  fr(v.st.sPre);
  fr(v.st.sBlk);
  fr(v.st.sdata);
  fr(v.st.swork);
  v.st = SliceStore{};
  fr(v.full.sSizes);
  fr(v.full.sLq);
  fr(v.full.sPre);
  fr(v.full.sBlk);
  fr(v.full.sdata);
  fr(v.full.swork);
  v.full = SliceStore{};
  v.full_valid = false;
  fr(v.rowmap[0]);
  fr(v.rowmap[1]);
  fr(v.in_view[0]);
  fr(v.in_view[1]);
  fr(v.blk);
  fr(v.viewpos);
  fr(v.desc);
  v.cap_rows = v.cap_flags = v.cap_blk = v.cap_pos = 0;
  v.valid = false;
  v.nrows = 0;
}

// a view can exist at all: slices with C == pattern(M); column shards: the
// bytes of all shards are known (the policy's cost model must be the same on every rank)
bool rowview_possible(const Ctx* h) {
  static const bool env_off = [] {
    const char* e = std::getenv("CLIPPER_HIP_ROW_VIEW");
    return e && std::atoi(e) == 0;
  }();
  return !env_off && h->rv_mode != 1 && h->csc_valid && !h->explicitC &&
         h->m >= RV_MIN_M &&
         (csc_single(h) || h->total_slice_bytes > 0.0);
}

// Column shards: the bytes all shards' slices hold, by one all-gather of a one-slot block at build time
// (each shard's count rides in the first element of its block) — every rank gets the same sum, so
// every rank's policy takes the same decisions.
int gather_slice_bytes(Ctx* h) {
  h->total_slice_bytes = 0.0;
  if (csc_single(h) || !h->csc_valid) return 0;
  if (h->multiproc && !h->comm && !h->xchg_fn) return 0;  // no exchange yet: this matrix is solved without views
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    const double v = static_cast<double>(s.s_bytes);
    HIPCHK(hipMemcpy(s.ab + static_cast<int64_t>(s.slot) * h->W, &v, sizeof(double), hipMemcpyHostToDevice));
  }
  if (int rc = exchange(h, 1)) return rc;
  if (int rc = sync_all(h)) return rc;
  Shard& s0 = h->sh[0];
  HIPCHK(hipSetDevice(s0.device));
  double total = 0.0;
  for (int p = 0; p < h->world; ++p) {
    double v = 0.0;
    HIPCHK(hipMemcpy(&v, s0.ab + static_cast<int64_t>(p) * h->W, sizeof(double), hipMemcpyDeviceToHost));
    total += v;
  }
  h->total_slice_bytes = total;
  return 0;
}

// The cost model the device-side policy (view_wanted, k_solver.hip.h) works with, for the matrix at
// hand. A pass that streams r of the m rows: what the launch costs besides the bytes (decision,
// prologue, the slowest workgroup's tail) + r rows' share of the slices at the rate the pass sustains.
// Building a view of r rows: drain + compaction + directory round trip + planning, and the rectangular
// fill (r * m pairs at the rate of the tile kernels, which have no mirror image to share).
ViewPolicy rowview_policy(const Ctx* h) {
  static const double scale_env = std::getenv("CLIPPER_HIP_RV_BUILD_SCALE") ? std::atof(std::getenv("CLIPPER_HIP_RV_BUILD_SCALE")) : 1.0;
  ViewPolicy p{};
  p.on = rowview_possible(h) ? 1 : 0;
  p.max_builds = RV_MAX_BUILDS;
  p.pass_fixed = 8e-6;
  // (column shards: a shard's share of the bytes all shards hold — the same figure on every rank)
  const double bytes = csc_single(h) ? static_cast<double>(h->sh[0].s_bytes)
                                     : h->total_slice_bytes / static_cast<double>(std::max(1, h->world));
  p.pass_per_row = bytes / static_cast<double>(h->m) / 3.3e12;
  if (rowview_cost_of_filter(h)) {  // one read of the shard's slices whatever the rows + the rows' slices written:
                                    // 1.1 TB/s over both, measured (profiles/r03_view_by_filter.txt)
    p.build_fixed = (100e-6 + bytes / 1.1e12) * scale_env;
    p.build_per_row = bytes / static_cast<double>(h->m) / 1.1e12 * scale_env;
  } else {
    // (a shard fills its own columns of the rows; fp64 values: 64-wide tiles, twice the image per pair — measured
    // twice the time per row: 313 ms for all 300k rows against 144 ms as a rectangular fp32 fill)
    p.build_fixed = 60e-6 * scale_env;
    p.build_per_row = static_cast<double>(h->W) * (h->esize() == 8 ? 9.0e-12 : 4.5e-12) * scale_env;
  }
  return p;
}

template <typename T>
int rv_grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap && p) return 0;
  if (p) hipFree(p);
  p = nullptr;
  cap = 0;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(need, 1) * sizeof(T)));
  cap = need;
  return 0;
}

// Builds the view from the state the NEXT iteration decides from (state copy h->par). The stream need not be
// idle: whatever is queued behind the hold only moves the state between its two copies, and the build's launches
// queue behind that. built = false: the live rows were too many to be worth it — the view in use, if any, stays
// as it is, unless its store was already overwritten by a speculative fill whose count turned out wrong (never
// seen; forced by CLIPPER_HIP_RV_TEST_MISCOUNT): then the build goes through with the real count whatever the
// cost model says, and only an EMPTY list leaves the shard without a view (the passes stream M: always correct).
template <int V>
int rowview_build_shard(Ctx* h, Shard& s, bool& built);

// every local shard builds its own view (its columns of the same rows: the row list is a function of the
// solver state, which is identical on every shard and rank)
template <int V>
int rowview_build_v(Ctx* h, bool& built) {
  built = false;
  bool first = true;
  for (auto& s : h->sh) {
    bool b = false;
    if (int rc = rowview_build_shard<V>(h, s, b)) return rc;
    if (first) built = b;
    else if (b != built) return fail(CLIPPER_HIP_E_HIP, "row view: the shards disagree");
    first = false;
  }
  if (built) {
    h->rv_stats.builds += 1;
    h->rv_stats.rows = h->sh[0].rv.nrows;
    h->rv_stats.bytes = static_cast<int64_t>(h->sh[0].rv.st.s_bytes);
  }
  return 0;
}

template <int V>
int rowview_build_shard(Ctx* h, Shard& s, bool& built) {
  built = false;
  RowView& v = s.rv;
  const auto t0 = std::chrono::high_resolution_clock::now();
  HIPCHK(hipSetDevice(s.device));
  const int64_t m = h->m, mp = h->mp;
  const int nblk = static_cast<int>(ceil_div(m, RV_BLK));
  int rc;
  {
    size_t c0 = v.cap_flags, c1 = v.cap_flags;
    if ((rc = rv_grow(v.in_view[0], c0, static_cast<size_t>(mp)))) return rc;
    if ((rc = rv_grow(v.in_view[1], c1, static_cast<size_t>(mp)))) return rc;
    v.cap_flags = static_cast<size_t>(mp);
    size_t r0 = v.cap_rows, r1 = v.cap_rows;
    if ((rc = rv_grow(v.rowmap[0], r0, static_cast<size_t>(mp)))) return rc;
    if ((rc = rv_grow(v.rowmap[1], r1, static_cast<size_t>(mp)))) return rc;
    v.cap_rows = static_cast<size_t>(mp);
    if ((rc = rv_grow(v.blk, v.cap_blk, static_cast<size_t>(nblk) + 2))) return rc;
    if ((rc = rv_grow(v.viewpos, v.cap_pos, static_cast<size_t>(mp)))) return rc;
  }
  if (!h->rv_count) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->rv_count), 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->rv_count_dev), h->rv_count, 0));
  }
  const int next = v.cur ^ 1;
  *h->rv_count = -1;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  if (mp <= RV_ONE_MAX) {  // flags, scan and scatter by one workgroup: one launch instead of three
    hipLaunchKernelGGL((k_rv_list_one<V>), dim3(1), dim3(1024), 0, s.stream, s.st + h->par, s.pt, mp, m, v.in_view[next],
                       v.rowmap[next], static_cast<int64_t>(v.cap_rows), v.viewpos, h->rv_count_dev);
  } else {
    hipLaunchKernelGGL((k_rv_flags<V>), dim3(static_cast<unsigned>(nblk)), dim3(256), 0, s.stream,
                       s.st + h->par, s.pt, mp, m, v.in_view[next], v.blk);
    hipLaunchKernelGGL(k_rv_scan, dim3(1), dim3(1024), 0, s.stream, v.blk, nblk, h->rv_count_dev);
    hipLaunchKernelGGL(k_rv_scatter, dim3(static_cast<unsigned>(nblk)), dim3(256), 0, s.stream,
                       v.in_view[next], m, v.blk, v.rowmap[next], static_cast<int64_t>(v.cap_rows), v.viewpos, mp);
  }
  static const bool host_timing = std::getenv("CLIPPER_HIP_HOST_TIMING") != nullptr;
  auto lap = [&](const char* what) {
    if (host_timing)
      std::fprintf(stderr, "[view] %s %.1f us\n", what,
                   std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count());
  };
  // The device asked (view_wanted, k_solver.hip.h) with the live count its tail summed, and the list being
  // built is that very set: its length is known before the three launches above have run. The fill is
  // therefore sized and queued right behind them — ONE wait for the whole build instead of one for the
  // count and one for the fill — and the count is only checked afterwards. Should it ever differ (it never
  // has: then the same cost model is evaluated again with the rows the view would really have), the build
  // is repeated with the real length.
  const int64_t asked = h->mirror->hold_nlive;
  int64_t nrows = asked;
  static const bool test_miscount = std::getenv("CLIPPER_HIP_RV_TEST_MISCOUNT") != nullptr;  // (test knob: the first fill
  if (test_miscount && asked > 1) nrows = asked - 1;                                          // is sized one row short)
  bool counted = false;
  bool store_gone = false;  // the store of the view in use was overwritten by this build
  bool early_done = false;  // the hold was lifted and the decide-only iteration queued behind the first fill (below)
  const bool early_ok = h->enqueue_one != nullptr && h->sh.size() == 1 && !h->multiproc;
  h->early_decide_done = false;
  if (asked <= 0 || asked > m) {
    HIPCHK(hipStreamSynchronize(s.stream));
    nrows = *h->rv_count;
    counted = true;
    lap("row list");
  }
  const double rows_now = v.valid ? static_cast<double>(v.nrows) : static_cast<double>(m);
  const ViewPolicy pol = h->rvp;
  const double horizon = std::max<double>(12.0, static_cast<double>(h->mirror->iters + 1));
  for (int attempt = 0;; ++attempt) {
    if (counted) {
      if (nrows < 0) return fail(CLIPPER_HIP_E_HIP, "row view: the row count did not arrive");
      const bool as_asked = nrows > 0 && nrows == asked;
      if (!as_asked &&
          (nrows == 0 ||
           (!store_gone && (static_cast<double>(nrows) > RV_ROWS_RATIO * rows_now ||
                            RV_GAIN_MARGIN * (pol.build_fixed + pol.build_per_row * static_cast<double>(nrows)) >=
                                horizon * (rows_now - static_cast<double>(nrows)) * pol.pass_per_row)))) {
        // (a refusal AFTER the early hand-over — the hold is lifted and the decide-only iteration is in the
        // stream already: the caller must count it, and the state has moved on: no second resume)
        if (!early_done)
          hipLaunchKernelGGL(k_rv_resume, dim3(1), dim3(64), 0, s.stream, s.st + h->par, s.shared, static_cast<int>(nrows),
                             static_cast<uint32_t*>(nullptr));
        h->early_decide_done = early_done;
        h->rv_stats.build_ms +=
            std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        return 0;
      }
    }
    v.valid = false;  // its store is about to be overwritten
    v.full_valid = false;
    store_gone = true;
    SliceOut O{};
    if ((rc = emit_prepare(h, s, v.st, nrows, O))) return rc;
    if (rowview_fill_by_filter(h)) rc = launch_filter(h, s, v.rowmap[next], v.viewpos, nrows, O);
    else rc = launch_rect(h, s, v.rowmap[next], nrows, O);
    if (rc) return rc;
    if ((rc = emit_enqueue(h, s, v.st))) return rc;
    // EARLY HAND-OVER (round 5). A view the resident solver may take, sized from the count the device asked with:
    // everything the decide-only iteration needs of the view is known before the fill has run — the row flags and
    // the list (queued above), the rows' number, the store's arrays — so the descriptor, the lift of the hold and
    // the decide-only iteration itself are queued HERE, behind the fill, and run on the device while the host
    // sleeps in its one wait and plans the units: the resident launch is the only thing left to queue afterwards
    // (round 4: all of it after the wait — 25 us of launches and small kernels with the device idle). Should the
    // fill turn out unusable (an arena that overflowed: the first view of a size; a count that differed: never
    // seen) the view is built again below as before — the state has moved on to a prepared pass by then, which is
    // what the launches after ANY build start from (iteration_head), on the view if there is one, on M if not.
    if (early_ok && attempt == 0 && !counted && rvr_candidate(h, nrows)) {
      const int cur0 = v.cur;
      const int64_t nrows0 = v.nrows;
      v.cur = next;
      v.nrows = nrows;
      v.valid = true;  // (provisional: the launches queued right here read the flags, the list and the rows' number)
      v.st.s_nwork = 0;
      if ((rc = rowview_put_descriptor(h, s))) return rc;
      hipLaunchKernelGGL(k_rv_resume, dim3(1), dim3(64), 0, s.stream, s.st + h->par, s.shared, 0, rvr_ctl_block(h));
      h->rv_fresh = true;
      h->decide_only = true;
      rc = h->enqueue_one ? h->enqueue_one() : CLIPPER_HIP_E_INTERNAL;
      h->decide_only = false;
      if (rc) return rc;
      early_done = true;
      v.valid = false;
      v.cur = cur0;
      v.nrows = nrows0;
    }
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipGetLastError());
    lap("filled");
    if (!counted) {
      counted = true;
      if (*h->rv_count != nrows) {  // (never seen) the list is not what the device counted: again, with its real length
        nrows = *h->rv_count;
        continue;
      }
    }
    bool again = false;
    // (one shard, a view small enough for the resident solver: the work list of the STREAMED pass on it — what the
    // launches behind the resident one fall back to — is planned while that launch runs: rowview_finish_plan)
    const bool defer = rvr_candidate(h, nrows);
    if ((rc = emit_check(h, s, v.st, false, again, !defer))) return rc;
    lap("planned");
    if (!again) {
      v.plan_pending = defer;
      break;
    }
    if (attempt >= 3) return fail(CLIPPER_HIP_E_HIP, "row view: the build keeps overflowing");
  }
  v.cur = next;
  v.nrows = nrows;
  v.valid = true;
  built = true;
  if ((rc = rvr_plan(h, s, false))) return rc;  // (the view's directory is still in the pinned staging)
  lap("resident plan");
  if (v.plan_pending && !h->vres.ready) {  // the resident solver does not take it after all
    if ((rc = slices_plan(h, s, v.st, false))) return rc;
    v.plan_pending = false;
  }
  if ((rc = rowview_put_descriptor(h, s))) return rc;  // (no work list yet while its plan is pending: no item for anybody)
  if (!early_done) hipLaunchKernelGGL(k_rv_resume, dim3(1), dim3(64), 0, s.stream, s.st + h->par, s.shared, 0, rvr_ctl_block(h));
  h->early_decide_done = early_done;
  h->rv_stats.build_ms +=
      std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
  return 0;
}

// the descriptor a pass on the view reads, to the device
int rowview_put_descriptor(Ctx* h, Shard& s) {
  RowView& v = s.rv;
  if (!v.desc) HIPCHK(hipMalloc(reinterpret_cast<void**>(&v.desc), sizeof(SliceView)));
  if (!h->rv_desc_host) {  // two staging slots, used in turn: a view whose work list is planned behind the resident
                           // launch sends a second descriptor while the copy of the first may still be queued
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->rv_desc_host), 2 * sizeof(SliceView), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->rv_desc_host_dev), h->rv_desc_host, 0));
  }
  const int slot = (h->rv_desc_slot ^= 1);
  h->rv_desc_host[slot] = row_view(h, s);
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipLaunchKernelGGL(k_copy_words, dim3(1), dim3(64), 0, s.stream, reinterpret_cast<const uint4*>(h->rv_desc_host_dev + slot),
                     reinterpret_cast<uint4*>(v.desc), static_cast<int64_t>(sizeof(SliceView) / 16));
  // the pinned staging slot is the context's: the copy has to be through before the next shard of an
  // in-process group writes ITS descriptor there
  if (h->sh.size() > 1) HIPCHK(hipStreamSynchronize(s.stream));
  return 0;
}

// The work list of the streamed pass on a view whose plan was put off (see rowview_build_shard): queued behind
// the resident launch, in front of the streaming launches that may need it.
int rowview_finish_plan(Ctx* h) {
  for (auto& s : h->sh) {
    RowView& v = s.rv;
    if (!v.valid || !v.plan_pending) continue;
    HIPCHK(hipSetDevice(s.device));
    // (the view's directory words are still in the pinned staging: nothing was built since)
    if (int rc = slices_plan(h, s, v.st, false)) return rc;
    v.plan_pending = false;
    if (int rc = rowview_put_descriptor(h, s)) return rc;
  }
  return 0;
}

int rowview_build(Ctx* h, bool& built) {
  int rc = 0;
  dispatch_window(h, [&](auto v) { rc = rowview_build_v<decltype(v)::value>(h, built); });
  return rc;
}

}  // namespace
