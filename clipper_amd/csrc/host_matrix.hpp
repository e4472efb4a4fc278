// host_matrix.hpp — the compressed copy of M (build, plan, expand), the affinity driver, gathers
// Part of clipper_hip.hip (one translation unit; included there, in order).
#pragma once

namespace {

// ---- the compressed storage (CLIPPER_HIP_STORE_F32_CSC / _F64_CSC): groups -> slices ----------
bool csc_applies(const Ctx* h) { return csc_possible(h) && !h->explicitC; }
constexpr int SL_H = 1;  // sub-blocks per chunk (R = 256 rows): see k_slices.hip.h / DESIGN.md

SliceView slice_view(const Ctx* h, const Shard& s) {
  SliceView M;
  M.data = s.sdata;
  M.Pre = s.sPre;
  M.work = s.swork;
  M.nchunks = s.s_nchunks;
  M.ncg = s.s_ncg;
  M.nwork = s.s_nwork;
  M.rowmap = nullptr;
  M.nrows = h->m;
  M.pad = 0;
  return M;
}

// the row view of this shard's columns (data == null: none in use)
SliceView row_view(const Ctx* h, const Shard& s) {
  SliceView R{};
  if (!s.rv.valid || !h->csc_valid) return R;
  R.data = s.rv.st.sdata;
  R.Pre = s.rv.st.sPre;
  R.work = s.rv.st.swork;
  R.nchunks = s.rv.st.s_nchunks;
  R.ncg = s.rv.st.s_ncg;
  R.nwork = s.rv.st.s_nwork;
  R.rowmap = s.rv.rowmap[s.rv.cur];
  R.nrows = s.rv.nrows;
  return R;
}

// the replica of the view over ALL columns (column shards; data == null: none)
SliceView row_view_full(const Ctx* h, const Shard& s) {
  SliceView R{};
  if (!s.rv.valid || !s.rv.full_valid || !h->csc_valid) return R;
  R.data = s.rv.full.sdata;
  R.Pre = s.rv.full.sPre;
  R.work = nullptr;
  R.nchunks = s.rv.full.s_nchunks;
  R.ncg = s.rv.full.s_ncg;
  R.nwork = 0;
  R.rowmap = s.rv.rowmap[s.rv.cur];
  R.nrows = s.rv.nrows;
  return R;
}

// calls f(value type tag) for the storage's element type
template <typename F>
void dispatch_vt(const Ctx* h, F&& f) {
  if (h->storage == CLIPPER_HIP_STORE_F64) f(double{});
  else f(float{});
}

// The dense store of every local shard, allocated if it is not; with `from_csc` its content is
// materialised from the slices when they are all there is.
int ensure_dense(Ctx* h, bool from_csc) {
  for (auto& s : h->sh) {
    if (s.S) continue;
    HIPCHK(hipSetDevice(s.device));
    if (hipMalloc(&s.S, s.bytes_S) != hipSuccess) {
      s.S = nullptr;
      (void)hipGetLastError();
      return fail(CLIPPER_HIP_E_NOMEM, "dense store of %zu bytes (this call needs one) does not fit",
                  s.bytes_S);
    }
    if (from_csc && h->csc_valid) {
      const int64_t nsl = static_cast<int64_t>(s.s_ncg) * s.s_nchunks;
      dim3 grid(static_cast<unsigned>(ceil_div(nsl, 4))), block(256);
      dispatch_vt(h, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL((k_slice_expand<T, SL_H, T>), grid, block, 0, s.stream, slice_view(h, s),
                           static_cast<T*>(s.S), h->W, h->m);
      });
      HIPCHK(hipStreamSynchronize(s.stream));
    }
  }
  return 0;
}

void drop_dense(Ctx* h) {
  for (auto& s : h->sh) {
    if (!s.S) continue;
    hipSetDevice(s.device);
    hipFree(s.S);
    s.S = nullptr;
  }
}

template <typename T>
int grow_dev(T*& p, size_t& cap, size_t need, size_t elem = sizeof(T)) {
  if (need <= cap && p) return 0;
  if (p) hipFree(p);
  p = nullptr;
  cap = 0;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(need, 1) * elem));
  cap = need;
  return 0;
}

// the per-slice arrays (directory, sizes, costs) of the store `st` (this shard's columns x `nrows`
// rows) and the pinned staging of this build
// (wcols: the store's columns, a multiple of 64 — the shard's pitch W unless the store is a REPLICA of a row view
// over all columns, host_rowview.hpp)
int slices_arrays(Ctx* h, Shard& sh, SliceStore& s, int64_t nrows, int64_t wcols = -1) {
  HIPCHK(hipSetDevice(sh.device));
  if (!sh.cctl) HIPCHK(hipMalloc(&sh.cctl, CSC_ARENAS * sizeof(CscBuildCtl)));
  s.s_ncg = static_cast<int>((wcols > 0 ? wcols : h->W) / SL_W);
  s.s_nchunks = static_cast<int>(ceil_div(nrows, SL_SUB * SL_H));
  const size_t nsl = static_cast<size_t>(s.s_ncg) * s.s_nchunks;
  if (nsl > s.scap_slices) {
    for (void** p : {reinterpret_cast<void**>(&s.sSizes), reinterpret_cast<void**>(&s.sLq),
                     reinterpret_cast<void**>(&s.sPre), reinterpret_cast<void**>(&s.sBlk)}) {
      if (*p) hipFree(*p);
      *p = nullptr;
    }
    s.scap_slices = 0;
    HIPCHK(hipMalloc(&s.sSizes, nsl * sizeof(uint32_t)));
    // (room behind the directory words for the arena counters of a direct emission: one read-back)
    HIPCHK(hipMalloc(&s.sLq, round_up(nsl, 32) * sizeof(uint32_t) + CSC_ARENAS * sizeof(CscBuildCtl)));
    HIPCHK(hipMalloc(&s.sPre, nsl * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&s.sBlk, (ceil_div(nsl, SCAN_BLK) + 4) * sizeof(uint64_t)));
    s.scap_slices = nsl;
  }
  if (nsl > h->csc_hcap_slices) {
    if (h->csc_hLq) hipHostFree(h->csc_hLq);
    h->csc_hLq = nullptr;
    h->csc_hcap_slices = 0;
    HIPCHK(hipHostMalloc(&h->csc_hLq, round_up(nsl, 32) * sizeof(uint32_t) + CSC_ARENAS * sizeof(CscBuildCtl),
                         hipHostMallocDefault));
    h->csc_hcap_slices = nsl;
  }
  if (!h->csc_hctl)
    HIPCHK(hipHostMalloc(&h->csc_hctl, 2 * CSC_ARENAS * sizeof(CscBuildCtl), hipHostMallocDefault));
  if (!h->csc_htotal) HIPCHK(hipHostMalloc(&h->csc_htotal, 2 * sizeof(uint64_t), hipHostMallocDefault));
  return 0;
}
int slices_arrays(Ctx* h, Shard& s) { return slices_arrays(h, s, s, h->m); }

// Before a fill that writes the slices itself (k_affinity_sym): the arenas of the slice store
// reset. out.Pre == null: compressed storage not in use.
int emit_prepare(Ctx* h, Shard& sh, SliceStore& s, int64_t nrows, SliceOut& out, int64_t wcols = -1) {
  out = SliceOut{};
  if (!csc_applies(h)) return 0;
  if (int rc = slices_arrays(h, sh, s, nrows, wcols)) return rc;
  const size_t units = s.scap_bytes >= SL_TAILPAD ? (s.scap_bytes - SL_TAILPAD) / 16 : 0;
  CscBuildCtl* init = h->csc_hctl + CSC_ARENAS;  // second half: what the device starts from
  for (int k = 0; k < CSC_ARENAS; ++k) {
    init[k].cursor = 0;
    init[k].capacity = units / CSC_ARENAS;
    init[k].origin = static_cast<unsigned long long>(k) * (units / CSC_ARENAS);
    init[k].overflow = 0;
  }
  // the counters live right behind this build's directory words: both come back in ONE copy
  CscBuildCtl* ectl = reinterpret_cast<CscBuildCtl*>(
      s.sLq + round_up(static_cast<int64_t>(s.s_ncg) * s.s_nchunks, 32));
  void* idev = nullptr;
  if (hipHostGetDevicePointer(&idev, init, 0) == hipSuccess && idev) {  // 2 KB: a launch beats a DMA copy
    hipLaunchKernelGGL(k_copy_words, dim3(1), dim3(256), 0, sh.stream, reinterpret_cast<const uint4*>(idev),
                       reinterpret_cast<uint4*>(ectl), static_cast<int64_t>(CSC_ARENAS * sizeof(CscBuildCtl) / 16));
  } else {
    (void)hipGetLastError();
    HIPCHK(hipMemcpyAsync(ectl, init, CSC_ARENAS * sizeof(CscBuildCtl), hipMemcpyHostToDevice, sh.stream));
  }
  out.Pre = s.sPre;
  out.Lq = s.sLq;
  out.data = s.sdata;
  out.ctl = ectl;
  out.nchunks = s.s_nchunks;
  out.ncg = s.s_ncg;
  out.stamps = h->stamps_dev ? h->stamps_dev + 8192 : nullptr;
  return 0;
}
int emit_prepare(Ctx* h, Shard& s, SliceOut& out) {
  h->csc_valid = false;
  h->csc_emitted = false;
  return emit_prepare(h, s, s, h->m, out);
}

// After such a fill: the counters to pinned host memory; emit_check() reads them once the stream
// was synchronised, grows the arena if a slice did not fit (`again`), else plans the passes.
int emit_enqueue(Ctx* h, Shard& sh, SliceStore& s) {
  HIPCHK(hipSetDevice(sh.device));
  const size_t nsl8 = static_cast<size_t>(round_up(static_cast<int64_t>(s.s_ncg) * s.s_nchunks, 32));
  const size_t bytes = nsl8 * sizeof(uint32_t) + CSC_ARENAS * sizeof(CscBuildCtl);
  void* hdev = nullptr;
  if (bytes <= (1u << 20) && hipHostGetDevicePointer(&hdev, h->csc_hLq, 0) == hipSuccess && hdev) {
    const int64_t n16 = static_cast<int64_t>(bytes / 16);  // (both terms are multiples of 32 bytes)
    hipLaunchKernelGGL(k_copy_words, dim3(static_cast<unsigned>(std::min<int64_t>(ceil_div(n16, 256), 64))),
                       dim3(256), 0, sh.stream, reinterpret_cast<const uint4*>(s.sLq),
                       reinterpret_cast<uint4*>(hdev), n16);
  } else {
    (void)hipGetLastError();
    HIPCHK(hipMemcpyAsync(h->csc_hLq, s.sLq, bytes, hipMemcpyDeviceToHost, sh.stream));
  }
  return 0;
}
int emit_enqueue(Ctx* h, Shard& s) { return emit_enqueue(h, s, s); }

int slices_plan(Ctx* h, Shard& sh, SliceStore& s, bool whole);
int slices_plan(Ctx* h, Shard& s) { return slices_plan(h, s, s, true); }
int resident_plan(Ctx* h, Shard& s);

int emit_check(Ctx* h, Shard& sh, SliceStore& s, bool whole, bool& again, bool plan = true) {
  again = false;
  HIPCHK(hipSetDevice(sh.device));
  bool over = false;
  size_t worst = 0;
  uint64_t sum = 0;
  const CscBuildCtl* rb = reinterpret_cast<const CscBuildCtl*>(
      h->csc_hLq + round_up(static_cast<int64_t>(s.s_ncg) * s.s_nchunks, 32));
  for (int k = 0; k < CSC_ARENAS; ++k) {
    over = over || rb[k].overflow != 0;
    worst = std::max(worst, static_cast<size_t>(rb[k].cursor));
    sum += rb[k].cursor;
  }
  if (over) {
    const size_t need = worst * CSC_ARENAS;  // every arena as large as the fullest one
    if (s.sdata) hipFree(s.sdata);
    s.sdata = nullptr;
    s.scap_bytes = 0;
    const size_t units = (need + need / 8 + 256 * CSC_ARENAS) / CSC_ARENAS * CSC_ARENAS;
    const size_t cap = units * 16 + SL_TAILPAD;
    HIPCHK(hipMalloc(&s.sdata, cap));
    HIPCHK(hipMemsetAsync(s.sdata + units * 16, 0, SL_TAILPAD, sh.stream));
    s.scap_bytes = cap;
    again = true;
    return 0;
  }
  s.s_bytes = sum * 16;
  if (!plan) {  // (a row view the resident solver takes: its work list is planned behind that launch, host_rowview.hpp)
    s.s_nwork = 0;
    return 0;
  }
  return slices_plan(h, sh, s, whole);
}
int emit_check(Ctx* h, Shard& s, bool& again) { return emit_check(h, s, s, true, again); }

// Before a build through groups: the group directory, the arenas' cursors reset, the per-slice arrays. Returns
// what a kernel that emits groups needs; out.Goff == null: compressed storage not in use.
template <typename VT>
int groups_prepare(Ctx* h, Shard& s, GroupOut<VT>& out) {
  out = GroupOut<VT>{};
  h->csc_valid = false;
  h->csc_emitted = false;
  if (!csc_applies(h)) return 0;
  HIPCHK(hipSetDevice(s.device));
  h->csc_nblocks = static_cast<int>(ceil_div(h->m, GR_RB));
  h->csc_nstrips = static_cast<int>(ceil_div(h->W, GR_CW));
  const size_t G = static_cast<size_t>(h->csc_nstrips) * static_cast<size_t>(h->csc_nblocks);
  if (G > s.gcap_groups) {
    if (s.gOff) hipFree(s.gOff);
    if (s.gPre) hipFree(s.gPre);
    s.gOff = nullptr;
    s.gPre = nullptr;
    s.gcap_groups = 0;
    HIPCHK(hipMalloc(&s.gOff, G * GR_OFFS * sizeof(uint16_t)));
    HIPCHK(hipMalloc(&s.gPre, G * sizeof(uint64_t)));
    s.gcap_groups = G;
  }
  if (int rc = slices_arrays(h, s)) return rc;
  CscBuildCtl* init = h->csc_hctl + CSC_ARENAS;  // second half: what the device starts from
  for (int k = 0; k < CSC_ARENAS; ++k) {
    init[k].cursor = 0;
    init[k].capacity = s.gcap_units / CSC_ARENAS;
    init[k].origin = static_cast<unsigned long long>(k) * (s.gcap_units / CSC_ARENAS);
    init[k].overflow = 0;
  }
  HIPCHK(hipMemcpyAsync(s.cctl, init, CSC_ARENAS * sizeof(CscBuildCtl), hipMemcpyHostToDevice,
                        s.stream));
  out.Goff = s.gOff;
  out.Gpre = s.gPre;
  out.vals = static_cast<VT*>(s.gvals);
  out.rows = s.grows;
  out.ctl = s.cctl;
  out.nblocks = h->csc_nblocks;
  return 0;
}

// count -> scan -> pack (the pack returns at once if the groups overflowed or the slice arena is
// too small) -> copies of the counters to pinned host memory; slices_check() reads them once
// the stream was synchronised.
template <typename VT, typename Source>
int slices_enqueue(Ctx* h, Shard& s, const Source& src, const CscBuildCtl* ctl) {
  HIPCHK(hipSetDevice(s.device));
  const int64_t nsl = static_cast<int64_t>(s.s_ncg) * s.s_nchunks;
  const int64_t nblk = ceil_div(nsl, SCAN_BLK);
  dim3 g4(static_cast<unsigned>(ceil_div(nsl, 4))), b256(256);
  hipLaunchKernelGGL((k_slice_count<VT, SL_H, Source>), g4, b256, 0, s.stream, src, s.s_ncg,
                     s.s_nchunks, s.sSizes, s.sLq, ctl, s.sBlk + nblk + 1);
  if (nsl <= 32768) {
    hipLaunchKernelGGL(k_slice_scan_small, dim3(1), dim3(1024), 0, s.stream, s.sSizes, nsl, s.sPre,
                       s.sBlk + nblk);
  } else {
    hipLaunchKernelGGL(k_slice_scan_local, dim3(static_cast<unsigned>(nblk)), b256, 0, s.stream,
                       s.sSizes, nsl, s.sPre, s.sBlk);
    hipLaunchKernelGGL(k_slice_scan_blocks, dim3(1), b256, 0, s.stream, s.sBlk, nblk);
    hipLaunchKernelGGL(k_slice_scan_add, dim3(static_cast<unsigned>(nblk)), b256, 0, s.stream, s.sPre,
                       nsl, s.sBlk);
  }
  const uint64_t cap_units = s.scap_bytes >= SL_TAILPAD ? (s.scap_bytes - SL_TAILPAD) / 16 : 0;
  int heavy_only = 0;
  if constexpr (std::is_same<Source, GroupSource<VT>>::value && SL_H == 1) {
    // the common case through LDS (coalesced), only the slices of dense blocks by per-lane gathers
    hipLaunchKernelGGL((k_slice_pack_staged<VT>), g4, b256, 0, s.stream, src, s.s_ncg, s.s_nchunks,
                       s.sPre, s.sdata, s.sBlk + nblk, cap_units);
    heavy_only = 1;
  }
  hipLaunchKernelGGL((k_slice_pack<VT, SL_H, Source>),
                     dim3(static_cast<unsigned>(std::min<int64_t>(nsl, 1 << 20))),
                     dim3(SL_PACKW * 64), 0, s.stream, src, s.s_ncg, s.s_nchunks, s.sPre, s.sdata,
                     s.sBlk + nblk, cap_units, s.sLq, heavy_only);
  if (ctl)
    HIPCHK(hipMemcpyAsync(h->csc_hctl, s.cctl, CSC_ARENAS * sizeof(CscBuildCtl),
                          hipMemcpyDeviceToHost, s.stream));
  HIPCHK(hipMemcpyAsync(h->csc_htotal, s.sBlk + nblk, sizeof(uint64_t), hipMemcpyDeviceToHost,
                        s.stream));
  HIPCHK(hipMemcpyAsync(h->csc_hLq, s.sLq, static_cast<size_t>(nsl) * sizeof(uint32_t),
                        hipMemcpyDeviceToHost, s.stream));
  return 0;
}

template <typename VT>
GroupSource<VT> group_source(const Ctx* h, const Shard& s) {
  GroupSource<VT> g{};
  g.Goff = s.gOff;
  g.Gpre = s.gPre;
  g.vals = static_cast<const VT*>(s.gvals);
  g.rows = s.grows;
  g.nblocks = h->csc_nblocks;
  g.ncols = static_cast<int64_t>(h->csc_nstrips) * GR_CW;
  return g;
}

// What a pass's workgroups do (SliceWork), planned per matrix from the slices' costs: per strip
// of SL_NW column groups, runs of chunks of about equal cost (cost of a chunk: its slowest
// slice — the waves of a workgroup meet at every chunk — plus a constant for the staging); a
// chunk that alone exceeds the target is split by step range. Most expensive first.
int slices_plan(Ctx* h, Shard& sh, SliceStore& s, bool whole) {
  static const double target_env = std::getenv("CLIPPER_HIP_CSC_WGS") ? std::max(1.0, std::atof(std::getenv("CLIPPER_HIP_CSC_WGS"))) : 0.0;
  static_assert(sizeof(clipper_plan::Work) == sizeof(SliceWork), "the planner's work item is the kernel's");
  static thread_local clipper_plan::PassPlan plan;  // (buffers kept from build to build)
  clipper_plan::plan_pass(h->csc_hLq, s.s_ncg, s.s_nchunks, clipper_plan::PassConsts{SL_NW, SL_SO, SL_OCC, whole ? 12 : 8},
                          h->cus, target_env, 2.0, plan);
  s.s_entries = plan.entries;
  const int nslots = plan.nslots;
  const size_t nw = plan.work.size();
  if (nw > h->csc_hcap_work) {
    if (h->csc_hwork) hipHostFree(h->csc_hwork);
    h->csc_hwork = nullptr;
    h->csc_hcap_work = 0;
    HIPCHK(hipHostMalloc(&h->csc_hwork, (nw + 1024) * sizeof(SliceWork), hipHostMallocDefault));
    h->csc_hcap_work = nw + 1024;
  }
  std::memcpy(h->csc_hwork, plan.work.data(), nw * sizeof(SliceWork));
  HIPCHK(hipSetDevice(sh.device));
  int rc = grow_dev(s.swork, s.scap_work, nw);
  if (rc) return rc;
  {
    void* wdev = nullptr;
    const size_t bytes = nw * sizeof(SliceWork);  // 32 bytes each
    if (bytes <= (1u << 20) && hipHostGetDevicePointer(&wdev, h->csc_hwork, 0) == hipSuccess && wdev) {
      hipLaunchKernelGGL(k_copy_words, dim3(static_cast<unsigned>(std::min<size_t>(ceil_div(bytes / 16, 256), 64))),
                         dim3(256), 0, sh.stream, reinterpret_cast<const uint4*>(wdev),
                         reinterpret_cast<uint4*>(s.swork), static_cast<int64_t>(bytes / 16));
    } else {
      (void)hipGetLastError();
      HIPCHK(hipMemcpyAsync(s.swork, h->csc_hwork, bytes, hipMemcpyHostToDevice, sh.stream));
    }
    // the pinned staging is shared by every store and shard of the context: the copy has to be
    // through before the next plan overwrites it (one shard's own builds are ordered by its stream
    // and by the wait that precedes every plan)
    if (h->sh.size() > 1) HIPCHK(hipStreamSynchronize(sh.stream));
  }
  const size_t NSLOT = static_cast<size_t>(nslot(h->V));
  if (static_cast<size_t>(nslots) > sh.part_tiles) {
    HIPCHK(hipStreamSynchronize(sh.stream));
    HIPCHK(hipFree(sh.part));
    sh.part = nullptr;
    sh.part_tiles = static_cast<size_t>(nslots) + 8;
    HIPCHK(hipMalloc(&sh.part, sh.part_tiles * NSLOT * static_cast<size_t>(h->W) * sizeof(double)));
  }
  s.s_nwork = static_cast<int>(nw);
  s.s_nslots = nslots;
  return whole ? resident_plan(h, sh) : 0;  // small problems: the whole solve as one launch
}

// After the stream was synchronised: did everything fit? If not (always the case for the first
// matrix of a size) the buffers are grown and `again` is set — the caller repeats the step that
// produces the groups; otherwise the work list is planned and the slices are valid.
template <typename VT>
int slices_check(Ctx* h, Shard& s, bool with_groups, bool& again) {
  again = false;
  if (!csc_applies(h)) return 0;
  HIPCHK(hipSetDevice(s.device));
  if (with_groups) {
    bool over = false;
    size_t worst = 0;
    for (int k = 0; k < CSC_ARENAS; ++k) {
      over = over || h->csc_hctl[k].overflow != 0;
      worst = std::max(worst, static_cast<size_t>(h->csc_hctl[k].cursor));
    }
    if (over) {
      const size_t need = worst * CSC_ARENAS;  // every arena as large as the fullest one
      if (s.gvals) hipFree(s.gvals);
      if (s.grows) hipFree(s.grows);
      s.gvals = nullptr;
      s.grows = nullptr;
      s.gcap_units = 0;
      const size_t units = (need + need / 8 + 64 * CSC_ARENAS) / CSC_ARENAS * CSC_ARENAS;
      HIPCHK(hipMalloc(&s.gvals, units * 4 * sizeof(VT)));
      HIPCHK(hipMalloc(&s.grows, units * 4));
      s.gcap_units = units;
      again = true;
    }
  }
  const uint64_t bytes = h->csc_htotal[0] * 16;
  if (bytes + SL_TAILPAD > s.scap_bytes) {
    if (s.sdata) hipFree(s.sdata);
    s.sdata = nullptr;
    s.scap_bytes = 0;
    const size_t cap = static_cast<size_t>(bytes) + static_cast<size_t>(bytes) / 16 + SL_TAILPAD + 4096;
    HIPCHK(hipMalloc(&s.sdata, cap));
    HIPCHK(hipMemsetAsync(s.sdata + bytes, 0, cap - bytes, s.stream));
    s.scap_bytes = cap;
    again = true;
  }
  if (again) return 0;
  s.s_bytes = bytes;
  return slices_plan(h, s);  // the caller declares the slices valid once every shard has them
}

int gather_slice_bytes(Ctx* h);

// groups from the dense store(s) + pack + wait + plan: the setMatrixData paths, and every fill
// that went through a dense store. Shard by shard (the pinned staging is shared).
int csc_rebuild(Ctx* h) {
  h->csc_valid = false;
  h->total_slice_bytes = 0.0;
  if (!csc_applies(h)) return 0;
  int rc = 0;
  dispatch_vt(h, [&](auto t) {
    using VT = decltype(t);
    for (auto& s : h->sh) {
      bool again = true;
      for (int attempt = 0; again; ++attempt) {
        if (attempt >= 3) {
          rc = fail(CLIPPER_HIP_E_HIP, "compressed storage: the build keeps overflowing");
          return;
        }
        GroupOut<VT> O;
        if ((rc = groups_prepare<VT>(h, s, O))) return;
        dim3 grid(h->csc_nstrips, static_cast<unsigned>(ceil_div(h->csc_nblocks, 2))), block(256);
        hipLaunchKernelGGL((k_groups_from_dense<VT>), grid, block, 0, s.stream,
                           static_cast<const VT*>(s.S), h->W, h->m, O);
        if ((rc = slices_enqueue<VT>(h, s, group_source<VT>(h, s), s.cctl))) return;
        if (hipStreamSynchronize(s.stream) != hipSuccess) {
          rc = fail(CLIPPER_HIP_E_HIP, "compressed storage: build failed: %s",
                    hipGetErrorString(hipGetLastError()));
          return;
        }
        if ((rc = slices_check<VT>(h, s, true, again))) return;
      }
    }
  });
  if (rc) return rc;
  h->csc_valid = true;
  drop_dense(h);  // M lives in the slices from here on (getters materialise a dense copy on demand)
  if ((rc = sync_all(h))) return rc;
  return gather_slice_bytes(h);  // column shards: the row-view policy's cost model is about THIS matrix
}

bool rect_fill_possible(const Ctx* h);
int gather_slice_bytes(Ctx* h);
int launch_rect(Ctx* h, Shard& s, const int32_t* rowmap, int64_t nrows, const SliceOut& O, int64_t col0 = -1, int64_t wcols = -1);

// Every compressed build the symmetric kernel cannot serve (fp64 values, column shards): the
// rectangular tile kernel writes each shard's slices straight from its LDS images — no dense store, no
// groups, no packers, whatever the size. The shards fill concurrently; their directories come back one
// after the other (the pinned staging is the context's).
int run_affinity_rect(Ctx* h, double& kernel_ms) {
  drop_dense(h);
  Shard& s0 = h->sh[0];
  HIPCHK(hipSetDevice(s0.device));
  if (!h->ev_aff[0]) {
    HIPCHK(hipEventCreate(&h->ev_aff[0]));
    HIPCHK(hipEventCreate(&h->ev_aff[1]));
  }
  std::vector<char> pending(h->sh.size(), 1);
  kernel_ms = 0.0;
  for (int attempt = 0;; ++attempt) {
    if (attempt >= 3) return fail(CLIPPER_HIP_E_HIP, "compressed storage: the build keeps overflowing");
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipEventRecord(h->ev_aff[0], s0.stream));
    std::vector<SliceOut> outs(h->sh.size());
    for (size_t k = 0; k < h->sh.size(); ++k) {
      if (!pending[k]) continue;
      Shard& s = h->sh[k];
      HIPCHK(hipSetDevice(s.device));
      int rc = emit_prepare(h, s, s, h->m, outs[k]);
      if (rc) return rc;
      // (emit_prepare stages the arenas' start values in pinned memory shared by all shards)
      if (h->sh.size() > 1) HIPCHK(hipStreamSynchronize(s.stream));
      if ((rc = launch_rect(h, s, nullptr, h->m, outs[k]))) return rc;
    }
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipEventRecord(h->ev_aff[1], s0.stream));
    bool any = false;
    for (size_t k = 0; k < h->sh.size(); ++k) {
      if (!pending[k]) continue;
      Shard& s = h->sh[k];
      int rc = emit_enqueue(h, s, s);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(s.stream));
      HIPCHK(hipGetLastError());
      bool again = false;
      if ((rc = emit_check(h, s, s, csc_single(h), again))) return rc;
      pending[k] = again ? 1 : 0;
      any = any || again;
    }
    float ms = 0.f;
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipEventElapsedTime(&ms, h->ev_aff[0], h->ev_aff[1]));
    kernel_ms = ms;  // the last round of fills (the first of a problem size only sizes the arenas)
    if (!any) break;
  }
  return sync_all(h);
}

// `emits`: the fill kernel `launch` starts writes the slices itself when asked to
// (k_affinity_sym, fp32) — then neither a dense store nor groups are needed
template <typename Launch>
int run_affinity(Ctx* h, bool emits, Launch launch) {
  h->has_matrix = false;  // until the build has succeeded (a failed rebuild leaves no matrix)
  h->csc_valid = false;
  h->total_slice_bytes = 0.0;  // (column shards: gathered again once this build's slices exist)
  for (auto& s : h->sh) s.rv.valid = false;  // a row view of the previous matrix
  h->nodes.clear();
  // explicit constraint storage is not needed on this path: C == pattern(M)
  for (auto& s : h->sh) {
    if (s.Cs) {
      hipSetDevice(s.device);
      hipFree(s.Cs);
      s.Cs = nullptr;
    }
  }
  h->explicitC = false;
  plan_tiles(h);
  int rc = 0;
  const bool emit = csc_applies(h) && csc_single(h) && emits;  // (fp32 values, or fp64 values: k_affinity_sym<.., VT>)
  if (!emit && csc_applies(h) && rect_fill_possible(h) && !h->plain_affinity && !h->strip_affinity) {
    double kms = 0.0;
    if ((rc = run_affinity_rect(h, kms))) return rc;
    h->csc_valid = true;
    h->csc_emitted = true;
    if ((rc = gather_slice_bytes(h))) return rc;
    h->tm.affinity_kernel_ms = kms;
    h->tm.affinity_bytes = static_cast<double>(h->sh[0].s_bytes);
    h->has_matrix = true;
    return 0;
  }
  if (emit) drop_dense(h);  // a materialised copy would be stale
  else if ((rc = ensure_dense(h, false))) return rc;
  Shard& s0 = h->sh[0];
  HIPCHK(hipSetDevice(s0.device));
  if (!h->ev_aff[0]) {
    HIPCHK(hipEventCreate(&h->ev_aff[0]));
    HIPCHK(hipEventCreate(&h->ev_aff[1]));
  }
  hipEvent_t e0 = h->ev_aff[0], e1 = h->ev_aff[1];
  double build_ms = 0.0;
  for (int attempt = 0;; ++attempt) {
    CscOut O{};
    if (emit) {
      rc = emit_prepare(h, s0, O);
      if (rc) return rc;
    } else {
      h->csc_valid = false;
      h->csc_emitted = false;
    }
    h->csc_out = O;
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipEventRecord(e0, s0.stream));
    for (auto& s : h->sh) {
      HIPCHK(hipSetDevice(s.device));
      launch(s);  // k_affinity_sym writes the slices itself and sets csc_emitted
    }
    if (emit) {
      rc = emit_enqueue(h, s0);
      if (rc) return rc;
    }
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipEventRecord(e1, s0.stream));
    const auto th0 = std::chrono::high_resolution_clock::now();
    rc = sync_all(h);
    if (rc) return rc;
    const auto th1 = std::chrono::high_resolution_clock::now();
    if (!emit) {
      // dense slices (column shards, the other fill kernels, fp64 storage): slices from them
      const auto t0 = std::chrono::high_resolution_clock::now();
      rc = csc_rebuild(h);
      if (rc) return rc;
      build_ms = std::chrono::duration<double, std::milli>(
                     std::chrono::high_resolution_clock::now() - t0).count();
      break;
    }
    bool again = false;
    rc = emit_check(h, s0, again);
    if (rc) return rc;
    if (std::getenv("CLIPPER_HIP_HOST_TIMING")) {
      const auto th2 = std::chrono::high_resolution_clock::now();
      float kms = 0.f;
      (void)hipEventElapsedTime(&kms, e0, e1);
      std::fprintf(stderr, "[affinity] enqueue->synced %.1f us (events %.1f us), check+plan %.1f us\n",
                   std::chrono::duration<double, std::micro>(th1 - th0).count(), kms * 1e3,
                   std::chrono::duration<double, std::micro>(th2 - th1).count());
    }
    if (!again) {
      h->csc_valid = true;
      break;
    }
    if (attempt >= 2) return fail(CLIPPER_HIP_E_HIP, "compressed storage: the build keeps overflowing");
  }
  float ms = 0.f;
  HIPCHK(hipSetDevice(s0.device));
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  h->tm.affinity_kernel_ms = ms + (h->csc_valid ? build_ms : 0.0);
  h->tm.affinity_bytes = h->csc_valid ? static_cast<double>(h->sh[0].s_bytes)
                                      : static_cast<double>(h->sh[0].bytes_S);
  h->has_matrix = true;
  return 0;
}

constexpr int AFF_ROWS_PER_BLK = 32;

// ALGORITHMIC bytes one mat-vec launch of shard 0 must move: s * m * (valid owned columns)
// (= s*m^2 on one GPU; the zero padding up to the 64-column pitch is not counted), doubled
// when an explicit constraint matrix is read as well.
double algorithmic_gemv_bytes(const Ctx* h, bool dense = false) {
  if (h->csc_valid && !dense)  // the slices (headers, lengths, step offsets, quads) + their directory
    return static_cast<double>(h->sh[0].s_bytes) +
           static_cast<double>(h->sh[0].s_ncg) * h->sh[0].s_nchunks * 8.0;
  const int64_t c0 = static_cast<int64_t>(h->sh[0].slot) * h->W;
  const int64_t valid = std::max<int64_t>(0, std::min<int64_t>(h->W, h->m - c0));
  return static_cast<double>(h->esize()) * static_cast<double>(h->m) *
         static_cast<double>(valid) * (h->explicitC ? 2.0 : 1.0);
}

// dsd::solve(M_, S) (dsd.cpp:274-320): gathers the sub-matrix induced by S from the device and runs
// Goldberg's algorithm on the host (dsd_host.h). Nodes come back ascending. With M in slices the gather
// walks the slices themselves (k_slice_gather_sub: one read of M, no dense copy — at m = 300 000 there
// could not be one); a dense store is gathered by index (k_gather_sub).
int densest_subgraph_of(Ctx* h, const std::vector<int32_t>& S, std::vector<int32_t>& nodes) {
  nodes.clear();
  const int k = static_cast<int>(S.size());
  if (k < 2) return 0;
  // k x k doubles on the host (twice while the shards' parts are added, once more inside the flow) and
  // on the device: a list of 100 000 nodes would ask for 80 GB each — refused, not thrown
  std::vector<double> Wsub, tmp;
  std::vector<int32_t> pos;
  try {
    Wsub.assign(static_cast<size_t>(k) * k, 0.0);
    tmp.resize(static_cast<size_t>(k) * k);
    pos.assign(static_cast<size_t>(h->m), -1);  // position of every node in the list
  } catch (const std::bad_alloc&) {
    return fail(CLIPPER_HIP_E_NOMEM, "densest subgraph of %d nodes: %.1f GB of host memory per copy of the sub-matrix", k,
                static_cast<double>(k) * k * 8e-9);
  }
  // a list that names a node twice is gathered by index
  bool listed_once = true;
  for (int a = 0; a < k; ++a) {
    int32_t& p = pos[static_cast<size_t>(S[static_cast<size_t>(a)])];
    listed_once = listed_once && p < 0;
    p = a;
  }
  const bool from_slices = h->csc_valid && listed_once;
  const auto t0 = std::chrono::high_resolution_clock::now();
  if (!from_slices)
    if (int rc = ensure_dense(h, true)) return rc;
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    int32_t* didx = nullptr;
    double* dout = nullptr;
    const size_t nidx = from_slices ? pos.size() : static_cast<size_t>(k);
    HIPCHK(hipMalloc(&didx, nidx * sizeof(int32_t)));
    HIPCHK(hipMalloc(&dout, tmp.size() * sizeof(double)));
    HIPCHK(hipMemcpyAsync(didx, from_slices ? pos.data() : S.data(), nidx * sizeof(int32_t),
                          hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemsetAsync(dout, 0, tmp.size() * sizeof(double), s.stream));
    const int64_t c0 = static_cast<int64_t>(s.slot) * h->W;
    if (from_slices) {
      const int64_t nsl = static_cast<int64_t>(s.s_ncg) * s.s_nchunks;
      dim3 grid(static_cast<unsigned>(ceil_div(nsl, 4))), block(256);
      dispatch_vt(h, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL((k_slice_gather_sub<T, SL_H>), grid, block, 0, s.stream, slice_view(h, s), didx,
                           c0, h->m, k, dout);
      });
    } else {
      dim3 grid(static_cast<unsigned>(ceil_div(static_cast<int64_t>(k) * k, 256))), block(256);
      if (h->storage == CLIPPER_HIP_STORE_F64)
        hipLaunchKernelGGL((k_gather_sub<double>), grid, block, 0, s.stream,
                           static_cast<const double*>(s.S), h->W, c0, h->W, didx, k, dout);
      else
        hipLaunchKernelGGL((k_gather_sub<float>), grid, block, 0, s.stream,
                           static_cast<const float*>(s.S), h->W, c0, h->W, didx, k, dout);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(tmp.data(), dout, tmp.size() * sizeof(double), hipMemcpyDeviceToHost,
                          s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    hipFree(didx);
    hipFree(dout);
    for (size_t e = 0; e < tmp.size(); ++e) Wsub[e] += tmp[e];  // disjoint column sets
  }
  if (h->csc_valid) drop_dense(h);  // a copy materialised for this gather only: M lives in the slices
  const auto t1 = std::chrono::high_resolution_clock::now();
  int flows = 0;
  tmp = std::vector<double>();  // (the flow's residual matrix takes its place)
  try {
    for (int32_t a : dsd::densest_subgraph(Wsub, k, h->m, &flows)) nodes.push_back(S[static_cast<size_t>(a)]);
  } catch (const std::bad_alloc&) {
    return fail(CLIPPER_HIP_E_NOMEM, "densest subgraph of %d nodes: the flow network does not fit the host's memory", k);
  }
  if (std::getenv("CLIPPER_HIP_HOST_TIMING"))
    std::fprintf(stderr, "[dsd] k = %d: gather (%s) %.2f ms, host %.2f ms, %d maximum flows%s, %zu nodes\n", k,
                 from_slices ? "slices" : "dense store", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t1).count(),
                 flows < 0 ? -flows : flows, flows < 0 ? " (plain bisection)" : "", nodes.size());
  return 0;
}

}  // namespace
