// host_matrix.hpp — the compressed copy of M (build, plan, expand), the affinity driver, gathers
// Part of clipper_hip.hip (one translation unit; included there, in order).
#pragma once

namespace {

// ---- the column-compressed copy (CLIPPER_HIP_STORE_F32_CSC) ---------------------------------
bool csc_applies(const Ctx* h) { return csc_possible(h) && !h->explicitC; }

CscView csc_view(const Ctx* h, const Shard& s) {
  CscView M;
  M.vals = s.cvals;
  M.rows = s.crows;
  M.Lc = s.cLc;
  M.Pre = s.cPre;
  M.tb = s.ctb;
  M.nblocks = h->csc_nblocks;
  M.ntmax = s.c_ntmax;
  return M;
}

// The dense store of every local shard, allocated if it is not; with `from_csc` its content is
// materialised from the compressed copy when that is all there is.
int ensure_dense(Ctx* h, bool from_csc) {
  for (auto& s : h->sh) {
    if (s.S) continue;
    HIPCHK(hipSetDevice(s.device));
    if (hipMalloc(&s.S, s.bytes_S) != hipSuccess) {
      s.S = nullptr;
      return fail(CLIPPER_HIP_E_NOMEM, "dense store of %zu bytes (this call needs one) does not fit",
                  s.bytes_S);
    }
    if (from_csc && h->csc_valid) {
      dim3 grid(h->csc_nstrips, static_cast<unsigned>(ceil_div(h->csc_nblocks, 2))), block(256);
      hipLaunchKernelGGL(k_csc_expand, grid, block, 0, s.stream, csc_view(h, s),
                         static_cast<float*>(s.S), h->W, h->m);
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

// Before the fill: buffers of the group directory, the arenas' cursors reset. Returns what a
// kernel that emits groups needs (k_affinity_sym, k_csc_build); out.Lc == null: not in use.
int csc_prepare(Ctx* h, Shard& s, CscOut& out) {
  out = CscOut{};
  h->csc_valid = false;
  h->csc_emitted = false;
  if (!csc_applies(h)) return 0;
  HIPCHK(hipSetDevice(s.device));
  const int nblocks = static_cast<int>(ceil_div(h->m, CSC_RB));
  h->csc_nstrips = static_cast<int>(ceil_div(h->W, CSC_CW));
  const size_t G = static_cast<size_t>(h->csc_nstrips) * static_cast<size_t>(nblocks);
  h->csc_nblocks = nblocks;
  if (G > s.ccap_groups) {
    if (s.cLc) hipFree(s.cLc);
    if (s.cPre) hipFree(s.cPre);
    s.cLc = nullptr;
    s.cPre = nullptr;
    HIPCHK(hipMalloc(&s.cLc, G * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&s.cPre, G * sizeof(uint64_t)));
    s.ccap_groups = G;
  }
  if (!s.cctl) HIPCHK(hipMalloc(&s.cctl, CSC_ARENAS * sizeof(CscBuildCtl)));
  if (G > h->csc_hcap_groups) {
    if (h->csc_hLc) hipHostFree(h->csc_hLc);
    h->csc_hLc = nullptr;
    HIPCHK(hipHostMalloc(&h->csc_hLc, G * sizeof(uint32_t), hipHostMallocDefault));
    h->csc_hcap_groups = G;
  }
  if (!h->csc_hctl) {
    HIPCHK(hipHostMalloc(&h->csc_hctl, 2 * CSC_ARENAS * sizeof(CscBuildCtl), hipHostMallocDefault));
  }
  CscBuildCtl* init = h->csc_hctl + CSC_ARENAS;  // second half: what the device starts from
  for (int k = 0; k < CSC_ARENAS; ++k) {
    init[k].cursor = 0;
    init[k].capacity = s.ccap_units / CSC_ARENAS;
    init[k].origin = static_cast<unsigned long long>(k) * (s.ccap_units / CSC_ARENAS);
    init[k].overflow = 0;
  }
  HIPCHK(hipMemcpyAsync(s.cctl, init, CSC_ARENAS * sizeof(CscBuildCtl), hipMemcpyHostToDevice,
                        s.stream));
  out.Lc = s.cLc;
  out.Pre = s.cPre;
  out.vals = s.cvals;
  out.rows = s.crows;
  out.ctl = s.cctl;
  out.nblocks = nblocks;
  return 0;
}

// After the fill: the build from the dense store unless the fill kernel emitted the groups
// itself, then the copies of the counters to pinned host memory (csc_finish() reads them once
// the stream was synchronised).
int csc_enqueue(Ctx* h, Shard& s, const CscOut& O) {
  if (O.Lc == nullptr) return 0;
  HIPCHK(hipSetDevice(s.device));
  if (!h->csc_emitted) {
    dim3 grid(h->csc_nstrips, static_cast<unsigned>(ceil_div(h->csc_nblocks, 2))), block(256);
    hipLaunchKernelGGL(k_csc_build, grid, block, 0, s.stream, static_cast<const float*>(s.S), h->W,
                       h->m, O);
  }
  const size_t G = static_cast<size_t>(h->csc_nstrips) * static_cast<size_t>(h->csc_nblocks);
  HIPCHK(hipMemcpyAsync(h->csc_hctl, s.cctl, CSC_ARENAS * sizeof(CscBuildCtl),
                        hipMemcpyDeviceToHost, s.stream));
  HIPCHK(hipMemcpyAsync(h->csc_hLc, s.cLc, G * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
  return 0;
}

// Row tiles of equal cost per strip (cost of a block: its padded list length + a constant for
// the staging of its x rows). The number of workgroups aims at whole waves of co-resident ones
// (two 8-wave workgroups per CU measured best: every workgroup repeats the decision), at most
// ~32 blocks each.
int csc_plan(Ctx* h, Shard& s) {
  const int nstrips = h->csc_nstrips, nblocks = h->csc_nblocks;
  const uint32_t* L = h->csc_hLc;
  const double slots = static_cast<double>(h->cus) * 2.0;
  const double G = static_cast<double>(nstrips) * nblocks;
  double target = slots * std::max(1.0, std::ceil(G / (slots * 32.0)));
  if (const char* e = std::getenv("CLIPPER_HIP_CSC_WGS")) target = std::max(1.0, std::atof(e));
  std::vector<double> tot(static_cast<size_t>(nstrips), 0.0);
  double total = 0.0;
  for (int st = 0; st < nstrips; ++st) {
    double t = 0.0;
    for (int b = 0; b < nblocks; ++b) t += static_cast<double>(L[static_cast<size_t>(st) * nblocks + b]) + 2.0;
    tot[static_cast<size_t>(st)] = t;
    total += t;
  }
  const double Q = total / target;
  std::vector<int> nts(static_cast<size_t>(nstrips));
  int ntmax = 1;
  for (int st = 0; st < nstrips; ++st) {
    int n = static_cast<int>(std::max(1.0, std::floor(tot[static_cast<size_t>(st)] / Q + 0.5)));
    n = std::min(n, nblocks);
    nts[static_cast<size_t>(st)] = n;
    ntmax = std::max(ntmax, n);
  }
  const size_t ntb = static_cast<size_t>(nstrips) * static_cast<size_t>(ntmax + 1);
  if (ntb > h->csc_hcap_tb) {
    if (h->csc_htb) hipHostFree(h->csc_htb);
    h->csc_htb = nullptr;
    HIPCHK(hipHostMalloc(&h->csc_htb, ntb * sizeof(int), hipHostMallocDefault));
    h->csc_hcap_tb = ntb;
  }
  for (int st = 0; st < nstrips; ++st) {
    int* t = h->csc_htb + static_cast<size_t>(st) * (ntmax + 1);
    const int n = nts[static_cast<size_t>(st)];
    const double T = tot[static_cast<size_t>(st)];
    double run = 0.0;
    int k = 1;
    t[0] = 0;
    for (int b = 0; b < nblocks; ++b) {
      run += static_cast<double>(L[static_cast<size_t>(st) * nblocks + b]) + 2.0;
      while (k < n && run >= T * k / n) t[k++] = b + 1;
    }
    for (; k <= ntmax; ++k) t[k] = nblocks;
  }
  HIPCHK(hipSetDevice(s.device));
  if (ntb > s.ccap_tb) {
    if (s.ctb) hipFree(s.ctb);
    s.ctb = nullptr;
    HIPCHK(hipMalloc(&s.ctb, ntb * sizeof(int)));
    s.ccap_tb = ntb;
  }
  HIPCHK(hipMemcpyAsync(s.ctb, h->csc_htb, ntb * sizeof(int), hipMemcpyHostToDevice, s.stream));
  const size_t NSLOT = static_cast<size_t>(nslot(h->V));
  if (static_cast<size_t>(ntmax) > s.part_tiles) {
    HIPCHK(hipFree(s.part));
    s.part = nullptr;
    s.part_tiles = static_cast<size_t>(ntmax) + 8;
    HIPCHK(hipMalloc(&s.part, s.part_tiles * NSLOT * static_cast<size_t>(h->W) * sizeof(double)));
  }
  s.c_ntmax = ntmax;
  return 0;
}

// After the stream was synchronised: did the lists fit? If not (always the case for the first
// matrix of a size) the buffers are grown and `again` is set — the caller repeats the step that
// produces the groups; otherwise the tiles are planned and the copy is valid.
int csc_check(Ctx* h, Shard& s, bool& again) {
  again = false;
  if (!csc_applies(h)) return 0;
  HIPCHK(hipSetDevice(s.device));
  bool over = false;
  size_t worst = 0;
  uint64_t sum = 0;
  for (int k = 0; k < CSC_ARENAS; ++k) {
    over = over || h->csc_hctl[k].overflow != 0;
    worst = std::max(worst, static_cast<size_t>(h->csc_hctl[k].cursor));
    sum += h->csc_hctl[k].cursor;
  }
  if (over) {
    const size_t need = worst * CSC_ARENAS;  // every arena as large as the fullest one
    if (s.cvals) hipFree(s.cvals);
    if (s.crows) hipFree(s.crows);
    s.cvals = nullptr;
    s.crows = nullptr;
    s.ccap_units = (need + need / 8 + 64 * CSC_ARENAS) / CSC_ARENAS * CSC_ARENAS;
    HIPCHK(hipMalloc(&s.cvals, s.ccap_units * 128 * sizeof(float)));
    HIPCHK(hipMalloc(&s.crows, s.ccap_units * 128));
    again = true;
    return 0;
  }
  s.c_units = sum;
  return csc_plan(h, s);  // the caller declares the copy valid once every shard has one
}

// build from the dense store(s) + wait + plan: the setMatrixData paths, and every fill of
// column shards. Shard by shard (the pinned staging of the counters is shared).
int csc_rebuild(Ctx* h) {
  h->csc_valid = false;
  if (!csc_applies(h)) return 0;
  for (auto& s : h->sh) {
    bool again = true;
    for (int attempt = 0; again; ++attempt) {
      if (attempt >= 3) return fail(CLIPPER_HIP_E_HIP, "compressed copy: the build keeps overflowing");
      CscOut O;
      int rc = csc_prepare(h, s, O);
      if (rc) return rc;
      rc = csc_enqueue(h, s, O);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(s.stream));
      rc = csc_check(h, s, again);
      if (rc) return rc;
    }
  }
  h->csc_valid = true;
  return 0;
}

// `emits`: the fill kernel `launch` starts writes the compressed copy itself when asked to
// (k_affinity_sym) — then no dense store is needed at all
template <typename Launch>
int run_affinity(Ctx* h, bool emits, Launch launch) {
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
  const bool emit = csc_applies(h) && csc_single(h) && emits;
  if (emit) drop_dense(h);  // a materialised copy would be stale
  else if ((rc = ensure_dense(h, false))) return rc;
  hipEvent_t e0, e1;
  Shard& s0 = h->sh[0];
  HIPCHK(hipSetDevice(s0.device));
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  double build_ms = 0.0;
  for (int attempt = 0;; ++attempt) {
    CscOut O{};
    if (emit) {
      rc = csc_prepare(h, s0, O);
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
      launch(s);  // k_affinity_sym emits the compressed copy itself and sets csc_emitted
    }
    if (emit) {
      rc = csc_enqueue(h, s0, O);  // counted as part of the affinity build
      if (rc) return rc;
    }
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipEventRecord(e1, s0.stream));
    rc = sync_all(h);
    if (rc) return rc;
    if (!emit) {
      // dense slices (column shards, the other fill kernels): the compressed copies from them
      const auto t0 = std::chrono::high_resolution_clock::now();
      rc = csc_rebuild(h);
      if (rc) return rc;
      build_ms = std::chrono::duration<double, std::milli>(
                     std::chrono::high_resolution_clock::now() - t0).count();
      break;
    }
    bool again = false;
    rc = csc_check(h, s0, again);
    if (rc) return rc;
    if (!again) {
      h->csc_valid = true;
      break;
    }
    if (attempt >= 2) return fail(CLIPPER_HIP_E_HIP, "compressed copy: the build keeps overflowing");
  }
  float ms = 0.f;
  HIPCHK(hipSetDevice(s0.device));
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  h->tm.affinity_kernel_ms = ms + (h->csc_valid ? build_ms : 0.0);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  h->has_matrix = true;
  return 0;
}

constexpr int AFF_ROWS_PER_BLK = 32;

// ALGORITHMIC bytes one mat-vec launch of shard 0 must move: s * m * (valid owned columns)
// (= s*m^2 on one GPU; the zero padding up to the 64-column pitch is not counted), doubled
// when an explicit constraint matrix is read as well.
double algorithmic_gemv_bytes(const Ctx* h, bool dense = false) {
  if (h->csc_valid && !dense)  // the compressed copy: 5 bytes per (padded) entry + the group directory
    return static_cast<double>(h->sh[0].c_units) * 128.0 * 5.0 +
           static_cast<double>(h->csc_nstrips) * h->csc_nblocks * 12.0;
  const int64_t c0 = static_cast<int64_t>(h->sh[0].slot) * h->W;
  const int64_t valid = std::max<int64_t>(0, std::min<int64_t>(h->W, h->m - c0));
  return static_cast<double>(h->esize()) * static_cast<double>(h->m) *
         static_cast<double>(valid) * (h->explicitC ? 2.0 : 1.0);
}

// dsd::solve(M_, S) (dsd.cpp:274-320): gathers the sub-matrix induced by S from the device
// slices and runs Goldberg's algorithm on the host (dsd_host.h). Nodes come back ascending.
int densest_subgraph_of(Ctx* h, const std::vector<int32_t>& S, std::vector<int32_t>& nodes) {
  nodes.clear();
  const int k = static_cast<int>(S.size());
  if (k < 2) return 0;
  std::vector<double> Wsub(static_cast<size_t>(k) * k, 0.0), tmp(static_cast<size_t>(k) * k);
  if (int rc = ensure_dense(h, true)) return rc;
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    int32_t* didx = nullptr;
    double* dout = nullptr;
    HIPCHK(hipMalloc(&didx, static_cast<size_t>(k) * sizeof(int32_t)));
    HIPCHK(hipMalloc(&dout, tmp.size() * sizeof(double)));
    HIPCHK(hipMemcpyAsync(didx, S.data(), static_cast<size_t>(k) * sizeof(int32_t),
                          hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemsetAsync(dout, 0, tmp.size() * sizeof(double), s.stream));
    dim3 grid(static_cast<unsigned>(ceil_div(static_cast<int64_t>(k) * k, 256))), block(256);
    const int64_t c0 = static_cast<int64_t>(s.slot) * h->W;
    if (h->storage == CLIPPER_HIP_STORE_F64)
      hipLaunchKernelGGL((k_gather_sub<double>), grid, block, 0, s.stream,
                         static_cast<const double*>(s.S), h->W, c0, h->W, didx, k, dout);
    else
      hipLaunchKernelGGL((k_gather_sub<float>), grid, block, 0, s.stream,
                         static_cast<const float*>(s.S), h->W, c0, h->W, didx, k, dout);
    HIPCHK(hipMemcpyAsync(tmp.data(), dout, tmp.size() * sizeof(double), hipMemcpyDeviceToHost,
                          s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    hipFree(didx);
    hipFree(dout);
    for (size_t e = 0; e < tmp.size(); ++e) Wsub[e] += tmp[e];  // disjoint column sets
  }
  for (int32_t a : dsd::densest_subgraph(Wsub, k, h->m)) nodes.push_back(S[static_cast<size_t>(a)]);
  return 0;
}

}  // namespace
