// clipper_hip.hip — host side of the C ABI declared in include/clipper_hip.h.
//
// Owns device memory, streams and the solve loop; all arithmetic runs in the kernels of
// kernels.hip.h. One translation unit: host_state.hpp (context, shards, RCCL binding),
// host_solver.hpp (planning, dispatch, one iteration), host_matrix.hpp (compressed copy, affinity
// driver) are included below, then the extern "C" entry points. There is no CPU fallback anywhere in this file: if HIP is unusable the
// entry points return an error.
//
// Solve loop (CLIPPER::solve -> findDenseClique, /root/reference/src/clipper.cpp:172-323):
// the whole state machine — windowed line search, convergence tests, penalty homotopy — lives
// in device memory (SolverState). One solver iteration = k_gemv (every workgroup decides what
// the previous iteration's results mean, then streams M against a window of V candidate
// vectors), then k_tail (grid: blocks x V). The host only enqueues iterations, a few ahead of
// what the device reports as started in a pinned progress record, and stops when `done` shows
// up there; kernels launched after convergence return immediately.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <queue>
#include <string>
#include <type_traits>
#include <utility>
#include <map>
#include <mutex>
#include <new>
#include <exception>
#include <vector>

#include "../../include/clipper_hip.h"
#include "kernels.hip.h"
#include "dsd_host.h"
#include "host_batch.hpp"
#include "host_plan.hpp"

using namespace clipper_hip;

#include "host_state.hpp"
#include "host_solver.hpp"
#include "host_matrix.hpp"
#include "host_rowview.hpp"
#include "host_resident.hpp"
#include "host_rv_resident.hpp"
#include "host_subproblem.hpp"
#include "host_registration.hpp"

// ============================================================================================
// brute-force nearest neighbours: launch of the two kernels for one (K, D)
namespace {

template <int K, int D>
int knn_run(const double* dP0, int64_t n0, const double* dP1, int64_t n1, int S, int64_t chunk,
            double* pd, int32_t* pi, double* od, int32_t* oi, hipStream_t st) {
  dim3 g(static_cast<unsigned>(ceil_div(n0, 256)), static_cast<unsigned>(S));
  hipLaunchKernelGGL((k_knn_partial<K, D>), g, dim3(256), 0, st, dP0, n0, dP1, n1, chunk, pd, pi);
  hipLaunchKernelGGL((k_knn_merge<K>), dim3(static_cast<unsigned>(ceil_div(n0, 256))), dim3(256), 0,
                     st, pd, pi, n0, S, od, oi);
  return 0;
}

}  // namespace

extern "C" {

const char* clipper_hip_last_error(void) try {
  return g_err.c_str();
} CLIPPER_HIP_GUARD_STR

int clipper_hip_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
} CLIPPER_HIP_GUARD_INT

clipper_hip_t* clipper_hip_create(int device, int storage) try {
  return make_ctx(&device, 1, storage, 1, 0, false);
} CLIPPER_HIP_GUARD_PTR

clipper_hip_t* clipper_hip_create_group(const int* devices, int nshards, int storage) try {
  if (!devices || nshards < 1) {
    fail(CLIPPER_HIP_E_INVALID, "invalid shard list");
    return nullptr;
  }
  return make_ctx(devices, nshards, storage, nshards, 0, false);
} CLIPPER_HIP_GUARD_PTR

clipper_hip_t* clipper_hip_create_rank(int device, int storage, int rank, int world) try {
  if (world < 1 || rank < 0 || rank >= world) {
    fail(CLIPPER_HIP_E_INVALID, "rank %d / world %d invalid", rank, world);
    return nullptr;
  }
  // CLIPPER_HIP_FORCE_RCCL=1 routes even a 1-rank world through ncclAllGather, so that the
  // communicator plumbing can be exercised on a single-GPU box
  const char* force = std::getenv("CLIPPER_HIP_FORCE_RCCL");
  const bool multiproc = world > 1 || (force && force[0] == '1');
  return make_ctx(&device, 1, storage, world, rank, multiproc);
} CLIPPER_HIP_GUARD_PTR

int clipper_hip_comm_unique_id(void* id128) try {
  if (!id128) return fail(CLIPPER_HIP_E_INVALID, "null id buffer");
  int rc = load_rccl();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(CLIPPER_HIP_E_COMM, "ncclGetUniqueId failed (%d)", (int)r);
  std::memcpy(id128, &id, sizeof(id));
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_comm_init(clipper_hip_t* h, const void* id128) try {
  if (!h || !id128) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->multiproc) return 0;  // nothing to exchange
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  HIPCHK(hipSetDevice(h->sh[0].device));
  ncclResult_t r = g_rccl.CommInitRank(&h->comm, h->world, id, h->sh[0].slot);
  if (r != ncclSuccess)
    return fail(CLIPPER_HIP_E_COMM, "ncclCommInitRank: %s",
                g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_comm_init_callback(clipper_hip_t* h, clipper_hip_allgather_fn fn, void* user) try {
  if (!h || !fn) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->multiproc) return 0;  // nothing to exchange
  h->xchg_fn = fn;
  h->xchg_user = user;
  return 0;
} CLIPPER_HIP_GUARD_INT

void clipper_hip_destroy(clipper_hip_t* h) try {
  if (!h) return;
  for (auto& s : h->sh) {
    hipSetDevice(s.device);
    if (s.stream) hipStreamSynchronize(s.stream);
  }
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  sub_free(h);  // (the live sub-problem's context borrows this one's stream: before the stream goes)
  for (auto& s : h->sh) {
    free_shard_buffers(s);
    if (s.ev_reduced) hipEventDestroy(s.ev_reduced);
    if (s.ev_copied) hipEventDestroy(s.ev_copied);
    if (s.stream) hipStreamDestroy(s.stream);
  }
  if (!h->sh.empty()) hipSetDevice(h->sh[0].device);
  for (hipEvent_t e : h->ev_pairs) hipEventDestroy(e);
  for (hipEvent_t e : h->ev_xchg) hipEventDestroy(e);
  if (h->ev_poll[0]) hipEventDestroy(h->ev_poll[0]);
  if (h->ev_poll[1]) hipEventDestroy(h->ev_poll[1]);
  if (h->host_state) hipHostFree(h->host_state);
  if (h->mirror) hipHostFree(h->mirror);
  if (h->kind) hipHostFree(h->kind);
  if (h->u_pinned) hipHostFree(h->u_pinned);
  if (h->stamps_dev) hipFree(h->stamps_dev);
  if (h->xchg_send) hipHostFree(h->xchg_send);
  if (h->xchg_recv) hipHostFree(h->xchg_recv);
  if (h->ev_aff[0]) hipEventDestroy(h->ev_aff[0]);
  if (h->ev_aff[1]) hipEventDestroy(h->ev_aff[1]);
  if (h->csc_hLq) hipHostFree(h->csc_hLq);
  if (h->csc_hctl) hipHostFree(h->csc_hctl);
  if (h->csc_htotal) hipHostFree(h->csc_htotal);
  resident_free(h);
  rvr_free(h);
  if (h->csc_hwork) hipHostFree(h->csc_hwork);
  if (h->rv_count) hipHostFree(h->rv_count);
  if (h->rv_desc_host) hipHostFree(h->rv_desc_host);
  delete h;
} CLIPPER_HIP_GUARD_VOID

// ---- affinity --------------------------------------------------------------------------

int clipper_hip_stage_inputs(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                             const double* D2, int64_t n2, const int32_t* A, int64_t m) try {
  return stage_inputs(h, D1, d, n1, D2, n2, A, m);
} CLIPPER_HIP_GUARD_INT

int clipper_hip_affinity_euclidean_staged(clipper_hip_t* h, double sigma, double epsilon,
                                          double mindist, double affinityeps) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->staged_d < 1) return fail(CLIPPER_HIP_E_STATE, "clipper_hip_stage_inputs not called");
  const EuclidParams prm{sigma, epsilon, mindist, affinityeps};
  const int64_t mm = h->m, W = h->W, pstride = h->staged_pstride;
  const int d = h->staged_d;
  h->fill_kind = 1;  // what a row view of this matrix is filled with later (host_rowview.hpp)
  h->fill_e = prm;
  h->fill_n = PointNormalParams{};
  h->fill_E2 = guarded_threshold_sq(guarded_threshold(epsilon, h->staged_maxabs, d));
  return run_affinity(h, use_sym_fill(h) && (d == 2 || d == 3), [&](Shard& s) {
    dim3 grid(static_cast<unsigned>(ceil_div(W, 1024)),
              static_cast<unsigned>(ceil_div(mm, AFF_ROWS_PER_BLK))),
        block(256);
    const int64_t c0 = static_cast<int64_t>(s.slot) * W;
    const int32_t* A0 = s.Adev;
    const int32_t* A1 = s.Adev + mm;
#define LAUNCH_EUCLID(T, D)                                                                 \
  hipLaunchKernelGGL((k_affinity_euclid<T, D>), grid, block, 0, s.stream,                   \
                     static_cast<T*>(s.S), W, mm, c0, AFF_ROWS_PER_BLK, d, s.P1, s.P2, pstride, \
                     A0, A1, prm)
#define LAUNCH_EUCLID_COMPACT(T, D)                                                         \
  hipLaunchKernelGGL((k_affinity_euclid_compact<T, D>), grid, block, 0, s.stream,           \
                     static_cast<T*>(s.S), W, mm, c0, AFF_ROWS_PER_BLK, s.P1, s.P2, s.P1f,  \
                     s.P2f, pstride, A0, A1, prm, thr)
    const float thr = guarded_threshold(epsilon, h->staged_maxabs, d);
    if (use_sym_fill(h) && (d == 2 || d == 3)) {
      const int nT = static_cast<int>(ceil_div(mm, AT));
      dim3 g(static_cast<unsigned>(static_cast<int64_t>(nT) * (nT + 1) / 2));
      const PointNormalParams none{};
      const float E2 = guarded_threshold_sq(thr);
      if (h->storage == CLIPPER_HIP_STORE_F64) {  // (slices with fp64 values: no dense store on this route)
        if (d == 3)
          launch_sym<double>(k_affinity_sym<3, false, double>, g, s.stream, static_cast<double*>(nullptr), W, mm, nT, s,
                             pstride, A0, A1, prm, none, E2, h->csc_out);
        else
          launch_sym<double>(k_affinity_sym<2, false, double>, g, s.stream, static_cast<double*>(nullptr), W, mm, nT, s,
                             pstride, A0, A1, prm, none, E2, h->csc_out);
      } else if (d == 3)
        launch_sym<float>(k_affinity_sym<3, false>, g, s.stream, static_cast<float*>(s.S), W, mm, nT, s,
                          pstride, A0, A1, prm, none, E2, h->csc_out);
      else
        launch_sym<float>(k_affinity_sym<2, false>, g, s.stream, static_cast<float*>(s.S), W, mm, nT, s,
                          pstride, A0, A1, prm, none, E2, h->csc_out);
      h->csc_emitted = (h->csc_out.Pre != nullptr);
      return;
    }
    const bool compact = !h->plain_affinity && (d == 2 || d == 3);
    if (h->storage == CLIPPER_HIP_STORE_F64) {
      if (compact && d == 3) LAUNCH_EUCLID_COMPACT(double, 3);
      else if (compact && d == 2) LAUNCH_EUCLID_COMPACT(double, 2);
      else if (d == 3) LAUNCH_EUCLID(double, 3);
      else if (d == 2) LAUNCH_EUCLID(double, 2);
      else LAUNCH_EUCLID(double, 0);
    } else {
      if (compact && d == 3) LAUNCH_EUCLID_COMPACT(float, 3);
      else if (compact && d == 2) LAUNCH_EUCLID_COMPACT(float, 2);
      else if (d == 3) LAUNCH_EUCLID(float, 3);
      else if (d == 2) LAUNCH_EUCLID(float, 2);
      else LAUNCH_EUCLID(float, 0);
    }
#undef LAUNCH_EUCLID_COMPACT
#undef LAUNCH_EUCLID
  });
} CLIPPER_HIP_GUARD_INT

int clipper_hip_affinity_pointnormal_staged(clipper_hip_t* h, double sigp, double epsp,
                                            double sign, double epsn, double affinityeps) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->staged_d != 6)
    return fail(CLIPPER_HIP_E_STATE, "PointNormalDistance needs staged inputs with d == 6");
  const PointNormalParams prm{sigp, epsp, sign, epsn, affinityeps};
  const int64_t mm = h->m, W = h->W, pstride = h->staged_pstride;
  h->fill_kind = 2;
  h->fill_e = EuclidParams{};
  h->fill_n = prm;
  h->fill_E2 = guarded_threshold_sq(guarded_threshold(epsp, h->staged_maxabs, 3));
  return run_affinity(h, use_sym_fill(h), [&](Shard& s) {
    dim3 grid(static_cast<unsigned>(ceil_div(W, 1024)),
              static_cast<unsigned>(ceil_div(mm, AFF_ROWS_PER_BLK))),
        block(256);
    const int64_t c0 = static_cast<int64_t>(s.slot) * W;
    const float thr = guarded_threshold(epsp, h->staged_maxabs, 3);
    if (use_sym_fill(h)) {
      const int nT = static_cast<int>(ceil_div(mm, AT));
      dim3 g(static_cast<unsigned>(static_cast<int64_t>(nT) * (nT + 1) / 2));
      const EuclidParams none{};
      if (h->storage == CLIPPER_HIP_STORE_F64)
        launch_sym<double>(k_affinity_sym<3, true, double>, g, s.stream, static_cast<double*>(nullptr), W, mm, nT, s,
                           pstride, s.Adev, s.Adev + mm, none, prm, guarded_threshold_sq(thr), h->csc_out);
      else
        launch_sym<float>(k_affinity_sym<3, true>, g, s.stream, static_cast<float*>(s.S), W, mm, nT, s,
                          pstride, s.Adev, s.Adev + mm, none, prm, guarded_threshold_sq(thr), h->csc_out);
      h->csc_emitted = (h->csc_out.Pre != nullptr);
      return;
    }
    if (h->plain_affinity) {
      if (h->storage == CLIPPER_HIP_STORE_F64)
        hipLaunchKernelGGL((k_affinity_pointnormal<double>), grid, block, 0, s.stream,
                           static_cast<double*>(s.S), W, mm, c0, AFF_ROWS_PER_BLK, s.P1, s.P2,
                           pstride, s.Adev, s.Adev + mm, prm);
      else
        hipLaunchKernelGGL((k_affinity_pointnormal<float>), grid, block, 0, s.stream,
                           static_cast<float*>(s.S), W, mm, c0, AFF_ROWS_PER_BLK, s.P1, s.P2,
                           pstride, s.Adev, s.Adev + mm, prm);
    } else {
      if (h->storage == CLIPPER_HIP_STORE_F64)
        hipLaunchKernelGGL((k_affinity_pointnormal_compact<double>), grid, block, 0, s.stream,
                           static_cast<double*>(s.S), W, mm, c0, AFF_ROWS_PER_BLK, s.P1, s.P2,
                           s.P1f, s.P2f, pstride, s.Adev, s.Adev + mm, prm, thr);
      else
        hipLaunchKernelGGL((k_affinity_pointnormal_compact<float>), grid, block, 0, s.stream,
                           static_cast<float*>(s.S), W, mm, c0, AFF_ROWS_PER_BLK, s.P1, s.P2,
                           s.P1f, s.P2f, pstride, s.Adev, s.Adev + mm, prm, thr);
    }
  });
} CLIPPER_HIP_GUARD_INT

int clipper_hip_affinity_euclidean(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                                   const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                   double sigma, double epsilon, double mindist,
                                   double affinityeps) try {
  const auto t0 = std::chrono::high_resolution_clock::now();
  int rc = stage_inputs(h, D1, d, n1, D2, n2, A, m);
  if (rc) return rc;
  rc = clipper_hip_affinity_euclidean_staged(h, sigma, epsilon, mindist, affinityeps);
  if (rc) return rc;
  h->tm.affinity_total_ms =
      std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0)
          .count();
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_affinity_pointnormal(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                                     const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                     double sigp, double epsp, double sign, double epsn,
                                     double affinityeps) try {
  if (d != 6) return fail(CLIPPER_HIP_E_INVALID, "PointNormalDistance needs d == 6");
  const auto t0 = std::chrono::high_resolution_clock::now();
  int rc = stage_inputs(h, D1, d, n1, D2, n2, A, m);
  if (rc) return rc;
  rc = clipper_hip_affinity_pointnormal_staged(h, sigp, epsp, sign, epsn, affinityeps);
  if (rc) return rc;
  h->tm.affinity_total_ms =
      std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0)
          .count();
  return 0;
} CLIPPER_HIP_GUARD_INT

int64_t clipper_hip_num_associations(const clipper_hip_t* h) try {
  return h ? h->m : 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_get_associations(const clipper_hip_t* h, int32_t* A_out) try {
  if (!h || !A_out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->A.size() != static_cast<size_t>(2 * h->m))
    return fail(CLIPPER_HIP_E_STATE, "no association list is held");
  std::memcpy(A_out, h->A.data(), h->A.size() * sizeof(int32_t));
  return 0;
} CLIPPER_HIP_GUARD_INT

// ---- matrix set / get ------------------------------------------------------------------

int clipper_hip_set_matrix(clipper_hip_t* h, const double* M, const double* C, int64_t m) try {
  if (!h || !M || !C || m < 1) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->A.size() != static_cast<size_t>(2 * m)) h->A.clear();
  h->nodes.clear();
  h->has_matrix = false;  // until the new matrix is complete
  h->csc_valid = false;
  h->fill_kind = 0;  // no points behind this matrix: no row view
  rowview_drop(h);
  int rc = ensure_problem(h, m);
  if (rc) return rc;
  h->has_matrix = false;
  h->csc_valid = false;
  if ((rc = ensure_dense(h, false))) return rc;
  const size_t bytes = static_cast<size_t>(m) * m * sizeof(double);
  const int64_t W = h->W;
  // device temporaries of this call, released on every path
  struct Temps {
    std::vector<std::pair<int, void*>> v;
    ~Temps() {
      for (auto& p : v) {
        hipSetDevice(p.first);
        hipFree(p.second);
      }
    }
    int alloc(int dev, void** p, size_t n) {
      HIPCHK(hipMalloc(p, n));
      v.emplace_back(dev, *p);
      return 0;
    }
  } tmp;
  // pass 1: fill S and detect whether C is anything other than pattern(M)
  std::vector<double*> dM(h->sh.size(), nullptr), dC(h->sh.size(), nullptr);
  std::vector<int*> dflag(h->sh.size(), nullptr);
  int mismatch = 0;
  const unsigned gy = static_cast<unsigned>(std::min<int64_t>(m, 65535));
  for (size_t k = 0; k < h->sh.size(); ++k) {
    Shard& s = h->sh[k];
    HIPCHK(hipSetDevice(s.device));
    if ((rc = tmp.alloc(s.device, reinterpret_cast<void**>(&dM[k]), bytes))) return rc;
    if ((rc = tmp.alloc(s.device, reinterpret_cast<void**>(&dC[k]), bytes))) return rc;
    if ((rc = tmp.alloc(s.device, reinterpret_cast<void**>(&dflag[k]), sizeof(int)))) return rc;
    HIPCHK(hipMemsetAsync(dflag[k], 0, sizeof(int), s.stream));
    HIPCHK(hipMemcpyAsync(dM[k], M, bytes, hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemcpyAsync(dC[k], C, bytes, hipMemcpyHostToDevice, s.stream));
    if (s.Cs) {
      hipFree(s.Cs);
      s.Cs = nullptr;
    }
    dim3 grid(static_cast<unsigned>(ceil_div(W, 256)), gy), block(256);
    const int64_t c0 = static_cast<int64_t>(s.slot) * W;
    if (h->storage == CLIPPER_HIP_STORE_F64)
      hipLaunchKernelGGL((k_from_dense_upper<double>), grid, block, 0, s.stream,
                         static_cast<double*>(s.S), W, m, c0, dM[k], dC[k],
                         static_cast<double*>(nullptr), dflag[k]);
    else
      hipLaunchKernelGGL((k_from_dense_upper<float>), grid, block, 0, s.stream,
                         static_cast<float*>(s.S), W, m, c0, dM[k], dC[k],
                         static_cast<float*>(nullptr), dflag[k]);
    int f = 0;
    HIPCHK(hipMemcpyAsync(&f, dflag[k], sizeof(int), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipGetLastError());
    mismatch |= f;
  }
  // NOTE: in multi-process mode every rank sees the whole (M, C), so `mismatch` agrees.
  // The full upper triangle must be inspected, not only owned columns: do it on the host
  // cheaply when sharded (owned columns cover all (lo,hi) pairs with hi or lo owned only).
  if (h->world > 1 && !mismatch) {
    for (int64_t hi = 1; hi < m && !mismatch; ++hi)
      for (int64_t lo = 0; lo < hi; ++lo) {
        const double mv = M[lo + hi * m], cv = C[lo + hi * m];
        if (cv != ((mv != 0.0) ? 1.0 : 0.0)) {
          mismatch = 1;
          break;
        }
      }
  }
  h->explicitC = (mismatch != 0);
  plan_tiles(h);
  if (h->explicitC) {
    for (size_t k = 0; k < h->sh.size(); ++k) {
      Shard& s = h->sh[k];
      HIPCHK(hipSetDevice(s.device));
      HIPCHK(hipMalloc(&s.Cs, s.bytes_S));
      dim3 grid(static_cast<unsigned>(ceil_div(W, 256)), gy), block(256);
      const int64_t c0 = static_cast<int64_t>(s.slot) * W;
      if (h->storage == CLIPPER_HIP_STORE_F64)
        hipLaunchKernelGGL((k_from_dense_upper<double>), grid, block, 0, s.stream,
                           static_cast<double*>(s.S), W, m, c0, dM[k], dC[k],
                           static_cast<double*>(s.Cs), static_cast<int*>(nullptr));
      else
        hipLaunchKernelGGL((k_from_dense_upper<float>), grid, block, 0, s.stream,
                           static_cast<float*>(s.S), W, m, c0, dM[k], dC[k],
                           static_cast<float*>(s.Cs), static_cast<int*>(nullptr));
    }
  }
  if ((rc = sync_all(h))) return rc;
  if ((rc = csc_rebuild(h))) return rc;
  h->has_matrix = true;
  return 0;
} CLIPPER_HIP_GUARD_INT

// setSparseMatrixData (clipper.cpp:162-166). Every stored (i, j) with i < j stands for the symmetric
// pair, as selfadjointView<Upper> reads it; entries below the diagonal are not read (the reference
// never does); the diagonal is implicit. With compressed storage and C == pattern(M) the slices
// are packed straight from the lists (no dense intermediate: O(nnz) memory); otherwise through the
// dense store.
int clipper_hip_set_sparse(clipper_hip_t* h, int64_t m, const int64_t* Mcolptr,
                           const int32_t* Mrow, const double* Mval, const int64_t* Ccolptr,
                           const int32_t* Crow, const double* Cval) try {
  if (!h || !Mcolptr || !Ccolptr || m < 1) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  // the caller's arrays are not trusted: structure first
  auto check_csc = [&](const char* what, const int64_t* cp, const int32_t* ri, const double* va) -> int {
    if (cp[0] != 0) return fail(CLIPPER_HIP_E_INVALID, "%s: colptr[0] must be 0", what);
    for (int64_t c = 0; c < m; ++c)
      if (cp[c + 1] < cp[c]) return fail(CLIPPER_HIP_E_INVALID, "%s: colptr decreases at column %lld", what, static_cast<long long>(c));
    const int64_t nnz = cp[m];
    if (nnz > 0 && (!ri || !va)) return fail(CLIPPER_HIP_E_INVALID, "%s: null CSC arrays", what);
    for (int64_t p = 0; p < nnz; ++p)
      if (ri[p] < 0 || ri[p] >= m)
        return fail(CLIPPER_HIP_E_INVALID, "%s: row index %d out of range at entry %lld", what, ri[p], static_cast<long long>(p));
    return 0;
  };
  int rc;
  if ((rc = check_csc("M", Mcolptr, Mrow, Mval))) return rc;
  if ((rc = check_csc("C", Ccolptr, Crow, Cval))) return rc;
  // The reference keeps what it is handed (clipper.cpp:162-166) and reads it through
  // selfadjointView<Eigen::Upper> (clipper.cpp:194-271): an entry BELOW the diagonal is never read — a
  // full symmetric SpAffinity counts through its upper half, a lower-triangular one is an empty matrix —
  // and a stored diagonal would count once on top of the implicit identity. Same here for the lower
  // triangle (dropped, whatever it holds); the diagonal is implicit in every storage of this library, so a
  // stored non-zero diagonal — outside the reference's own contract, clipper.h:137-138 — is refused
  // rather than silently dropped.
  struct Csc {
    std::vector<int64_t> cp;
    std::vector<int32_t> ri;
    std::vector<double> va;
  } Mn, Cn;
  int64_t dropped_below = 0;  // entries below the diagonal, which the reference never reads: dropped, and reported
  auto upper_only = [&](const char* what, const int64_t*& cp, const int32_t*& ri, const double*& va, Csc& out) -> int {
    bool strict = true;
    for (int64_t j = 0; j < m && strict; ++j)
      for (int64_t p = cp[j]; p < cp[j + 1]; ++p)
        if (ri[p] >= j) {
          strict = false;
          break;
        }
    if (strict) return 0;  // (what Eigen hands over from a strictly upper matrix: nothing to copy)
    out.cp.assign(static_cast<size_t>(m) + 1, 0);
    for (int64_t j = 0; j < m; ++j) {
      for (int64_t p = cp[j]; p < cp[j + 1]; ++p) {
        const int64_t i = ri[p];
        if (i > j) {
          ++dropped_below;
          continue;
        }
        if (i == j) {
          if (va[p] != 0.0)
            return fail(CLIPPER_HIP_E_INVALID, "%s: a stored diagonal entry (%lld,%lld) — the matrices must not have diagonal values set",
                        what, static_cast<long long>(i), static_cast<long long>(j));
          continue;
        }
        out.ri.push_back(static_cast<int32_t>(i));
        out.va.push_back(va[p]);
      }
      out.cp[static_cast<size_t>(j) + 1] = static_cast<int64_t>(out.ri.size());
    }
    cp = out.cp.data();
    ri = out.ri.data();
    va = out.va.data();
    return 0;
  };
  if ((rc = upper_only("M", Mcolptr, Mrow, Mval, Mn))) return rc;
  if ((rc = upper_only("C", Ccolptr, Crow, Cval, Cn))) return rc;
  // (a warning, not an error — the call goes on and returns 0 unless something else fails: clipper_hip_last_error()
  // tells a caller who handed over both triangles, or only the lower one, what became of them)
  if (dropped_below > 0)
    (void)fail(0, "warning: %lld stored entries below the diagonal were ignored (the matrices are read through their upper "
                  "triangle, as the reference's selfadjointView<Upper> does: clipper.cpp:194-271)", static_cast<long long>(dropped_below));
  const int64_t nnzM = Mcolptr[m], nnzC = Ccolptr[m];
  if (h->A.size() != static_cast<size_t>(2 * m)) h->A.clear();
  h->nodes.clear();
  h->has_matrix = false;
  h->csc_valid = false;
  h->total_slice_bytes = 0.0;
  h->fill_kind = 0;  // no points behind this matrix: its row views are filtered out of its own slices
  rowview_drop(h);
  if ((rc = ensure_problem(h, m))) return rc;
  h->has_matrix = false;
  h->csc_valid = false;
  // C == pattern(M)?  (same structure, every stored C equal to 1, every stored M non-zero)
  bool pattern = (nnzM == nnzC) && std::equal(Mcolptr, Mcolptr + m + 1, Ccolptr) &&
                 (nnzM == 0 || std::equal(Mrow, Mrow + nnzM, Crow));
  for (int64_t p = 0; pattern && p < nnzM; ++p) pattern = (Cval[p] == 1.0) && (Mval[p] != 0.0);
  h->explicitC = !pattern;
  plan_tiles(h);
  const int64_t W = h->W;
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    if (s.Cs) {
      hipFree(s.Cs);
      s.Cs = nullptr;
    }
  }
  struct Temps {
    std::vector<std::pair<int, void*>> v;
    ~Temps() {
      for (auto& p : v) {
        hipSetDevice(p.first);
        hipFree(p.second);
      }
    }
    int alloc(int dev, void** p, size_t n) {
      HIPCHK(hipMalloc(p, std::max<size_t>(n, 16)));
      v.emplace_back(dev, *p);
      return 0;
    }
  } tmp;

  if (csc_applies(h)) {
    // ---- lists -> slices. Host: the full symmetric lists (both triangles), rows ascending.
    drop_dense(h);
    std::vector<int64_t> cp(static_cast<size_t>(m) + 1, 0);
    for (int64_t j = 0; j < m; ++j)
      for (int64_t p = Mcolptr[j]; p < Mcolptr[j + 1]; ++p) {
        const int64_t i = Mrow[p];
        if (i == j) continue;
        ++cp[static_cast<size_t>(i) + 1];
        ++cp[static_cast<size_t>(j) + 1];
      }
    for (int64_t c = 0; c < m; ++c) cp[static_cast<size_t>(c) + 1] += cp[static_cast<size_t>(c)];
    const int64_t nnz2 = cp[static_cast<size_t>(m)];
    std::vector<int32_t> ri(static_cast<size_t>(nnz2));
    std::vector<double> va(static_cast<size_t>(nnz2));
    {
      std::vector<int64_t> cur(cp.begin(), cp.end() - 1);
      for (int64_t j = 0; j < m; ++j)
        for (int64_t p = Mcolptr[j]; p < Mcolptr[j + 1]; ++p) {
          const int64_t i = Mrow[p];
          if (i == j) continue;
          int64_t& a = cur[static_cast<size_t>(j)];
          ri[static_cast<size_t>(a)] = static_cast<int32_t>(i);
          va[static_cast<size_t>(a)] = Mval[p];
          ++a;
          int64_t& b = cur[static_cast<size_t>(i)];
          ri[static_cast<size_t>(b)] = static_cast<int32_t>(j);
          va[static_cast<size_t>(b)] = Mval[p];
          ++b;
        }
    }
    // strictly-upper input with ascending rows (what Eigen hands over) comes out sorted; anything
    // else is sorted here; an entry given twice (e.g. in both triangles) is an error
    std::vector<std::pair<int32_t, double>> buf;
    for (int64_t c = 0; c < m; ++c) {
      const int64_t a = cp[static_cast<size_t>(c)], b = cp[static_cast<size_t>(c) + 1];
      bool sorted = true;
      for (int64_t p = a + 1; p < b && sorted; ++p) sorted = ri[static_cast<size_t>(p - 1)] < ri[static_cast<size_t>(p)];
      if (sorted) continue;
      buf.clear();
      for (int64_t p = a; p < b; ++p) buf.emplace_back(ri[static_cast<size_t>(p)], va[static_cast<size_t>(p)]);
      std::sort(buf.begin(), buf.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      for (size_t q = 1; q < buf.size(); ++q)
        if (buf[q - 1].first == buf[q].first)
          return fail(CLIPPER_HIP_E_INVALID, "entry (%d,%lld) is stored more than once", buf[q].first,
                      static_cast<long long>(c));
      for (int64_t p = a; p < b; ++p) {
        ri[static_cast<size_t>(p)] = buf[static_cast<size_t>(p - a)].first;
        va[static_cast<size_t>(p)] = buf[static_cast<size_t>(p - a)].second;
      }
    }
    for (auto& s : h->sh) {
      HIPCHK(hipSetDevice(s.device));
      int64_t* dcp = nullptr;
      int32_t* dri = nullptr;
      double* dva = nullptr;
      if ((rc = tmp.alloc(s.device, reinterpret_cast<void**>(&dcp), cp.size() * sizeof(int64_t)))) return rc;
      if ((rc = tmp.alloc(s.device, reinterpret_cast<void**>(&dri), ri.size() * sizeof(int32_t)))) return rc;
      if ((rc = tmp.alloc(s.device, reinterpret_cast<void**>(&dva), va.size() * sizeof(double)))) return rc;
      HIPCHK(hipMemcpyAsync(dcp, cp.data(), cp.size() * sizeof(int64_t), hipMemcpyHostToDevice, s.stream));
      if (nnz2 > 0) {
        HIPCHK(hipMemcpyAsync(dri, ri.data(), ri.size() * sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
        HIPCHK(hipMemcpyAsync(dva, va.data(), va.size() * sizeof(double), hipMemcpyHostToDevice, s.stream));
      }
      const int64_t c0 = static_cast<int64_t>(s.slot) * W;
      dispatch_vt(h, [&](auto t) {
        using VT = decltype(t);
        bool again = true;
        for (int attempt = 0; again && !rc; ++attempt) {
          if (attempt >= 3) {
            rc = fail(CLIPPER_HIP_E_HIP, "compressed storage: the build keeps overflowing");
            break;
          }
          GroupOut<VT> unused;
          if ((rc = groups_prepare<VT>(h, s, unused))) break;  // sizes the per-slice arrays
          CscSource<VT> src{};
          src.colptr = dcp + std::min<int64_t>(c0, m);
          src.rowidx = dri;
          src.values = dva;
          src.ncols = std::max<int64_t>(0, std::min<int64_t>(W, m - c0));
          if ((rc = slices_enqueue<VT>(h, s, src, nullptr))) break;
          if (hipStreamSynchronize(s.stream) != hipSuccess) {
            rc = fail(CLIPPER_HIP_E_HIP, "set_sparse: %s", hipGetErrorString(hipGetLastError()));
            break;
          }
          rc = slices_check<VT>(h, s, false, again);
        }
      });
      if (rc) return rc;
    }
    if ((rc = sync_all(h))) return rc;
    h->csc_valid = true;
    if ((rc = gather_slice_bytes(h))) return rc;  // column shards: the row-view policy's cost model
    h->has_matrix = true;
    return 0;
  }

  // ---- through the dense store (dense storage modes, or an explicit C) ----------------------
  if ((rc = ensure_dense(h, false))) return rc;
  auto scatter = [&](Shard& s, void* dst, const int64_t* cp, const int32_t* ri, const double* va,
                     int64_t nnz) -> int {
    int64_t* dcp = nullptr;
    int32_t* dri = nullptr;
    double* dva = nullptr;
    int r;
    if ((r = tmp.alloc(s.device, reinterpret_cast<void**>(&dcp), static_cast<size_t>(m + 1) * sizeof(int64_t)))) return r;
    if ((r = tmp.alloc(s.device, reinterpret_cast<void**>(&dri), static_cast<size_t>(nnz) * sizeof(int32_t)))) return r;
    if ((r = tmp.alloc(s.device, reinterpret_cast<void**>(&dva), static_cast<size_t>(nnz) * sizeof(double)))) return r;
    HIPCHK(hipMemcpyAsync(dcp, cp, static_cast<size_t>(m + 1) * sizeof(int64_t),
                          hipMemcpyHostToDevice, s.stream));
    if (nnz > 0) {
      HIPCHK(hipMemcpyAsync(dri, ri, static_cast<size_t>(nnz) * sizeof(int32_t),
                            hipMemcpyHostToDevice, s.stream));
      HIPCHK(hipMemcpyAsync(dva, va, static_cast<size_t>(nnz) * sizeof(double),
                            hipMemcpyHostToDevice, s.stream));
    }
    HIPCHK(hipMemsetAsync(dst, 0, s.bytes_S, s.stream));
    const int64_t c0 = static_cast<int64_t>(s.slot) * W;
    dim3 grid(static_cast<unsigned>(std::min<int64_t>(m, 1 << 20))), block(256);
    if (h->storage == CLIPPER_HIP_STORE_F64)
      hipLaunchKernelGGL((k_from_csc<double>), grid, block, 0, s.stream,
                         static_cast<double*>(dst), W, m, c0, W, dcp, dri, dva);
    else
      hipLaunchKernelGGL((k_from_csc<float>), grid, block, 0, s.stream, static_cast<float*>(dst),
                         W, m, c0, W, dcp, dri, dva);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s.stream));
    return 0;
  };
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    rc = scatter(s, s.S, Mcolptr, Mrow, Mval, nnzM);
    if (rc) return rc;
    if (h->explicitC) {
      HIPCHK(hipMalloc(&s.Cs, s.bytes_S));
      rc = scatter(s, s.Cs, Ccolptr, Crow, Cval, nnzC);
      if (rc) return rc;
    }
  }
  rc = csc_rebuild(h);
  if (rc) return rc;
  h->has_matrix = true;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_get_matrix(clipper_hip_t* h, double* M_out, double* C_out) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
  if (h->multiproc)
    return fail(CLIPPER_HIP_E_STATE, "get_matrix is not available on a multi-process shard");
  const int64_t m = h->m, W = h->W;
  const bool f64 = (h->storage == CLIPPER_HIP_STORE_F64);
  if (int rc = ensure_dense(h, true)) return rc;
  std::vector<unsigned char> buf;
  auto fetch = [&](Shard& s, const void* src, double* out, bool as_pattern) -> int {
    buf.resize(s.bytes_S);
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipMemcpy(buf.data(), src, s.bytes_S, hipMemcpyDeviceToHost));
    const int64_t c0 = static_cast<int64_t>(s.slot) * W;
    for (int64_t c = 0; c < W; ++c) {
      const int64_t g = c0 + c;
      if (g >= m) break;
      for (int64_t j = 0; j < m; ++j) {
        const double v = f64 ? reinterpret_cast<const double*>(buf.data())[j * W + c]
                             : static_cast<double>(
                                   reinterpret_cast<const float*>(buf.data())[j * W + c]);
        double o = as_pattern ? ((v != 0.0) ? 1.0 : 0.0) : v;
        if (j == g) o += 1.0;  // clipper.cpp:133-134, 142-143: identity added
        out[j + g * m] = o;
      }
    }
    return 0;
  };
  for (auto& s : h->sh) {
    int rc;
    if (M_out && (rc = fetch(s, s.S, M_out, false))) return rc;
    if (C_out) {
      rc = h->explicitC ? fetch(s, s.Cs, C_out, false) : fetch(s, s.S, C_out, true);
      if (rc) return rc;
    }
  }
  if (h->csc_valid) drop_dense(h);  // the copy was materialised for this call only: M lives in the slices
  return 0;
} CLIPPER_HIP_GUARD_INT

// ---- solver ------------------------------------------------------------------------------

int clipper_hip_stage_u0(clipper_hip_t* h, const double* u0) try {
  if (!h || !u0) return fail(CLIPPER_HIP_E_INVALID, "u0 is required");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
  const size_t vbytes = static_cast<size_t>(h->m) * sizeof(double);
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipMemcpyAsync(s.u0, u0, vbytes, hipMemcpyHostToDevice, s.stream));
  }
  int rc = sync_all(h);
  if (rc) return rc;
  h->u0_staged = true;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_solve(clipper_hip_t* h, const double* u0, const clipper_params_t* P,
                      double* u_out, clipper_solve_info_t* info) try {
  if (!h || !u0 || !P) return fail(CLIPPER_HIP_E_INVALID, "u0 and params are required");
  const auto t0 = std::chrono::high_resolution_clock::now();
  int rc = clipper_hip_stage_u0(h, u0);
  if (rc) return rc;
  rc = clipper_hip_solve_staged(h, P, u_out, info);
  if (rc) return rc;
  const double secs =
      std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  h->tm.solve_total_ms = secs * 1e3;
  if (info) info->seconds = secs;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_solve_staged(clipper_hip_t* h, const clipper_params_t* P, double* u_out,
                             clipper_solve_info_t* info) try {
  if (!h || !P) return fail(CLIPPER_HIP_E_INVALID, "params are required");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
  if (!h->u0_staged) return fail(CLIPPER_HIP_E_STATE, "clipper_hip_stage_u0 not called");
  if (P->rounding == CLIPPER_ROUNDING_DSD && h->multiproc)
    return fail(CLIPPER_HIP_E_SCOPE,
                "Rounding::DSD needs the induced sub-matrix on one host: not available on a "
                "multi-process shard");
  if (P->rounding != CLIPPER_ROUNDING_NONZERO && P->rounding != CLIPPER_ROUNDING_DSD_HEU &&
      P->rounding != CLIPPER_ROUNDING_DSD)
    return fail(CLIPPER_HIP_E_INVALID, "unknown rounding mode %d", P->rounding);
  if (P->maxlsiters < 1) return fail(CLIPPER_HIP_E_INVALID, "maxlsiters must be >= 1");

  const auto t0 = std::chrono::high_resolution_clock::now();
  const int64_t m = h->m;
  const size_t vbytes = static_cast<size_t>(m) * sizeof(double);
  // CLIPPER_HIP_HOST_TIMING: where the host side of a solve goes (us since entry, to stderr)
  static const bool host_timing = std::getenv("CLIPPER_HIP_HOST_TIMING") != nullptr;
  std::vector<std::pair<const char*, double>> marks_t;
  auto mark_t = [&](const char* what) {
    if (host_timing)
      marks_t.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count());
  };

  h->rv_stats = clipper_hip_view_stats_t{};
  h->ev_used = 0;
  std::fill(h->ev_xchg_used.begin(), h->ev_xchg_used.end(), 0);
  if (h->profiling)  // marks of the previous solve
  {
    const size_t n = static_cast<size_t>(std::min<int64_t>(h->launch_counter + 1, KIND_CAP));
    std::memset(h->kind, 0, n);
    HIPCHK(hipSetDevice(h->sh[0].device));
    HIPCHK(hipMemsetAsync(h->sh[0].marks, 0, n, h->sh[0].stream));
  }
  h->launch_counter = 0;

  SolverParams prm;
  prm.tol_u = P->tol_u;
  prm.tol_F = P->tol_F;
  prm.beta = P->beta;
  prm.eps = P->eps;
  prm.maxiniters = P->maxiniters;
  prm.maxoliters = P->maxoliters;
  prm.maxlsiters = P->maxlsiters;

  SolverState init;
  std::memset(&init, 0, sizeof(init));
  init.alpha = 1.0;
  for (int l = 0; l < VS; ++l) init.nrm[l] = 1.0;
  init.nlive = init.nout = static_cast<int32_t>(std::min<int64_t>(m, 0x7fffffff));  // unknown until a tail counts
  init.rv_last = -100;
  init.weff = 0;
  init.zero_run = 2;  // (no line search has rejected anything yet: the first windows multiply candidate 0 alone)
  rowview_drop(h);  // a solve starts without a view: what it builds is a function of this solve alone
  rvr_begin_solve(h);
  sub_begin_solve(h);
  h->rvp = rowview_policy(h);
  // with rescaling the first iteration runs the pair pass on u0; without, it only normalises
  init.phase = P->rescale_u0 ? PH_RESCALE : PH_NORMALIZE;
  init.stage = P->rescale_u0 ? ST_PASS : ST_RESULTS;
  if (h->u_pinned_cap < vbytes) {
    if (h->u_pinned) hipHostFree(h->u_pinned);
    h->u_pinned = nullptr;
    h->u_pinned_dev = nullptr;
    h->u_pinned_cap = 0;
    HIPCHK(hipSetDevice(h->sh[0].device));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->u_pinned), vbytes,
                         hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->u_pinned_dev), h->u_pinned, 0));
    h->u_pinned_cap = vbytes;
  }
  Shard& s0 = h->sh[0];
  SolveShared fin;
  std::memset(&fin, 0, sizeof(fin));
  int rc = 0;
  bool resident = false;
  if ((rc = resident_solve(h, prm, P->rescale_u0 != 0, fin, resident))) return rc;
  h->last_solver = resident ? 1 : 0;
  if (!resident) {
  // prologue, one launch per shard: pending vector = u0 (T pair 0, nrm = 1), state, counters
  h->par = 0;
  std::memset(h->mirror, 0, sizeof(HostMirror));
  std::atomic_thread_fence(std::memory_order_seq_cst);
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    const SolveArgs a = solve_args(h, s, prm, 0);
    hipLaunchKernelGGL(k_init, dim3(static_cast<unsigned>(ceil_div(m, 256))), dim3(256), 0,
                       s.stream, a, init, s.st, s.X[0]);
  }
  mark_t("init queued");
  if (!h->multiproc) {
    // One process: the deciding workgroup reports progress into pinned host memory; the host
    // keeps RUN_AHEAD iterations queued ahead of what the device has retired and stops
    // queueing the moment `done` shows up — no memcpy, no event, no host wait in the loop.
    volatile HostMirror* hm = h->mirror;
    int64_t queued = 0;
    uint64_t spins = 0;
    h->rv_fresh = false;
    constexpr int run_ahead = RUN_AHEAD;
    h->enqueue_one = [h, &prm]() { return enqueue_iteration(h, prm); };
    // (while the solve runs on the live sub-problem the same launches go out with the child context's arguments)
    auto enqueue_next = [&]() { return enqueue_iteration(h->sub.active ? h->sub.use : h, prm); };
    auto sub_account = [&]() {  // the passes since the hand-over ran on the sub-problem
      h->sub.sub_passes += std::max<int64_t>(0, hm->n_passes - h->sub.passes_at_entry);
    };
    struct ClearEnqueue {
      Ctx* c;
      ~ClearEnqueue() { c->enqueue_one = nullptr; }
    } clear_enqueue{h};
    while (!hm->done) {
      if (hm->hold) {
        // The decision asked for a row view (k_solver.hip.h, LIVE ROWS) and put the solve on hold:
        // whatever was queued behind it does nothing. Drain, build the view from exactly the state
        // that asked, lift the hold, go on.
        mark_t("hold seen");
        if (host_timing)
          std::fprintf(stderr, "[solve] hold: iters %lld passes %lld trials %lld view passes %lld live %d of which outside the view %d\n",
                       static_cast<long long>(hm->iters), static_cast<long long>(hm->n_passes), static_cast<long long>(hm->n_trials),
                       static_cast<long long>(hm->n_view_passes), static_cast<int>(hm->hold_nlive), static_cast<int>(hm->nout));
        // (No drain: the iterations queued behind the hold do nothing but move the state between its two
        // copies, in stream order — the build's launches queue behind them and read copy h->par, where the last
        // of them leaves it; the build itself waits for the stream once. An in-process GROUP holds on every
        // shard: its streams are drained, the shards' builds are not ordered with each other otherwise.)
        if (h->sh.size() > 1 && (rc = sync_all(h))) return rc;
        std::atomic_thread_fence(std::memory_order_acquire);
        const int hold_reason = hm->hold;
        hm->hold = 0;
        queued = hm->iters;              // the iterations that did nothing never counted
        h->launch_counter = hm->iters;   // (profiling: launch index = the device's iteration count)
        while (h->ev_used > 0 && h->ev_launch_index[static_cast<size_t>(h->ev_used - 1)] >= hm->iters)
          --h->ev_used;                  // event pairs around launches that did nothing
        if (hold_reason == 2) {  // the hand-over to the live sub-problem (host_subproblem.hpp)
          if ((rc = sub_enter(h, prm))) return rc;
          ++queued;  // (its decide-only iteration)
          mark_t("sub-problem entered");
          continue;
        }
        if (hold_reason == 3) {  // ... and the way back
          sub_account();
          if ((rc = sub_leave(h))) return rc;
          mark_t("sub-problem left");
          continue;
        }
        bool built = false;
        if ((rc = rowview_build(h, built))) return rc;
        mark_t("view built");
        const bool early = h->early_decide_done;  // the decide-only iteration went out behind the fill (host_rowview.hpp):
        h->early_decide_done = false;             // the hold is lifted, rv_fresh was used by it
        if (early) ++queued;
        else h->rv_fresh = built;
        if (built && h->vres.ready) {
          // The view fits the LDS of the chip: the iterations on it run as ONE launch (k_rv_resident.hip.h).
          // A decide-only iteration turns the held decision into a prepared pass; the resident launch starts
          // from it and leaves a prepared pass (or the end of the solve) for whatever is queued behind it.
          if (!early) {
            h->decide_only = true;
            rc = enqueue_iteration(h, prm);
            h->decide_only = false;
            if (rc) return rc;
            ++queued;
          }
          bool launched = false;
          if ((rc = rvr_enqueue(h, prm, launched))) return rc;
          mark_t("resident queued");
        }
        if ((rc = rowview_finish_plan(h))) return rc;  // (a work list put off while the resident launch was prepared)
        // a view the resident solver does not take: the live sub-problem of its rows is prepared now, and entered
        // when the decision finds that nothing outside it can come back to life
        if (built && !h->vres.ready && !early) {
          // (an optimisation: if it cannot be prepared — no memory for the child's buffers — the solve goes on without)
          if (int r2 = sub_prepare(h)) {
            if (r2 != CLIPPER_HIP_E_NOMEM) return r2;
            (void)hipGetLastError();
            h->sub.ready = false;
          }
          mark_t("sub-problem prepared");
        }
        continue;
      }
      if (hm->iters > queued) queued = hm->iters;  // (a resident launch retired many iterations at once)
      if (queued - hm->iters < run_ahead) {
        if ((rc = enqueue_next())) return rc;
        ++queued;
        spins = 0;
      } else if ((++spins & 0xfffff) == 0) {
        // the device has not retired an iteration for a long time: make sure it is still alive
        hipError_t q = hipStreamQuery(s0.stream);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(CLIPPER_HIP_E_HIP, "solver stream failed: %s", hipGetErrorString(q));
        if (q == hipSuccess && !hm->done && queued - hm->iters >= run_ahead)
          return fail(CLIPPER_HIP_E_HIP, "solver made no progress (iters %lld of %lld queued)",
                      static_cast<long long>(hm->iters), static_cast<long long>(queued));
      }
    }
    mark_t("done seen");
    std::atomic_thread_fence(std::memory_order_acquire);
    if (h->sub.active) {  // the solve ended on the live sub-problem (its deciding workgroup wrote u through the list of S)
      sub_account();
      h->sub.active = false;
    }
    fin.F = hm->F;
    fin.d = hm->d;
    fin.n_passes = hm->n_passes;
    fin.n_trials = hm->n_trials;
    fin.ifinal = hm->ifinal;
    fin.ubp = hm->ubp;
    fin.ubv = hm->ubv;
    h->rv_stats.view_passes = hm->n_view_passes;
  } else {
    // Multi-process: every rank must queue the same number of iterations (each holds a
    // collective), so the decision to stop rests on state snapshots only, which are
    // bit-identical on all ranks. Batch n+1 is queued before the snapshot after batch n is read.
    int batch = SOLVE_BATCH;
    if (const char* e = std::getenv("CLIPPER_HIP_SOLVE_BATCH")) batch = std::max(1, std::atoi(e));  // tuning knob, same on every rank
    HIPCHK(hipSetDevice(s0.device));
    h->rv_fresh = false;
    rc = run_batched_with_holds(
        batch, [&]() { return enqueue_iteration(h, prm); },
        [&](int slot) -> int {
          HIPCHK(hipSetDevice(s0.device));
          HIPCHK(hipMemcpyAsync(&h->host_state[slot], s0.shared, sizeof(SolveShared),
                                hipMemcpyDeviceToHost, s0.stream));
          HIPCHK(hipEventRecord(h->ev_poll[slot], s0.stream));
          return 0;
        },
        [&](int slot, int& st) -> int {
          HIPCHK(hipEventSynchronize(h->ev_poll[slot]));
          st = h->host_state[slot].done != 0 ? 1 : (h->host_state[slot].hold != 0 ? 2 : 0);
          return 0;
        },
        [&]() -> int {
          // every rank reads the hold from the same snapshot: all of them have queued the same
          // iterations, so draining cannot wait for a peer; then each builds its columns of the view
          if (int r2 = sync_all(h)) return r2;
          h->mirror->hold = 0;
          h->launch_counter = h->mirror->iters;
          while (h->ev_used > 0 && h->ev_launch_index[static_cast<size_t>(h->ev_used - 1)] >= h->mirror->iters)
            --h->ev_used;  // event pairs around launches that did nothing
          bool built = false;
          if (int r2 = rowview_build(h, built)) return r2;
          h->rv_fresh = built;
          // a view small enough for the resident solver: every rank runs it on a replica of the view (no exchange
          // for the iterations inside the launch; host_rv_resident.hpp)
          if (built)
            if (int r2 = rvr_replica_handover(h, prm)) return r2;
          return 0;
        },
        nullptr);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipMemcpy(&fin, s0.shared, sizeof(fin), hipMemcpyDeviceToHost));
    h->rv_stats.view_passes = h->mirror->n_view_passes;
  }

  rvr_end_solve(h);  // (launches of the resident solver on a view that gave up: counted, the context backs off)
  }  // !resident

  // final u
  HIPCHK(hipSetDevice(s0.device));
  if (h->multiproc) {
    const double* u_dev =
        s0.pt + ((static_cast<int64_t>(fin.ubp & 1) * h->V + fin.ubv) * 2 + 0) * h->mp;
    HIPCHK(hipMemcpyAsync(h->u_pinned, u_dev, vbytes, hipMemcpyDeviceToHost, s0.stream));
    if (h->profiling)
      HIPCHK(hipMemcpyAsync(h->kind, s0.marks,
                            static_cast<size_t>(std::min<int64_t>(h->launch_counter, KIND_CAP)),
                            hipMemcpyDeviceToHost, s0.stream));
    if ((rc = sync_all(h))) return rc;
  } else {
    // one process: the deciding workgroup wrote u into the pinned buffer before it raised `done`;
    // the few no-op launches still queued drain behind the caller's back (stream order keeps
    // every later call behind them)
    HIPCHK(hipGetLastError());
  }
  std::vector<double>& u = h->u_host;  // (kept from solve to solve: no allocation on the way out)
  u.assign(h->u_pinned, h->u_pinned + m);

  // rounding — clipper.cpp:287-310 with utils.cpp:33-68, on the host
  std::vector<int32_t> nodes;
  if (P->rounding == CLIPPER_ROUNDING_NONZERO) {
    for (int64_t i = 0; i < m; ++i)
      if (u[static_cast<size_t>(i)] > 0.0) nodes.push_back(static_cast<int32_t>(i));
  } else if (P->rounding == CLIPPER_ROUNDING_DSD) {
    // :294-300 — exact densest subgraph of the graph induced by the non-zero entries of u
    std::vector<int32_t> S;
    for (int64_t i = 0; i < m; ++i)
      if (u[static_cast<size_t>(i)] > 0.0) S.push_back(static_cast<int32_t>(i));
    if ((rc = densest_subgraph_of(h, S, nodes))) return rc;
  } else {
    const int omega = static_cast<int>(std::round(fin.F));  // :305
    nodes = indices_of_k_largest(u, omega);                 // :308
  }
  h->nodes = nodes;
  if (u_out) std::memcpy(u_out, u.data(), vbytes);
  mark_t("rounded");
  if (host_timing) {
    std::fprintf(stderr, "[solve]");
    for (const auto& mk : marks_t) std::fprintf(stderr, " %s %.1f |", mk.first, mk.second);
    std::fprintf(stderr, "\n");
  }

  const double secs =
      std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  h->tm.solve_total_ms = secs * 1e3;
  if (info) {
    info->score = fin.F;
    info->seconds = secs;
    info->d = fin.d;
    info->ifinal = fin.ifinal;
    info->num_nodes = static_cast<int32_t>(nodes.size());
    info->n_passes = fin.n_passes;
    info->n_trials = fin.n_trials;
  }
  h->rv_stats.passes = fin.n_passes;
  h->rv_stats.sub_leaves = h->sub.leaves;
  h->rv_stats.sub_passes = h->sub.sub_passes;
  h->rv_stats.sub_build_ms = h->sub.build_ms;
  if (h->rv_stats.sub_entries > 0 && h->sub.use) {
    h->rv_stats.sub_rows = h->sub.nS;
    // what a pass on it streams: the slices, or (a mostly non-zero sub-problem) the dense fp32 store
    h->rv_stats.sub_bytes = h->sub.use->csc_valid ? static_cast<int64_t>(h->sub.use->sh[0].s_bytes)
                                                   : static_cast<int64_t>(algorithmic_gemv_bytes(h->sub.use));
    h->rv_stats.sub_dense = h->sub.use->csc_valid ? 0 : 1;
  }

  // mat-vec timings from the event pairs
  h->tm.gemv_avg_us = h->tm.gemv_min_us = 0.0;
  h->tm.gemv_launches = 0;
  h->tm.gemv_bytes = algorithmic_gemv_bytes(h);
  h->tm.gemv_useful_bytes = h->csc_valid ? static_cast<double>(h->sh[0].s_entries) * (h->esize() + 1.0) : h->tm.gemv_bytes;
  if (h->profiling && h->ev_used > 0) {
    // only launches that streamed M count: the device marked every iteration as pass (1) or
    // transition (0); launches queued past convergence have no mark
    const uint8_t* kind = h->kind;
    const int64_t iters_run = std::min<int64_t>(h->launch_counter, KIND_CAP);
    double sum = 0.0, mn = 1e30, vsum = 0.0, ssum = 0.0;
    int64_t nreal = 0, nview = 0, nsub = 0;
    for (int k = 0; k < h->ev_used; ++k) {
      const int64_t li = h->ev_launch_index[static_cast<size_t>(k)];
      if (li >= iters_run || !kind[static_cast<size_t>(li)]) continue;
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, h->ev_pairs[2 * k], h->ev_pairs[2 * k + 1]));
      if (kind[static_cast<size_t>(li)] == 2) {  // the launch streamed the row view, not M
        vsum += ms;
        ++nview;
        continue;
      }
      if (kind[static_cast<size_t>(li)] == 4) {  // a window pass on the live sub-problem
        ssum += ms;
        ++nsub;
        continue;
      }
      if (kind[static_cast<size_t>(li)] == 3 || kind[static_cast<size_t>(li)] == 5) continue;  // a pair-mode pass (one vector): not the window pass the roofline is about
      sum += ms;
      mn = std::min<double>(mn, ms);
      ++nreal;
    }
    // the exchanges of the same iterations (column shards)
    h->tm.exchange_avg_us = 0.0;
    h->tm.exchange_samples = 0;
    h->tm.exchange_bytes = static_cast<double>(nslot(h->V)) * static_cast<double>(h->W) * sizeof(double);
    {
      double xs = 0.0;
      int64_t nx = 0;
      for (int k = 0; k < h->ev_used; ++k) {
        const int64_t li = h->ev_launch_index[static_cast<size_t>(k)];
        if (!h->ev_xchg_used[static_cast<size_t>(k)] || li >= iters_run || !kind[static_cast<size_t>(li)]) continue;
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_xchg[2 * k], h->ev_xchg[2 * k + 1]));
        xs += ms;
        ++nx;
      }
      if (nx > 0) {
        h->tm.exchange_avg_us = xs / static_cast<double>(nx) * 1e3;
        h->tm.exchange_samples = nx;
      }
    }
    if (nview > 0) {
      h->rv_stats.view_pass_avg_us = vsum / static_cast<double>(nview) * 1e3;
      h->rv_stats.view_pass_samples = nview;
    }
    if (nsub > 0) {
      h->rv_stats.sub_pass_avg_us = ssum / static_cast<double>(nsub) * 1e3;
      h->rv_stats.sub_pass_samples = nsub;
    }
    if (nreal > 0) {
      h->tm.gemv_avg_us = sum / static_cast<double>(nreal) * 1e3;
      h->tm.gemv_min_us = mn * 1e3;
      h->tm.gemv_launches = nreal;
    }
  }
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_get_nodes(const clipper_hip_t* h, int32_t* out, int32_t capacity) try {
  if (!h || !out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  const int32_t k = static_cast<int32_t>(h->nodes.size());
  if (capacity < k) return fail(CLIPPER_HIP_E_INVALID, "capacity %d < %d nodes", capacity, k);
  if (k) std::memcpy(out, h->nodes.data(), static_cast<size_t>(k) * sizeof(int32_t));
  return k;
} CLIPPER_HIP_GUARD_INT

// utils::selectInlierAssociations — utils.cpp:101-108
int clipper_hip_get_selected_associations(const clipper_hip_t* h, int32_t* A_out,
                                          int32_t capacity) try {
  if (!h || !A_out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  const int32_t k = static_cast<int32_t>(h->nodes.size());
  if (capacity < k) return fail(CLIPPER_HIP_E_INVALID, "capacity %d < %d nodes", capacity, k);
  if (k == 0) return 0;
  if (h->A.size() != static_cast<size_t>(2 * h->m))
    return fail(CLIPPER_HIP_E_STATE, "no association list is held");
  for (int32_t r = 0; r < k; ++r) {
    const size_t n = static_cast<size_t>(h->nodes[static_cast<size_t>(r)]);
    A_out[r] = h->A[n];
    A_out[k + r] = h->A[static_cast<size_t>(h->m) + n];
  }
  return k;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_densest_subgraph(clipper_hip_t* h, const int32_t* S, int32_t k, int32_t* nodes_out,
                                 int32_t capacity) try {
  if (!h || !nodes_out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
  if (h->multiproc)
    return fail(CLIPPER_HIP_E_SCOPE, "not available on a multi-process shard");
  std::vector<int32_t> sub;
  if (S == nullptr || k <= 0) {  // dsd.cpp:279-284: the whole graph
    sub.resize(static_cast<size_t>(h->m));
    for (int64_t i = 0; i < h->m; ++i) sub[static_cast<size_t>(i)] = static_cast<int32_t>(i);
  } else {
    sub.assign(S, S + k);
    for (int32_t v : sub)
      if (v < 0 || v >= h->m) return fail(CLIPPER_HIP_E_INVALID, "node %d out of range", v);
  }
  std::vector<int32_t> nodes;
  int rc = densest_subgraph_of(h, sub, nodes);
  if (rc) return rc;
  const int32_t n = static_cast<int32_t>(nodes.size());
  if (capacity < n) return fail(CLIPPER_HIP_E_INVALID, "capacity %d < %d nodes", capacity, n);
  if (n) std::memcpy(nodes_out, nodes.data(), static_cast<size_t>(n) * sizeof(int32_t));
  return n;
} CLIPPER_HIP_GUARD_INT

// ---- putative associations (before the path): brute-force nearest neighbours -------------------

int clipper_hip_knn(int device, const double* P0, int64_t n0, const double* P1, int64_t n1, int d,
                    int knn, int32_t* idx_out, double* sqd_out) try {
  if (!P0 || !P1 || !idx_out || n0 < 1 || n1 < 1) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (d != 2 && d != 3) return fail(CLIPPER_HIP_E_INVALID, "points must have 2 or 3 coordinates");
  if (knn < 1 || knn > 16) return fail(CLIPPER_HIP_E_INVALID, "knn must be in 1..16");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(CLIPPER_HIP_E_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(CLIPPER_HIP_E_INVALID, "device %d out of range", device);
  HIPCHK(hipSetDevice(device));
  const int K = knn <= 1 ? 1 : (knn <= 2 ? 2 : (knn <= 4 ? 4 : (knn <= 8 ? 8 : 16)));
  // enough workgroups to fill the chip: split pcd1 into S chunks of whole tiles
  const int64_t qblocks = ceil_div(n0, 256);
  const int64_t tiles = ceil_div(n1, KNN_TILE);
  const int S = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(tiles, ceil_div(512, qblocks))));
  const int64_t chunk = ceil_div(tiles, S) * KNN_TILE;
  double *dP0 = nullptr, *dP1 = nullptr, *pd = nullptr, *od = nullptr;
  int32_t *pi = nullptr, *oi = nullptr;
  auto cleanup = [&]() {
    hipFree(dP0); hipFree(dP1); hipFree(pd); hipFree(od); hipFree(pi); hipFree(oi);
  };
  const size_t b0 = static_cast<size_t>(n0) * d * sizeof(double), b1 = static_cast<size_t>(n1) * d * sizeof(double);
  const size_t np = static_cast<size_t>(S) * n0 * K, no = static_cast<size_t>(n0) * K;
  if (hipMalloc(&dP0, b0) != hipSuccess || hipMalloc(&dP1, b1) != hipSuccess ||
      hipMalloc(&pd, np * sizeof(double)) != hipSuccess || hipMalloc(&pi, np * sizeof(int32_t)) != hipSuccess ||
      hipMalloc(&od, no * sizeof(double)) != hipSuccess || hipMalloc(&oi, no * sizeof(int32_t)) != hipSuccess) {
    cleanup();
    return fail(CLIPPER_HIP_E_NOMEM, "device allocation failed");
  }
  hipStream_t st = nullptr;  // the default stream: a stand-alone call
  bool ok = hipMemcpyAsync(dP0, P0, b0, hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(dP1, P1, b1, hipMemcpyHostToDevice, st) == hipSuccess;
  if (ok) {
#define KNN_CASE(KK)                                                                        \
  case KK:                                                                                  \
    if (d == 3) knn_run<KK, 3>(dP0, n0, dP1, n1, S, chunk, pd, pi, od, oi, st);             \
    else knn_run<KK, 2>(dP0, n0, dP1, n1, S, chunk, pd, pi, od, oi, st);                    \
    break
    switch (K) {
      KNN_CASE(1);
      KNN_CASE(2);
      KNN_CASE(4);
      KNN_CASE(8);
      default:
        if (d == 3) knn_run<16, 3>(dP0, n0, dP1, n1, S, chunk, pd, pi, od, oi, st);
        else knn_run<16, 2>(dP0, n0, dP1, n1, S, chunk, pd, pi, od, oi, st);
        break;
    }
#undef KNN_CASE
    std::vector<double> hd(no);
    std::vector<int32_t> hi(no);
    ok = hipMemcpy(hd.data(), od, no * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(hi.data(), oi, no * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess &&
         hipGetLastError() == hipSuccess;
    if (ok) {
      for (int64_t i = 0; i < n0; ++i)
        for (int k = 0; k < knn; ++k) {
          idx_out[i * knn + k] = hi[static_cast<size_t>(i) * K + k];
          if (sqd_out) sqd_out[i * knn + k] = hd[static_cast<size_t>(i) * K + k];
        }
    }
  }
  cleanup();
  if (!ok) return fail(CLIPPER_HIP_E_HIP, "nearest-neighbour search failed: %s", hipGetErrorString(hipGetLastError()));
  return 0;
} CLIPPER_HIP_GUARD_INT

int64_t clipper_hip_distance_based_correspondences(int device, const double* P0, int64_t n0,
                                                   const double* P1, int64_t n1, int d, int knn,
                                                   double radius, int enforce_1to1, int32_t* A_out,
                                                   int64_t capacity) try {
  if (!A_out && capacity > 0) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  std::vector<int32_t> idx(static_cast<size_t>(std::max<int64_t>(n0, 0)) * std::max(knn, 0));
  std::vector<double> sqd(idx.size());
  int rc = clipper_hip_knn(device, P0, n0, P1, n1, d, knn, idx.data(), sqd.data());
  if (rc) return rc;
  // bm_utils.cpp:187-229: rows (i, nn_j(i)) for i ascending, neighbours by distance, kept if within
  // the radius; one-to-one: per point of pcd1 (ascending) the FIRST closest of its claimants
  const double r2 = radius * radius;
  std::vector<std::pair<int32_t, int32_t>> rows;
  std::map<int32_t, std::vector<std::pair<int32_t, double>>> claim;
  for (int64_t i = 0; i < n0; ++i)
    for (int k = 0; k < knn; ++k) {
      const int32_t c1 = idx[static_cast<size_t>(i) * knn + k];
      const double sd = sqd[static_cast<size_t>(i) * knn + k];
      if (c1 < 0) continue;  // fewer than knn points in pcd1
      if (sd <= r2) {
        rows.emplace_back(static_cast<int32_t>(i), c1);
        if (enforce_1to1) claim[c1].emplace_back(static_cast<int32_t>(i), sd);
      }
    }
  if (enforce_1to1) {
    rows.clear();
    for (const auto& it : claim) {
      size_t best = 0;
      for (size_t q = 1; q < it.second.size(); ++q)
        if (it.second[q].second < it.second[best].second) best = q;  // std::min_element: first minimum
      rows.emplace_back(it.second[best].first, it.first);
    }
  }
  const int64_t n = static_cast<int64_t>(rows.size());
  if (n > capacity) return fail(CLIPPER_HIP_E_INVALID, "capacity %lld < %lld associations",
                                static_cast<long long>(capacity), static_cast<long long>(n));
  for (int64_t r = 0; r < n; ++r) {  // column-major n x 2, as clipper::Association
    A_out[r] = rows[static_cast<size_t>(r)].first;
    A_out[n + r] = rows[static_cast<size_t>(r)].second;
  }
  return n;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_set_window(clipper_hip_t* h, int window) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (window != 0 && window != 1 && window != 4 && window != 6 && window != 8)
    return fail(CLIPPER_HIP_E_INVALID, "window must be 0 (automatic), 1, 4, 6 or 8");
  h->V_forced = window;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_window(const clipper_hip_t* h) try {
  return h ? h->V : 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_set_resident(clipper_hip_t* h, int mode) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (mode != 0 && mode != 1 && mode != 2)
    return fail(CLIPPER_HIP_E_INVALID, "mode must be 0 (automatic), 1 (never) or 2 (the sub-problem always as slices)");
  h->resident_mode = mode;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_last_solver(const clipper_hip_t* h) try {
  return h ? h->last_solver : -1;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_set_row_view(clipper_hip_t* h, int mode) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (mode != 0 && mode != 1 && mode != 2)
    return fail(CLIPPER_HIP_E_INVALID, "mode must be 0 (automatic), 1 (never) or 2 (views, but streamed: never the resident solver on one)");
  h->rv_mode = mode;
  if (mode == 1) rowview_drop(h);
  if (mode != 0) h->vres.ready = false;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_set_subproblem(clipper_hip_t* h, int mode) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (mode != 0 && mode != 1 && mode != 2)
    return fail(CLIPPER_HIP_E_INVALID, "mode must be 0 (automatic), 1 (never) or 2 (the sub-problem always as slices)");
  h->sub_mode = mode;
  if (mode == 1) h->sub.ready = false;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_get_view_stats(const clipper_hip_t* h, clipper_hip_view_stats_t* out) try {
  if (!h || !out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  *out = h->rv_stats;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_storage_in_use(const clipper_hip_t* h) try {
  if (!h) return -1;
  if (!h->csc_valid) return h->storage;
  return h->storage == CLIPPER_HIP_STORE_F64 ? CLIPPER_HIP_STORE_F64_CSC : CLIPPER_HIP_STORE_F32_CSC;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_matvec(clipper_hip_t* h, const double* x, double* yM, double* yC) try {
  if (!h || !x) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
  const int64_t m = h->m, W = h->W;
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    // x -> candidate 0 of table 0 (staged through the u0 buffer)
    HIPCHK(hipMemcpyAsync(s.u0, x, static_cast<size_t>(m) * sizeof(double),
                          hipMemcpyHostToDevice, s.stream));
    hipLaunchKernelGGL(k_spread, dim3(static_cast<unsigned>(ceil_div(m, 256))), dim3(256), 0,
                       s.stream, s.u0, m, s.X[0]);
  }
  h->u0_staged = false;
  int rc = h->csc_valid ? 0 : ensure_dense(h, true);
  if (rc) return rc;
  if ((rc = enqueue_gemv_plain(h))) return rc;
  if ((rc = enqueue_reduce_exchange(h))) return rc;
  if ((rc = sync_all(h))) return rc;
  std::vector<double> ab(static_cast<size_t>(h->world) * 2 * W);
  Shard& s0 = h->sh[0];
  HIPCHK(hipSetDevice(s0.device));
  HIPCHK(hipMemcpy(ab.data(), s0.ab, ab.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < m; ++i) {
    const int64_t p = i / W, off = i - p * W;
    if (yM) yM[i] = ab[static_cast<size_t>(p * 2 * W + off)];
    if (yC) yC[i] = ab[static_cast<size_t>(p * 2 * W + W + off)];
  }
  return 0;
} CLIPPER_HIP_GUARD_INT

// The products of clipper_hip_matvec through a ROW VIEW of the given rows: yM = M_off[:, rows] x[rows],
// yC likewise — what a pass of the solver computes when it streams the view instead of M. Builds the
// slices of M[rows, :] with the rectangular fill kernel from the staged points (the view of a later
// solve is built anew). For tests: equal to clipper_hip_matvec of x with every other entry zeroed, up
// to the order of the partial sums.
int clipper_hip_view_matvec(clipper_hip_t* h, const int32_t* rows, int64_t nrows, const double* x,
                            double* yM, double* yC) try {
  if (!h || !rows || !x || nrows < 1) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->has_matrix || !h->csc_valid || !csc_single(h) || !rect_fill_possible(h))
    return fail(CLIPPER_HIP_E_STATE, "a row view needs slices of a matrix scored from staged points on one device");
  const int64_t m = h->m, W = h->W;
  for (int64_t r = 0; r < nrows; ++r)
    if (rows[r] < 0 || rows[r] >= m || (r > 0 && rows[r] <= rows[r - 1]))
      return fail(CLIPPER_HIP_E_INVALID, "rows must be ascending association indices");
  Shard& s = h->sh[0];
  RowView& v = s.rv;
  HIPCHK(hipSetDevice(s.device));
  HIPCHK(hipStreamSynchronize(s.stream));
  v.valid = false;
  int rc;
  {
    size_t r0 = v.cap_rows, r1 = v.cap_rows;
    if ((rc = rv_grow(v.rowmap[0], r0, static_cast<size_t>(h->mp)))) return rc;
    if ((rc = rv_grow(v.rowmap[1], r1, static_cast<size_t>(h->mp)))) return rc;
    v.cap_rows = static_cast<size_t>(h->mp);
  }
  HIPCHK(hipMemcpyAsync(v.rowmap[0], rows, static_cast<size_t>(nrows) * sizeof(int32_t), hipMemcpyHostToDevice, s.stream));
  for (int attempt = 0;; ++attempt) {
    SliceOut O{};
    if ((rc = emit_prepare(h, s, v.st, nrows, O))) return rc;
    if ((rc = launch_rect(h, s, v.rowmap[0], nrows, O))) return rc;
    if ((rc = emit_enqueue(h, s, v.st))) return rc;
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipGetLastError());
    bool again = false;
    if ((rc = emit_check(h, s, v.st, false, again))) return rc;
    if (!again) break;
    if (attempt >= 2) return fail(CLIPPER_HIP_E_HIP, "row view: the build keeps overflowing");
  }
  HIPCHK(hipMemcpyAsync(s.u0, x, static_cast<size_t>(m) * sizeof(double), hipMemcpyHostToDevice, s.stream));
  hipLaunchKernelGGL(k_spread, dim3(static_cast<unsigned>(ceil_div(m, 256))), dim3(256), 0, s.stream, s.u0, m, s.X[0]);
  h->u0_staged = false;
  SliceView R{};
  R.data = v.st.sdata;
  R.Pre = v.st.sPre;
  R.work = v.st.swork;
  R.nchunks = v.st.s_nchunks;
  R.ncg = v.st.s_ncg;
  R.nwork = v.st.s_nwork;
  R.rowmap = v.rowmap[0];
  R.nrows = nrows;
  R.pad = 0;
  dim3 grid(static_cast<unsigned>(R.nwork)), block(SL_NW * 64);
  if (h->storage == CLIPPER_HIP_STORE_F64)
    hipLaunchKernelGGL((k_gemv_slices_plain<double, 1>), grid, block, 0, s.stream, R, W, m, s.X[0], s.part);
  else
    hipLaunchKernelGGL((k_gemv_slices_plain<float, 1>), grid, block, 0, s.stream, R, W, m, s.X[0], s.part);
  hipLaunchKernelGGL(k_reduce, dim3(static_cast<unsigned>(ceil_div(2 * W, 256))), dim3(256), 0, s.stream, s.part,
                     v.st.s_nslots, 2, W, s.ab);
  std::vector<double> ab(static_cast<size_t>(2 * W));
  HIPCHK(hipMemcpyAsync(ab.data(), s.ab, ab.size() * sizeof(double), hipMemcpyDeviceToHost, s.stream));
  HIPCHK(hipStreamSynchronize(s.stream));
  HIPCHK(hipGetLastError());
  for (int64_t i = 0; i < m; ++i) {
    if (yM) yM[i] = ab[static_cast<size_t>(i)];
    if (yC) yC[i] = ab[static_cast<size_t>(W + i)];
  }
  return 0;
} CLIPPER_HIP_GUARD_INT

// ---- measurement ---------------------------------------------------------------------------

int clipper_hip_set_profiling(clipper_hip_t* h, int on) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  h->profiling = (on != 0);
  h->profiling_level = on;  // 2: also an event pair around every launch of the resident solver on a view
  if (h->profiling && h->ev_pairs.empty()) {  // here, not inside the first profiled solve (~1 ms)
    HIPCHK(hipSetDevice(h->sh[0].device));
    h->ev_pairs.resize(2 * MAX_EVENT_PAIRS);
    h->ev_launch_index.assign(MAX_EVENT_PAIRS, 0);
    for (auto& e : h->ev_pairs) HIPCHK(hipEventCreate(&e));
    h->ev_xchg.resize(2 * MAX_EVENT_PAIRS);
    h->ev_xchg_used.assign(MAX_EVENT_PAIRS, 0);
    for (auto& e : h->ev_xchg) HIPCHK(hipEventCreate(&e));
  }
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_get_timings(const clipper_hip_t* h, clipper_hip_timings_t* out) try {
  if (!h || !out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  *out = h->tm;
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_bench_matvec(clipper_hip_t* h, int reps, double* avg_us) try {
  if (!h || reps < 1 || !avg_us) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
  if (!h->csc_valid)
    if (int rc = ensure_dense(h, true)) return rc;
  Shard& s = h->sh[0];
  HIPCHK(hipSetDevice(s.device));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) launch_plain(h, s, s.X[0]);
  HIPCHK(hipEventRecord(e0, s.stream));
  for (int r = 0; r < reps; ++r) launch_plain(h, s, s.X[0]);
  HIPCHK(hipEventRecord(e1, s.stream));
  HIPCHK(hipStreamSynchronize(s.stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *avg_us = static_cast<double>(ms) * 1e3 / reps;
  h->tm.gemv_bytes = algorithmic_gemv_bytes(h, /*dense=*/!h->csc_valid);
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_debug_stamps(clipper_hip_t* h, int64_t* out, int capacity) try {
  if (!h || !out || capacity < 0) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->stamps_dev) return fail(CLIPPER_HIP_E_STATE, "CLIPPER_HIP_STAMPS was not set when the context was created");
  const int n = std::min(capacity, h->stamps_rows * 4);
  HIPCHK(hipSetDevice(h->sh[0].device));
  HIPCHK(hipStreamSynchronize(h->sh[0].stream));
  HIPCHK(hipMemcpy(out, h->stamps_dev, static_cast<size_t>(n) * sizeof(long long), hipMemcpyDeviceToHost));
  return n;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_debug_occupy(int device, int workgroups, int lds_bytes, double milliseconds) try {
  if (workgroups < 1 || lds_bytes < 0 || lds_bytes > static_cast<int>(RS_LDS_MAX) || !(milliseconds >= 0.0) || milliseconds > 2000.0)
    return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  HIPCHK(hipSetDevice(device));
  if (!raise_dynamic_lds(reinterpret_cast<const void*>(k_debug_occupy), device, static_cast<int>(RS_LDS_MAX)))
    return fail(CLIPPER_HIP_E_HIP, "the device refuses %u bytes of dynamic LDS", RS_LDS_MAX);
  hipStream_t st = nullptr;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_debug_occupy, dim3(static_cast<unsigned>(workgroups)), dim3(64), static_cast<size_t>(lds_bytes), st,
                     static_cast<long long>(milliseconds * 1e5));
  const hipError_t e = hipStreamSynchronize(st);
  hipStreamDestroy(st);
  HIPCHK(e);
  return 0;
} CLIPPER_HIP_GUARD_INT

int clipper_hip_device_info(const clipper_hip_t* h, char* name64, int* cus, int64_t* hbm_bytes) try {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, h->sh[0].device));
  if (name64) {
    std::snprintf(name64, 64, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cus) *cus = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
  return 0;
} CLIPPER_HIP_GUARD_INT

}  // extern "C"
