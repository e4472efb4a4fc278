// clipper_hip.hip — host side of the C ABI declared in include/clipper_hip.h.
//
// Owns device memory, streams and the solve loop; all arithmetic runs in the kernels of
// kernels.hip.h. There is no CPU fallback anywhere in this file: if HIP is unusable the
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
#include <vector>

#include "../../include/clipper_hip.h"
#include "kernels.hip.h"
#include "dsd_host.h"

using namespace clipper_hip;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(e_ == hipErrorOutOfMemory ? CLIPPER_HIP_E_NOMEM : CLIPPER_HIP_E_HIP, \
                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                  __LINE__);                                                          \
  } while (0)

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// mat-vec kernel geometry (tuned on MI355X)
constexpr int GEMV_NW = 8;      // waves per workgroup
constexpr int GEMV_WG_PER_CU = 2;
// rows in flight per wave: 8 x 16 B per lane for fp32 storage and windows up to 6 vectors (one
// accumulator set per candidate: 48 registers at V = 6); the 8-vector window keeps 4 rows in
// flight (tools/mv_tune.hip); halved for fp64 storage (32 B per lane and row) and again with an
// explicit C matrix
constexpr int gemv_unr(int V, int esize, bool hasc) {
  int u = (V <= 6) ? 8 : 4;
  if (esize == 8) u /= 2;
  if (hasc) u /= 2;
  return u < 1 ? 1 : u;
}
// line-search candidates per pass: 6 from m = 6000 on, 4 from m = 2000 on, else 1 (an iteration
// is latency-bound there: at m = 100 and 1k the 10 % fewer passes of a window of 4 cost 10 %
// more per iteration; at m = 5k it is 20 % fewer for 15 %); CLIPPER_HIP_WINDOW = 1|4|6|8 overrides
constexpr int64_t WINDOW_MIN_M = 6000;
constexpr int64_t WINDOW4_MIN_M = 2000;
// multi-process: iterations queued between two state snapshots. Up to two batches of no-op
// iterations (each still holds its all-gather) run past convergence: keep them short. 16 -> 4
// changes nothing on a 1-rank world (tools/rank1_probe.py).
constexpr int SOLVE_BATCH = 4;
constexpr int RUN_AHEAD = 4;     // one process: iterations kept queued ahead of the device
constexpr int MAX_EVENT_PAIRS = 256;  // per solve; created when profiling is switched on
// time every 20th iteration's mat-vec: an event pair costs ~30 us of stream time (the launches
// around it no longer pipeline) — every 8th was 0.13 ms of a 2.0 ms step at m = 10k
constexpr int PROFILE_EVERY = 20;

// ---- RCCL, bound at run time so the single-GPU path never loads librccl -----------------
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(CLIPPER_HIP_E_COMM, "cannot load librccl: %s", dlerror());
  auto sym = [&](const char* s) { return dlsym(lib, s); };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.GetErrorString =
      reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
    return fail(CLIPPER_HIP_E_COMM, "librccl is missing required symbols");
  g_rccl.lib = lib;
  return 0;
}

// ---- one column slice of M on one device ------------------------------------------------
struct Shard {
  int device = 0;
  int slot = 0;  // global shard index: owns columns [slot*W, slot*W + W)
  hipStream_t stream = nullptr;
  void* S = nullptr;   // m x W, element = float | double
  void* Cs = nullptr;  // explicit constraint matrix, same shape (only when C != pattern(M))
  double* part = nullptr;  // [ntiles][2][W]
  double* u0 = nullptr;
  double* pt = nullptr;    // point slots [2][V][2][mp]
  double* cab = nullptr;   // (a, b) of the last pair-mode pass [2][mp]
  double* X[2] = {nullptr, nullptr};  // candidate tables [V+1][mp][VS], see SolveArgs
  double* ab = nullptr;    // [P][NSLOT][W]
  double* scal = nullptr;  // [nwg][Q] partial scalars of k_tail
  SolverState* st = nullptr;      // ST[2], see SolverState
  uint8_t* marks = nullptr;       // [KIND_CAP] per-iteration pass marks (profiling), see SolveArgs
  SolveShared* shared = nullptr;
  // affinity inputs (staged once, reused while the sizes fit)
  double *P1 = nullptr, *P2 = nullptr;  // gathered point tables [d][pstride]
  float *P1f = nullptr, *P2f = nullptr; // the same, rounded to fp32 (prefilter input)
  size_t capPf = 0;
  int32_t* Adev = nullptr;              // [2][m]
  double *dD1 = nullptr, *dD2 = nullptr;  // raw D1, D2 as uploaded
  size_t capP = 0, capA = 0, capD1 = 0, capD2 = 0;
  hipEvent_t ev_reduced = nullptr, ev_copied = nullptr;
  size_t bytes_S = 0;
  size_t part_tiles = 0;  // row tiles `part` has room for
  // column-compressed copy of M (CLIPPER_HIP_STORE_F32_CSC), see kernels.hip.h
  uint32_t* cLc = nullptr;
  uint64_t* cPre = nullptr;
  float* cvals = nullptr;
  uint8_t* crows = nullptr;
  int* ctb = nullptr;
  CscBuildCtl* cctl = nullptr;
  size_t ccap_units = 0, ccap_groups = 0, ccap_tb = 0;
  int c_ntmax = 0;        // row tiles per strip of this shard's plan
  uint64_t c_units = 0;   // sum of the padded list lengths (units of 128 entries)
};

}  // namespace

struct clipper_hip_ctx {
  int storage = CLIPPER_HIP_STORE_F32;
  int world = 1;        // total shards P
  bool multiproc = false;
  std::vector<Shard> sh;  // local shards
  ncclComm_t comm = nullptr;

  int64_t m = 0;  // associations / matrix dimension
  int64_t W = 0;  // shard pitch
  int64_t alloc_m = 0, alloc_W = 0;
  bool has_matrix = false;
  bool explicitC = false;
  bool compressed = false;   // CLIPPER_HIP_STORE_F32_CSC was asked for
  bool csc_valid = false;    // ... and the compressed copy of the current matrix exists
  bool csc_emitted = false;  // the fill kernel of this build wrote the groups itself
  CscOut csc_out{};          // what that kernel was given
  int csc_nblocks = 0, csc_nstrips = 0;
  uint32_t* csc_hLc = nullptr;     // pinned host copy of Lc
  CscBuildCtl* csc_hctl = nullptr; // pinned host copy of the build's counters
  int* csc_htb = nullptr;          // pinned staging of the tile boundaries
  size_t csc_hcap_groups = 0, csc_hcap_tb = 0;
  int staged_d = 0;          // dimension of the staged point tables (0 = nothing staged)
  double staged_maxabs = 0;  // max |coordinate| of D1, D2: bounds the fp32 prefilter's error
  bool plain_affinity = false;  // CLIPPER_HIP_AFFINITY=plain: non-compacting fill kernels
  bool strip_affinity = false;  // CLIPPER_HIP_AFFINITY=strip: compacting strip kernels even where
                                // the symmetric tile kernel applies (one shard, fp32 storage)
  int64_t staged_pstride = 0;
  bool u0_staged = false;
  int ntiles = 1, rows_per_tile = 0, nstrips = 0;
  int cus = 256;

  std::vector<int32_t> A;  // column-major m x 2 (host copy)
  std::vector<int32_t> nodes;

  SolveShared* host_state = nullptr;  // pinned, 2 slots (multi-process snapshots)
  hipEvent_t ev_poll[2] = {nullptr, nullptr};
  HostMirror* mirror = nullptr;      // pinned + coherent: progress record written by the device
  HostMirror* mirror_dev = nullptr;  // its device address
  uint8_t* kind = nullptr;           // pinned + coherent: per-iteration pass / transition marks
  uint8_t* kind_dev = nullptr;       // (profiling only), written by the device
  double* u_pinned = nullptr;        // pinned staging of the final u (the device writes it)
  double* u_pinned_dev = nullptr;    // its device address
  size_t u_pinned_cap = 0;
  int V = 6;               // line-search window: candidate vectors per pass
  int V_forced = 0;        // CLIPPER_HIP_WINDOW
  int64_t mp = 0;          // rows of a candidate table
  int par = 0;             // which table set the next launch reads

  bool profiling = false;
  std::vector<hipEvent_t> ev_pairs;  // 2*MAX_EVENT_PAIRS, created by clipper_hip_set_profiling
  std::vector<int64_t> ev_launch_index;  // which mat-vec launch of the solve each pair timed
  int ev_used = 0;
  int64_t launch_counter = 0;
  clipper_hip_timings_t tm{};

  size_t esize() const { return storage == CLIPPER_HIP_STORE_F64 ? 8 : 4; }
};

namespace {

using Ctx = clipper_hip_ctx;

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
  fr(s.cLc);
  fr(s.cPre);
  fr(s.cvals);
  fr(s.crows);
  fr(s.ctb);
  fr(s.cctl);
  s.ccap_units = s.ccap_groups = s.ccap_tb = 0;
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

// CLIPPER_HIP_STORE_F32_CSC (C == pattern(M) is checked per matrix). On one unsharded device
// M exists ONLY compressed (csc_single: the fill kernel emits the groups, no dense store); column
// shards keep their dense slice and build a compressed copy of it for the solver's passes.
bool csc_possible(const Ctx* h) { return h->compressed && h->storage == CLIPPER_HIP_STORE_F32; }
bool csc_single(const Ctx* h) { return csc_possible(h) && h->world == 1 && !h->multiproc; }

int plan_unr(const Ctx* h) {
  return gemv_unr(h->V, static_cast<int>(h->esize()), h->explicitC);
}

// largest row-tile count plan_tiles considers for this (m, W)
int64_t max_tiles(const Ctx* h) {
  const int64_t slots = static_cast<int64_t>(h->cus) * GEMV_WG_PER_CU;
  int64_t nt = std::max<int64_t>(16, ceil_div(slots, std::max(1, h->nstrips)) + 1);
  if (const char* e = std::getenv("CLIPPER_HIP_TILES")) nt = std::max<int64_t>(nt, std::atoll(e));
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
  if (const char* e = std::getenv("CLIPPER_HIP_TILES")) {  // tuning knob (measurements only)
    const int64_t v = std::atoll(e);
    if (v > 0) best = std::min<int64_t>(v, std::max<int64_t>(1, ceil_div(h->m, chunk)));
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
  const int V = h->V_forced ? h->V_forced : (m >= WINDOW_MIN_M ? 6 : (m >= WINDOW4_MIN_M ? 4 : 1));
  const bool same = (h->alloc_m == m && h->alloc_W == W && h->V == V);
  h->V = V;
  plan_tiles(h);
  if (same) return 0;
  for (auto& s : h->sh) {
    free_shard_buffers(s);
    HIPCHK(hipSetDevice(s.device));
    const size_t bytesS = static_cast<size_t>(m) * static_cast<size_t>(W) * h->esize();
    s.bytes_S = bytesS;
    // CLIPPER_HIP_STORE_F32_CSC keeps M compressed: the dense store exists only while a path
    // that needs it is in use (ensure_dense)
    if (!csc_single(h)) HIPCHK(hipMalloc(&s.S, bytesS));
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
    const size_t Q = V * (2 + 2 * V) + 2 * V + 2;
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

// the pair-mode mat-vec alone on table X (matvec API, micro-benchmark)
void launch_plain(Ctx* h, Shard& s, const double* X) {
  dispatch_storage(h, [&](auto t, auto c) {
    launch_plain_t<decltype(t), decltype(c)::value>(h, s, X);
  });
}

// G on the compressed copy of M (one shard, C == pattern(M), fp32)
CscView csc_view(const Ctx* h, const Shard& s);

template <int V>
void launch_pass_csc(Ctx* h, Shard& s, const SolveArgs& a) {
  const CscView M = csc_view(h, s);
  dim3 grid(h->csc_nstrips, s.c_ntmax), block(GEMV_NW * 64);
  hipLaunchKernelGGL((k_gemv_csc<V, GEMV_NW>), grid, block, 0, s.stream, M, a);
}

// calls f(integral_constant<V>) for the context's window size
template <typename F>
void dispatch_window(const Ctx* h, F&& f) {
  switch (h->V) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    default: f(std::integral_constant<int, 6>{}); break;
  }
}

// plain reduction of `nslots` partial slots into this shard's block (matvec API)
void launch_reduce(Ctx* h, Shard& s, int nslots) {
  dim3 grid(static_cast<unsigned>(ceil_div(static_cast<int64_t>(nslots) * h->W, 256))), block(256);
  hipLaunchKernelGGL(k_reduce, grid, block, 0, s.stream, s.part, h->ntiles, nslots, h->W,
                     s.ab + static_cast<int64_t>(s.slot) * nslots * h->W);
}

// exchange of the per-shard blocks [nslots][W] so that every shard holds the gathered sums
int exchange(Ctx* h, int nslots) {
  if (h->world == 1 && !h->multiproc) return 0;
  const int64_t blk_elems = static_cast<int64_t>(nslots) * h->W;
  const size_t blk = static_cast<size_t>(blk_elems) * sizeof(double);
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
  a.ntiles = h->csc_valid ? s.c_ntmax : h->ntiles;
  a.slot = s.slot;
  a.scal = s.scal;
  a.nwg = static_cast<int>(ceil_div(h->m, TAIL_THREADS));
  a.scal_in = s.scal;
  a.nwg_in = a.nwg;
  if (a.nwg > SCAL_FOLD_MIN) {  // folded copy behind the partials themselves
    a.scal_in = s.scal + static_cast<int64_t>(a.nwg) * (h->V * (2 + 2 * h->V) + 2 * h->V + 2);
    a.nwg_in = static_cast<int>(ceil_div(a.nwg, SCAL_FOLD));
  }
  a.marks = (h->profiling && &s == &h->sh[0]) ? s.marks : nullptr;
  a.kind = (a.marks && !h->multiproc) ? h->kind_dev : nullptr;
  a.host_u = (!h->multiproc && &s == &h->sh[0]) ? h->u_pinned_dev : nullptr;
  return a;
}

// One full solver iteration:
//   one shard : k_gemv[_csc] (decision + pass) -> k_tail<V, true> (adds the tile partials itself)
//   sharded   : k_gemv[_csc] -> k_reduce_pass (tile partials -> own block) -> exchange -> k_tail<V, false>
template <int V>
int enqueue_iteration_v(Ctx* h, const SolverParams& prm) {
  const int par = h->par;
  h->par ^= 1;
  // CLIPPER_HIP_FORCE_SHARDED: test / measurement knob — the column-shard protocol (reduce launch,
  // exchange, k_tail<V, false>) on a single unsharded device
  static const bool force_sharded = std::getenv("CLIPPER_HIP_FORCE_SHARDED") != nullptr;
  const bool sharded = !(h->world == 1 && !h->multiproc) || force_sharded;
  // timing events cost ~5-10 us of stream time each: sample every 8th launch only
  Shard& s0 = h->sh[0];
  static const int every = std::getenv("CLIPPER_HIP_PROFILE_EVERY") ? std::max(4, std::atoi(std::getenv("CLIPPER_HIP_PROFILE_EVERY"))) : PROFILE_EVERY;
  // iterations 4, 11, then every `every`-th: short solves (20 iterations) still get samples, and
  // one of them is a pass (3 and 9 both hit transitions at cfg4)
  const bool prof = h->profiling && (h->launch_counter % every == 4 || h->launch_counter == 11) &&
                    h->ev_used < MAX_EVENT_PAIRS;
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    const SolveArgs a = solve_args(h, s, prm, par);
    if (prof && &s == &s0) HIPCHK(hipEventRecord(h->ev_pairs[2 * h->ev_used], s.stream));
    if (h->csc_valid) launch_pass_csc<V>(h, s, a);
    else launch_pass<V>(h, s, a);
    if (prof && &s == &s0) {
      HIPCHK(hipEventRecord(h->ev_pairs[2 * h->ev_used + 1], s.stream));
      h->ev_launch_index[h->ev_used] = h->launch_counter;
      ++h->ev_used;
    }
  }
  ++h->launch_counter;
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
  }
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    const SolveArgs a = solve_args(h, s, prm, par);
    dim3 grid(static_cast<unsigned>(a.nwg), V), block(TAIL_THREADS);
    if (sharded) hipLaunchKernelGGL((k_tail<V, false>), grid, block, 0, s.stream, a);
    else hipLaunchKernelGGL((k_tail<V, true>), grid, block, 0, s.stream, a);
    if (a.nwg_in != a.nwg)
      hipLaunchKernelGGL(k_scal_fold, dim3(static_cast<unsigned>(a.nwg_in)), dim3(128), 0, s.stream,
                         a.scal, a.nwg, V * (2 + 2 * V) + 2 * V + 2,
                         const_cast<double*>(a.scal_in), a.shared);
  }
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
  std::priority_queue<T, std::vector<T>, std::greater<T>> q;
  for (size_t i = 0; i < x.size(); ++i) {
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
      storage != CLIPPER_HIP_STORE_F32_CSC) {
    fail(CLIPPER_HIP_E_INVALID, "storage must be CLIPPER_HIP_STORE_F32, _F64 or _F32_CSC");
    return nullptr;
  }
  Ctx* h = new Ctx();
  h->compressed = (storage == CLIPPER_HIP_STORE_F32_CSC);
  h->storage = h->compressed ? CLIPPER_HIP_STORE_F32 : storage;
  h->world = world;
  h->multiproc = multiproc;
  h->sh.resize(static_cast<size_t>(nlocal));
  for (int p = 0; p < nlocal; ++p) {
    Shard& s = h->sh[static_cast<size_t>(p)];
    s.device = devices[p];
    s.slot = first_slot + p;
    if (s.device < 0 || s.device >= ndev) {
      fail(CLIPPER_HIP_E_INVALID, "device %d out of range (%d visible)", s.device, ndev);
      delete h;
      return nullptr;
    }
    if (hipSetDevice(s.device) != hipSuccess ||
        hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev_reduced, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming) != hipSuccess) {
      fail(CLIPPER_HIP_E_HIP, "cannot create stream/events on device %d", s.device);
      delete h;
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
    delete h;
    return nullptr;
  }
  // progress record the deciding workgroup writes straight into host memory (coherent, mapped)
  if (hipHostMalloc(reinterpret_cast<void**>(&h->mirror), sizeof(HostMirror),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h->mirror_dev), h->mirror, 0) !=
          hipSuccess) {
    fail(CLIPPER_HIP_E_HIP, "cannot allocate the pinned progress record");
    delete h;
    return nullptr;
  }
  std::memset(h->mirror, 0, sizeof(HostMirror));
  if (hipHostMalloc(reinterpret_cast<void**>(&h->kind), KIND_CAP,
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&h->kind_dev), h->kind, 0) != hipSuccess) {
    fail(CLIPPER_HIP_E_HIP, "cannot allocate the pinned iteration marks");
    delete h;
    return nullptr;
  }
  std::memset(h->kind, 0, KIND_CAP);
  // CLIPPER_HIP_WINDOW = 1 | 4 | 6 | 8: line-search candidates multiplied per pass over M
  if (const char* w = std::getenv("CLIPPER_HIP_WINDOW")) {
    const int v = std::atoi(w);
    if (v == 1 || v == 4 || v == 6 || v == 8) h->V_forced = v;
  }
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
  int rc = ensure_problem(h, m);
  if (rc) return rc;
  const int64_t pstride = round_up(m, 64);
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

bool use_sym_fill(const Ctx* h) {
  return !h->plain_affinity && !h->strip_affinity && h->world == 1 && !h->multiproc &&
         h->storage == CLIPPER_HIP_STORE_F32;
}

// k_affinity_sym needs more dynamic LDS than the 64 KiB a kernel gets by default
template <typename K>
void launch_sym(K kernel, dim3 grid, hipStream_t stream, float* S, int64_t W, int64_t mm, int nT,
                const Shard& s, int64_t pstride, const int32_t* A0, const int32_t* A1,
                const EuclidParams& e, const PointNormalParams& n, float E2, const CscOut& O) {
  static std::vector<const void*> raised;  // once per kernel instantiation and device
  const void* fn = reinterpret_cast<const void*>(kernel);
  if (std::find(raised.begin(), raised.end(), fn) == raised.end()) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, AT_SYM_LDS_BYTES);
    raised.push_back(fn);
  }
  hipLaunchKernelGGL(kernel, grid, dim3(AT_WAVES * 64), AT_SYM_LDS_BYTES, stream, S, W, mm, nT, s.P1, s.P2,
                     s.P1f, s.P2f, pstride, A0, A1, e, n, E2, O);
}

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

const char* clipper_hip_last_error(void) { return g_err.c_str(); }

int clipper_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

clipper_hip_t* clipper_hip_create(int device, int storage) {
  return make_ctx(&device, 1, storage, 1, 0, false);
}

clipper_hip_t* clipper_hip_create_group(const int* devices, int nshards, int storage) {
  if (!devices || nshards < 1) {
    fail(CLIPPER_HIP_E_INVALID, "invalid shard list");
    return nullptr;
  }
  return make_ctx(devices, nshards, storage, nshards, 0, false);
}

clipper_hip_t* clipper_hip_create_rank(int device, int storage, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) {
    fail(CLIPPER_HIP_E_INVALID, "rank %d / world %d invalid", rank, world);
    return nullptr;
  }
  // CLIPPER_HIP_FORCE_RCCL=1 routes even a 1-rank world through ncclAllGather, so that the
  // communicator plumbing can be exercised on a single-GPU box
  const char* force = std::getenv("CLIPPER_HIP_FORCE_RCCL");
  const bool multiproc = world > 1 || (force && force[0] == '1');
  return make_ctx(&device, 1, storage, world, rank, multiproc);
}

int clipper_hip_comm_unique_id(void* id128) {
  if (!id128) return fail(CLIPPER_HIP_E_INVALID, "null id buffer");
  int rc = load_rccl();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(CLIPPER_HIP_E_COMM, "ncclGetUniqueId failed (%d)", (int)r);
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

int clipper_hip_comm_init(clipper_hip_t* h, const void* id128) {
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
}

void clipper_hip_destroy(clipper_hip_t* h) {
  if (!h) return;
  for (auto& s : h->sh) {
    hipSetDevice(s.device);
    if (s.stream) hipStreamSynchronize(s.stream);
  }
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  for (auto& s : h->sh) {
    free_shard_buffers(s);
    if (s.ev_reduced) hipEventDestroy(s.ev_reduced);
    if (s.ev_copied) hipEventDestroy(s.ev_copied);
    if (s.stream) hipStreamDestroy(s.stream);
  }
  if (!h->sh.empty()) hipSetDevice(h->sh[0].device);
  for (hipEvent_t e : h->ev_pairs) hipEventDestroy(e);
  if (h->ev_poll[0]) hipEventDestroy(h->ev_poll[0]);
  if (h->ev_poll[1]) hipEventDestroy(h->ev_poll[1]);
  if (h->host_state) hipHostFree(h->host_state);
  if (h->mirror) hipHostFree(h->mirror);
  if (h->kind) hipHostFree(h->kind);
  if (h->u_pinned) hipHostFree(h->u_pinned);
  if (h->csc_hLc) hipHostFree(h->csc_hLc);
  if (h->csc_hctl) hipHostFree(h->csc_hctl);
  if (h->csc_htb) hipHostFree(h->csc_htb);
  delete h;
}

// ---- affinity --------------------------------------------------------------------------

int clipper_hip_stage_inputs(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                             const double* D2, int64_t n2, const int32_t* A, int64_t m) {
  return stage_inputs(h, D1, d, n1, D2, n2, A, m);
}

int clipper_hip_affinity_euclidean_staged(clipper_hip_t* h, double sigma, double epsilon,
                                          double mindist, double affinityeps) {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->staged_d < 1) return fail(CLIPPER_HIP_E_STATE, "clipper_hip_stage_inputs not called");
  const EuclidParams prm{sigma, epsilon, mindist, affinityeps};
  const int64_t mm = h->m, W = h->W, pstride = h->staged_pstride;
  const int d = h->staged_d;
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
      if (d == 3)
        launch_sym(k_affinity_sym<3, false>, g, s.stream, static_cast<float*>(s.S), W, mm, nT, s,
                   pstride, A0, A1, prm, none, E2, h->csc_out);
      else
        launch_sym(k_affinity_sym<2, false>, g, s.stream, static_cast<float*>(s.S), W, mm, nT, s,
                   pstride, A0, A1, prm, none, E2, h->csc_out);
      h->csc_emitted = (h->csc_out.Lc != nullptr);
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
}

int clipper_hip_affinity_pointnormal_staged(clipper_hip_t* h, double sigp, double epsp,
                                            double sign, double epsn, double affinityeps) {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->staged_d != 6)
    return fail(CLIPPER_HIP_E_STATE, "PointNormalDistance needs staged inputs with d == 6");
  const PointNormalParams prm{sigp, epsp, sign, epsn, affinityeps};
  const int64_t mm = h->m, W = h->W, pstride = h->staged_pstride;
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
      launch_sym(k_affinity_sym<3, true>, g, s.stream, static_cast<float*>(s.S), W, mm, nT, s,
                 pstride, s.Adev, s.Adev + mm, none, prm, guarded_threshold_sq(thr), h->csc_out);
      h->csc_emitted = (h->csc_out.Lc != nullptr);
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
}

int clipper_hip_affinity_euclidean(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                                   const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                   double sigma, double epsilon, double mindist,
                                   double affinityeps) {
  const auto t0 = std::chrono::high_resolution_clock::now();
  int rc = stage_inputs(h, D1, d, n1, D2, n2, A, m);
  if (rc) return rc;
  rc = clipper_hip_affinity_euclidean_staged(h, sigma, epsilon, mindist, affinityeps);
  if (rc) return rc;
  h->tm.affinity_total_ms =
      std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0)
          .count();
  return 0;
}

int clipper_hip_affinity_pointnormal(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                                     const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                     double sigp, double epsp, double sign, double epsn,
                                     double affinityeps) {
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
}

int64_t clipper_hip_num_associations(const clipper_hip_t* h) { return h ? h->m : 0; }

int clipper_hip_get_associations(const clipper_hip_t* h, int32_t* A_out) {
  if (!h || !A_out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->A.size() != static_cast<size_t>(2 * h->m))
    return fail(CLIPPER_HIP_E_STATE, "no association list is held");
  std::memcpy(A_out, h->A.data(), h->A.size() * sizeof(int32_t));
  return 0;
}

// ---- matrix set / get ------------------------------------------------------------------

int clipper_hip_set_matrix(clipper_hip_t* h, const double* M, const double* C, int64_t m) {
  if (!h || !M || !C || m < 1) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (h->A.size() != static_cast<size_t>(2 * m)) h->A.clear();
  h->nodes.clear();
  int rc = ensure_problem(h, m);
  if (rc) return rc;
  h->csc_valid = false;
  if ((rc = ensure_dense(h, false))) return rc;
  const size_t bytes = static_cast<size_t>(m) * m * sizeof(double);
  const int64_t W = h->W;
  // pass 1: fill S and detect whether C is anything other than pattern(M)
  std::vector<double*> dM(h->sh.size(), nullptr), dC(h->sh.size(), nullptr);
  std::vector<int*> dflag(h->sh.size(), nullptr);
  int mismatch = 0;
  for (size_t k = 0; k < h->sh.size(); ++k) {
    Shard& s = h->sh[k];
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipMalloc(&dM[k], bytes));
    HIPCHK(hipMalloc(&dC[k], bytes));
    HIPCHK(hipMalloc(&dflag[k], sizeof(int)));
    HIPCHK(hipMemsetAsync(dflag[k], 0, sizeof(int), s.stream));
    HIPCHK(hipMemcpyAsync(dM[k], M, bytes, hipMemcpyHostToDevice, s.stream));
    HIPCHK(hipMemcpyAsync(dC[k], C, bytes, hipMemcpyHostToDevice, s.stream));
    if (s.Cs) {
      hipFree(s.Cs);
      s.Cs = nullptr;
    }
    dim3 grid(static_cast<unsigned>(ceil_div(W, 256)), static_cast<unsigned>(m)), block(256);
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
      dim3 grid(static_cast<unsigned>(ceil_div(W, 256)), static_cast<unsigned>(m)), block(256);
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
  rc = sync_all(h);
  for (size_t k = 0; k < h->sh.size(); ++k) {
    hipSetDevice(h->sh[k].device);
    hipFree(dM[k]);
    hipFree(dC[k]);
    hipFree(dflag[k]);
  }
  if (rc) return rc;
  rc = csc_rebuild(h);
  if (rc) return rc;
  h->has_matrix = true;
  return 0;
}

int clipper_hip_set_sparse(clipper_hip_t* h, int64_t m, const int64_t* Mcolptr,
                           const int32_t* Mrow, const double* Mval, const int64_t* Ccolptr,
                           const int32_t* Crow, const double* Cval) {
  if (!h || !Mcolptr || !Ccolptr || m < 1) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  const int64_t nnzM = Mcolptr[m], nnzC = Ccolptr[m];
  if ((nnzM > 0 && (!Mrow || !Mval)) || (nnzC > 0 && (!Crow || !Cval)))
    return fail(CLIPPER_HIP_E_INVALID, "null CSC arrays");
  if (h->A.size() != static_cast<size_t>(2 * m)) h->A.clear();
  h->nodes.clear();
  int rc = ensure_problem(h, m);
  if (rc) return rc;
  h->csc_valid = false;
  if ((rc = ensure_dense(h, false))) return rc;
  // C == pattern(M)?  (same structure, every stored C equal to 1, every stored M non-zero)
  bool pattern = (nnzM == nnzC) && std::equal(Mcolptr, Mcolptr + m + 1, Ccolptr) &&
                 std::equal(Mrow, Mrow + nnzM, Crow);
  for (int64_t p = 0; pattern && p < nnzM; ++p) pattern = (Cval[p] == 1.0) && (Mval[p] != 0.0);
  h->explicitC = !pattern;
  plan_tiles(h);
  const int64_t W = h->W;
  auto scatter = [&](Shard& s, void* dst, const int64_t* cp, const int32_t* ri, const double* va,
                     int64_t nnz) -> int {
    int64_t* dcp = nullptr;
    int32_t* dri = nullptr;
    double* dva = nullptr;
    HIPCHK(hipMalloc(&dcp, static_cast<size_t>(m + 1) * sizeof(int64_t)));
    HIPCHK(hipMalloc(&dri, std::max<size_t>(1, static_cast<size_t>(nnz)) * sizeof(int32_t)));
    HIPCHK(hipMalloc(&dva, std::max<size_t>(1, static_cast<size_t>(nnz)) * sizeof(double)));
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
    dim3 grid(static_cast<unsigned>(m)), block(256);
    if (h->storage == CLIPPER_HIP_STORE_F64)
      hipLaunchKernelGGL((k_from_csc<double>), grid, block, 0, s.stream,
                         static_cast<double*>(dst), W, m, c0, W, dcp, dri, dva);
    else
      hipLaunchKernelGGL((k_from_csc<float>), grid, block, 0, s.stream, static_cast<float*>(dst),
                         W, m, c0, W, dcp, dri, dva);
    HIPCHK(hipStreamSynchronize(s.stream));
    hipFree(dcp);
    hipFree(dri);
    hipFree(dva);
    return 0;
  };
  for (auto& s : h->sh) {
    HIPCHK(hipSetDevice(s.device));
    if (s.Cs) {
      hipFree(s.Cs);
      s.Cs = nullptr;
    }
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
}

int clipper_hip_get_matrix(clipper_hip_t* h, double* M_out, double* C_out) {
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
  return 0;
}

// ---- solver ------------------------------------------------------------------------------

int clipper_hip_stage_u0(clipper_hip_t* h, const double* u0) {
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
}

int clipper_hip_solve(clipper_hip_t* h, const double* u0, const clipper_params_t* P,
                      double* u_out, clipper_solve_info_t* info) {
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
}

int clipper_hip_solve_staged(clipper_hip_t* h, const clipper_params_t* P, double* u_out,
                             clipper_solve_info_t* info) {
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

  h->ev_used = 0;
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
  int rc = 0;

  Shard& s0 = h->sh[0];
  SolveShared fin;
  std::memset(&fin, 0, sizeof(fin));
  if (!h->multiproc) {
    // One process: the deciding workgroup reports progress into pinned host memory; the host
    // keeps RUN_AHEAD iterations queued ahead of what the device has retired and stops
    // queueing the moment `done` shows up — no memcpy, no event, no host wait in the loop.
    volatile HostMirror* hm = h->mirror;
    int64_t queued = 0;
    uint64_t spins = 0;
    while (!hm->done) {
      if (queued - hm->iters < RUN_AHEAD) {
        if ((rc = enqueue_iteration(h, prm))) return rc;
        ++queued;
        spins = 0;
      } else if ((++spins & 0xfffff) == 0) {
        // the device has not retired an iteration for a long time: make sure it is still alive
        hipError_t q = hipStreamQuery(s0.stream);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(CLIPPER_HIP_E_HIP, "solver stream failed: %s", hipGetErrorString(q));
        if (q == hipSuccess && !hm->done && queued - hm->iters >= RUN_AHEAD)
          return fail(CLIPPER_HIP_E_HIP, "solver made no progress (iters %lld of %lld queued)",
                      static_cast<long long>(hm->iters), static_cast<long long>(queued));
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    fin.F = hm->F;
    fin.d = hm->d;
    fin.n_passes = hm->n_passes;
    fin.n_trials = hm->n_trials;
    fin.ifinal = hm->ifinal;
    fin.ubp = hm->ubp;
    fin.ubv = hm->ubv;
  } else {
    // Multi-process: every rank must queue the same number of iterations (each holds a
    // collective), so the decision to stop rests on state snapshots only, which are
    // bit-identical on all ranks. Batch n+1 is queued before the snapshot after batch n is read.
    int slot = 0;
    bool have_prev = false;
    bool done = false;
    int batch = SOLVE_BATCH;
    if (const char* e = std::getenv("CLIPPER_HIP_SOLVE_BATCH")) batch = std::max(1, std::atoi(e));  // tuning knob, same on every rank
    while (!done) {
      for (int it = 0; it < batch; ++it) {
        if ((rc = enqueue_iteration(h, prm))) return rc;
      }
      HIPCHK(hipSetDevice(s0.device));
      HIPCHK(hipMemcpyAsync(&h->host_state[slot], s0.shared, sizeof(SolveShared),
                            hipMemcpyDeviceToHost, s0.stream));
      HIPCHK(hipEventRecord(h->ev_poll[slot], s0.stream));
      if (have_prev) {
        HIPCHK(hipEventSynchronize(h->ev_poll[slot ^ 1]));
        if (h->host_state[slot ^ 1].done) done = true;
      }
      have_prev = true;
      slot ^= 1;
    }
    if ((rc = sync_all(h))) return rc;
    HIPCHK(hipSetDevice(s0.device));
    HIPCHK(hipMemcpy(&fin, s0.shared, sizeof(fin), hipMemcpyDeviceToHost));
  }

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
  std::vector<double> u(h->u_pinned, h->u_pinned + m);

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

  // mat-vec timings from the event pairs
  h->tm.gemv_avg_us = h->tm.gemv_min_us = 0.0;
  h->tm.gemv_launches = 0;
  h->tm.gemv_bytes = algorithmic_gemv_bytes(h);
  if (h->profiling && h->ev_used > 0) {
    // only launches that streamed M count: the device marked every iteration as pass (1) or
    // transition (0); launches queued past convergence have no mark
    const uint8_t* kind = h->kind;
    const int64_t iters_run = std::min<int64_t>(h->launch_counter, KIND_CAP);
    double sum = 0.0, mn = 1e30;
    int64_t nreal = 0;
    for (int k = 0; k < h->ev_used; ++k) {
      const int64_t li = h->ev_launch_index[static_cast<size_t>(k)];
      if (li >= iters_run || !kind[static_cast<size_t>(li)]) continue;
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, h->ev_pairs[2 * k], h->ev_pairs[2 * k + 1]));
      sum += ms;
      mn = std::min<double>(mn, ms);
      ++nreal;
    }
    if (nreal > 0) {
      h->tm.gemv_avg_us = sum / static_cast<double>(nreal) * 1e3;
      h->tm.gemv_min_us = mn * 1e3;
      h->tm.gemv_launches = nreal;
    }
  }
  return 0;
}

int clipper_hip_get_nodes(const clipper_hip_t* h, int32_t* out, int32_t capacity) {
  if (!h || !out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  const int32_t k = static_cast<int32_t>(h->nodes.size());
  if (capacity < k) return fail(CLIPPER_HIP_E_INVALID, "capacity %d < %d nodes", capacity, k);
  if (k) std::memcpy(out, h->nodes.data(), static_cast<size_t>(k) * sizeof(int32_t));
  return k;
}

// utils::selectInlierAssociations — utils.cpp:101-108
int clipper_hip_get_selected_associations(const clipper_hip_t* h, int32_t* A_out,
                                          int32_t capacity) {
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
}

int clipper_hip_densest_subgraph(clipper_hip_t* h, const int32_t* S, int32_t k, int32_t* nodes_out,
                                 int32_t capacity) {
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
}

// ---- putative associations (before the path): brute-force nearest neighbours -------------------

int clipper_hip_knn(int device, const double* P0, int64_t n0, const double* P1, int64_t n1, int d,
                    int knn, int32_t* idx_out, double* sqd_out) {
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
}

int64_t clipper_hip_distance_based_correspondences(int device, const double* P0, int64_t n0,
                                                   const double* P1, int64_t n1, int d, int knn,
                                                   double radius, int enforce_1to1, int32_t* A_out,
                                                   int64_t capacity) {
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
}

int clipper_hip_set_window(clipper_hip_t* h, int window) {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (window != 0 && window != 1 && window != 4 && window != 6 && window != 8)
    return fail(CLIPPER_HIP_E_INVALID, "window must be 0 (automatic), 1, 4, 6 or 8");
  h->V_forced = window;
  return 0;
}

int clipper_hip_window(const clipper_hip_t* h) { return h ? h->V : 0; }

int clipper_hip_storage_in_use(const clipper_hip_t* h) {
  if (!h) return -1;
  return h->csc_valid ? CLIPPER_HIP_STORE_F32_CSC : h->storage;
}

int clipper_hip_matvec(clipper_hip_t* h, const double* x, double* yM, double* yC) {
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
  int rc = ensure_dense(h, true);
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
}

// ---- measurement ---------------------------------------------------------------------------

int clipper_hip_set_profiling(clipper_hip_t* h, int on) {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  h->profiling = (on != 0);
  if (h->profiling && h->ev_pairs.empty()) {  // here, not inside the first profiled solve (~1 ms)
    HIPCHK(hipSetDevice(h->sh[0].device));
    h->ev_pairs.resize(2 * MAX_EVENT_PAIRS);
    h->ev_launch_index.assign(MAX_EVENT_PAIRS, 0);
    for (auto& e : h->ev_pairs) HIPCHK(hipEventCreate(&e));
  }
  return 0;
}

int clipper_hip_get_timings(const clipper_hip_t* h, clipper_hip_timings_t* out) {
  if (!h || !out) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  *out = h->tm;
  return 0;
}

int clipper_hip_bench_matvec(clipper_hip_t* h, int reps, double* avg_us) {
  if (!h || reps < 1 || !avg_us) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  if (!h->has_matrix) return fail(CLIPPER_HIP_E_STATE, "no matrix has been built or set");
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
  h->tm.gemv_bytes = algorithmic_gemv_bytes(h, /*dense=*/true);
  return 0;
}

int clipper_hip_device_info(const clipper_hip_t* h, char* name64, int* cus, int64_t* hbm_bytes) {
  if (!h) return fail(CLIPPER_HIP_E_INVALID, "invalid argument");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, h->sh[0].device));
  if (name64) {
    std::snprintf(name64, 64, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cus) *cus = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
  return 0;
}

}  // extern "C"
