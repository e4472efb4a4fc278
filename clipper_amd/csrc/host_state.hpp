// host_state.hpp — host-side state: constants, the RCCL binding, Shard (one column slice on one device), the context
// Part of clipper_hip.hip (one translation unit; included there, in order).
#pragma once

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

// ---- no exception crosses the C ABI (SURVEY 8b "Error conventions") -------------------------------
// Every entry point is a function-try-block closed by one of these: std::bad_alloc (host vectors sized by
// the caller's nnz / m / k) becomes CLIPPER_HIP_E_NOMEM, anything else CLIPPER_HIP_E_INTERNAL, with the
// message in clipper_hip_last_error(). tests/test_abi_exports.py scans the sources for the pairing.
int guard_fail(int code, const char* what) noexcept {
  try {
    g_err = what;
  } catch (...) {  // (even the message may not allocate: the code alone goes back)
  }
  return code;
}
#define CLIPPER_HIP_GUARD_CATCH(on_nomem, on_other)                                                     \
  catch (const std::bad_alloc&) { on_nomem; }                                                            \
  catch (const std::exception& e) { (void)e; on_other; }                                                 \
  catch (...) { on_other; }
#define CLIPPER_HIP_GUARD_INT                                                                            \
  CLIPPER_HIP_GUARD_CATCH(return guard_fail(CLIPPER_HIP_E_NOMEM, "out of host memory inside the library"), \
                          return guard_fail(CLIPPER_HIP_E_INTERNAL, "unexpected exception inside the library"))
#define CLIPPER_HIP_GUARD_PTR                                                                            \
  CLIPPER_HIP_GUARD_CATCH({ guard_fail(CLIPPER_HIP_E_NOMEM, "out of host memory inside the library"); return nullptr; }, \
                          { guard_fail(CLIPPER_HIP_E_INTERNAL, "unexpected exception inside the library"); return nullptr; })
#define CLIPPER_HIP_GUARD_STR CLIPPER_HIP_GUARD_CATCH(return "", return "")
#define CLIPPER_HIP_GUARD_VOID CLIPPER_HIP_GUARD_CATCH(return, return)


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
constexpr int64_t WINDOW_MIN_M = 6000;       // dense storages: a window of 6 from here on
constexpr int64_t WINDOW_MIN_M_CSC = 8500;   // slices: measured crossover of the windows of 4 and 6 (profiles/r02e_window_sweep.txt)
constexpr int64_t WINDOW4_MIN_M = 2000;
// multi-process: iterations queued between two state snapshots. Up to two batches of no-op
// iterations (each still holds its all-gather) run past convergence: keep them short. 16 -> 4
// changes nothing on a 1-rank world (tools/rank1_probe.py).
// the row view (host_rowview.hpp): problems the resident solver does not take anyway; views per solve
constexpr int64_t RV_MIN_M = 3000;
constexpr int RV_MAX_BUILDS = 6;
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

// ---- one set of slices (k_slices.hip.h): the data, its directory, the work list of a pass ----
struct SliceStore {
  uint32_t* sSizes = nullptr;   // [nslices] size / 16
  uint32_t* sLq = nullptr;      // [nslices] maxq | entries << 8
  uint64_t* sPre = nullptr;     // [nslices]
  uint64_t* sBlk = nullptr;     // scan block sums [nblk + 1]; the last one = total units
  uint8_t* sdata = nullptr;
  SliceWork* swork = nullptr;
  size_t scap_slices = 0, scap_bytes = 0, scap_work = 0;
  int s_ncg = 0, s_nchunks = 0;
  int s_nwork = 0;        // workgroups of a pass
  int s_nslots = 0;       // partial-sum slots per column (what the tail adds)
  uint64_t s_bytes = 0;   // bytes of the slices
  uint64_t s_entries = 0; // stored entries (both triangles)
};

// ---- the row view of M (k_solver.hip.h, LIVE ROWS; host_rowview.hpp) ---------------------------
struct RowView {
  SliceStore st;               // the slices of M[rows, :] (this shard's columns)
  int32_t* rowmap[2] = {nullptr, nullptr};  // [cap_rows] association of view row r': in use / being built
  uint8_t* in_view[2] = {nullptr, nullptr};  // [mp] row flags: of the view in use / of the one being built
  int cur = 0;                 // which of the two the view in use owns
  SliceView* desc = nullptr;   // device copy of the view's descriptor (what a pass on the view reads)
  uint32_t* blk = nullptr;     // [nblk + 2] per-block live counts, then their offsets
  int32_t* viewpos = nullptr;  // [mp] position of a row of M in the view being built, or -1
  size_t cap_rows = 0, cap_flags = 0, cap_blk = 0, cap_pos = 0;
  int64_t nrows = 0;
  bool valid = false;
  bool plan_pending = false;    // the work list of the streamed pass is not planned yet (the resident solver took the view)
  // Column shards: the REPLICA of a small view — the slices of M[rows, ALL columns], on every rank (a few MB, scored
  // from the replicated points) — which the resident solver on a view runs on, redundantly and identically on every
  // rank, while `st` (this shard's columns) serves the streamed passes around it (host_rv_resident.hpp)
  SliceStore full;
  bool full_valid = false;
};

// ---- one column slice of M on one device ------------------------------------------------
struct Shard : SliceStore {
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
  // compressed storage of M (CLIPPER_HIP_STORE_F32_CSC / _F64_CSC), see k_csc.hip.h, k_slices.hip.h
  // -- groups: what the fill kernels emit (transient input of the packers, kept for reuse)
  uint16_t* gOff = nullptr;     // [ngroups][GR_OFFS]
  uint64_t* gPre = nullptr;     // [ngroups]
  void* gvals = nullptr;        // float | double [4 * gcap_units]
  uint8_t* grows = nullptr;     // [4 * gcap_units]
  CscBuildCtl* cctl = nullptr;  // [CSC_ARENAS]
  size_t gcap_units = 0, gcap_groups = 0;
  // -- slices: what the passes stream: the SliceStore this shard IS (all rows), and the row view
  RowView rv;
};

// the resident solver (k_resident.hip.h): plan of the current slices and its buffers
struct Resident {
  bool ready = false;    // the current slices fit: `plan` is valid
  bool failed = false;   // a launch gave up on this matrix (the streaming solver took over)
  int last_error = 0;
  int V = 0, E = 0, nunits = 0, maxslots = 0;
  uint32_t lds_slices = 0, lds_total = 0;
  uint8_t* host_plan = nullptr;      // pinned + mapped: units, then slots per column group
  uint8_t* host_plan_dev = nullptr;
  size_t host_plan_cap = 0;
  size_t off_np = 0, off_wc = 0, off_pc = 0;  // where the piece counts / the wave map / the pieces start in the plan
  unsigned long long* xb = nullptr;  // exchange buffer [2][maxslots][V+1][mp][2] (granules)
  size_t xb_cap = 0;
  unsigned long long* ctl = nullptr;  // [4]: the error word, the two counters of the one-XCD mode
  unsigned long long epoch = 0;
  int V_forced = 0;                  // (a window other than 1: measured 2.3 x slower per pass)
  bool xcd_off = false;              // the one-XCD mode was refused once (or CLIPPER_HIP_RESIDENT_XCD=0)
  int home = -1;                     // this context's home XCD in that mode
};

// the resident solver on a row view (k_rv_resident.hip.h): plan of the view in use and its buffers
struct ViewResident {
  bool ready = false;        // the view in use fits: `plan` is valid
  int nunits = 0;
  uint32_t lds_slices = 0;
  uint8_t* host_plan = nullptr;      // pinned + mapped: units, piece counts, wave map, pieces
  uint8_t* host_plan_dev = nullptr;
  size_t host_plan_cap = 0;
  size_t off_np = 0, off_wc = 0, off_pc = 0;
  unsigned long long* xb = nullptr;  // exchange buffer (granules), zero at allocation
  size_t xb_bytes = 0;
  uint32_t* ctl = nullptr;           // [RVR_GIVEUP_SLOTS][16]: per launch of a solve [0] error word, [1] arrivals at the exit, [4..5] unit 0's start
  unsigned long long epoch = 0;      // every launch takes 2^20 epochs
  int target_units = 0;              // CLIPPER_HIP_VIEW_RESIDENT_WGS (0: automatic)
  // Launches that gave up (an exchange that timed out: a unit that did not become resident in time — another tenant
  // on the device —, an LDS plan the device refused). The kernel leaves its error word in pinned memory, one word per
  // launch of the solve; the host reads them when the solve is over, counts, and BACKS OFF: the next `cooldown` solves
  // of this context stream their views, twice as many after every solve with a give-up (at most 64), one again
  // after a solve whose launches all ran.
  uint32_t* giveup_host = nullptr;   // pinned + mapped [RVR_GIVEUP_SLOTS][4]: per launch of the solve {error word, iterations,
                                     // duration in wall-clock ticks (lo, hi)} — RvrArgs::giveup_host
  hipEvent_t ev[2 * 16] = {};        // profiling level 2: event pairs around the launches of a solve
  int ev_n = 0;
  uint64_t plan_entries = 0;         // stored entries of the view the plan is of
  uint32_t* giveup_host_dev = nullptr;
  int launches_this_solve = 0;
  int cooldown = 0, cooldown_next = 1;
  int max_units_device = -1;         // workgroups of the kernel the device holds at once (occupancy query; -1: not asked yet)
  bool on_replica = false;           // the plan is of RowView::full (column shards), not of the shard's own view
  uint8_t* backup = nullptr;         // column shards: SolverState + SolveShared as they were before a launch, restored on
                                     // the ranks whose launch ran when another rank's gave up (all ranks go on alike)
};
constexpr int RVR_GIVEUP_SLOTS = 16;

// the live sub-problem (k_subproblem.hip.h, host_subproblem.hpp): the associations that can still be selected, as a
// CLIPPER problem of their own — a child context on the parent's device and stream, kept from solve to solve
constexpr int SUB_MAX_ENTRIES = 4;   // hand-overs per solve (each leave costs a host round trip)
constexpr int64_t SUB_MIN_M = 12000; // full problems below this keep their views (the resident solver on a view takes them)
struct SubProblem {
  struct clipper_hip_ctx* ctx = nullptr;        // M[S,S] as slices
  struct clipper_hip_ctx* ctx_dense = nullptr;  // M[S,S] as a dense fp32 store (where it is mostly non-zero: the inlier block)
  struct clipper_hip_ctx* use = nullptr;        // the one the solve is handed over to
  int32_t* cnt = nullptr;     // [mp] entries of every column among the view's rows (an upper bound: quads x 4)
  uint8_t* flags = nullptr;   // [mp] 1 = the association is in the sub-problem
  int32_t* colmap = nullptr;  // [mp] sub-problem index -> association
  int32_t* pos = nullptr;     // [mp] association -> sub-problem index, or -1
  uint32_t* blk = nullptr;    // block counts / offsets of the list build
  int32_t* nout_acc = nullptr;  // device counter of k_sub_leave
  size_t cap = 0, cap_blk = 0;
  SubRecord* rec = nullptr;   // pinned + mapped: what the selection found
  SubRecord* rec_dev = nullptr;
  bool ready = false;    // a sub-problem of the view in use stands ready: the decision may ask for the hand-over
  bool active = false;   // the solve's launches run on it
  int entries = 0;       // hand-overs of this solve
  int launches_since_entry = 0;  // launches on it since the last hand-over (test knob CLIPPER_HIP_SUB_TEST_LEAVE)
  int64_t nS = 0;        // its associations
  double ncol = 0.0;     // the largest entry count among the view's rows of a column outside it
  // reporting (clipper_hip_view_stats_t)
  int64_t passes_at_entry = 0, sub_passes = 0, leaves = 0;
  double build_ms = 0.0;
};

}  // namespace

struct clipper_hip_ctx {
  int storage = CLIPPER_HIP_STORE_F32;
  int world = 1;        // total shards P
  bool multiproc = false;
  std::vector<Shard> sh;  // local shards
  ncclComm_t comm = nullptr;
  // exchange through the caller (clipper_hip_comm_init_callback) instead of RCCL
  clipper_hip_allgather_fn xchg_fn = nullptr;
  void* xchg_user = nullptr;
  double* xchg_send = nullptr;  // pinned staging of this rank's block / of the gathered blocks
  double* xchg_recv = nullptr;
  size_t xchg_cap = 0;

  int64_t m = 0;  // associations / matrix dimension
  int64_t W = 0;  // shard pitch
  int64_t alloc_m = 0, alloc_W = 0;
  bool has_matrix = false;
  bool explicitC = false;
  bool compressed = false;   // CLIPPER_HIP_STORE_F32_CSC / _F64_CSC was asked for
  bool csc_valid = false;    // ... and the slices of the current matrix exist
  bool csc_emitted = false;  // the fill kernel of this build wrote the groups itself
  CscOut csc_out{};          // what that kernel was given
  int csc_nblocks = 0, csc_nstrips = 0;  // 64-row blocks, 128-column strips of the groups
  uint32_t* csc_hLq = nullptr;     // pinned host copy of sLq
  CscBuildCtl* csc_hctl = nullptr; // pinned host copy of the build's counters
  uint64_t* csc_htotal = nullptr;  // pinned: total slice units of the last count
  SliceWork* csc_hwork = nullptr;  // pinned staging of the work list
  size_t csc_hcap_slices = 0, csc_hcap_work = 0;
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
  std::vector<double> u_host;  // the last solve's u on the host (rounding works on it)

  SolveShared* host_state = nullptr;  // pinned, 2 slots (multi-process snapshots)
  hipEvent_t ev_poll[2] = {nullptr, nullptr};
  hipEvent_t ev_aff[2] = {nullptr, nullptr};  // timing of the affinity build
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
  Resident res;
  ViewResident vres;
  bool decide_only = false;  // the next iteration's G launch only decides (hand-over to the resident solver on a view)
  std::function<int()> enqueue_one;  // set by a one-process solve while it runs: queues one solver iteration (the view
                                     // build uses it to queue the decide-only iteration behind its fill, host_rowview.hpp)
  bool early_decide_done = false;    // ... and did: the hold is lifted, the decide-only iteration is in the stream
  int resident_mode = 0;   // 0 = use the resident solver where the slices fit, 1 = never
  int last_solver = 0;     // what the last solve ran on: 0 = streaming launches, 1 = resident

  // what built the matrix, kept so that a row view can be filled from the same points later:
  // 0 = nothing (setMatrixData: no view), 1 = EuclideanDistance, 2 = PointNormalDistance
  int fill_kind = 0;
  EuclidParams fill_e{};
  PointNormalParams fill_n{};
  float fill_E2 = 0.f;
  // row-view policy (host_rowview.hpp)
  double total_slice_bytes = 0.0;  // column shards: bytes of all shards' slices (gather_slice_bytes)
  ViewPolicy rvp{};           // the cost model the device-side policy works with (host_rowview.hpp)
  bool rv_fresh = false;      // the next iteration is the first after a view was built
  int rv_mode = 0;            // 0 = automatic, 1 = never (clipper_hip_set_row_view / CLIPPER_HIP_ROW_VIEW=0),
                              // 2 = views, but never the resident solver on one (CLIPPER_HIP_VIEW_RESIDENT=0)
  SliceView* rv_desc_host = nullptr;     // pinned + mapped staging of a view's descriptor (two slots, used in turn)
  int rv_desc_slot = 0;
  SliceView* rv_desc_host_dev = nullptr;
  int32_t* rv_count = nullptr;      // pinned + mapped: rows of the view being built
  int32_t* rv_count_dev = nullptr;
  clipper_hip_view_stats_t rv_stats{};
  // the live sub-problem: this context's (`sub`), or the context this one IS the sub-problem of (`parent`: its
  // launches report into the parent's progress record and borrow the parent's stream)
  SubProblem sub;
  struct clipper_hip_ctx* parent = nullptr;
  bool adaptive_window = true;   // CLIPPER_HIP_ADAPTIVE_WINDOW=0: every window pass multiplies all V candidates
  int sub_mode = 0;           // 0 = automatic, 1 = never (clipper_hip_set_subproblem / CLIPPER_HIP_SUBPROBLEM=0), 2 = always as slices

  long long* stamps_dev = nullptr;  // CLIPPER_HIP_STAMPS=1: [4096][4], see SolveArgs::stamps; =2: [16384][4]
  int stamps_rows = 4096;
  bool profiling = false;
  int profiling_level = 0;
  std::vector<hipEvent_t> ev_pairs;  // 2*MAX_EVENT_PAIRS, created by clipper_hip_set_profiling
  std::vector<int64_t> ev_launch_index;  // which mat-vec launch of the solve each pair timed
  std::vector<hipEvent_t> ev_xchg;       // 2*MAX_EVENT_PAIRS: around the exchange of the same iterations
  std::vector<char> ev_xchg_used;
  int ev_used = 0;
  int64_t launch_counter = 0;
  clipper_hip_timings_t tm{};

  size_t esize() const { return storage == CLIPPER_HIP_STORE_F64 ? 8 : 4; }
};
