// host_resident.hpp — planning and launch of the resident solver (k_resident.hip.h)
// Part of clipper_hip.hip (one translation unit; included there, in order).
#pragma once

namespace {

constexpr uint32_t RS_LDS_MAX = 159u * 1024u;  // dynamic LDS of a workgroup: the 160 KB of a gfx950 CU less the kernel's few static words
constexpr int RS_MAX_UNITS = 64;  // beyond (dense problems of m ~ 2000: > 5 MB of slices) the streaming
                                  // launches with their window of 4 win (profiles/r02c_bm_table*.md)

bool rs_debug() {
  static const bool on = std::getenv("CLIPPER_HIP_RESIDENT_DEBUG") != nullptr;
  return on;
}

void resident_free(Ctx* h) {
  Resident& r = h->res;
  if (r.host_plan) hipHostFree(r.host_plan);
  if (r.xb) hipFree(r.xb);
  if (r.ctl) hipFree(r.ctl);
  const int vf = r.V_forced;
  const bool xo = r.xcd_off;
  const int hm = r.home;
  r = Resident{};
  r.V_forced = vf;
  r.xcd_off = xo;
  r.home = hm;
}

// Decide whether the current slices fit the resident solver and lay out its units (host_plan.hpp).
// Called when a build's directory (csc_hLq) is on the host. Leaves r.ready = false when the problem
// does not fit.
int resident_plan(Ctx* h, Shard& s) {
  Resident& r = h->res;
  r.ready = false;
  r.failed = false;
  if (h->resident_mode == 1 || h->V_forced != 0 || h->world != 1 || h->multiproc || h->explicitC) return 0;
  static_assert(sizeof(clipper_plan::Unit) == sizeof(ResidentUnit), "the planner's unit is the kernel's");
  static_assert(clipper_plan::so_bytes(17, SL_SO) == sl_so_bytes(17) && clipper_plan::so_bytes(32, SL_SO) == sl_so_bytes(32) &&
                    clipper_plan::so_bytes(0, SL_SO) == sl_so_bytes(0),
                "the planner sizes the step-offset table as the format does");
  const clipper_plan::ResidentConsts K{RS_NT, RS_NWV, RS_TMAX, RS_PMAX, RS_MAXE, RS_LDS_MAX,
                                       RS_RED_BYTES, RS_TAB_BYTES, RS_SLICE_PAD, SL_SO};
  static thread_local clipper_plan::ResidentPlan plan;
  const int ncg = s.s_ncg, nchunks = s.s_nchunks;
  clipper_plan::plan_resident(h->csc_hLq, ncg, nchunks, h->m, h->mp, static_cast<int>(h->esize()),
                              std::min(RS_MAX_UNITS, h->cus - 8), r.V_forced, K, plan);
  if (rs_debug())
    std::fprintf(stderr, "[resident] plan m=%lld ncg=%d nchunks=%d total_ub=%llu -> V=%d E=%d units=%zu ok=%d\n",
                 static_cast<long long>(h->m), ncg, nchunks, static_cast<unsigned long long>(plan.total_bound),
                 plan.V, plan.E, plan.units.size(), plan.ok ? 1 : 0);
  if (!plan.ok) return 0;
  const int V = plan.V, E = plan.E, maxslots = plan.maxslots;
  const int64_t mp = h->mp;
  const std::vector<clipper_plan::Unit>& units = plan.units;
  const std::vector<uint8_t>&nsl = plan.nsl, &npieces = plan.npieces, &wave_cg = plan.wave_cg;
  const std::vector<uint32_t>& pieces = plan.pieces;
  const uint32_t lds_slices = plan.lds_slices, lds_total = RS_LDS_MAX;

  HIPCHK(hipSetDevice(s.device));
  // plan in mapped pinned memory: the kernel reads it through the bus once, nothing is copied
  const size_t off_nsl = units.size() * sizeof(ResidentUnit);
  const size_t off_np = off_nsl + static_cast<size_t>(round_up(ncg, 16));
  const size_t off_wc = off_np + static_cast<size_t>(round_up(static_cast<int64_t>(npieces.size()), 16));
  const size_t off_pc = off_wc + static_cast<size_t>(round_up(static_cast<int64_t>(wave_cg.size()), 16));
  const size_t plan_bytes = off_pc + pieces.size() * sizeof(uint32_t) + 64;
  if (plan_bytes > r.host_plan_cap) {
    if (r.host_plan) hipHostFree(r.host_plan);
    r.host_plan = nullptr;
    r.host_plan_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&r.host_plan), plan_bytes + 4096,
                         hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.host_plan_dev), r.host_plan, 0));
    r.host_plan_cap = plan_bytes + 4096;
  }
  std::memcpy(r.host_plan, units.data(), units.size() * sizeof(ResidentUnit));
  std::memcpy(r.host_plan + off_nsl, nsl.data(), static_cast<size_t>(ncg));
  std::memcpy(r.host_plan + off_np, npieces.data(), npieces.size());
  std::memcpy(r.host_plan + off_wc, wave_cg.data(), wave_cg.size());
  r.off_wc = off_wc;
  std::memcpy(r.host_plan + off_pc, pieces.data(), pieces.size() * sizeof(uint32_t));
  r.off_np = off_np;
  r.off_pc = off_pc;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const size_t xb_bytes = 2ull * maxslots * (V + 1) * static_cast<size_t>(mp) * 2 * sizeof(unsigned long long);
  if (xb_bytes > r.xb_cap) {
    if (r.xb) HIPCHK(hipFree(r.xb));
    r.xb = nullptr;
    HIPCHK(hipMalloc(&r.xb, xb_bytes));
    r.xb_cap = xb_bytes;
    HIPCHK(hipMemsetAsync(r.xb, 0, xb_bytes, s.stream));  // no granule may carry a future epoch
  }
  if (!r.ctl) {  // the error word and the two counters of the one-XCD mode
    HIPCHK(hipMalloc(&r.ctl, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(r.ctl, 0, 4 * sizeof(unsigned long long), s.stream));
  }
  r.V = V;
  r.E = E;
  r.nunits = static_cast<int>(units.size());
  r.maxslots = maxslots;
  r.lds_slices = lds_slices;
  r.lds_total = lds_total;
  r.ready = true;
  return 0;
}

template <typename VT, int V, int E>
int resident_launch_t(Ctx* h, Shard& s, const ResidentArgs& a) {
  const unsigned grid = static_cast<unsigned>(a.xcd_mode ? 8 * a.nunits : a.nunits);
  Resident& r = h->res;
  auto kern = k_solve_resident<VT, V, E>;
  static std::atomic<bool> attr_set[64] = {};  // per instantiation and device (contexts may solve concurrently)
  const int dv = (s.device >= 0 && s.device < 64) ? s.device : 0;
  if (!attr_set[dv]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(RS_LDS_MAX));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (rs_debug()) {
        hipFuncAttributes fa{};
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
        int optin = 0, perblk = 0;
        (void)hipDeviceGetAttribute(&perblk, hipDeviceAttributeMaxSharedMemoryPerBlock, s.device);
        (void)hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, s.device);
        std::fprintf(stderr, "[resident] hipFuncSetAttribute: %s (static LDS %zu, max dynamic %d, device per block %d, optin %d)\n",
                     hipGetErrorString(e), fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes, perblk, optin);
      }
      return 1;  // this device does not give a workgroup 159 KB of LDS: not an error, no resident solver
    }
    attr_set[dv] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(RS_NT), r.lds_total, s.stream, a);
  if (const hipError_t e = hipGetLastError(); e != hipSuccess) {  // refused: the streaming launches take over
    if (rs_debug()) std::fprintf(stderr, "[resident] launch failed: %s\n", hipGetErrorString(e));
    return 1;
  }
  return 0;
}

template <typename VT>
int resident_launch_v(Ctx* h, Shard& s, const ResidentArgs& a) {
  const Resident& r = h->res;
  switch (r.V * 10 + r.E) {
    case 21: return resident_launch_t<VT, 2, 1>(h, s, a);
    case 22: return resident_launch_t<VT, 2, 2>(h, s, a);
    case 24: return resident_launch_t<VT, 2, 4>(h, s, a);
    case 11: return resident_launch_t<VT, 1, 1>(h, s, a);
    case 12: return resident_launch_t<VT, 1, 2>(h, s, a);
    case 14: return resident_launch_t<VT, 1, 4>(h, s, a);
    default: return 1;
  }
}

// Runs the whole solve as one launch. ran = false: the resident solver did not apply or gave up
// (nothing of the solver state was touched: the caller runs the streaming solver).
int resident_solve(Ctx* h, const SolverParams& prm, bool rescale, SolveShared& fin, bool& ran) {
  ran = false;
  Resident& r = h->res;
  if (rs_debug())
    std::fprintf(stderr, "[resident] solve: ready=%d failed=%d csc_valid=%d mode=%d Vf=%d\n", r.ready, r.failed,
                 h->csc_valid, h->resident_mode, h->V_forced);
  if (!r.ready || r.failed || !h->csc_valid || h->resident_mode == 1 || h->V_forced != 0) return 0;
  Shard& s = h->sh[0];
  HIPCHK(hipSetDevice(s.device));
  ResidentArgs a;
  a.M = slice_view(h, s);
  a.units = reinterpret_cast<const ResidentUnit*>(r.host_plan_dev);
  a.nunits = r.nunits;
  a.nslots_of_cg = r.host_plan_dev + static_cast<size_t>(r.nunits) * sizeof(ResidentUnit);
  a.npieces = r.host_plan_dev + r.off_np;
  a.wave_cg = r.host_plan_dev + r.off_wc;
  a.pieces = reinterpret_cast<const uint32_t*>(r.host_plan_dev + r.off_pc);
  a.maxslots = r.maxslots;
  a.m = h->m;
  a.mp = h->mp;
  a.prm = prm;
  a.rescale = rescale ? 1 : 0;
  a.u0 = s.u0;
  a.xb = r.xb;
  a.epoch0 = r.epoch;
  a.err = reinterpret_cast<uint32_t*>(r.ctl);
  a.lds_slices = r.lds_slices;
  a.u_dev = s.pt;  // point slot (0, 0), array u
  a.host_u = h->u_pinned_dev;
  a.host = h->mirror_dev;
  a.shared = s.shared;
  a.stamps = h->stamps_dev;
  a.ctrs = r.ctl + 1;
  // contexts take their home XCD in turn, so that concurrent solves of several contexts do not
  // queue for the 32 CUs of one XCD
  static std::atomic<int> next_home{0};
  if (r.home < 0) r.home = next_home.fetch_add(1) & 7;
  a.home = r.home;
  long long timeout_override = 0;
  // test knobs: a home XCD that does not exist (the one-XCD launch must be refused and repeated in
  // the placement-free mode), a time-out of a few ticks (the streaming launches must take over)
  if (const char* e = std::getenv("CLIPPER_HIP_RESIDENT_HOME")) a.home = std::atoi(e);
  if (const char* e = std::getenv("CLIPPER_HIP_RESIDENT_TIMEOUT_TICKS")) timeout_override = std::atoll(e);
  // One-XCD mode: with at most 28 units (an XCD has 32 CUs) the launch is 8 x units workgroups and the
  // ones on XCD 0 do the work, exchanging through their common L2 (per pass ~2.5 us less). Refused
  // once (the hardware did not put enough workgroups there), it is not tried again.
  const char* xe = std::getenv("CLIPPER_HIP_RESIDENT_XCD");
  const bool xcd_env_off = xe && std::atoi(xe) == 0;
  volatile HostMirror* hm = h->mirror;
  for (int attempt = 0;; ++attempt) {
    a.xcd_mode = (!r.xcd_off && !xcd_env_off && r.nunits >= 2 && r.nunits <= 28) ? 1 : 0;
    // longest wait for another unit's sums of ONE pass on the 100 MHz wall clock: 5 ms — five hundred passes' worth
    // (a pass is 3-11 us); in the one-XCD mode 2 ms (its units compete for one XCD's CUs with whatever else runs
    // there: rather launch again). A unit that is not resident by then is behind another tenant's work: the
    // streaming launches take over, and this matrix is not tried again (r.failed). Round 4 waited 0.5 s — a thousand
    // solves' worth — before a 0.3 ms solve went on.
    // (ten times as long for a context's first launch — cold code object, clock ramp-up, a serialising profiler)
    a.timeout_ticks = timeout_override ? timeout_override : (a.xcd_mode ? 200000ll : 500000ll) * (r.epoch == 0 ? 10 : 1);
    a.epoch0 = r.epoch;
    std::memset(h->mirror, 0, sizeof(HostMirror));
    std::atomic_thread_fence(std::memory_order_seq_cst);
    int lr = 1;
    dispatch_vt(h, [&](auto tag) {
      using VT = decltype(tag);
      lr = resident_launch_v<VT>(h, s, a);
    });
    if (lr != 0) {
      if (rs_debug()) std::fprintf(stderr, "[resident] launch refused (V=%d E=%d)\n", r.V, r.E);
      r.failed = true;
      return 0;
    }
    uint64_t spins = 0;
    bool finished = true;
    while (!hm->done) {
      if ((++spins & 0x3fff) == 0) {
        hipError_t q = hipStreamQuery(s.stream);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(CLIPPER_HIP_E_HIP, "resident solver failed: %s", hipGetErrorString(q));
        if (q == hipSuccess && !hm->done) {  // the launch is over and did not finish the solve
          finished = false;
          break;
        }
      }
    }
    if (finished) break;
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, a.err, sizeof(err), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(r.ctl, 0, 4 * sizeof(unsigned long long)));  // error word, counters
    HIPCHK(hipMemset(r.xb, 0, r.xb_cap));  // granules of the abandoned solve
    r.epoch += 1ull << 20;
    r.last_error = static_cast<int>(err);
    if (rs_debug()) std::fprintf(stderr, "[resident] gave up: error %u (one-XCD mode %d)\n", err, a.xcd_mode);
    if ((err == RS_ERR_PLAN || err == RS_ERR_TIMEOUT) && a.xcd_mode && attempt == 0 && timeout_override == 0) {
      r.xcd_off = true;  // placement-free mode from now on
      continue;
    }
    r.failed = true;  // until the next build
    return 0;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  fin.F = hm->F;
  fin.d = hm->d;
  fin.n_passes = hm->n_passes;
  fin.n_trials = hm->n_trials;
  fin.ifinal = hm->ifinal;
  fin.ubp = 0;
  fin.ubv = 0;
  r.epoch += static_cast<unsigned long long>(hm->iters) + 8ull;
  // the granules carry the low 32 bits of the epoch: long before they wrap, start over on a clean buffer
  if ((r.epoch & 0xffffffffull) > 0xf0000000ull) {
    HIPCHK(hipMemset(r.xb, 0, r.xb_cap));
    r.epoch = (r.epoch & ~0xffffffffull) + (1ull << 32);
  }
  if (rs_debug()) std::fprintf(stderr, "[resident] solved: units=%d one-XCD mode=%d passes=%lld\n", r.nunits, a.xcd_mode,
                               static_cast<long long>(fin.n_passes));
  ran = true;
  return 0;
}

}  // namespace
