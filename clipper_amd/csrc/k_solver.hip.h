// k_solver.hip.h — solver state, the decision at the head of a pass (decide), k_init, k_tail, k_scal_fold
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>


namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// The line-search WINDOW (what makes this solver an MI355X design rather than a port)
//
// One pass over M costs s*m^2 bytes of HBM traffic and, for a single vector, ~20 % of the
// fp64 VALU time that fits under it. The reference's backtracking line search
// (clipper.cpp:234-251) tries alpha = 1, beta, beta^2, ... one mat-vec pair at a time — at
// m = 10k half of all passes are rejected trials. Every one of those candidates
//     c_l = max(u + alpha*beta^l * gradF, 0)                         (clipper.cpp:235-236)
// is known BEFORE the first of them is evaluated, so a pass here multiplies M by a window of
// V consecutive candidates at once (V accumulator sets per lane, one read of M): the decision
// then walks the window in the reference's order and takes the first candidate the reference
// would have accepted. Same trials, same arithmetic per trial, same result — in 1/2 to 1/3 of
// the passes. Candidates live interleaved in "tables" X[row][VS] (64-byte rows) so that the
// V wave-uniform multipliers of a row arrive with one scalar load.
//
// One solver ITERATION is two launches, and no workgroup ever waits for another:
//   G  k_gemv   every workgroup first DECIDES, redundantly and identically, what the results
//               of the previous iteration mean (a few KB of partial scalars from L2 — the
//               loads overlap the first rows of M) and then streams its tile of M against the
//               pending window. Workgroup (0,0) also records the state it decided on.
//   T  k_tail   once per candidate v (grid = blocks x V): gradFnew_v, the partial sums of
//               Fnew_v and ||x_v - u||^2, (x_v, gradFnew_v) into point slot (ubp^1, v), and —
//               speculatively — the NEXT window for the outcome "candidate v was accepted"
//               (table v of Xout: max(x_v + beta^l gradFnew_v, 0), l < V) with the partial sums
//               of its norms; the v = 0 workgroups also build the outcome "all V rejected"
//               (table V: the next V step sizes from the unchanged (u, gradF)).
// The rare steps that sweep whole vectors (initialisation clipper.cpp:193-220, penalty update
// :268-280, a new outer iteration :219-220) take an iteration of their own: every workgroup of
// G sees that the decision needs one, workgroup (0,0) alone performs it, the others exit, and T
// finds nothing to do.
// ------------------------------------------------------------------------------------------

constexpr int VS = 8;       // doubles per table row; window sizes V <= VS
// partial-sum slots per row tile: slot 0 = a = M_off x_0, slots 1..V-1 = g_v of the other
// candidates, slot V = b = C_off x_0 (a pair-mode pass fills slots 0 and V only)
constexpr int nslot(int V) { return V + 1; }
// what a tail workgroup sums per candidate: Fnew, ||x - u||^2, the V (z, sum) pairs of the next
// window's norms, and the LIVE CODE of the point the candidate would become (below)
constexpr int tail_nr(int V) { return 3 + 2 * V; }
// ... and per launch: V of those, the V pairs of the "all rejected" window, the two penalty sums
constexpr int tail_q(int V) { return V * tail_nr(V) + 2 * V + 2; }

// ------------------------------------------------------------------------------------------
// LIVE ROWS and the ROW VIEW (what lets a pass skip the rows that cannot contribute)
//
// A window is built from a point (u', g') = (u, gradF): candidate l = max(u' + alpha_l g', 0)
// (clipper.cpp:235-236). u' >= 0 always, so row r of EVERY candidate is exactly 0 unless
//     live(r) :=  u'[r] > 0  or  g'[r] > 0,
// and a row of zeros adds exact zeros to every sum of the pass: it can be skipped without
// changing a bit of any partial sum it is skipped from. The projected gradient ascent of
// clipper.cpp:226-262 drives the outliers' entries of u to zero within a few iterations and the
// penalty keeps their gradient negative: at the headline problem (m = 10k, 95 % outliers) 45 %
// of the rows are live after the first step, 36 % for the rest of the first outer iteration and
// 5 % from the second outer iteration on (52 of 66 trials).
// The host therefore builds, when the live rows have become few, a ROW VIEW of M: the slices of
// M[R, :] for a row list R (host_rowview.hpp, k_affinity_rect) — and a pass streams the view
// instead of M whenever the view is known to cover the live rows of the window it multiplies:
//   * the tail counts, for the point every candidate v would become, its live rows and those of
//     them that lie outside the view (one double: live + 2^26 * outside — exact integer sums);
//   * the decision keeps the code of the accepted candidate in the state (nlive, nout) and
//     plans the pass on the view iff nout == 0 (PassPlan::view); otherwise the pass runs on M
//     itself, which is always correct;
//   * WHEN a view is built is decided on the device too, as a function of the solver state alone
//     (view_wanted below: a cost model of the passes to come against the fill): the decision then
//     puts the solve on HOLD — this and every later iteration already queued do nothing — and tells
//     the host, which drains the stream, builds the view from exactly the state that asked for it,
//     lifts the hold and goes on queueing. Nothing depends on how far ahead the host was: a solve is
//     bit-reproducible from run to run, views included.
// ------------------------------------------------------------------------------------------
constexpr double LIVE_OUT = 67108864.0;  // 2^26: the "outside the view" digit of a live code
constexpr double RV_GAIN_MARGIN = 1.3;   // the modelled gain of a view over its modelled cost, at least
constexpr double RV_ROWS_RATIO = 0.8;    // a new view must drop at least a fifth of the rows a pass streams today
                                         // (one constant for the device's request and the host's acceptance)

// cost model of the view policy (seconds; set by the host from the matrix at hand, see host_rowview.hpp)
struct ViewPolicy {
  int32_t on;           // 0: no views for this matrix
  int32_t max_builds;
  double pass_fixed, pass_per_row;    // a pass that streams r rows: pass_fixed + r * pass_per_row
  double build_fixed, build_per_row;  // building a view of r rows
};

enum Phase : int32_t {
  PH_NORMALIZE = 0,  // no rescale: u = u0/||u0||, no pass consumed       (clipper.cpp:196-198)
  PH_RESCALE = 1,    // pass on x = u0: u = M_off u0 + u0, normalise      (clipper.cpp:193-198)
  PH_INIT = 2,       // pass on x = u: initial d, first gradient          (clipper.cpp:200-220)
  PH_TRIAL = 3,      // pass on a window of trial vectors                 (clipper.cpp:234-262)
  PH_PENALTY = 4,    // pass on x = the inner loop's final u: penalty update (:268-280)
  PH_BUILD = 5       // no pass: the tail forms gradF, F and the first window of an outer
                     // iteration from (u, a, b) and the new penalty (:219-220, :235-236)
};
// PH_TRIAL passes run the mat-vec in window mode (candidate 0: a and b apart, the others
// g_v = (M_off + d*C_off) x_v), all others in pair mode (a, b of candidate 0) — see k_gemv.

enum Stage : int32_t {
  ST_PASS = 0,     // the tables hold the pending vectors of `phase`: run the pass
  ST_RESULTS = 1   // the pass of `phase` and its tail have run: decide what they mean first
};

// Candidates are kept UN-normalised: x_l = Xin[sel][.][l] / nrm[l]. The mat-vec multiplies M
// by the raw table; the tail divides the sums by nrm[l].
// Two copies ST[2] alternate: iteration k reads ST[k & 1], its workgroup (0,0) writes the state
// it decided on to ST[(k+1) & 1], which the tail of iteration k and iteration k+1 read.
struct SolverState {
  double d;        // penalty
  double F;        // objective at u
  double alpha;    // step size of candidate 0 of the pending window
  double s;        // sum(u)
  double nrm[VS];  // ||candidate l|| of the pending window (1 for an already normalised vector)
  double sx[VS];   // sum(x_l)
  int32_t sel;     // which table of Xin holds the pending window
  int32_t ubp, ubv;  // point slot that holds the current (u, gradF)
  int32_t phase;
  int32_t stage;
  int32_t i, j, k;  // outer / inner / line-search counters (clipper.cpp:217)
  int64_t n_passes;
  int64_t n_trials;  // trials the reference would have evaluated (window slots past the
                     // accepted candidate do not count)
  int64_t n_iters;   // iterations (G, T) the device has started
  int32_t nlive;     // live rows of the current point (u, gradF), see LIVE ROWS above
  int32_t nout;      // ... of which outside the row view the tail counted against
  int32_t view;      // the pass of this iteration ran on the row view (1) or on M (0): which
                     // partial-sum slots the tail adds
  int32_t hold;      // the decision wants a row view built before it goes on: nothing runs until the
                     // host has built it (or refused) and cleared this
  int64_t n_view_passes;  // passes that ran on the row view
  int32_t rv_builds;   // views asked for so far
  int32_t rv_last;     // n_iters when the last one was asked for
  int32_t rv_backoff;  // the host refused a view of this many rows (0: none): ask again only well below
  int32_t hold_slot;   // while on hold: the point slot p * V + v the decision ends on — the window it
                       // will leave pending is built from that point, the view must cover ITS live rows
  int32_t hold_nlive;  // ... and the live rows of that point the request was made with
  int32_t resume;      // stage == ST_PASS only: 0 = the pass a transition iteration prepared (pair mode on
                       // table `sel`); 1 = a WINDOW pass prepared from point slot (ubp, ubv) with step
                       // `alpha`, norms nrm / sx (phase PH_TRIAL); 2 = a pair-mode pass on the u array of
                       // that slot (phase PH_PENALTY) — what a decide-only launch and the resident solver
                       // on a row view (k_rv_resident.hip.h) leave behind; 3 = as 1, but nrm / sx hold the RAW
                       // sums (z_l, s_l) over the view's rows only: the resident solver left because a row
                       // outside its view became live (nout != 0), and that row's candidates are not zero —
                       // the launch that runs the pass adds the rows outside the view (iteration_head)
  // THE WINDOW IN USE (round 6). A window of V candidates pays when line searches reject; while they do not — the whole
  // first two outer iterations of every problem looked at: the penalty is small, alpha = 1 is accepted every time —
  // candidates 1 .. V-1 are never looked at, and a pass that multiplies candidate 0 alone (the pair-mode loop: one LDS
  // gather per entry instead of three, two fmas instead of seven) is 30 % shorter at m = 10k. `weff` = how many
  // candidates of the pending window the pass of this iteration multiplies and its tail evaluates (1 or V; 0 = V): the
  // decision walks exactly those. A pass on candidate 0 alone whose candidate is REJECTED was a wrong guess: the decision
  // discards it and the same window is multiplied again, whole — so every candidate that is ever walked sits at the
  // window index, and is formed by the expressions, it has without any of this: the solve is bit for bit the one with
  // full windows, a wrong guess costs one pass (one or two per solve: the policy below).
  int32_t weff;
  int32_t zero_run;    // line searches in a row that accepted their first trial (the policy: weff = 1 from two on)
  int32_t n_redo;      // windows multiplied a second time so far. The view policy counts time in iterations; it is given
                       // n_iters - n_redo, so that a solve asks for its views (and is handed over to its sub-problem) at
                       // the same points with and without the guesses — which keeps the two bit-identical
  int32_t pad_;
};

// What outlives the alternating state: the end of the solve. Kernels launched after
// convergence see `done` and return immediately.
struct SolveShared {
  double F, d;
  int64_t n_passes, n_trials;
  int32_t ifinal, ubp, ubv;
  int32_t done;
  int32_t hold;  // the solve waits for a row view (SolverState::hold): what a multi-process host reads in
  int32_t pad;   // its state snapshots — bit-identical on every rank, like `done`
};

// Host-visible progress record in pinned, coherent host memory. Workgroup (0,0) of G writes it
// with system-scope stores; the host spins on `iters` / `done` instead of issuing memcpy +
// event round trips, and keeps only a few iterations queued ahead of the device.
struct HostMirror {
  double F, d;
  int64_t n_passes, n_trials;
  int64_t iters;
  int32_t ifinal, ubp, ubv;
  int32_t done;
  int32_t nlive, nout;    // live rows of the current point / of them outside the view (reporting)
  int64_t n_view_passes;
  int32_t hold;           // the solve waits for the host to build a row view (SolverState::hold)
  int32_t hold_nlive;     // ... of this many rows (SolverState::hold_nlive)
};

struct SolverParams {
  double tol_u, tol_F, beta, eps;
  int32_t maxiniters, maxoliters, maxlsiters;
};

constexpr int TAIL_THREADS = 256;  // elements per tail workgroup
constexpr int TAIL_WAVES = TAIL_THREADS / 64;
// One shard (FUSED_REDUCE): the tail adds the pass's partial-sum slots itself. TAIL_SPLIT groups
// of TAIL_THREADS threads could each add a part of the slots and combine through LDS — measured
// at m = 10k with 4 groups (1024-thread workgroups): 12.6 us instead of 9.5; one group it is.
constexpr int TAIL_SPLIT = 1;

struct SolveArgs {
  const SolverState* st_cur;  // ST[k & 1]: what iteration k starts from
  SolverState* st_next;       // ST[(k+1) & 1]: what it decided on (read by its tail)
  SolveShared* shared;
  HostMirror* host;  // device address of the pinned progress record (may be null)
  SolverParams prm;
  int64_t m;    // problem size
  int64_t W;    // shard pitch: element i lives in block p = i / W of `ab`
  int64_t mp;   // rows of a candidate table / pitch of a point-slot array (>= m)
  const double* u0;
  double* pt;   // point slots [2][V][2][mp]: u, gradF
  double* cab;  // [2][mp]: a = M_off x, b = C_off x of the last pair-mode pass / of candidate 0
  // candidate tables [V+1][mp][VS]. Iteration k READS the pending window from Xin and WRITES
  // the windows of every outcome to Xout; the host swaps the two from iteration to iteration.
  const double* Xin;
  double* Xout;
  double* ab;     // column-sharded M: gathered RAW sums [P][NSLOT][W], NSLOT = V + 1
  double* part;   // [ntiles][NSLOT][W] row-tile partials of this shard
  int ntiles;
  int slot;       // this shard's block of `ab`
  double* scal;   // [nwg][Q] partial scalars of the tail, Q = V*(2+2V) + 2V + 2
  int nwg;        // tail workgroups per candidate = ceil(m / TAIL_THREADS)
  // what the decision sums: scal itself, or (large m: every workgroup of G repeats the decision,
  // nwg*Q doubles each) the SCAL_FOLD-fold pre-reduction k_scal_fold makes of it
  const double* scal_in;
  int nwg_in;
  // profiling only: iteration n_iters ran a pass. `marks` [KIND_CAP] is device memory (a store to
  // pinned host memory at the end of EVERY pass launch cost ~2 us each); the deciding workgroup
  // copies the marks to `kind` (pinned host memory) once, when the solve ends
  uint8_t* marks;
  uint8_t* kind;
  double* host_u; // pinned host memory [m] (may be null): the final u, written before `done`
  long long* stamps;  // measurement only (may be null): per workgroup {start, head done, end} of the
                      // last pass launch, 100 MHz wall clock (clipper_hip_debug_stamps)
  int stamps_wide;    // CLIPPER_HIP_STAMPS=2: every workgroup of the pass is stamped (16384 rows, and where it
                      // ran: XCD and HW_ID), the tail is not
  // the row view (null / 0: none), see LIVE ROWS
  const uint8_t* in_view;  // [m] 1 = the row is in the view
  int rv_nslots;           // partial-sum slots of a pass on the view (ntiles: of a pass on M)
  int rv_nwork;            // workgroups of a pass on the view
  int rv_fresh;            // the view was built from exactly the state this iteration decides from:
                           // it covers the live rows of every outcome, whatever the tail counted
  int rv_rows;             // rows of the view
  ViewPolicy rvp;
  int decide_only;         // this G launch only DECIDES: a decision that ends in a pass records that pass as
                           // prepared (SolverState::resume) instead of running it — the hand-over to the
                           // resident solver on a row view (k_rv_resident.hip.h), which starts from a
                           // prepared pass and leaves one behind
  int adaptive_window;     // 1: the decision may plan a pass on candidate 0 alone (SolverState::weff; the pass on the slices of
                           // one shard); 0: every window pass multiplies all V candidates
  // the live sub-problem (k_subproblem.hip.h)
  int sub_state;           // 0: none. 1: a sub-problem stands ready — the decision puts the solve on hold (hold = 2)
                           // for the hand-over once no column outside it can come back to life. 2: these launches
                           // RUN on the sub-problem — the decision hands the solve back (hold = 3) when one could
  double sub_ncol;         // stored entries among the live rows of any column outside the sub-problem, at most
  const int32_t* colmap;   // sub_state == 2: element i of the vectors is association colmap[i] of the full problem
};
constexpr int KIND_CAP = 1 << 16;

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// array k (0 = u, 1 = gradF) of point slot (p, v)
__device__ __forceinline__ double* pt_arr(const SolveArgs& A, int V, int p, int v, int k) {
  return A.pt + ((static_cast<int64_t>(p) * V + v) * 2 + k) * A.mp;
}

// Wave-level sum with DPP moves (VALU speed, no LDS crossbar): after the six steps lane 63
// holds the total of the 64 lanes; the order of the additions is fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xf>(v);  // row_mirror: every lane holds the sum of its row of 16
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 into rows 1, 3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 into rows 2, 3: lane 63 holds the wave total
  return v;
}

// Sum over the NWAVES waves of the workgroup; every thread must call it, every thread gets the
// totals. Fixed tree: bit-reproducible.
template <int N, int NWAVES>
__device__ __forceinline__ void block_reduce(double (&v)[N], double* lds /* [NWAVES*N] */) {
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_sum_to_lane63(v[q]);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
#pragma unroll
    for (int q = 0; q < N; ++q) lds[wave * N + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < N; ++q) {
    double acc = lds[q];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) acc += lds[w * N + q];
    v[q] = acc;
  }
}

// The same sums, left in LDS: thread t < N returns total t (its own registers are never indexed
// at run time, which would push the array to scratch), other threads return 0.
template <int N, int NWAVES>
__device__ __forceinline__ double block_reduce_pick(double (&v)[N], double* lds) {
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_sum_to_lane63(v[q]);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
#pragma unroll
    for (int q = 0; q < N; ++q) lds[wave * N + q] = v[q];
  }
  __syncthreads();
  double acc = 0.0;
  if (threadIdx.x < N) {
    acc = lds[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) acc += lds[w * N + threadIdx.x];
  }
  return acc;
}

__device__ __forceinline__ void store_row(double* row, const double (&c)[VS]) {
  double4* q = reinterpret_cast<double4*>(row);
  q[0] = make_double4(c[0], c[1], c[2], c[3]);
  q[1] = make_double4(c[4], c[5], c[6], c[7]);
}

constexpr int pow2_at_least(int x) {
  int p = 1;
  while (p < x) p *= 2;
  return p;
}

// ------------------------------------------------------------------------------------------
// decide — the head of every G launch (NT threads per workgroup). Returns true when this
// iteration performs a pass (plan = what to stream against), false when the workgroup has
// nothing more to do (transition iteration or end of the solve).
//
// Common case (a window pass and its tail have run): EVERY workgroup adds the tail's partial
// scalars in the same fixed order and walks the window exactly like the reference's line
// search (clipper.cpp:244-251, 261) — all workgroups reach the same decision on their own, no
// communication. Workgroup (0,0) records the decided state in st_next.
// Transitions (everything that sweeps whole vectors): workgroup (0,0) alone, the others leave.
// ------------------------------------------------------------------------------------------

// The workgroup that records the decided state (and alone performs the transition sweeps): the
// LAST one of the grid — on the slices the workgroups are ordered most expensive first, so this
// is the one with the least to stream.
__device__ __forceinline__ bool is_writer_block() {
  return blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;
}

// The view policy: build a view of the `nlive` live rows now? Rows a pass streams today: the view's
// while it covers the live rows, else all of M's; as many passes to come as there have been
// iterations (at least 12: the penalty homotopy's later outer iterations are the long ones).
__device__ __forceinline__ bool view_wanted_v(const ViewPolicy& rvp, int64_t m, bool have_view, int rv_rows,
                                              int nlive, int nout, int64_t n_iters, int builds, int last,
                                              int backoff) {
  if (builds >= rvp.max_builds || n_iters < 2 || n_iters - last < 3) return false;
  if (nlive <= 0 || nlive >= m) return false;
  const double rows_now = (have_view && nout == 0) ? static_cast<double>(rv_rows) : static_cast<double>(m);
  const double r = static_cast<double>(nlive);
  if (r > RV_ROWS_RATIO * rows_now) return false;
  if (backoff > 0 && r > RV_ROWS_RATIO * static_cast<double>(backoff)) return false;  // nothing much changed since a refusal
  const double horizon = n_iters > 12 ? static_cast<double>(n_iters) : 12.0;
  const double gain = horizon * ((rows_now - r) * rvp.pass_per_row);
  // (a margin: a view that only just pays by this model does not — at m = 100k / 300k a marginal middle view between
  // the first, large one and the last, small one cost 10 / 89 ms of fill for passes that the next view took over a
  // few iterations later: profiles/r04_view_policy_margin.txt)
  return gain > RV_GAIN_MARGIN * (rvp.build_fixed + r * rvp.build_per_row);
}
__device__ __forceinline__ bool view_wanted(const SolveArgs& A, int nlive, int nout, int64_t n_iters,
                                            int builds, int last, int backoff) {
  return view_wanted_v(A.rvp, A.m, A.in_view != nullptr, A.rv_rows, nlive, nout, n_iters, builds, last, backoff);
}

// dst = *src by the threads of a workgroup, 8 bytes each
__device__ __forceinline__ void copy_state(SolverState* dst, const SolverState* src, int tid, int nt) {
  static_assert(sizeof(SolverState) % 8 == 0, "copied in 8-byte words");
  const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long* d8 = reinterpret_cast<unsigned long long*>(dst);
  for (int k = tid; k < static_cast<int>(sizeof(SolverState) / 8); k += nt) d8[k] = s8[k];
}

struct PassPlan {
  int view;    // 1: stream the row view instead of M
  int phase;
  int sel;     // table of Xin that holds the pending window, or
  int from_u;  // -1, or (p*V + v): pair-mode pass straight on the u array of point slot (p, v)
  double d;
  // the pending window without its table: candidate l = max(u' + alpha0 beta^l g', 0) with (u', g')
  // the arrays of point slot `src` (p*V + v) — what the pass on the slices stages (k_slices.hip.h)
  int src;
  double alpha0;
  int weff;    // candidates of the window this pass multiplies (1 or V)
};

constexpr int VU = 4;  // elements per thread per sweep step (all NT threads of the workgroup sweep)
#define VEC_CHUNKS(base) for (int64_t base = tid; base < m; base += NT * VU)
#define VEC_EACH(k, i, base)            \
  _Pragma("unroll") for (int k = 0; k < VU; ++k) \
    if (const int64_t i = base + static_cast<int64_t>(k) * NT; i < m)

// What the head of a launch loads before it knows whether it needs it: the scalar state and this
// thread's chain of the tail's partial scalars. Requested together, up front — one round trip
// to L2 instead of three dependent ones (done/stage -> state -> partials).
struct HeadLoads {
  int done, stage, phase;
  double d, F, alpha, s;
  int i, j, k, ubp, ubv, sel;
  int64_t n_passes, n_trials, n_iters, n_view_passes;
  int nlive, nout, hold, rv_builds, rv_last, rv_backoff, weff, zero_run, n_redo;
  double chain;  // this thread's share of sum_w scal[w][q]
};

template <int V, int NT>
__device__ __forceinline__ void head_loads(const SolveArgs& A, HeadLoads& L) {
  constexpr int Q = tail_q(V);
  constexpr int QPAD = pow2_at_least(Q);
  constexpr int NCH = NT / QPAD;
  const SolverState* st = A.st_cur;
  L.done = A.shared->done;
  L.stage = st->stage;
  L.phase = st->phase;
  L.d = st->d;
  L.F = st->F;
  L.alpha = st->alpha;
  L.s = st->s;
  L.i = st->i;
  L.j = st->j;
  L.k = st->k;
  L.ubp = st->ubp;
  L.ubv = st->ubv;
  L.sel = st->sel;
  L.n_passes = st->n_passes;
  L.n_trials = st->n_trials;
  L.n_iters = st->n_iters;
  L.n_view_passes = st->n_view_passes;
  L.nlive = st->nlive;
  L.nout = st->nout;
  L.hold = st->hold;
  L.rv_builds = st->rv_builds;
  L.rv_last = st->rv_last;
  L.rv_backoff = st->rv_backoff;
  L.weff = st->weff;
  L.zero_run = st->zero_run;
  L.n_redo = st->n_redo;
  // sums[q] = sum over the tail workgroups w of scal[w][q]: NCH interleaved chains per
  // quantity (w = c, c + NCH, ...), added in chain order
  const int tid = threadIdx.x;
  const int q = tid & (QPAD - 1), c = tid / QPAD;
  double acc = 0.0;
  if (q < Q) {
    constexpr int U = 20;  // loads in flight per chain (all of them up to m = 10k)
    const double* p = A.scal_in + q;
    int w = c;
    for (; w + NCH * (U - 1) < A.nwg_in; w += NCH * U) {
      double x[U];
#pragma unroll
      for (int k = 0; k < U; ++k) x[k] = p[static_cast<int64_t>(w + NCH * k) * Q];
#pragma unroll
      for (int k = 0; k < U; ++k) acc += x[k];
    }
    // the rest of the chain (fewer than U rows): again every load in flight at once, then the additions in chain
    // order — a loop of load-and-add here was a chain of 7 dependent round trips at m = 100k (13 rows after the fold)
    // and of 19 at m = 300k: 4 and 10 us of every workgroup's head (round 4)
    if (w < A.nwg_in) {
      auto rest = [&](auto un) __attribute__((always_inline)) {
        constexpr int R = decltype(un)::value;
        double x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int ww = w + NCH * k;
          x[k] = p[static_cast<int64_t>(ww < A.nwg_in ? ww : w) * Q];
        }
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (w + NCH * k < A.nwg_in) acc += x[k];
      };
      if (w + NCH * 4 >= A.nwg_in) rest(std::integral_constant<int, 4>{});
      else if (w + NCH * 10 >= A.nwg_in) rest(std::integral_constant<int, 10>{});
      else rest(std::integral_constant<int, U>{});
    }
  }
  L.chain = acc;
}

template <int V, int NT>
__device__ __forceinline__ bool decide(const SolveArgs& A, const HeadLoads& L, double* red,
                                       SolverState* stash, PassPlan& plan) {
  constexpr int NR = tail_nr(V);
  constexpr int PEN = V * NR + 2 * V;  // speculative penalty sums of candidate 0
  constexpr int Q = tail_q(V);
  constexpr int QPAD = pow2_at_least(Q);
  constexpr int NCH = NT / QPAD;  // interleaved summation chains per quantity
  constexpr int NWV = NT / 64;
  static_assert(QPAD <= NT && V <= VS, "window too large");
  double* scratch = red + NT;     // block_reduce scratch, NWV * 2 doubles at most
  const SolverState* st = A.st_cur;
  const int tid = threadIdx.x;
  const int64_t m = A.m;
  const SolverParams P = A.prm;
  const bool writer = is_writer_block();

  const int phase = L.phase;
  double d = L.d, F = L.F, alpha = L.alpha, s = L.s;
  int i_ = L.i, j_ = L.j, k_ = L.k, ubp = L.ubp, ubv = L.ubv, sel = L.sel;
  const int wprev = (L.weff >= 1 && L.weff <= V) ? L.weff : V;  // candidates the pass just evaluated
  int zero_run = L.zero_run;
  int64_t n_passes = L.n_passes, n_trials = L.n_trials;
  const int64_t n_iters = L.n_iters + 1;
  // live rows of the current point / of them outside the row view: unchanged while the point is
  int nlive = L.nlive, nout = L.nout;
  auto live_code = [&](double code) {  // live + 2^26 * outside, summed exactly
    const double o = floor(code * (1.0 / LIVE_OUT));
    nout = static_cast<int>(o);
    nlive = static_cast<int>(code - o * LIVE_OUT);
  };
  // The norms of the window a pass iteration leaves pending go straight into the LDS copy of the state
  // it records (`stash`: a pass iteration parks its state there, see below) — 4 V registers held from
  // here to the record were what tipped the streaming loop's allocation into scratch. Every other
  // outcome records the defaults (1, 0).
  if (writer && tid == 0) {
#pragma unroll
    for (int l = 0; l < V; ++l) {
      stash->nrm[l] = 1.0;
      stash->sx[l] = 0.0;
    }
  }
  // what this iteration does next
  enum { ACT_PASS, ACT_BUILD, ACT_SLOW, ACT_DONE };
  int action = ACT_SLOW;
  int next_phase = PH_TRIAL;
  bool need_pair = false;  // the pass of this iteration is a pair-mode pass on the accepted x
  bool redo = false;       // the pass on candidate 0 alone guessed wrong: the same window again, whole
  // The live sub-problem (k_subproblem.hip.h): no column outside it can come back to life under ANY candidate of the
  // window this decision leaves pending — d^2 s_l^2 >= kappa (1 + d)^2 N z_l for its raw sums (z_l, s_l) =
  // (||x_l||^2, sum x_l), which the tail left at sums[nb + 2 l], sums[nb + 2 l + 1]. Every workgroup evaluates it.
  bool sub_ok = true;
  auto sub_bound = [&](const double* sums, int nb) {
    if (A.sub_state == 0) return;
    const double kap = (A.sub_state == 1 ? 1.10 : 1.01) * A.sub_ncol;  // SUB_ENTER_MARGIN / SUB_STAY_MARGIN
    const double lhs = d * d, rhs = (1.0 + d) * (1.0 + d) * kap;
#pragma unroll
    for (int l = 0; l < V; ++l) {
      const double z = sums[nb + 2 * l], sl = sums[nb + 2 * l + 1];
      sub_ok = sub_ok && (lhs * sl * sl >= rhs * z);
    }
  };

  if (phase == PH_TRIAL || phase == PH_BUILD) {
    // sums[q] = the chains of head_loads(), added in chain order
    {
      red[tid] = L.chain;
      __syncthreads();
      double tot = 0.0;
      if (tid < QPAD) {
        tot = red[tid];
#pragma unroll
        for (int c2 = 1; c2 < NCH; ++c2) tot += red[c2 * QPAD + tid];
      }
      __syncthreads();
      if (tid < QPAD) red[tid] = tot;
      __syncthreads();
    }
    const double* sums = red;
    if (phase == PH_TRIAL) {
      // the decisions of clipper.cpp:244-262, candidate by candidate
      int jstar = -1;
      double Fnew = 0.0, deltaF = 0.0;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (jstar < 0 && v < wprev) {
          ++n_trials;
          Fnew = sums[v * NR + 0];
          deltaF = Fnew - F;  // :244
          bool accept = true;
          if (deltaF < -P.eps) {  // :246-248
            alpha = alpha * P.beta;
            ++k_;
            if (k_ < P.maxlsiters) accept = false;  // :234 loop bound; the last trial is kept
          }
          if (accept) jstar = v;
        }
      }
      if (jstar < 0 && wprev < V) {
        // A pass on candidate 0 ALONE (SolverState::weff) whose candidate was rejected: the guess was wrong. Nothing of
        // it is used: the SAME window is multiplied again, whole — every candidate then sits at the index and is formed
        // by the expressions it would have had without the guess, and the solve is bit for bit the one with full windows
        // (the trial is counted when that pass is walked). The pending window's norms go on unchanged.
        alpha = L.alpha;
        k_ = L.k;
        n_trials = L.n_trials;
        zero_run = 0;
        redo = true;
        action = ACT_PASS;
        // (the live sub-problem's bound for that window, from the norms the state carries: z = nrm^2, s = sx nrm — this
        // decision may be the first one that is asked, e.g. right after a view and its sub-problem were built)
        if (A.sub_state != 0) {
          const double kap = (A.sub_state == 1 ? 1.10 : 1.01) * A.sub_ncol;
          const double rhs = (1.0 + d) * (1.0 + d) * kap;
#pragma unroll
          for (int l = 0; l < V; ++l) {
            const double dsx = d * st->sx[l];
            sub_ok = sub_ok && (dsx * dsx >= rhs);
          }
        }
        if (writer && tid == 0) {
#pragma unroll
          for (int l = 0; l < V; ++l) {
            stash->nrm[l] = st->nrm[l];
            stash->sx[l] = st->sx[l];
          }
        }
      } else if (jstar < 0) {
        // all V candidates rejected: the v = 0 tail already built the next V step sizes from
        // the unchanged (u, g) in table V; alpha was multiplied by beta V times above
        sel = V;
        action = ACT_PASS;
        sub_bound(sums, V * NR);
        if (writer && tid == 0) {  // the norms are only recorded (for the tail): no other workgroup needs them
#pragma unroll
          for (int l = 0; l < V; ++l) {
            const double z = sums[V * NR + 2 * l];
            const double nl = (z > 0.0) ? sqrt(z) : 1.0;  // Eigen normalize(): only if squaredNorm > 0
            stash->nrm[l] = nl;
            stash->sx[l] = sums[V * NR + 2 * l + 1] / nl;
          }
        }
      } else {
        zero_run = (k_ == 0) ? (zero_run < 1000000 ? zero_run + 1 : zero_run) : 0;  // this line search is over
        const double deltau = sqrt(sums[jstar * NR + 1]);
        live_code(sums[jstar * NR + NR - 1]);  // the point becomes candidate jstar
        s = st->sx[jstar];
        F = Fnew;  // :256-258 — u <- x, gradF <- gradFnew: the point slot the tail filled
        ubp ^= 1;
        ubv = jstar;
        ++j_;
        if (deltau < P.tol_u || fabs(deltaF) < P.tol_F || j_ >= P.maxiniters) {  // :261, :226
          // end of the inner loop: the penalty update (:268-280) needs M_off u and C_off u apart
          if (jstar == 0) {
            // candidate 0 carries them (cab) and the tail already summed its penalty terms
            const double cnt = sums[PEN], rs = sums[PEN + 1];
            if (cnt > 0.0) {
              d += rs / cnt;  // :276
              ++i_;           // :218 loop increment
              action = (i_ >= P.maxoliters) ? ACT_DONE : ACT_BUILD;
            } else {
              action = ACT_DONE;  // :278-280 break
            }
          } else {
            // pair-mode pass straight on the accepted x (already normalised, in its point slot)
            need_pair = true;
            action = ACT_PASS;
            next_phase = PH_PENALTY;
          }
        } else {
          alpha = 1.0;  // :227
          k_ = 0;
          sel = jstar;  // the tail already built max(x + beta^l gradFnew, 0) in table jstar
          action = ACT_PASS;
          sub_bound(sums, jstar * NR + 2);
          if (writer && tid == 0) {
#pragma unroll
            for (int l = 0; l < V; ++l) {
              const double z = sums[jstar * NR + 2 + 2 * l];
              const double nl = (z > 0.0) ? sqrt(z) : 1.0;
              stash->nrm[l] = nl;
              stash->sx[l] = sums[jstar * NR + 3 + 2 * l] / nl;
            }
          }
        }
      }
    } else {  // PH_BUILD: the tail formed gradF, F and the first window of an outer iteration
      F = sums[0];  // :220
      live_code(sums[NR - 1]);  // the gradient changed with the penalty
      j_ = 0;
      if (P.maxiniters <= 0) {
        action = ACT_SLOW;  // empty inner loop: u unchanged, its (a, b) in cab are still valid
      } else {
        alpha = 1.0;
        k_ = 0;
        sel = 0;
        action = ACT_PASS;
        sub_bound(sums, 2);
        if (writer && tid == 0) {
#pragma unroll
          for (int l = 0; l < V; ++l) {
            const double z = sums[2 + 2 * l];
            const double nl = (z > 0.0) ? sqrt(z) : 1.0;
            stash->nrm[l] = nl;
            stash->sx[l] = sums[3 + 2 * l] / nl;
          }
        }
      }
    }
  }

  if (action == ACT_SLOW) {
    // ---- sweeps over whole vectors: workgroup (0,0) alone -----------------------------------
    if (!writer) return false;
    nlive = nout = static_cast<int>(m < 0x7fffffff ? m : 0x7fffffff);  // unknown until a tail counts again
    const double* ca_ = A.cab;         // a = M_off x of the last pair-mode pass / of candidate 0
    const double* cb_ = A.cab + A.mp;  // b = C_off x
    if (phase == PH_NORMALIZE || phase == PH_RESCALE) {
      // clipper.cpp:193-198 — u = M_off*u0 + u0 (or u0), then u /= u.norm()
      double* u = pt_arr(A, V, ubp, ubv, 0);
      double z[1] = {0.0};
      VEC_CHUNKS(base) {
        double uv[VU], av[VU];
        VEC_EACH(k, i, base) {
          uv[k] = A.u0[i];
          if (phase == PH_RESCALE) av[k] = ca_[i];
        }
        VEC_EACH(k, i, base) {
          const double ui = (phase == PH_RESCALE) ? av[k] + uv[k] : uv[k];
          u[i] = ui;
          z[0] += ui * ui;
        }
      }
      block_reduce<1, NWV>(z, scratch);
      const double n0 = sqrt(z[0]);
      VEC_CHUNKS(base) {
        double uv[VU];
        VEC_EACH(k, i, base) uv[k] = u[i];
        VEC_EACH(k, i, base) {
          const double ui = uv[k] / n0;
          u[i] = ui;
          // next pass (pair mode) runs on x = u, already normalised: candidate 0 of table 0
          const double row[VS] = {ui, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          store_row(A.Xout + i * VS, row);
        }
      }
      sel = 0;
      next_phase = PH_INIT;  // action stays ACT_SLOW: "a pass was prepared", see the record below
    } else {
      if (phase == PH_INIT) {
        // clipper.cpp:200-209 — initial d from the pair-mode pass on u
        const double* u = pt_arr(A, V, ubp, ubv, 0);
        double sv[1] = {0.0};
        VEC_CHUNKS(base) {
          double uv[VU];
          VEC_EACH(k, i, base) uv[k] = u[i];
          VEC_EACH(k, i, base) sv[0] += uv[k];
        }
        block_reduce<1, NWV>(sv, scratch);
        s = sv[0];
        double ca[2] = {0.0, 0.0};  // count, sum of ratios
        VEC_CHUNKS(base) {
          double uv[VU], av[VU], bv[VU];
          VEC_EACH(k, i, base) {
            uv[k] = u[i];
            av[k] = ca_[i];
            bv[k] = cb_[i];
          }
          VEC_EACH(k, i, base) {
            const double cbu = s - bv[k] - uv[k];  // :202
            if (cbu > P.eps && uv[k] > P.eps) {    // :203
              ca[0] += 1.0;
              ca[1] += (av[k] + uv[k]) / cbu;  // :205-208
            }
          }
        }
        block_reduce<2, NWV>(ca, scratch);
        d = (ca[0] > 0.0) ? ca[1] / ca[0] : 0.0;
        i_ = 0;
        action = (i_ >= P.maxoliters) ? ACT_DONE : ACT_BUILD;  // :218 loop bound
      } else {
        // PH_PENALTY (the pair-mode pass on the inner loop's final u has run), or an empty
        // inner loop: penalty update :268-280
        const double* u = pt_arr(A, V, ubp, ubv, 0);
        double ca[2] = {0.0, 0.0};
        VEC_CHUNKS(base) {
          double uv[VU], av[VU], bv[VU];
          VEC_EACH(k, i, base) {
            uv[k] = u[i];
            av[k] = ca_[i];
            bv[k] = cb_[i];
          }
          VEC_EACH(k, i, base) {
            const double cbu = s - bv[k] - uv[k];  // :268
            if (cbu > P.eps && uv[k] > P.eps) {    // :269
              ca[0] += 1.0;
              ca[1] += fabs((av[k] + uv[k]) / cbu);  // :271-274
            }
          }
        }
        block_reduce<2, NWV>(ca, scratch);
        if (ca[0] > 0.0) {
          d += ca[1] / ca[0];  // :276
          ++i_;                // :218 loop increment
          action = (i_ >= P.maxoliters) ? ACT_DONE : ACT_BUILD;
        } else {
          action = ACT_DONE;  // :278-280 break
        }
      }
    }
  }

  // ---- the end: the deciding workgroup hands the final u to the host itself (pinned memory), so
  // that the host needs neither a copy nor a wait on the stream once it sees `done`
  if (writer && action == ACT_DONE && A.host_u != nullptr) {
    const double* u = pt_arr(A, V, ubp, ubv, 0);
    // (on the live sub-problem element i is association colmap[i]; the host zeroed the rest at the hand-over)
    for (int64_t i = tid; i < m; i += NT)
      __hip_atomic_store(A.host_u + (A.colmap != nullptr ? static_cast<int64_t>(A.colmap[i]) : i), u[i], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (A.kind != nullptr && A.marks != nullptr) {
      const int64_t n = (n_iters < KIND_CAP) ? n_iters : KIND_CAP;
      for (int64_t i = tid; i < n; i += NT)
        __hip_atomic_store(A.kind + i, A.marks[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
  }

  // The pass of this iteration runs on the row view when the view covers every live row of what it
  // multiplies: a window built from the decided point, or (pair mode) that point's u itself.
  // (a view built from exactly this state covers every outcome by construction — whatever the tail,
  // which ran before the view existed, counted)
  if (A.rv_fresh != 0 && action != ACT_SLOW) nout = 0;
  // HOLD: this iteration decides nothing — the state it read goes on unchanged, marked with what the host is asked for
  // (1: a row view of the decided point's live rows; 2: the hand-over to the live sub-problem)
  auto put_on_hold = [&](int reason) {
    if (writer) {  // (block-uniform; word by word: a struct copy by one thread is 66 registers)
      copy_state(A.st_next, st, tid, NT);
      __syncthreads();
      if (tid == 0) {
        A.st_next->hold = reason;
        A.st_next->hold_slot = ubp * V + ubv;  // (after the decision: the accepted candidate's slot, or the unchanged point's)
        A.st_next->hold_nlive = nlive;
        A.shared->hold = reason;
        if (A.host != nullptr) {
          __hip_atomic_store(&A.host->hold_nlive, nlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&A.host->hold, reason, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  };
  // Is it time to build a (smaller) row view? A function of the state alone; see LIVE ROWS.
  if (action == ACT_PASS && next_phase == PH_TRIAL && A.rvp.on != 0 && A.rv_fresh == 0 && A.sub_state != 2 &&
      view_wanted(A, nlive, nout, n_iters - (L.n_redo + (redo ? 1 : 0)), L.rv_builds, L.rv_last, L.rv_backoff)) {
    put_on_hold(1);
    return false;
  }
  const bool on_view = A.in_view != nullptr && nout == 0 && action == ACT_PASS &&
                       (next_phase == PH_TRIAL || need_pair);
  // The live sub-problem stands ready and the window this decision leaves pending cannot bring a column outside it back
  // to life (and the view covers the live rows: they all lie inside it): hold for the hand-over.
  if (A.sub_state == 1 && A.decide_only == 0 && on_view && next_phase == PH_TRIAL && !need_pair && sub_ok) {
    put_on_hold(2);
    return false;
  }
  // ... or these launches run ON it and the pending window could: the pass is left prepared and the solve goes back
  const bool sub_leave = A.sub_state == 2 && action == ACT_PASS && !sub_ok;
  // How many candidates of the window the pass multiplies (SolverState::weff): one while line searches have been
  // accepting their first trial (two in a row, or none has run yet) and this one has not rejected anything either.
  const int wnext = (A.adaptive_window != 0 && !redo && k_ == 0 && zero_run >= 2) ? 1 : V;
  // ---- record the decided state (workgroup (0,0), one thread) ------------------------------
  // A pass iteration parks it in LDS and writes it out AFTER the streaming loop (flush_state):
  // a global store ahead of the loop would make the compiler treat the table rows as possibly
  // clobbered and turn their scalar loads into per-lane vector loads.
  if (writer && tid == 0) {
    // (two call sites so that each store keeps its address space: LDS or global, never flat)
    auto record = [&](SolverState* o, bool norms_in_place) {
      o->d = d;
      o->F = F;
      o->alpha = alpha;
      o->s = s;
      if (!norms_in_place) {
#pragma unroll
        for (int l = 0; l < V; ++l) {
          o->nrm[l] = 1.0;
          o->sx[l] = 0.0;
        }
      }
      o->sel = sel;
      o->ubp = ubp;
      o->ubv = ubv;
      // ACT_PASS : this iteration streams, its results are due next time
      // ACT_BUILD: no pass; the tail of this iteration forms gradF, F and the first window
      // ACT_SLOW : (normalisation) the next iteration runs the pass this one prepared
      o->phase = (action == ACT_BUILD) ? static_cast<int>(PH_BUILD) : next_phase;
      o->stage = (action == ACT_SLOW) ? ST_PASS : ST_RESULTS;
      o->i = i_;
      o->j = j_;
      o->k = k_;
      o->n_passes = n_passes + (action == ACT_PASS ? 1 : 0);
      o->n_trials = n_trials;
      o->n_iters = n_iters;
      o->nlive = nlive;
      o->nout = nout;
      o->view = (action == ACT_PASS && on_view) ? 1 : 0;
      o->hold = 0;
      o->n_view_passes = L.n_view_passes + ((action == ACT_PASS && on_view) ? 1 : 0);
      o->rv_builds = L.rv_builds;
      o->rv_last = L.rv_last;
      o->rv_backoff = L.rv_backoff;
      o->hold_slot = 0;
      o->hold_nlive = 0;
      o->resume = 0;
      o->weff = wnext;
      o->zero_run = zero_run;
      o->n_redo = L.n_redo + (redo ? 1 : 0);
      o->pad_ = 0;
    };
    if (action == ACT_PASS) record(stash, true);
    else record(A.st_next, false);
    if ((A.decide_only != 0 || sub_leave) && action == ACT_PASS) {
      // the pass is left PREPARED: window from point slot (ubp, ubv) with step `alpha` and the norms just
      // parked, or (penalty update) the pair-mode pass on that slot's u
      stash->stage = ST_PASS;
      stash->resume = need_pair ? 2 : 1;
      stash->n_passes = n_passes;
      stash->view = 0;
      stash->n_view_passes = L.n_view_passes;
      if (sub_leave) {  // (hold = 3: the host scatters the point back and goes on with the full problem's launches)
        stash->hold = 3;
        A.shared->hold = 3;
      }
    }
    if (action == ACT_DONE) {
      SolveShared* sh = A.shared;
      sh->F = F;
      sh->d = d;
      sh->n_passes = n_passes;
      sh->n_trials = n_trials;
      sh->ifinal = i_;
      sh->ubp = ubp;
      sh->ubv = ubv;
      sh->done = 1;
    }
    if (A.host != nullptr) {
      HostMirror* hm = A.host;
      if (action == ACT_DONE) {
        __hip_atomic_store(&hm->F, F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->d, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->n_passes, n_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->n_trials, n_trials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ifinal, i_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ubp, ubp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ubv, ubv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // every store above (and the vectors this workgroup wrote) before the flag
        __hip_atomic_store(&hm->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (action != ACT_PASS || A.decide_only != 0 || sub_leave) {
        __hip_atomic_store(&hm->n_passes, n_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->nlive, nlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->nout, nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->iters, n_iters, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (sub_leave) __hip_atomic_store(&hm->hold, 3, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if ((A.decide_only != 0 || sub_leave) && action == ACT_PASS) {  // (block-uniform) nothing streams: the prepared pass goes out as the state
    if (writer) {
      __syncthreads();
      copy_state(A.st_next, stash, tid, NT);
    }
    return false;
  }
  // the decision came out of LDS reads: tell the compiler it is wave-uniform, so that the
  // multipliers of the streaming loop stay scalar loads
  plan.view = __builtin_amdgcn_readfirstlane(on_view ? 1 : 0);
  plan.phase = __builtin_amdgcn_readfirstlane(next_phase);
  plan.sel = __builtin_amdgcn_readfirstlane(sel);
  plan.from_u = __builtin_amdgcn_readfirstlane(need_pair ? ubp * V + ubv : -1);
  plan.d = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(d)),
                            __builtin_amdgcn_readfirstlane(__double2loint(d)));
  // every window the decision can leave pending starts from the point it ends on: the accepted
  // candidate (alpha = 1), or the unchanged point with the V factors of beta the walk multiplied in
  plan.src = __builtin_amdgcn_readfirstlane(ubp * V + ubv);
  plan.alpha0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(alpha)),
                                 __builtin_amdgcn_readfirstlane(__double2loint(alpha)));
  plan.weff = __builtin_amdgcn_readfirstlane(wnext);
  return action == ACT_PASS;
}
#undef VEC_CHUNKS
#undef VEC_EACH

// What every G launch starts with. Returns false when this workgroup has nothing to stream.
// `stash`: LDS copy of the state a pass iteration decided on, see flush_state.
template <int V, int NT>
__device__ __forceinline__ bool iteration_head(const SolveArgs& A, double* lds,
                                               SolverState* stash, PassPlan& plan) {
  const SolverState* st = A.st_cur;
  HeadLoads L;
  head_loads<V, NT>(A, L);
  if (L.done) return false;
  if (L.hold) {  // waiting for the host to build a row view: the state goes on as it is
    if (is_writer_block()) copy_state(A.st_next, st, threadIdx.x, NT);
    return false;
  }
  if (L.stage == ST_RESULTS) return decide<V, NT>(A, L, lds, stash, plan);
  // the pass was prepared — by a transition iteration or k_init (pair mode on candidate 0 of table
  // `sel`: u0, the normalised u), or by a decide-only launch / the resident solver on a row view
  // (SolverState::resume: a window from the point slot, or the pair-mode pass on its u): run it as it stands
  const int resume = st->resume;
  const int slot = L.ubp * V + L.ubv;
  const int on_view = (resume != 0 && A.in_view != nullptr && L.nout == 0) ? 1 : 0;
  if (A.decide_only != 0) {  // nothing to decide: the state goes on as it is
    if (is_writer_block()) copy_state(A.st_next, st, threadIdx.x, NT);
    return false;
  }
  plan.view = __builtin_amdgcn_readfirstlane(on_view);
  plan.phase = st->phase;
  plan.sel = st->sel;
  plan.from_u = __builtin_amdgcn_readfirstlane(resume == 2 ? slot : -1);
  plan.d = st->d;
  plan.src = __builtin_amdgcn_readfirstlane(slot);  // (read by window passes only)
  plan.alpha0 = L.alpha;
  plan.weff = __builtin_amdgcn_readfirstlane((A.adaptive_window != 0 && L.weff == 1) ? 1 : V);
  if (is_writer_block()) {  // (word by word by the whole workgroup: one thread's struct copy is 70 registers)
    copy_state(stash, st, threadIdx.x, NT);
    __syncthreads();
    if (resume == 3) {  // (uniform, rare) the window's norms: the rows outside the view are missing
      // (no view at hand any more — nothing queues a launch like that today —: the raw sums over ALL rows, which is
      // what the two parts add up to; the view's part that came with the state is then left out below)
      const bool all_rows = A.in_view == nullptr;
      double r[2 * V];
#pragma unroll
      for (int q = 0; q < 2 * V; ++q) r[q] = 0.0;
      const double* u = pt_arr(A, V, L.ubp, L.ubv, 0);
      const double* g = pt_arr(A, V, L.ubp, L.ubv, 1);
      const double beta = A.prm.beta;
      for (int64_t i = threadIdx.x; i < A.m; i += NT) {
        if (all_rows || A.in_view[i] == 0) {
          const double ui = u[i], gi = g[i];
          double al = L.alpha;
#pragma unroll
          for (int l = 0; l < V; ++l) {
            double t = ui + al * gi;  // clipper.cpp:235-236
            t = (t > 0.0) ? t : 0.0;
            r[2 * l] += t * t;
            r[2 * l + 1] += t;
            al = al * beta;
          }
        }
      }
      block_reduce<2 * V, NT / 64>(r, lds);
      if (threadIdx.x == 0) {
#pragma unroll
        for (int l = 0; l < V; ++l) {
          const double z = (all_rows ? 0.0 : stash->nrm[l]) + r[2 * l], sm = (all_rows ? 0.0 : stash->sx[l]) + r[2 * l + 1];
          const double nl = (z > 0.0) ? sqrt(z) : 1.0;  // Eigen normalize(): only if squaredNorm > 0 (:237)
          stash->nrm[l] = nl;
          stash->sx[l] = sm / nl;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      stash->stage = ST_RESULTS;
      stash->n_passes = L.n_passes + 1;
      stash->n_iters = L.n_iters + 1;
      stash->view = on_view;
      stash->n_view_passes = L.n_view_passes + on_view;
      stash->resume = 0;
      stash->weff = plan.weff;  // (what the pass multiplies is what its tail evaluates and the next decision walks)
    }
  }
  return true;
}

// End of a pass iteration: workgroup (0,0) writes the state it decided on where the tail and
// the next iteration read it, marks the iteration as a pass and reports progress to the host.
__device__ __forceinline__ void flush_state(const SolveArgs& A, const SolverState* stash) {
  if (!is_writer_block()) return;
  copy_state(A.st_next, stash, threadIdx.x, blockDim.x);
  if (threadIdx.x == 0) {
    const int64_t n_iters = stash->n_iters;
    // what this launch streamed: 1 = a window pass on M, 2 = a pass on the row view, 3 = a pair-mode pass
    // on M (one vector: initialisation, penalty update), 4 / 5 = a window / pair-mode pass on the live
    // sub-problem — the pass timings keep them apart
    if (A.marks != nullptr && n_iters <= KIND_CAP)
      A.marks[n_iters - 1] = A.sub_state == 2 ? (stash->phase == PH_TRIAL ? 4 : 5) : stash->view ? 2 : (stash->phase == PH_TRIAL ? 1 : 3);
    if (A.host != nullptr) {
      __hip_atomic_store(&A.host->nlive, stash->nlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&A.host->nout, stash->nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&A.host->n_view_passes, stash->n_view_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&A.host->n_passes, stash->n_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&A.host->iters, n_iters, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Solve prologue, one launch: pending vector = u0 (candidate 0 of table 0, un-normalised,
// nrm = 1), initial state in ST[0].
__global__ __launch_bounds__(256) void k_init(SolveArgs A, SolverState init, SolverState* st0,
                                               double* X0) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < A.m) {
    const double row[VS] = {A.u0[i], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    store_row(X0 + i * VS, row);
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      *st0 = init;
      A.shared->done = 0;
      A.shared->hold = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// tail — grid (ceil(m/256), V): workgroup (blk, v) handles candidate v of 256 elements
// (one element per thread: more elements per thread only lengthens the latency chain — measured).
//   FUSED_REDUCE: sum the row-tile partials of the single shard here (else `ab` holds the
//                 gathered raw sums of all shards).
// Reads the state G decided on (st_next); communicates with nobody.
// ------------------------------------------------------------------------------------------
// k_scal_fold — out[b][q] = sum of scal[w][q] over the SCAL_FOLD tail workgroups w of block b, in
// order. Launched after the tail when nwg > SCAL_FOLD_MIN (m > 16k): the decision at the head of
// every workgroup of the next pass then reads nwg/SCAL_FOLD rows instead of nwg.
constexpr int SCAL_FOLD = 32;
constexpr int SCAL_FOLD_MIN = 64;
__global__ __launch_bounds__(128) void k_scal_fold(const double* __restrict__ scal, int nwg, int Q,
                                                    double* __restrict__ out,
                                                    const SolveShared* __restrict__ shared) {
  if (shared->done) return;
  const int w0 = blockIdx.x * SCAL_FOLD;
  const int w1 = (w0 + SCAL_FOLD < nwg) ? w0 + SCAL_FOLD : nwg;
  for (int q = threadIdx.x; q < Q; q += 128) {
    // all SCAL_FOLD loads in flight at once, then the additions in row order (a loop of load-and-add is a chain of
    // 32 round trips: the launch took 9.5 us for 43 KB — round 4)
    double x[SCAL_FOLD];
#pragma unroll
    for (int k = 0; k < SCAL_FOLD; ++k) {
      const int w = (w0 + k < w1) ? w0 + k : w1 - 1;
      x[k] = scal[static_cast<int64_t>(w) * Q + q];
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < SCAL_FOLD; ++k)
      if (w0 + k < w1) acc += x[k];
    out[static_cast<int64_t>(blockIdx.x) * Q + q] = acc;
  }
}

//   TABLES: write the candidate tables of the next windows (the dense pass reads its multipliers
//           from them with scalar loads). The pass on the slices builds a window from the point
//           slot itself while it stages the rows (PassPlan::src): no tables — 64 bytes less to
//           store per element and outcome, and nothing for the kernel boundary to write back.
template <int V, bool FUSED_REDUCE, bool TABLES = true>
__global__ __launch_bounds__(TAIL_THREADS * (FUSED_REDUCE ? TAIL_SPLIT : 1)) void k_tail(SolveArgs A) {
  constexpr int NR = tail_nr(V);
  constexpr int Q = tail_q(V);
  constexpr int NRED = NR + 2 * V + 2;  // what a v = 0 workgroup reduces
  constexpr int NSLOT = nslot(V);
  constexpr int NTW = TAIL_WAVES * (FUSED_REDUCE ? TAIL_SPLIT : 1);  // waves of the workgroup
  __shared__ double red[NTW * NRED > 2 * TAIL_SPLIT * TAIL_THREADS ? NTW * NRED : 2 * TAIL_SPLIT * TAIL_THREADS];
  const int v = blockIdx.y;
  const long long c0 = A.stamps ? wall_clock64() : 0;
  const SolverState* st = A.st_next;
  const int te = threadIdx.x & (TAIL_THREADS - 1);  // element of the workgroup
  const int grp = threadIdx.x / TAIL_THREADS;        // slot quarter (FUSED_REDUCE), else 0
  const int64_t i = static_cast<int64_t>(blockIdx.x) * TAIL_THREADS + te;
  const bool valid = i < A.m && grp == 0;
  // everything this launch needs of the solver state, requested up front: one round trip in the
  // shadow of the partial sums instead of a chain of dependent scalar loads after them
  const int done = A.shared->done;
  const int stage = st->stage, phase = st->phase;
  const int ubp = st->ubp, ubv = st->ubv, sel = st->sel;
  const double d = st->d, alpha = st->alpha, s_cur = st->s;
  const double nrmv = st->nrm[v], sxv = st->sx[v];
  const int weff = (st->weff >= 1 && st->weff <= V) ? st->weff : V;  // candidates the pass multiplied (SolverState::weff)
  // the pass streamed the row view: its own (fewer) partial-sum slots
  const int nslots_pass = st->view ? A.rv_nslots : A.ntiles;
  // "live and outside the view" weighs 1 + 2^26 in the live code
  const double live_w = (valid && A.in_view != nullptr && A.in_view[i] == 0) ? 1.0 + LIVE_OUT : 1.0;

  // raw sums of slot v (and of slot V, the b of candidate 0, for the v = 0 workgroups): these
  // loads do not depend on the solver state
  double p0 = 0.0, p1 = 0.0;
  if (FUSED_REDUCE) {  // single shard: W >= m; this group's quarter of the slots, in slot order
    const int64_t o1 = (v == 0) ? static_cast<int64_t>(V) * A.W : 0;  // slot V relative to slot 0
    const int per = (nslots_pass + TAIL_SPLIT - 1) / TAIL_SPLIT;
    const int t0 = grp * per, t1 = (t0 + per < nslots_pass) ? t0 + per : nslots_pass;
    if (i < A.m) {
      const double* p = A.part + static_cast<int64_t>(v) * A.W + i;
      const int64_t ts = static_cast<int64_t>(NSLOT) * A.W;
      // 16 slots per round trip, every load issued (a slot past the end re-reads the last one and
      // is not added): ~28 slots at m = 10k are two round trips (32 at once measured slower)
      for (int t = t0; t < t1; t += 16) {
        double va[16], vb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int tt = (t + q < t1) ? t + q : t1 - 1;
          va[q] = p[static_cast<int64_t>(tt) * ts];
          vb[q] = p[static_cast<int64_t>(tt) * ts + o1];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if (t + q < t1) {
            p0 += va[q];
            p1 += vb[q];
          }
        }
      }
    }
    if constexpr (TAIL_SPLIT > 1) {
      red[(grp * 2 + 0) * TAIL_THREADS + te] = p0;
      red[(grp * 2 + 1) * TAIL_THREADS + te] = p1;
      __syncthreads();
      p0 = red[te];
      p1 = red[TAIL_THREADS + te];
#pragma unroll
      for (int g = 1; g < TAIL_SPLIT; ++g) {
        p0 += red[(g * 2 + 0) * TAIL_THREADS + te];
        p1 += red[(g * 2 + 1) * TAIL_THREADS + te];
      }
      __syncthreads();  // `red` is reused by the reduction at the end
    }
  } else if (valid) {  // block pb = i / W of the gathered [P][NSLOT][W] layout (32-bit division)
    const int64_t o1 = (v == 0) ? static_cast<int64_t>(V) * A.W : 0;
    const uint32_t pb = static_cast<uint32_t>(i) / static_cast<uint32_t>(A.W);
    const int64_t off = i - static_cast<int64_t>(pb) * A.W;
    const double* blk = A.ab + (static_cast<int64_t>(pb) * NSLOT) * A.W;
    p0 = blk[static_cast<int64_t>(v) * A.W + off];
    p1 = blk[o1 + off];
  }
  if (done) return;
  if (st->hold) return;             // the solve waits for a row view: nothing was decided, nothing ran
  if (stage != ST_RESULTS) return;  // a pass was only prepared: nothing to evaluate
  if (phase == PH_TRIAL && v >= weff) return;  // a candidate the pass did not multiply: the decision will not look at it
  const long long c1 = A.stamps ? wall_clock64() + (p0 > 1e300 ? 1 : 0) : 0;

  if (phase != PH_TRIAL && phase != PH_BUILD) {
    // pair-mode passes carry one vector (candidate 0, nrm = 1): a = M_off x, b = C_off x
    if (v == 0 && valid) {
      A.cab[i] = p0;
      A.cab[A.mp + i] = p1;
    }
    return;
  }
  const double beta = A.prm.beta;
  double r[NRED];
#pragma unroll
  for (int q = 0; q < NRED; ++q) r[q] = 0.0;

  if (phase == PH_BUILD) {
    // start of an outer iteration (clipper.cpp:219-220, :235-236): gradF and F at the current u
    // under the new penalty, and the first window (alpha = 1, beta, ...) — v = 0 workgroups
    if (v != 0) return;
    if (valid) {
      const double ui = pt_arr(A, V, ubp, ubv, 0)[i];
      const double gi = (1 + d) * ui - d * s_cur + A.cab[i] + A.cab[A.mp + i] * d;  // :219
      pt_arr(A, V, ubp, ubv, 1)[i] = gi;
      r[0] = ui * gi;  // :220
      r[NR - 1] = (ui > 0.0 || gi > 0.0) ? live_w : 0.0;
      double row[VS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      double al = 1.0;
#pragma unroll
      for (int l = 0; l < V; ++l) {
        double t = ui + al * gi;
        t = (t > 0.0) ? t : 0.0;
        row[l] = t;
        r[2 + 2 * l] = t * t;
        r[3 + 2 * l] = t;
        al = al * beta;
      }
      if constexpr (TABLES) store_row(A.Xout + i * VS, row);
    }
  } else {
    if (valid) {
      const double ui = pt_arr(A, V, ubp, ubv, 0)[i];
      double xraw;
      if constexpr (TABLES) {
        xraw = A.Xin[(static_cast<int64_t>(sel) * A.mp + i) * VS + v];
      } else {  // candidate v of the window the pass staged: the same expression, the same bits
        double al = alpha;
        for (int l = 0; l < v; ++l) al = al * beta;
        const double t = ui + al * pt_arr(A, V, ubp, ubv, 1)[i];
        xraw = (t > 0.0) ? t : 0.0;
      }
      const double xi = xraw / nrmv;  // clipper.cpp:237
      double gn;
      if (v == 0) {
        // candidate 0: a and b apart, exactly the reference's expression (:238-241)
        const double an = p0 / nrmv, bn = p1 / nrmv;
        gn = (1 + d) * xi - d * sxv + an + bn * d;
        A.cab[i] = an;  // (a, b) of the current point if candidate 0 is accepted
        A.cab[A.mp + i] = bn;
        // its penalty terms (:268-274), in case the inner loop ends with it
        const double cbu = sxv - bn - xi;
        if (cbu > A.prm.eps && xi > A.prm.eps) {
          r[NR + 2 * V] = 1.0;
          r[NR + 2 * V + 1] = fabs((an + xi) / cbu);
        }
      } else {
        const double gs = p0 / nrmv;  // (M_off + d*C_off) x
        gn = (1 + d) * xi - d * sxv + gs;
      }
      pt_arr(A, V, ubp ^ 1, v, 0)[i] = xi;  // becomes (u, gradF) if candidate v is accepted
      pt_arr(A, V, ubp ^ 1, v, 1)[i] = gn;
      r[0] = xi * gn;  // :242
      const double du = xi - ui;
      r[1] = du * du;  // :253
      r[NR - 1] = (xi > 0.0 || gn > 0.0) ? live_w : 0.0;  // live rows of the point it would become
      // next window if candidate v is accepted: alpha = 1, beta, beta^2, ... (:227, :235-236)
      double row[VS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      double al = 1.0;
#pragma unroll
      for (int l = 0; l < V; ++l) {
        double t = xi + al * gn;
        t = (t > 0.0) ? t : 0.0;
        row[l] = t;
        r[2 + 2 * l] = t * t;
        r[3 + 2 * l] = t;
        al = al * beta;
      }
      if constexpr (TABLES) store_row(A.Xout + (static_cast<int64_t>(v) * A.mp + i) * VS, row);
      if (v == 0) {
        // next window if all V candidates are rejected: V more factors of beta (:248)
        // (after a pass on candidate 0 alone — weff = 1 — this outcome is not used: a rejection repeats the window)
        const double gi = pt_arr(A, V, ubp, ubv, 1)[i];
        double row2[VS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        al = alpha;
#pragma unroll
        for (int l = 0; l < V; ++l) al = al * beta;
#pragma unroll
        for (int l = 0; l < V; ++l) {
          double t = ui + al * gi;
          t = (t > 0.0) ? t : 0.0;
          row2[l] = t;
          r[NR + 2 * l] = t * t;
          r[NR + 2 * l + 1] = t;
          al = al * beta;
        }
        if constexpr (TABLES) store_row(A.Xout + (static_cast<int64_t>(V) * A.mp + i) * VS, row2);
      }
    }
  }
  const double tot = block_reduce_pick<NRED, NTW>(r, red);
  double* out = A.scal + static_cast<int64_t>(blockIdx.x) * Q;
  if (A.stamps && !A.stamps_wide && threadIdx.x == 0) {
    const int w = 1536 + static_cast<int>(blockIdx.y * gridDim.x + blockIdx.x);
    if (w < 2048) {
      A.stamps[w * 4 + 0] = c0;
      A.stamps[w * 4 + 1] = c1;
      A.stamps[w * 4 + 2] = wall_clock64();
      A.stamps[w * 4 + 3] = phase;
    }
  }
  if (threadIdx.x < NR) out[v * NR + threadIdx.x] = tot;
  if (v == 0 && threadIdx.x >= NR && threadIdx.x < NRED)
    out[V * NR + (threadIdx.x - NR)] = tot;  // "all rejected" window sums, then the penalty sums
}

}  // namespace clipper_hip
