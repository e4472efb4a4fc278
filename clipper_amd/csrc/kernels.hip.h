// kernels.hip.h — hand-written CDNA4 (gfx950, wave64) kernels of the CLIPPER hot path.
//
// Compiled with -ffp-contract=off: every fp64 expression rounds exactly as written and
// fused multiply-adds appear only where fma() is spelled out — the same convention as the
// CPU oracle, so that threshold decisions (c < epsilon, scr > affinityeps, deltaF < -eps)
// are taken on bit-identical operands wherever the operation order can be shared.
//
// Data layout (see include/clipper_hip.h): a shard owns global columns [c0, c0+W) of the
// symmetric matrix M_off and stores S[j][c] = M(j, c0+c), j = 0..m-1, row pitch ld = W
// (multiple of 64 elements, zero padded). Because M is symmetric, column c of S is row
// (c0+c) of M, so   (M_off x)[c0+c] = sum_j S[j][c] * x[j]:
// every lane owns output columns, walks down the rows with a wave-uniform x[j] (scalar
// load), and never needs a cross-lane reduction; a wave's 64 lanes read 64 x 16 B =
// 1 KiB of one row per load instruction — fully coalesced HBM streaming.
//
// Reference sites (relative to /root/reference):
//   k_affinity_*  : src/clipper.cpp:31-56 + src/invariants/euclidean_distance.cpp:13-31,
//                   src/invariants/pointnormal_distance.cpp:13-35
//   k_gemv / k_pass: every `M_.selfadjointView<Upper>() * v` / `C_...* v` in
//                   src/clipper.cpp:194,202,205,219,240-241,268,271 (one pass over M serves a
//                   whole window of line-search candidates), preceded by the decisions of
//                   findDenseClique's control flow :244-262, :268-280 (decide)
//   k_tail        : the O(m) algebra of findDenseClique, src/clipper.cpp:219-220 (gradient),
//                   :235-242, :253 (step, projection, trial objective), :268-274 (penalty terms)
//   k_affinity_sym: the same scores as k_affinity_*, upper block triangle + mirrored stores
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
};

// What outlives the alternating state: the end of the solve. Kernels launched after
// convergence see `done` and return immediately.
struct SolveShared {
  double F, d;
  int64_t n_passes, n_trials;
  int32_t ifinal, ubp, ubv;
  int32_t done;
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
};

struct SolverParams {
  double tol_u, tol_F, beta, eps;
  int32_t maxiniters, maxoliters, maxlsiters;
};

constexpr int TAIL_THREADS = 256;
constexpr int TAIL_WAVES = TAIL_THREADS / 64;

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
  int* cnt;       // column-sharded M: one arrival counter per column strip
  int nstrips;
  uint8_t* kind;  // pinned host memory [KIND_CAP], profiling only: iteration n_iters ran a pass
  double* host_u; // pinned host memory [m] (may be null): the final u, written before `done`
};
constexpr int KIND_CAP = 1 << 16;

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// array k (0 = u, 1 = gradF) of point slot (p, v)
__device__ __forceinline__ double* pt_arr(const SolveArgs& A, int V, int p, int v, int k) {
  return A.pt + ((static_cast<int64_t>(p) * V + v) * 2 + k) * A.mp;
}

// Last-arriver hand-off inside one launch (CDNA guide, section 6 guideline 16, counter form;
// the split-K recipe) — used only by k_pass (column-sharded M): every workgroup publishes what
// it stored — each wave drains its own stores, one lane issues the agent-scope release and
// draws a ticket — and the workgroup that draws the last ticket acquires at agent scope and
// continues with plain loads. Correct for any placement of the workgroups over the 8 XCDs
// (their L2s are not coherent with each other). The counter is zeroed before the first launch
// of a solve (k_init) and re-armed by the last arriver. Returns true in every thread of the
// last workgroup. `flag` is one int of LDS.
__device__ __forceinline__ bool arrive_last(int* counter, int expected, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop the fence's own wait
    const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == expected - 1) ? 1 : 0;
    if (last) {
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *flag = last;
  }
  __syncthreads();
  const bool last = (*flag != 0);
  __syncthreads();  // the flag word is free again
  return last;
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

struct PassPlan {
  int phase;
  int sel;     // table of Xin that holds the pending window, or
  int from_u;  // -1, or (p*V + v): pair-mode pass straight on the u array of point slot (p, v)
  double d;
};

constexpr int VU = 4;  // elements per thread per sweep step (all NT threads of the workgroup sweep)
#define VEC_CHUNKS(base) for (int64_t base = tid; base < m; base += NT * VU)
#define VEC_EACH(k, i, base)            \
  _Pragma("unroll") for (int k = 0; k < VU; ++k) \
    if (const int64_t i = base + static_cast<int64_t>(k) * NT; i < m)

template <int V, int NT>
__device__ __forceinline__ bool decide(const SolveArgs& A, double* red, SolverState* stash,
                                       PassPlan& plan) {
  constexpr int NR = 2 + 2 * V;
  constexpr int PEN = V * NR + 2 * V;  // speculative penalty sums of candidate 0
  constexpr int Q = PEN + 2;
  constexpr int QPAD = pow2_at_least(Q);
  constexpr int NCH = NT / QPAD;  // interleaved summation chains per quantity
  constexpr int NWV = NT / 64;
  static_assert(QPAD <= NT && V <= VS, "window too large");
  double* scratch = red + NT;     // block_reduce scratch, NWV * 2 doubles at most
  const SolverState* st = A.st_cur;
  const int tid = threadIdx.x;
  const int64_t m = A.m;
  const SolverParams P = A.prm;
  const bool writer = (blockIdx.x == 0 && blockIdx.y == 0);

  const int phase = st->phase;
  double d = st->d, F = st->F, alpha = st->alpha, s = st->s;
  int i_ = st->i, j_ = st->j, k_ = st->k, ubp = st->ubp, ubv = st->ubv, sel = st->sel;
  int64_t n_passes = st->n_passes, n_trials = st->n_trials;
  const int64_t n_iters = st->n_iters + 1;
  double nrm[V], sx[V];
#pragma unroll
  for (int l = 0; l < V; ++l) {
    nrm[l] = 1.0;
    sx[l] = 0.0;
  }
  // what this iteration does next
  enum { ACT_PASS, ACT_BUILD, ACT_SLOW, ACT_DONE };
  int action = ACT_SLOW;
  int next_phase = PH_TRIAL;
  bool need_pair = false;  // the pass of this iteration is a pair-mode pass on the accepted x

  if (phase == PH_TRIAL || phase == PH_BUILD) {
    // sums[q] = sum over the tail workgroups w of scal[w][q]: NCH interleaved chains per
    // quantity (w = c, c + NCH, ...), added in chain order
    {
      const int q = tid & (QPAD - 1), c = tid / QPAD;
      double acc = 0.0;
      if (q < Q) {
        constexpr int U = 10;  // loads in flight per chain
        const double* p = A.scal_in + q;
        int w = c;
        for (; w + NCH * (U - 1) < A.nwg_in; w += NCH * U) {
          double x[U];
#pragma unroll
          for (int k = 0; k < U; ++k) x[k] = p[static_cast<int64_t>(w + NCH * k) * Q];
#pragma unroll
          for (int k = 0; k < U; ++k) acc += x[k];
        }
        for (; w < A.nwg_in; w += NCH) acc += p[static_cast<int64_t>(w) * Q];
      }
      red[tid] = acc;
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
        if (jstar < 0) {
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
      if (jstar < 0) {
        // all V candidates rejected: the v = 0 tail already built the next V step sizes from
        // the unchanged (u, g) in table V; alpha was multiplied by beta V times above
        sel = V;
        action = ACT_PASS;
#pragma unroll
        for (int l = 0; l < V; ++l) {
          const double z = sums[V * NR + 2 * l];
          nrm[l] = (z > 0.0) ? sqrt(z) : 1.0;  // Eigen normalize(): only if squaredNorm > 0
          sx[l] = sums[V * NR + 2 * l + 1] / nrm[l];
        }
      } else {
        const double deltau = sqrt(sums[jstar * NR + 1]);
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
#pragma unroll
          for (int l = 0; l < V; ++l) {
            const double z = sums[jstar * NR + 2 + 2 * l];
            nrm[l] = (z > 0.0) ? sqrt(z) : 1.0;
            sx[l] = sums[jstar * NR + 3 + 2 * l] / nrm[l];
          }
        }
      }
    } else {  // PH_BUILD: the tail formed gradF, F and the first window of an outer iteration
      F = sums[0];  // :220
      j_ = 0;
      if (P.maxiniters <= 0) {
        action = ACT_SLOW;  // empty inner loop: u unchanged, its (a, b) in cab are still valid
      } else {
        alpha = 1.0;
        k_ = 0;
        sel = 0;
        action = ACT_PASS;
#pragma unroll
        for (int l = 0; l < V; ++l) {
          const double z = sums[2 + 2 * l];
          nrm[l] = (z > 0.0) ? sqrt(z) : 1.0;
          sx[l] = sums[3 + 2 * l] / nrm[l];
        }
      }
    }
  }

  if (action == ACT_SLOW) {
    // ---- sweeps over whole vectors: workgroup (0,0) alone -----------------------------------
    if (!writer) return false;
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
    for (int64_t i = tid; i < m; i += NT)
      __hip_atomic_store(A.host_u + i, u[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
  }

  // ---- record the decided state (workgroup (0,0), one thread) ------------------------------
  // A pass iteration parks it in LDS and writes it out AFTER the streaming loop (flush_state):
  // a global store ahead of the loop would make the compiler treat the table rows as possibly
  // clobbered and turn their scalar loads into per-lane vector loads.
  if (writer && tid == 0) {
    // (two call sites so that each store keeps its address space: LDS or global, never flat)
    auto record = [&](SolverState* o) {
      o->d = d;
      o->F = F;
      o->alpha = alpha;
      o->s = s;
#pragma unroll
      for (int l = 0; l < V; ++l) {
        o->nrm[l] = nrm[l];
        o->sx[l] = sx[l];
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
    };
    if (action == ACT_PASS) record(stash);
    else record(A.st_next);
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
      if (action != ACT_PASS)
        __hip_atomic_store(&hm->iters, n_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // the decision came out of LDS reads: tell the compiler it is wave-uniform, so that the
  // multipliers of the streaming loop stay scalar loads
  plan.phase = __builtin_amdgcn_readfirstlane(next_phase);
  plan.sel = __builtin_amdgcn_readfirstlane(sel);
  plan.from_u = __builtin_amdgcn_readfirstlane(need_pair ? ubp * V + ubv : -1);
  plan.d = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(d)),
                            __builtin_amdgcn_readfirstlane(__double2loint(d)));
  return action == ACT_PASS;
}
#undef VEC_CHUNKS
#undef VEC_EACH

// What every G launch starts with. Returns false when this workgroup has nothing to stream.
// `stash`: LDS copy of the state a pass iteration decided on, see flush_state.
template <int V, int NT>
__device__ __forceinline__ bool iteration_head(const SolveArgs& A, double* lds,
                                               SolverState* stash, PassPlan& plan) {
  if (A.shared->done) return false;
  const SolverState* st = A.st_cur;
  if (st->stage == ST_RESULTS) return decide<V, NT>(A, lds, stash, plan);
  // the pass was prepared by a transition iteration (or by k_init): run it as it stands
  plan.phase = st->phase;
  plan.sel = st->sel;
  plan.from_u = -1;
  plan.d = st->d;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    *stash = *st;
    stash->stage = ST_RESULTS;
    stash->n_passes = st->n_passes + 1;
    stash->n_iters = st->n_iters + 1;
  }
  return true;
}

// End of a pass iteration: workgroup (0,0) writes the state it decided on where the tail and
// the next iteration read it, marks the iteration as a pass and reports progress to the host.
__device__ __forceinline__ void flush_state(const SolveArgs& A, const SolverState* stash) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    *A.st_next = *stash;
    const int64_t n_iters = stash->n_iters;
    if (A.kind != nullptr && n_iters <= KIND_CAP)
      __hip_atomic_store(A.kind + (n_iters - 1), static_cast<uint8_t>(1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (A.host != nullptr)
      __hip_atomic_store(&A.host->iters, n_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Solve prologue, one launch: pending vector = u0 (candidate 0 of table 0, un-normalised,
// nrm = 1), initial state in ST[0], arrival counters zeroed.
__global__ __launch_bounds__(256) void k_init(SolveArgs A, SolverState init, SolverState* st0,
                                               double* X0) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < A.m) {
    const double row[VS] = {A.u0[i], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    store_row(X0 + i * VS, row);
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < A.nstrips; c += 256) A.cnt[c] = 0;
    if (threadIdx.x == 0) {
      *st0 = init;
      A.shared->done = 0;
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
    double acc = 0.0;
    for (int w = w0; w < w1; ++w) acc += scal[static_cast<int64_t>(w) * Q + q];
    out[static_cast<int64_t>(blockIdx.x) * Q + q] = acc;
  }
}

template <int V, bool FUSED_REDUCE>
__global__ __launch_bounds__(TAIL_THREADS) void k_tail(SolveArgs A) {
  constexpr int NR = 2 + 2 * V;
  constexpr int PEN = V * NR + 2 * V;
  constexpr int Q = PEN + 2;
  constexpr int NRED = NR + 2 * V + 2;  // what a v = 0 workgroup reduces
  constexpr int NSLOT = nslot(V);
  __shared__ double red[TAIL_WAVES * NRED];
  const int v = blockIdx.y;
  const SolverState* st = A.st_next;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * TAIL_THREADS + threadIdx.x;
  const bool valid = i < A.m;

  // raw sums of slot v (and of slot V, the b of candidate 0, for the v = 0 workgroups): these
  // loads do not depend on the solver state
  double p0 = 0.0, p1 = 0.0;
  if (valid) {
    const int64_t o1 = (v == 0) ? static_cast<int64_t>(V) * A.W : 0;  // slot V relative to slot 0
    if (FUSED_REDUCE) {  // single shard: W >= m; partials in tile order, 8 tiles in flight
      const double* p = A.part + static_cast<int64_t>(v) * A.W + i;
      const int64_t ts = static_cast<int64_t>(NSLOT) * A.W;
      int t = 0;
      for (; t + 8 <= A.ntiles; t += 8) {
        double va[8], vb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          va[q] = p[static_cast<int64_t>(t + q) * ts];
          vb[q] = p[static_cast<int64_t>(t + q) * ts + o1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          p0 += va[q];
          p1 += vb[q];
        }
      }
      for (; t < A.ntiles; ++t) {
        p0 += p[static_cast<int64_t>(t) * ts];
        p1 += p[static_cast<int64_t>(t) * ts + o1];
      }
    } else {  // block pb = i / W of the gathered [P][NSLOT][W] layout (32-bit division)
      const uint32_t pb = static_cast<uint32_t>(i) / static_cast<uint32_t>(A.W);
      const int64_t off = i - static_cast<int64_t>(pb) * A.W;
      const double* blk = A.ab + (static_cast<int64_t>(pb) * NSLOT) * A.W;
      p0 = blk[static_cast<int64_t>(v) * A.W + off];
      p1 = blk[o1 + off];
    }
  }
  if (A.shared->done) return;
  if (st->stage != ST_RESULTS) return;  // a pass was only prepared: nothing to evaluate
  const int phase = st->phase;
  const int ubp = st->ubp, ubv = st->ubv;

  if (phase != PH_TRIAL && phase != PH_BUILD) {
    // pair-mode passes carry one vector (candidate 0, nrm = 1): a = M_off x, b = C_off x
    if (v == 0 && valid) {
      A.cab[i] = p0;
      A.cab[A.mp + i] = p1;
    }
    return;
  }
  const double d = st->d, beta = A.prm.beta;
  double r[NRED];
#pragma unroll
  for (int q = 0; q < NRED; ++q) r[q] = 0.0;

  if (phase == PH_BUILD) {
    // start of an outer iteration (clipper.cpp:219-220, :235-236): gradF and F at the current u
    // under the new penalty, and the first window (alpha = 1, beta, ...) — v = 0 workgroups
    if (v != 0) return;
    if (valid) {
      const double ui = pt_arr(A, V, ubp, ubv, 0)[i];
      const double gi = (1 + d) * ui - d * st->s + A.cab[i] + A.cab[A.mp + i] * d;  // :219
      pt_arr(A, V, ubp, ubv, 1)[i] = gi;
      r[0] = ui * gi;  // :220
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
      store_row(A.Xout + i * VS, row);
    }
  } else {
    const double nrmv = st->nrm[v], sxv = st->sx[v];
    const double alpha = st->alpha;
    if (valid) {
      const double xraw = A.Xin[(static_cast<int64_t>(st->sel) * A.mp + i) * VS + v];
      const double ui = pt_arr(A, V, ubp, ubv, 0)[i];
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
      store_row(A.Xout + (static_cast<int64_t>(v) * A.mp + i) * VS, row);
      if (v == 0) {
        // next window if all V candidates are rejected: V more factors of beta (:248)
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
        store_row(A.Xout + (static_cast<int64_t>(V) * A.mp + i) * VS, row2);
      }
    }
  }
  const double tot = block_reduce_pick<NRED, TAIL_WAVES>(r, red);
  double* out = A.scal + static_cast<int64_t>(blockIdx.x) * Q;
  if (threadIdx.x < NR) out[v * NR + threadIdx.x] = tot;
  if (v == 0 && threadIdx.x >= NR && threadIdx.x < NRED)
    out[V * NR + (threadIdx.x - NR)] = tot;  // "all rejected" window sums, then the penalty sums
}

// ------------------------------------------------------------------------------------------
// k_gemv — ONE pass over the symmetric matrix, in one of two modes chosen by the solver state:
//
//   window mode (PH_TRIAL): the line search only ever needs a_v + d*b_v (clipper.cpp:238-241:
//     gradFnew = (1+d)x - d*sum(x) + M_off x + d * C_off x), and d is fixed during a pass. So
//     every element is turned ONCE into w = M + d*C (one fma with the 0/1 pattern indicator,
//     or with the explicit C value) and a candidate costs ONE fma per element:
//         g_v[c] = sum_r w[r][c] * x_v[r]
//     Only candidate 0 keeps a and b apart (two fmas): it is the one accepted when an inner
//     loop converges, and the penalty update that follows needs them apart. 5 + V fp64 ops per
//     element, which keeps a window of 6 under the HBM roofline (separate a/b pairs for all
//     would be 3 + 2V ops: VALU-bound from V = 5 on — measured, tools/mv_tune.hip).
//   pair mode (initialisation, penalty update; matvec API): a = M_off x and b = C_off x of ONE
//     vector separately — the products clipper.cpp:194,202,205,268,271 need them apart. Without
//     an explicit C, b += (M != 0) * x as an fma with the 0/1 indicator, which rounds exactly
//     like the addition it replaces.
//
// grid = (strips of 256 columns, row tiles). A workgroup of NW waves shares one column
// strip; wave w takes rows r0 + w*UNR + k*NW*UNR ... of its tile, UNR rows per iteration
// so UNR independent 16-byte loads per lane are in flight. The multipliers of a row are
// wave-uniform and contiguous (one 64-byte table row): scalar loads. Per-wave partials are
// combined through LDS in wave order and written to part[tile][slot][ld] (slot = candidate v,
// or 0 = a, 1 = b); the tail adds the tiles in tile order. Nothing is atomic: bit-reproducible
// from run to run and rank to rank.
//
// HBM-bound: s*m*W bytes per launch (s = sizeof(T)). MFMA has nothing to offer a product
// whose inner dimension is read exactly once (a 16x16x4 f64 tile would run 6/16 full and the
// operands would need a cross-lane transpose first).
// ------------------------------------------------------------------------------------------

#ifndef CLIPPER_GEMV_SLR
#define CLIPPER_GEMV_SLR 3
#endif
constexpr int GEMV_SLR = CLIPPER_GEMV_SLR;  // accumulator sets combined per LDS round (48 KiB at 3)

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  using type = float4;
};
template <>
struct Vec4<double> {
  using type = double4;
};

template <typename T>
__device__ __forceinline__ typename Vec4<T>::type load4(const T* p) {
  return *reinterpret_cast<const typename Vec4<T>::type*>(p);
}

// 0/1 pattern indicator, or the explicit constraint value, of the 4 elements of a lane
template <typename T, bool HASC>
__device__ __forceinline__ void indicator(const typename Vec4<T>::type& mv,
                                          const typename Vec4<T>::type& cv, double (&ii)[4]) {
  if (HASC) {
    ii[0] = static_cast<double>(cv.x);
    ii[1] = static_cast<double>(cv.y);
    ii[2] = static_cast<double>(cv.z);
    ii[3] = static_cast<double>(cv.w);
  } else {
    ii[0] = (mv.x != T(0)) ? 1.0 : 0.0;
    ii[1] = (mv.y != T(0)) ? 1.0 : 0.0;
    ii[2] = (mv.z != T(0)) ? 1.0 : 0.0;
    ii[3] = (mv.w != T(0)) ? 1.0 : 0.0;
  }
}

// The table rows are read through the CONSTANT address space: a launch never writes the table
// it reads (Xin; the writes go to Xout), and with a wave-uniform address a constant-space load
// is always a scalar load — independent of what the compiler can prove about the global stores
// workgroup (0,0) issues elsewhere in the kernel.
typedef const __attribute__((address_space(4))) double* const_f64_ptr;

// window mode: candidate 0 keeps a and b apart (acc[0] += M x_0, acc[V] += C x_0 — what a
// penalty update will need if it is the accepted one), the others acc[v] += (M + d*C) x_v
template <typename T, bool HASC, int V>
__device__ __forceinline__ void row_window(const typename Vec4<T>::type& mv,
                                           const typename Vec4<T>::type& cv, double d,
                                           const_f64_ptr xr, double (&acc)[V + 1][4]) {
  const double mm[4] = {static_cast<double>(mv.x), static_cast<double>(mv.y),
                        static_cast<double>(mv.z), static_cast<double>(mv.w)};
  double ii[4];
  indicator<T, HASC>(mv, cv, ii);
  const double x0 = xr[0];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[0][e] = fma(mm[e], x0, acc[0][e]);
    acc[V][e] = fma(ii[e], x0, acc[V][e]);
  }
  if (V > 1) {
    double w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = fma(d, ii[e], mm[e]);
#pragma unroll
    for (int v = 1; v < V; ++v) {
      const double xv = xr[v];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[v][e] = fma(w[e], xv, acc[v][e]);
    }
  }
}

// pair mode: acc[0][e] += M[e] * x, acc[1][e] += C[e] * x
template <typename T, bool HASC>
__device__ __forceinline__ void row_pair(const typename Vec4<T>::type& mv,
                                         const typename Vec4<T>::type& cv, double xv,
                                         double (&acc)[2][4]) {
  const double mm[4] = {static_cast<double>(mv.x), static_cast<double>(mv.y),
                        static_cast<double>(mv.z), static_cast<double>(mv.w)};
  double ii[4];
  indicator<T, HASC>(mv, cv, ii);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[0][e] = fma(mm[e], xv, acc[0][e]);
    acc[1][e] = fma(ii[e], xv, acc[1][e]);
  }
}

// The streaming part: this workgroup's (strip, row tile) partial sums -> part[tile][slot][ld].
// X: the pending table, X[row][VS]. NS accumulator sets: V + 1 (window mode) or 2 (pair mode);
// the last set (b) always goes to the last slot.
template <typename T, bool HASC, bool WINDOW, int NS, int NSLOT, int NW, int UNR>
__device__ __forceinline__ void gemv_core(const T* __restrict__ S, const T* __restrict__ Cs,
                                          int64_t ld, int64_t m, int rows_per_tile, double d,
                                          const double* __restrict__ Xg, int xstride,
                                          double* __restrict__ part, double* lds) {
  // X[row * xstride + v]: a table (xstride = VS) or, pair mode only, a plain vector (1)
  const const_f64_ptr X = (const_f64_ptr)Xg;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_tile;
  const int64_t r1 = (r0 + rows_per_tile < m) ? r0 + rows_per_tile : m;

  double acc[NS][4];
#pragma unroll
  for (int v = 0; v < NS; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[v][e] = 0.0;

  if (col < ld) {
    const T* p = S + col;
    const T* pc = HASC ? Cs + col : S + col;
    int64_t r = r0 + static_cast<int64_t>(wave) * UNR;
    for (; r + UNR <= r1; r += static_cast<int64_t>(NW) * UNR) {
      typename Vec4<T>::type mv[UNR];
      typename Vec4<T>::type cv[HASC ? UNR : 1];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        mv[q] = load4(p + (r + q) * ld);
        if (HASC) cv[q] = load4(pc + (r + q) * ld);
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        if constexpr (WINDOW) row_window<T, HASC, NS - 1>(mv[q], cv[HASC ? q : 0], d, X + (r + q) * VS, acc);
        else row_pair<T, HASC>(mv[q], cv[HASC ? q : 0], X[(r + q) * xstride], acc);
      }
    }
    // tail rows of this wave's last chunk
    for (int q = 0; q < UNR; ++q) {
      const int64_t rr = r + q;
      if (rr < r1) {
        const typename Vec4<T>::type mv = load4(p + rr * ld);
        const typename Vec4<T>::type cv = load4(pc + rr * ld);
        if constexpr (WINDOW) row_window<T, HASC, NS - 1>(mv, cv, d, X + rr * VS, acc);
        else row_pair<T, HASC>(mv, cv, X[rr * xstride], acc);
      }
    }
  }

  // cross-wave combine in wave order (fixed summation tree), GEMV_SLR slots per LDS round
  __syncthreads();  // the decision at the head of the launch used the same LDS
#pragma unroll
  for (int v0 = 0; v0 < NS; v0 += GEMV_SLR) {
#pragma unroll
    for (int j = 0; j < GEMV_SLR; ++j) {
      if (v0 + j < NS) {
        double* mine = lds + (wave * GEMV_SLR + j) * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[e] = acc[v0 + j][e];
      }
    }
    __syncthreads();
    constexpr int NOUT = GEMV_SLR * 256;
    for (int t = threadIdx.x; t < NOUT; t += NW * 64) {
      const int j = t >> 8, cl = t & 255;
      if (v0 + j < NS) {
        double sum = lds[j * 256 + cl];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += lds[(w * GEMV_SLR + j) * 256 + cl];
        const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + cl;
        const int v = v0 + j;
        const int slot = (v == NS - 1) ? NSLOT - 1 : v;
        if (c < ld) part[(static_cast<int64_t>(blockIdx.y) * NSLOT + slot) * ld + c] = sum;
      }
    }
    if (v0 + GEMV_SLR < NS) __syncthreads();
  }
}

constexpr int GEMV_LDS_DOUBLES(int NW) { return NW * GEMV_SLR * 256 + 2; }  // + the arrival flag

// window or pair mode by the plan of this iteration
template <typename T, bool HASC, int V, int NW, int UNR>
__device__ __forceinline__ void gemv_by_plan(const T* __restrict__ S, const T* __restrict__ Cs,
                                             int64_t ld, int64_t m, int rows_per_tile,
                                             const double* __restrict__ Xtab,
                                             const double* __restrict__ pt, int64_t mp,
                                             double* __restrict__ part, const PassPlan& plan,
                                             double* lds) {
  if (plan.phase == PH_TRIAL) {
    gemv_core<T, HASC, true, V + 1, nslot(V), NW, UNR>(
        S, Cs, ld, m, rows_per_tile, plan.d, Xtab + static_cast<int64_t>(plan.sel) * mp * VS, VS,
        part, lds);
  } else if (plan.from_u >= 0) {  // the u array of a point slot, see pt_arr
    gemv_core<T, HASC, false, 2, nslot(V), NW, UNR>(
        S, Cs, ld, m, rows_per_tile, 0.0, pt + static_cast<int64_t>(plan.from_u) * 2 * mp, 1,
        part, lds);
  } else {
    gemv_core<T, HASC, false, 2, nslot(V), NW, UNR>(
        S, Cs, ld, m, rows_per_tile, 0.0, Xtab + static_cast<int64_t>(plan.sel) * mp * VS, VS,
        part, lds);
  }
}

// two workgroups per CU (NW/2 waves per SIMD each): caps the registers at 128 per lane
template <typename T, bool HASC, int V, int NW, int UNR>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_gemv(const T* __restrict__ S,
                                                           const T* __restrict__ Cs,
                                                           int rows_per_tile, SolveArgs A) {
  static_assert(NW * 64 >= TAIL_THREADS && NW * 256 >= NW * 64 + (NW * 2 * V),
                "LDS of the mat-vec must hold the decision's scratch");
  __shared__ double lds[GEMV_LDS_DOUBLES(NW)];
  __shared__ SolverState stash;
  PassPlan plan;
  if (!iteration_head<V, NW * 64>(A, lds, &stash, plan)) return;
  gemv_by_plan<T, HASC, V, NW, UNR>(S, Cs, A.W, A.m, rows_per_tile, A.Xin, A.pt, A.mp, A.part,
                                    plan, lds);
  flush_state(A, &stash);
}

// k_pass — the same for a column-sharded M, with the reduction of the row-tile partials folded
// into the epilogue: the LAST row-tile workgroup of a column strip (arrival counter per strip)
// adds the strip's partials in tile order into this shard's block of the gathered layout
// ab[P][NSLOT][W] — what k_reduce would do in a launch of its own. The exchange and
// k_tail<V, false> follow. Every rank takes the same decision from the same bits.
template <typename T, bool HASC, int V, int NW, int UNR>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_pass(const T* __restrict__ S,
                                                           const T* __restrict__ Cs,
                                                           int rows_per_tile, SolveArgs A) {
  constexpr int NSLOT = nslot(V);
  __shared__ double lds[GEMV_LDS_DOUBLES(NW)];
  __shared__ SolverState stash;
  PassPlan plan;
  if (!iteration_head<V, NW * 64>(A, lds, &stash, plan)) return;
  const int64_t ld = A.W;
  gemv_by_plan<T, HASC, V, NW, UNR>(S, Cs, ld, A.m, rows_per_tile, A.Xin, A.pt, A.mp, A.part,
                                    plan, lds);
  flush_state(A, &stash);
  int* flag = reinterpret_cast<int*>(lds + GEMV_LDS_DOUBLES(NW) - 1);
  if (!arrive_last(A.cnt + blockIdx.x, gridDim.y, flag)) return;
  // ---- last workgroup of this column strip ------------------------------------------------
  double* ab_block = A.ab + static_cast<int64_t>(A.slot) * NSLOT * ld;
  const int64_t ts = static_cast<int64_t>(NSLOT) * ld;
  for (int t = threadIdx.x; t < NSLOT * 256; t += NW * 64) {
    const int sl = t >> 8;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + (t & 255);
    if (c < ld) {
      const double* p = A.part + sl * ld + c;
      double acc = 0.0;
      int tt = 0;
      for (; tt + 16 <= A.ntiles; tt += 16) {  // 16 tiles in flight: this workgroup is alone now
        double x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = p[static_cast<int64_t>(tt + q) * ts];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += x[q];
      }
      for (; tt < A.ntiles; ++tt) acc += p[static_cast<int64_t>(tt) * ts];
      ab_block[sl * ld + c] = acc;
    }
  }
}

// the pair-mode pass alone, on table 0 (matvec API, micro-benchmark): no solver state
template <typename T, bool HASC, int NW, int UNR>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_gemv_plain(const T* __restrict__ S,
                                                                 const T* __restrict__ Cs,
                                                                 int64_t ld, int64_t m,
                                                                 int rows_per_tile,
                                                                 const double* __restrict__ X,
                                                                 double* __restrict__ part) {
  __shared__ double lds[GEMV_LDS_DOUBLES(NW)];
  gemv_core<T, HASC, false, 2, 2, NW, UNR>(S, Cs, ld, m, rows_per_tile, 0.0, X, VS, part, lds);
}

// ------------------------------------------------------------------------------------------
// column-compressed copy of M for the solver's passes (CLIPPER_HIP_STORE_F32_CSC)
// ------------------------------------------------------------------------------------------
// M at the headline configuration is ~11 % dense; the dense pass spends its time multiplying
// zeros (it is VALU-bound before it is HBM-bound once a window of candidates shares one pass).
// The compressed copy stores, per GROUP = (128-column strip s, block b of 64 rows), every
// column's nonzeros as (row-in-block u8, value fp32). All 128 columns of a group are padded to
// the group's longest list, rounded up to 4 (padding: value 0, row 0 — adds exact zeros), and laid
// out [column-of-lane e = 0..1][quad kq][lane][4 entries]: lane l owns columns 2l, 2l+1 of the
// strip, and a wave reads 1 KiB of values + 256 B of rows per instruction. A lane multiplies
// only ITS columns' nonzeros; the x rows a block needs (64 table rows) are staged by the wave
// in LDS and gathered from there by row index.
//   Lc[g]   padded list length of group g = s * nblocks + b (multiple of 4)
//   Pre[g]  where the group's data starts, in units of 128 entries (vals: floats, rows: bytes)
//   tb      row-tile boundaries per strip [nstrips][ntmax + 1] in blocks: tiles of EQUAL COST
//           (sum of Lc), so that the dense inlier block at the end of the matrix does not land
//           in one workgroup; strips with fewer tiles have empty ones (they write zeros)
// The values are the fp32 M the dense store holds, the products are the same fp64 products, the
// zeros the dense pass adds are exact — only the summation order over the rows differs.
// A group is as wide as a tile of k_affinity_sym, which therefore emits the groups of the tiles
// it computes (and of their mirror images) straight from its LDS image; k_csc_build does the
// same from a dense store (the other fill kernels, setMatrixData).
constexpr int CSC_RB = 64;   // rows per block
constexpr int CSC_CW = 128;  // columns per strip
constexpr int CSC_MAXQ = 6;  // quads of one column phase in flight per lane

struct CscView {
  const float* vals;
  const uint8_t* rows;
  const uint32_t* Lc;
  const uint64_t* Pre;
  const int* tb;
  int nblocks;
  int ntmax;
};

// The space of a group is claimed with one atomic on the cursor of one of CSC_ARENAS arenas
// (same-address atomics serialise at ~25-50 ns each — thousands of groups on ONE cursor cost more
// than the build itself): the ORDER of the groups in memory varies from build to build; the
// content of a group, and with it every sum, does not. A build that does not fit an arena
// (always: the first one of a problem size, capacity 0) writes nothing but Lc and the totals;
// the host grows the buffers and builds again.
constexpr int CSC_ARENAS = 64;
struct alignas(128) CscArena {
  unsigned long long cursor;    // units of 128 entries claimed so far in this arena
  unsigned long long capacity;  // units available to it
  unsigned long long origin;    // where the arena starts, same units
  int overflow;
};
typedef CscArena CscBuildCtl;  // [CSC_ARENAS]

struct CscOut {
  uint32_t* Lc;
  uint64_t* Pre;
  float* vals;
  uint8_t* rows;
  CscBuildCtl* ctl;
  int nblocks;
};

// Emission of NG groups by one workgroup of NG*128 threads: thread t owns column (t & 127) of
// group (t >> 7); g = its group id or -1 (nothing to emit: the whole group, uniformly).
// csc_claim: the group's padded length from the columns' counts, its space claimed with one
// atomic. `red` [2*NG] ints and `base_s` [NG] are LDS scratch. Contains barriers. Returns false
// for a thread that has nothing to write.
template <int NG>
__device__ __forceinline__ bool csc_claim(int cnt, int64_t g, const CscOut& O, int* red,
                                          unsigned long long* base_s, int& LQ,
                                          unsigned long long& base) {
  const int t = threadIdx.x, gi = t >> 7, cl = t & 127;
  int mx = cnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(mx, o);
    mx = mx > other ? mx : other;
  }
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  const int w = red[2 * gi] > red[2 * gi + 1] ? red[2 * gi] : red[2 * gi + 1];
  if (cl == 0 && g >= 0) {
    const unsigned L = static_cast<unsigned>((w + 3) & ~3);
    O.Lc[g] = L;
    CscArena* ar = O.ctl + static_cast<int>((g * 11 + (g >> 6)) & (CSC_ARENAS - 1));
    unsigned long long bb = atomicAdd(&ar->cursor, static_cast<unsigned long long>(L));
    if (bb + L > ar->capacity) {
      ar->overflow = 1;
      bb = ~0ull;
    } else {
      bb += ar->origin;
    }
    O.Pre[g] = bb;
    base_s[gi] = bb;
  }
  __syncthreads();
  if (g < 0) return false;
  base = base_s[gi];
  LQ = ((w + 3) & ~3) >> 2;
  return base != ~0ull;
}

// the list of one column, written quad by quad
struct CscColumnWriter {
  float4* vq;
  uint32_t* rq;
  float v4[4];
  uint32_t r4;
  int k;
  __device__ __forceinline__ void open(const CscOut& O, unsigned long long base, int LQ) {
    const int cl = threadIdx.x & 127;
    const int lane = cl >> 1, e = cl & 1;
    vq = reinterpret_cast<float4*>(O.vals + base * 128) + static_cast<int64_t>(e) * LQ * 64 + lane;
    rq = reinterpret_cast<uint32_t*>(O.rows + base * 128) + static_cast<int64_t>(e) * LQ * 64 + lane;
    v4[0] = v4[1] = v4[2] = v4[3] = 0.f;
    r4 = 0;
    k = 0;
  }
  __device__ __forceinline__ void flush(int kq) {
    vq[kq * 64] = make_float4(v4[0], v4[1], v4[2], v4[3]);
    rq[kq * 64] = r4;
    v4[0] = v4[1] = v4[2] = v4[3] = 0.f;
    r4 = 0;
  }
  __device__ __forceinline__ void push(float v, uint32_t row) {
    const int j = k & 3;
    v4[0] = j == 0 ? v : v4[0];
    v4[1] = j == 1 ? v : v4[1];
    v4[2] = j == 2 ? v : v4[2];
    v4[3] = j == 3 ? v : v4[3];
    r4 |= row << (8 * j);
    ++k;
    if (j == 3) flush((k >> 2) - 1);
  }
  __device__ __forceinline__ void close(int LQ) {  // the open quad and the padding quads
    for (int kq = k >> 2; kq < LQ; ++kq) flush(kq);
  }
};

// from 64 values held in registers (k_csc_build)
template <int NG>
__device__ __forceinline__ void csc_emit(const float (&v)[CSC_RB], int64_t g, const CscOut& O,
                                         int* red, unsigned long long* base_s) {
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < CSC_RB; ++q) cnt += (v[q] != 0.f) ? 1 : 0;
  int LQ;
  unsigned long long base;
  if (!csc_claim<NG>(cnt, g, O, red, base_s, LQ, base)) return;
  CscColumnWriter w;
  w.open(O, base, LQ);
#pragma unroll
  for (int q = 0; q < CSC_RB; ++q)
    if (v[q] != 0.f) w.push(v[q], static_cast<uint32_t>(q));
  w.close(LQ);
}

// from a column of an LDS image (k_affinity_sym): col[q * stride], q = 0..63. A first sweep
// builds the bit mask of the nonzeros; only those are visited again (an ~11 % dense column: ~7
// of 64), the trip count of a wave being its longest column.
template <int NG>
__device__ __forceinline__ void csc_emit_lds(const float* col, int stride, int64_t g,
                                             const CscOut& O, int* red,
                                             unsigned long long* base_s) {
  uint32_t mlo = 0, mhi = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) mlo |= (col[q * stride] != 0.f ? 1u : 0u) << q;
#pragma unroll
  for (int q = 0; q < 32; ++q) mhi |= (col[(q + 32) * stride] != 0.f ? 1u : 0u) << q;
  const int cnt = __popc(mlo) + __popc(mhi);
  int LQ;
  unsigned long long base;
  if (!csc_claim<NG>(cnt, g, O, red, base_s, LQ, base)) return;
  CscColumnWriter w;
  w.open(O, base, LQ);
  while (mlo) {
    const int q = __ffs(mlo) - 1;
    mlo &= mlo - 1;
    w.push(col[q * stride], static_cast<uint32_t>(q));
  }
  while (mhi) {
    const int q = __ffs(mhi) - 1 + 32;
    mhi &= mhi - 1;
    w.push(col[q * stride], static_cast<uint32_t>(q));
  }
  w.close(LQ);
}

// k_csc_build — from a dense fp32 store: two groups (row blocks 2y, 2y+1 of strip x) per
// workgroup, one column per thread, the 64 rows of the block loaded at once.
__global__ __launch_bounds__(256) void k_csc_build(const float* __restrict__ S, int64_t ld,
                                                    int64_t m, CscOut O) {
  __shared__ int red[4];
  __shared__ unsigned long long base_s[2];
  const int s = blockIdx.x, t = threadIdx.x;
  const int b = 2 * blockIdx.y + (t >> 7);
  const int64_t c = static_cast<int64_t>(s) * CSC_CW + (t & 127);
  const int64_t r0 = static_cast<int64_t>(b) * CSC_RB;
  float v[CSC_RB];
#pragma unroll
  for (int q = 0; q < CSC_RB; ++q) {
    const int64_t r = r0 + q;
    v[q] = (c < ld && r < m) ? S[r * ld + c] : 0.f;
  }
  const int64_t g = (b < O.nblocks) ? static_cast<int64_t>(s) * O.nblocks + b : -1;
  csc_emit<2>(v, g, O, red, base_s);
}

// k_csc_expand — the dense fp32 store back from the compressed copy (getters, the exact DSD
// rounding's gather and the matvec API read a dense store; it is materialised on demand): one
// group per workgroup half, one column per thread — zeros first, then the column's entries.
__global__ __launch_bounds__(256) void k_csc_expand(CscView M, float* __restrict__ S, int64_t ld,
                                                     int64_t m) {
  const int s = blockIdx.x, t = threadIdx.x;
  const int b = 2 * blockIdx.y + (t >> 7);
  const int cl = t & 127;
  const int64_t c = static_cast<int64_t>(s) * CSC_CW + cl;
  if (b >= M.nblocks || c >= ld) return;
  const int64_t r0 = static_cast<int64_t>(b) * CSC_RB;
  for (int q = 0; q < CSC_RB; ++q)
    if (r0 + q < m) S[(r0 + q) * ld + c] = 0.f;
  const int64_t g = static_cast<int64_t>(s) * M.nblocks + b;
  const int LQ = static_cast<int>(M.Lc[g] >> 2);
  const int64_t base = static_cast<int64_t>(M.Pre[g]) * 128;
  const int lane = cl >> 1, e = cl & 1;
  const float4* vq = reinterpret_cast<const float4*>(M.vals + base) + static_cast<int64_t>(e) * LQ * 64 + lane;
  const uint32_t* rq = reinterpret_cast<const uint32_t*>(M.rows + base) + static_cast<int64_t>(e) * LQ * 64 + lane;
  for (int kq = 0; kq < LQ; ++kq) {
    const float4 v = vq[kq * 64];
    const uint32_t r = rq[kq * 64];
    const float vf[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (vf[j] != 0.f) S[(r0 + ((r >> (8 * j)) & 255u)) * ld + c] = vf[j];
  }
}

constexpr int csc_xpitch(int V) { return V <= 1 ? 2 : (V <= 6 ? 6 : 10); }  // doubles per staged row
constexpr int csc_lds_doubles(int V, int NW) {
  const int a = NW * CSC_RB * csc_xpitch(V), b = NW * (V + 1) * 64;
  return (a > b ? a : b) + 2;
}

// The streaming part on the compressed copy: this workgroup's (strip, tile) partial sums ->
// part[tile][slot][ld], the slots of gemv_core. Wave (e, h) of the workgroup: column e of every
// lane's two, blocks b0 + h, b0 + h + NW/2, ... of the tile — the column phases of a block cost
// the same by construction. WINDOW / pair mode as in gemv_core.
template <bool WINDOW, int V, int NSLOT, int NW>
__device__ __forceinline__ void csc_core(const CscView& M, int64_t ld, int64_t m, double d,
                                         const double* __restrict__ X, int xstride,
                                         double* __restrict__ part, double* lds) {
  constexpr int NS = WINDOW ? V + 1 : 2;
  constexpr int XP = WINDOW ? csc_xpitch(V) : 1;
  constexpr int NH = NW / 2;
  constexpr int XT = CSC_RB * XP;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = wave & 1, h = wave >> 1;
  const int s = blockIdx.x;
  const int b0 = M.tb[s * (M.ntmax + 1) + blockIdx.y];
  const int b1 = M.tb[s * (M.ntmax + 1) + blockIdx.y + 1];
  double* xs = lds + wave * XT;

  double acc[NS];
#pragma unroll
  for (int v = 0; v < NS; ++v) acc[v] = 0.0;

  __syncthreads();  // the decision at the head of the launch used the same LDS
  for (int b = b0 + h; b < b1; b += NH) {
    const int64_t g = static_cast<int64_t>(s) * M.nblocks + b;
    const int LQ = __builtin_amdgcn_readfirstlane(static_cast<int>(M.Lc[g] >> 2));
    const int64_t base = static_cast<int64_t>(M.Pre[g]) * 128;
    const float4* vq =
        reinterpret_cast<const float4*>(M.vals + base) + static_cast<int64_t>(e) * LQ * 64 + lane;
    const uint32_t* rq =
        reinterpret_cast<const uint32_t*>(M.rows + base) + static_cast<int64_t>(e) * LQ * 64 + lane;
    float4 mv[CSC_MAXQ];
    uint32_t rw[CSC_MAXQ];
#pragma unroll
    for (int q = 0; q < CSC_MAXQ; ++q) {
      if (q < LQ) {
        mv[q] = vq[q * 64];
        rw[q] = rq[q * 64];
      }
    }
    // stage the block's x rows (the wave's own tile: LDS operations of one wave stay in order)
    __builtin_amdgcn_wave_barrier();
    {
      const int64_t r = static_cast<int64_t>(b) * CSC_RB + lane;
      if constexpr (WINDOW) {
        double xr[VS];
#pragma unroll
        for (int v = 0; v < VS; ++v) xr[v] = 0.0;
        if (r < m) {
          const double2* xp = reinterpret_cast<const double2*>(X + r * VS);
#pragma unroll
          for (int v = 0; v < ((V + 1) & ~1); v += 2) {
            const double2 t2 = xp[v >> 1];
            xr[v] = t2.x;
            xr[v + 1] = t2.y;
          }
        }
#pragma unroll
        for (int v = 0; v < ((V + 1) & ~1); v += 2)
          *reinterpret_cast<double2*>(xs + lane * XP + v) = make_double2(xr[v], xr[v + 1]);
      } else {
        xs[lane] = (r < m) ? X[r * xstride] : 0.0;
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int k0 = 0; k0 < LQ; k0 += CSC_MAXQ) {
      if (k0 > 0) {
#pragma unroll
        for (int q = 0; q < CSC_MAXQ; ++q) {
          if (k0 + q < LQ) {
            mv[q] = vq[(k0 + q) * 64];
            rw[q] = rq[(k0 + q) * 64];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < CSC_MAXQ; ++q) {
        if (k0 + q < LQ) {
          const float mf[4] = {mv[q].x, mv[q].y, mv[q].z, mv[q].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double mm = static_cast<double>(mf[j]);
            const double ii = mf[j] != 0.f ? 1.0 : 0.0;
            const uint32_t row = (rw[q] >> (8 * j)) & 255u;
            if constexpr (WINDOW) {
              const double* xr = xs + row * XP;
              double xv[(V + 1) & ~1];
#pragma unroll
              for (int v = 0; v < ((V + 1) & ~1); v += 2) {
                const double2 t2 = *reinterpret_cast<const double2*>(xr + v);
                xv[v] = t2.x;
                xv[v + 1] = t2.y;
              }
              acc[0] = fma(mm, xv[0], acc[0]);
              acc[V] = fma(ii, xv[0], acc[V]);
              if (V > 1) {
                const double w = fma(d, ii, mm);
#pragma unroll
                for (int v = 1; v < V; ++v) acc[v] = fma(w, xv[v], acc[v]);
              }
            } else {
              const double xv = xs[row];
              acc[0] = fma(mm, xv, acc[0]);
              acc[1] = fma(ii, xv, acc[1]);
            }
          }
        }
      }
    }
  }

  // cross-wave combine: the NH waves of a column phase, in wave order
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NS; ++v) lds[(wave * NS + v) * 64 + lane] = acc[v];
  __syncthreads();
  for (int t = threadIdx.x; t < NS * CSC_CW; t += NW * 64) {
    const int v = t >> 7, cl = t & 127;
    const int ee = cl & 1, ln = cl >> 1;
    double sum = lds[(ee * NS + v) * 64 + ln];
#pragma unroll
    for (int hh = 1; hh < NH; ++hh) sum += lds[((hh * 2 + ee) * NS + v) * 64 + ln];
    const int64_t c = static_cast<int64_t>(blockIdx.x) * CSC_CW + cl;
    const int slot = (v == NS - 1) ? NSLOT - 1 : v;
    if (c < ld) part[(static_cast<int64_t>(blockIdx.y) * NSLOT + slot) * ld + c] = sum;
  }
}

// G of a solver iteration on the compressed copy (one shard): decision, then the pass
template <int V, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_gemv_csc(CscView M, SolveArgs A) {
  static_assert(csc_lds_doubles(V, NW) >= NW * 64 + NW * 2 * V,
                "LDS of the mat-vec must hold the decision's scratch");
  __shared__ double lds[csc_lds_doubles(V, NW)];
  __shared__ SolverState stash;
  PassPlan plan;
  if (!iteration_head<V, NW * 64>(A, lds, &stash, plan)) return;
  if (plan.phase == PH_TRIAL) {
    csc_core<true, V, nslot(V), NW>(M, A.W, A.m, plan.d,
                                    A.Xin + static_cast<int64_t>(plan.sel) * A.mp * VS, VS, A.part,
                                    lds);
  } else if (plan.from_u >= 0) {
    csc_core<false, V, nslot(V), NW>(M, A.W, A.m, 0.0,
                                     A.pt + static_cast<int64_t>(plan.from_u) * 2 * A.mp, 1,
                                     A.part, lds);
  } else {
    csc_core<false, V, nslot(V), NW>(M, A.W, A.m, 0.0,
                                     A.Xin + static_cast<int64_t>(plan.sel) * A.mp * VS, VS,
                                     A.part, lds);
  }
  flush_state(A, &stash);
}

// k_reduce — adds the row-tile partials in tile order and writes this shard's block of the
// gathered layout (matvec API only: the solver folds this into k_pass / k_tail). One thread per
// output element e = slot*ld + c; the loads of 8 tiles are issued before they are summed (the
// partials sit in L2 / MALL).
__global__ __launch_bounds__(256) void k_reduce(const double* __restrict__ part, int ntiles,
                                                 int nslots, int64_t ld,
                                                 double* __restrict__ ab_block) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t tstride = static_cast<int64_t>(nslots) * ld;
  if (e >= tstride) return;
  const double* p = part + e;
  double acc = 0.0;
  int t = 0;
  for (; t + 8 <= ntiles; t += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[static_cast<int64_t>(t + q) * tstride];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += v[q];
  }
  for (; t < ntiles; ++t) acc += p[static_cast<int64_t>(t) * tstride];
  ab_block[e] = acc;
}

// x[i] -> candidate 0 of a table row (matvec API)
__global__ __launch_bounds__(256) void k_spread(const double* __restrict__ x, int64_t m,
                                                 double* __restrict__ X) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < m) {
    const double row[VS] = {x[i], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    store_row(X + i * VS, row);
  }
}

// ------------------------------------------------------------------------------------------
// affinity fill
// ------------------------------------------------------------------------------------------

// P[k * pstride + i] = D[k + d * idx[i]] : per-association point table, structure of arrays,
// so that column data loads in the fill kernels are contiguous across lanes. Pf is the same
// table rounded to fp32 (input of the conservative prefilter of the compacting fill kernels).
__global__ __launch_bounds__(256) void k_gather_points(const double* __restrict__ D, int d,
                                                        const int32_t* __restrict__ idx,
                                                        int64_t m, int64_t pstride,
                                                        double* __restrict__ P,
                                                        float* __restrict__ Pf) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= pstride) return;
  const int64_t src = (i < m) ? idx[i] : 0;
  for (int k = 0; k < d; ++k) {
    const double v = (i < m) ? D[k + d * src] : 0.0;
    P[k * pstride + i] = v;
    Pf[k * pstride + i] = static_cast<float>(v);
  }
}

template <typename T>
__device__ __forceinline__ T store_score(double scr, double affinityeps) {
  // clipper.cpp:53-55 — keep the score only when it exceeds affinityeps.
  if (!(scr > affinityeps)) return T(0);
  T v = static_cast<T>(scr);
  // an fp32 underflow must not erase an entry from the pattern (C == pattern(M))
  if (v == T(0)) v = static_cast<T>(1.17549435e-38);
  return v;
}

template <typename T>
__device__ __forceinline__ void store4(T* p, T a, T b, T c, T d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<double>(double* p, double a, double b, double c,
                                               double d) {
  *reinterpret_cast<double4*>(p) = make_double4(a, b, c, d);
}

struct EuclidParams {
  double sigma, epsilon, mindist, affinityeps;
};

// One thread = 4 adjacent columns of S, looping down `rows_per_blk` rows; the 4 columns'
// points and association indices stay in registers for the whole loop, the row's point is
// wave-uniform (scalar loads). Each lane stores 4 consecutive elements, a wave 256: whole
// 1 KiB (fp32) row segments per store instruction.
// D > 0: compile-time dimension (2 or 3); D == 0: run-time dimension `d` (slow path).
template <typename T, int D>
__global__ __launch_bounds__(256) void k_affinity_euclid(
    T* __restrict__ S, int64_t ld, int64_t m, int64_t c0, int rows_per_blk, int d,
    const double* __restrict__ P1, const double* __restrict__ P2, int64_t pstride,
    const int32_t* __restrict__ A0, const int32_t* __restrict__ A1, EuclidParams prm) {
  const int64_t c = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (c >= ld) return;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_blk;
  const int64_t r1 = (r0 + rows_per_blk < m) ? r0 + rows_per_blk : m;
  constexpr int DD = (D > 0) ? D : 1;

  int64_t gi[4];
  bool valid[4];
  int32_t a0c[4], a1c[4];
  double p1c[4][DD], p2c[4][DD];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t g = c0 + c + q;
    valid[q] = g < m;
    gi[q] = valid[q] ? g : (m - 1);
    a0c[q] = A0[gi[q]];
    a1c[q] = A1[gi[q]];
    if (D > 0) {
#pragma unroll
      for (int k = 0; k < DD; ++k) {
        p1c[q][k] = P1[k * pstride + gi[q]];
        p2c[q][k] = P2[k * pstride + gi[q]];
      }
    }
  }

  for (int64_t r = r0; r < r1; ++r) {
    const int32_t a0r = A0[r], a1r = A1[r];
    double p1r[DD], p2r[DD];
    if (D > 0) {
#pragma unroll
      for (int k = 0; k < DD; ++k) {
        p1r[k] = P1[k * pstride + r];
        p2r[k] = P2[k * pstride + r];
      }
    }
    T out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double s1 = 0.0, s2 = 0.0;  // euclidean_distance.cpp:18-19, sequential fma chain
      if (D > 0) {
#pragma unroll
        for (int k = 0; k < DD; ++k) {
          const double t1 = p1r[k] - p1c[q][k];
          const double t2 = p2r[k] - p2c[q][k];
          s1 = fma(t1, t1, s1);
          s2 = fma(t2, t2, s2);
        }
      } else {
        for (int k = 0; k < d; ++k) {
          const double t1 = P1[k * pstride + r] - P1[k * pstride + gi[q]];
          const double t2 = P2[k * pstride + r] - P2[k * pstride + gi[q]];
          s1 = fma(t1, t1, s1);
          s2 = fma(t2, t2, s2);
        }
      }
      const double l1 = sqrt(s1), l2 = sqrt(s2);
      // clipper.cpp:35-38 distinctness; the diagonal (r == column) fails it by construction
      bool ok = valid[q] && (a0r != a0c[q]) && (a1r != a1c[q]);
      // euclidean_distance.cpp:23-25
      if (prm.mindist > 0 && (l1 < prm.mindist || l2 < prm.mindist)) ok = false;
      const double cc = fabs(l1 - l2);  // :28
      double scr = 0.0;
      if (ok && cc < prm.epsilon) scr = exp(-0.5 * cc * cc / (prm.sigma * prm.sigma));  // :30
      out[q] = store_score<T>(scr, prm.affinityeps);
    }
    store4<T>(S + r * ld + c, out[0], out[1], out[2], out[3]);
  }
}

struct PointNormalParams {
  double sigp, epsp, sign, epsn, affinityeps;
};

// PointNormalDistance: datum = [x y z nx ny nz] (pointnormal_distance.cpp:13-35).
template <typename T>
__global__ __launch_bounds__(256) void k_affinity_pointnormal(
    T* __restrict__ S, int64_t ld, int64_t m, int64_t c0, int rows_per_blk,
    const double* __restrict__ P1, const double* __restrict__ P2, int64_t pstride,
    const int32_t* __restrict__ A0, const int32_t* __restrict__ A1, PointNormalParams prm) {
  const int64_t c = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (c >= ld) return;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_blk;
  const int64_t r1 = (r0 + rows_per_blk < m) ? r0 + rows_per_blk : m;

  bool valid[4];
  int32_t a0c[4], a1c[4];
  double p1c[4][6], p2c[4][6];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t g = c0 + c + q;
    valid[q] = g < m;
    const int64_t gi = valid[q] ? g : (m - 1);
    a0c[q] = A0[gi];
    a1c[q] = A1[gi];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      p1c[q][k] = P1[k * pstride + gi];
      p2c[q][k] = P2[k * pstride + gi];
    }
  }

  for (int64_t r = r0; r < r1; ++r) {
    const int32_t a0r = A0[r], a1r = A1[r];
    double p1r[6], p2r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      p1r[k] = P1[k * pstride + r];
      p2r[k] = P2[k * pstride + r];
    }
    T out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double t1 = p1r[k] - p1c[q][k];
        const double t2 = p2r[k] - p2c[q][k];
        s1 = fma(t1, t1, s1);
        s2 = fma(t2, t2, s2);
      }
      const double l1 = sqrt(s1), l2 = sqrt(s2);  // :17-18
      const double dot1 = fma(p1r[5], p1c[q][5], fma(p1r[4], p1c[q][4], p1r[3] * p1c[q][3]));
      const double dot2 = fma(p2r[5], p2c[q][5], fma(p2r[4], p2c[q][4], p2r[3] * p2c[q][3]));
      const bool ok = valid[q] && (a0r != a0c[q]) && (a1r != a1c[q]);
      double scr = 0.0;
      if (ok) {
        const double alpha1 = acos(dot1);  // :21 (NaN when |dot| > 1, as in the reference)
        const double alpha2 = acos(dot2);  // :22
        const double dp = fabs(l1 - l2);          // :25
        const double dn = fabs(alpha1 - alpha2);  // :26
        if (dp < prm.epsp && dn < prm.epsn) {     // :28
          const double sp = exp(-0.5 * dp * dp / (prm.sigp * prm.sigp));  // :29
          const double sn = exp(-0.5 * dn * dn / (prm.sign * prm.sign));  // :30
          scr = sp * sn;                                                  // :31
        }
      }
      out[q] = store_score<T>(scr, prm.affinityeps);
    }
    store4<T>(S + r * ld + c, out[0], out[1], out[2], out[3]);
  }
}

// ------------------------------------------------------------------------------------------
// Compacting fill kernels.
//
// The exact score costs ~150 fp64-rate instructions per pair (two correctly rounded sqrt, one
// division, exp / acos), yet on registration data only ~10 % of the pairs pass `c < epsilon`
// — and with 64-lane waves a plain branch saves nothing. So every pair first goes through a
// CONSERVATIVE fp32 prefilter (|l1f - l2f| >= epsilon + guard  =>  certainly c >= epsilon; the
// guard bounds the fp32 error from the data's magnitude, incl. the 1-ulp raw v_sqrt_f32), survivors are compacted into a
// per-wave LDS queue with ballot/mbcnt (no atomics, no workgroup barrier), and only they are
// evaluated exactly in fp64 — with the same instruction sequence as the plain kernels, so the
// results are bit-identical to them. Scores are scattered into an LDS staging tile and leave
// as whole 1 KiB row segments, so the HBM store pattern is unchanged.
// Geometry: 4 waves per workgroup, wave w owns 256 columns (4 per lane); rows are processed in
// groups of AFF_RG = 8: queue 8 KiB + staging 8 (fp32) / 16 (fp64) KiB per wave.
// ------------------------------------------------------------------------------------------

constexpr int AFF_RG = 8;

__device__ __forceinline__ uint32_t lane_prefix(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

template <typename T, int D>
__device__ __forceinline__ double exact_euclid_score(const double* __restrict__ P1,
                                                     const double* __restrict__ P2,
                                                     int64_t pstride, int64_t r, int64_t g,
                                                     const EuclidParams& prm) {
  double s1 = 0.0, s2 = 0.0;  // euclidean_distance.cpp:18-19, sequential fma chain
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double t1 = P1[k * pstride + r] - P1[k * pstride + g];
    const double t2 = P2[k * pstride + r] - P2[k * pstride + g];
    s1 = fma(t1, t1, s1);
    s2 = fma(t2, t2, s2);
  }
  const double l1 = sqrt(s1), l2 = sqrt(s2);
  if (prm.mindist > 0 && (l1 < prm.mindist || l2 < prm.mindist)) return 0.0;  // :23-25
  const double cc = fabs(l1 - l2);                                            // :28
  return (cc < prm.epsilon) ? exp(-0.5 * cc * cc / (prm.sigma * prm.sigma)) : 0.0;  // :30
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_affinity_euclid_compact(
    T* __restrict__ S, int64_t ld, int64_t m, int64_t c0, int rows_per_blk,
    const double* __restrict__ P1, const double* __restrict__ P2,
    const float* __restrict__ P1f, const float* __restrict__ P2f, int64_t pstride,
    const int32_t* __restrict__ A0, const int32_t* __restrict__ A1, EuclidParams prm,
    float eps_guarded) {
  __shared__ uint32_t queue[4][AFF_RG * 256];
  __shared__ T stage[4][AFF_RG][256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t cw = static_cast<int64_t>(blockIdx.x) * 1024 + wave * 256;  // wave's first column
  const int64_t c = cw + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_blk;
  const int64_t r1 = (r0 + rows_per_blk < m) ? r0 + rows_per_blk : m;
  if (cw >= ld) return;  // whole wave outside the slice (wave-uniform)

  // column data (fp32) in registers for the prefilter
  bool valid[4];
  int32_t a0c[4], a1c[4];
  float p1c[4][D], p2c[4][D];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t g = c0 + c + q;
    valid[q] = (g < m) && (c + q < ld);
    const int64_t gi = (g < m) ? g : (m - 1);
    a0c[q] = A0[gi];
    a1c[q] = A1[gi];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      p1c[q][k] = P1f[k * pstride + gi];
      p2c[q][k] = P2f[k * pstride + gi];
    }
  }
#pragma unroll
  for (int r8 = 0; r8 < AFF_RG; ++r8)
#pragma unroll
    for (int q = 0; q < 4; ++q) stage[wave][r8][lane * 4 + q] = T(0);

  for (int64_t base = r0; base < r1; base += AFF_RG) {
    // ---- phase A: fp32 prefilter + compaction of the surviving (row, column) pairs --------
    uint32_t count = 0;  // wave-uniform
#pragma unroll
    for (int r8 = 0; r8 < AFF_RG; ++r8) {
      const int64_t r = base + r8;
      if (r < r1) {  // uniform
        const int32_t a0r = A0[r], a1r = A1[r];
        float p1r[D], p2r[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          p1r[k] = P1f[k * pstride + r];
          p2r[k] = P2f[k * pstride + r];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            const float t1 = p1r[k] - p1c[q][k];
            const float t2 = p2r[k] - p2c[q][k];
            s1 = fmaf(t1, t1, s1);
            s2 = fmaf(t2, t2, s2);
          }
          const float cf = fabsf(__builtin_amdgcn_sqrtf(s1) - __builtin_amdgcn_sqrtf(s2));
          // clipper.cpp:35-38 distinctness (also removes the diagonal) + conservative c < eps
          const bool cand = valid[q] && (a0r != a0c[q]) && (a1r != a1c[q]) && (cf < eps_guarded);
          const uint64_t mask = __ballot(cand);
          if (mask != 0) {  // uniform
            if (cand) queue[wave][count + lane_prefix(mask)] =
                (static_cast<uint32_t>(r8) << 16) | static_cast<uint32_t>(lane * 4 + q);
            count += static_cast<uint32_t>(__popcll(mask));
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- phase B: exact fp64 score of the survivors, scattered into the staging tile -------
    for (uint32_t e = lane; e < count; e += 64) {
      const uint32_t code = queue[wave][e];
      const int r8 = static_cast<int>(code >> 16);
      const int cl = static_cast<int>(code & 0xffffu);
      const double scr = exact_euclid_score<T, D>(P1, P2, pstride, base + r8, c0 + cw + cl, prm);
      stage[wave][r8][cl] = store_score<T>(scr, prm.affinityeps);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- phase C: whole row segments leave for HBM; the staging tile is re-zeroed ----------
    if (c < ld) {
#pragma unroll
      for (int r8 = 0; r8 < AFF_RG; ++r8) {
        const int64_t r = base + r8;
        if (r < r1) {
          T* sp = &stage[wave][r8][lane * 4];
          store4<T>(S + r * ld + c, sp[0], sp[1], sp[2], sp[3]);
          sp[0] = sp[1] = sp[2] = sp[3] = T(0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
__device__ __forceinline__ double exact_pointnormal_score(const double* __restrict__ P1,
                                                          const double* __restrict__ P2,
                                                          int64_t pstride, int64_t r, int64_t g,
                                                          const PointNormalParams& prm) {
  double p1r[6], p1g[6], p2r[6], p2g[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    p1r[k] = P1[k * pstride + r];
    p1g[k] = P1[k * pstride + g];
    p2r[k] = P2[k * pstride + r];
    p2g[k] = P2[k * pstride + g];
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double t1 = p1r[k] - p1g[k];
    const double t2 = p2r[k] - p2g[k];
    s1 = fma(t1, t1, s1);
    s2 = fma(t2, t2, s2);
  }
  const double l1 = sqrt(s1), l2 = sqrt(s2);  // :17-18
  const double dot1 = fma(p1r[5], p1g[5], fma(p1r[4], p1g[4], p1r[3] * p1g[3]));
  const double dot2 = fma(p2r[5], p2g[5], fma(p2r[4], p2g[4], p2r[3] * p2g[3]));
  const double alpha1 = acos(dot1);  // :21
  const double alpha2 = acos(dot2);  // :22
  const double dp = fabs(l1 - l2);          // :25
  const double dn = fabs(alpha1 - alpha2);  // :26
  if (dp < prm.epsp && dn < prm.epsn) {     // :28
    const double sp = exp(-0.5 * dp * dp / (prm.sigp * prm.sigp));  // :29
    const double sn = exp(-0.5 * dn * dn / (prm.sign * prm.sign));  // :30
    return sp * sn;                                                 // :31
  }
  return 0.0;
}

// PointNormalDistance: the prefilter tests only the point-distance residual dp (the normal
// residual needs acos); survivors get the full exact evaluation.
template <typename T>
__global__ __launch_bounds__(256) void k_affinity_pointnormal_compact(
    T* __restrict__ S, int64_t ld, int64_t m, int64_t c0, int rows_per_blk,
    const double* __restrict__ P1, const double* __restrict__ P2,
    const float* __restrict__ P1f, const float* __restrict__ P2f, int64_t pstride,
    const int32_t* __restrict__ A0, const int32_t* __restrict__ A1, PointNormalParams prm,
    float eps_guarded) {
  __shared__ uint32_t queue[4][AFF_RG * 256];
  __shared__ T stage[4][AFF_RG][256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t cw = static_cast<int64_t>(blockIdx.x) * 1024 + wave * 256;
  const int64_t c = cw + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_blk;
  const int64_t r1 = (r0 + rows_per_blk < m) ? r0 + rows_per_blk : m;
  if (cw >= ld) return;

  bool valid[4];
  int32_t a0c[4], a1c[4];
  float p1c[4][3], p2c[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t g = c0 + c + q;
    valid[q] = (g < m) && (c + q < ld);
    const int64_t gi = (g < m) ? g : (m - 1);
    a0c[q] = A0[gi];
    a1c[q] = A1[gi];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p1c[q][k] = P1f[k * pstride + gi];
      p2c[q][k] = P2f[k * pstride + gi];
    }
  }
#pragma unroll
  for (int r8 = 0; r8 < AFF_RG; ++r8)
#pragma unroll
    for (int q = 0; q < 4; ++q) stage[wave][r8][lane * 4 + q] = T(0);

  for (int64_t base = r0; base < r1; base += AFF_RG) {
    uint32_t count = 0;
#pragma unroll
    for (int r8 = 0; r8 < AFF_RG; ++r8) {
      const int64_t r = base + r8;
      if (r < r1) {
        const int32_t a0r = A0[r], a1r = A1[r];
        float p1r[3], p2r[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          p1r[k] = P1f[k * pstride + r];
          p2r[k] = P2f[k * pstride + r];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float t1 = p1r[k] - p1c[q][k];
            const float t2 = p2r[k] - p2c[q][k];
            s1 = fmaf(t1, t1, s1);
            s2 = fmaf(t2, t2, s2);
          }
          const float dpf = fabsf(__builtin_amdgcn_sqrtf(s1) - __builtin_amdgcn_sqrtf(s2));
          const bool cand = valid[q] && (a0r != a0c[q]) && (a1r != a1c[q]) && (dpf < eps_guarded);
          const uint64_t mask = __ballot(cand);
          if (mask != 0) {
            if (cand) queue[wave][count + lane_prefix(mask)] =
                (static_cast<uint32_t>(r8) << 16) | static_cast<uint32_t>(lane * 4 + q);
            count += static_cast<uint32_t>(__popcll(mask));
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e = lane; e < count; e += 64) {
      const uint32_t code = queue[wave][e];
      const int r8 = static_cast<int>(code >> 16);
      const int cl = static_cast<int>(code & 0xffffu);
      const double scr = exact_pointnormal_score<T>(P1, P2, pstride, base + r8, c0 + cw + cl, prm);
      stage[wave][r8][cl] = store_score<T>(scr, prm.affinityeps);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c < ld) {
#pragma unroll
      for (int r8 = 0; r8 < AFF_RG; ++r8) {
        const int64_t r = base + r8;
        if (r < r1) {
          T* sp = &stage[wave][r8][lane * 4];
          store4<T>(S + r * ld + c, sp[0], sp[1], sp[2], sp[3]);
          sp[0] = sp[1] = sp[2] = sp[3] = T(0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------
// Symmetric fill (one shard, fp32 storage): M is symmetric and the score of (i, j) is bit-equal
// to the score of (j, i) (squares and absolute differences only), so only the 128 x 128 tiles
// of the upper block triangle are evaluated — prefilter and exact scores cost half — and every
// off-diagonal tile leaves twice: as it stands, and transposed out of an LDS image with an odd
// row pitch (bank-conflict-free column reads), both as 512-byte row segments.
// The prefilter needs no square root: |l1 - l2| < E  <=>  t <= 0  or  t^2 < 4 s1 s2 with
// t = s1 + s2 - E^2 (s = squared lengths, E = the guarded threshold); the right-hand side
// carries a 2^-18 relative margin for the fp32 roundings of t, t^2 and s1 s2. Survivors get the
// same exact fp64 evaluation as in the other fill kernels: identical bits.
// Geometry: 8 waves; wave w owns tile rows [16w, 16w + 16), lane l tile columns 2l, 2l + 1.
// ------------------------------------------------------------------------------------------

constexpr int AT = 128;           // tile edge
constexpr int AT_PITCH = AT + 1;  // LDS image row pitch in floats
#ifndef CLIPPER_AT_WAVES
#define CLIPPER_AT_WAVES 8
#endif
constexpr int AT_WAVES = CLIPPER_AT_WAVES;       // waves per workgroup
constexpr int AT_ROWS_PER_WAVE = AT / AT_WAVES;  // tile rows a wave owns
constexpr int AT_QUEUE = 256;     // ring entries per wave: < 64 waiting + one row's 128 candidates
constexpr int AT_SYM_IMG_BYTES = (AT * AT_PITCH * 4 + 15) / 16 * 16;
constexpr int AT_SYM_LDS_BYTES = AT_SYM_IMG_BYTES + AT_WAVES * AT_QUEUE * 4;

// linear index t of the upper block triangle (row-major: (0,0) (0,1) ... (1,1) ...) -> (I, J)
__device__ __forceinline__ void tile_of(int t, int nT, int& I, int& J) {
  const float b = 2.0f * nT + 1.0f;
  int i = static_cast<int>((b - sqrtf(b * b - 8.0f * static_cast<float>(t))) * 0.5f);
  if (i < 0) i = 0;
  if (i > nT - 1) i = nT - 1;
  // first(i) = i*nT - i*(i-1)/2 is the index of tile (i, i); fix the float estimate
  while (i > 0 && i * nT - i * (i - 1) / 2 > t) --i;
  while (i + 1 < nT && (i + 1) * nT - (i + 1) * i / 2 <= t) ++i;
  I = i;
  J = i + (t - (i * nT - i * (i - 1) / 2));
}

template <int D, bool POINTNORMAL>
__global__ __launch_bounds__(AT_WAVES * 64, AT_WAVES / 2) void k_affinity_sym(
    float* __restrict__ S, int64_t ld, int64_t m, int nT, const double* __restrict__ P1,
    const double* __restrict__ P2, const float* __restrict__ P1f, const float* __restrict__ P2f,
    int64_t pstride, const int32_t* __restrict__ A0, const int32_t* __restrict__ A1,
    EuclidParams eprm, PointNormalParams nprm, float E2 /* guarded threshold squared, rounded up */,
    CscOut O /* O.Lc != null: also emit the tile's groups of the compressed copy */) {
  // 72.5 KiB of dynamic LDS (two workgroups per CU fit the 160 KiB): the image, then the queues
  extern __shared__ __attribute__((aligned(16))) char sym_smem[];
  float* img = reinterpret_cast<float*>(sym_smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* queue = reinterpret_cast<uint32_t*>(sym_smem + AT_SYM_IMG_BYTES) + wave * AT_QUEUE;
  int I, J;
  tile_of(blockIdx.x, nT, I, J);
  const int64_t r0 = static_cast<int64_t>(I) * AT, c0 = static_cast<int64_t>(J) * AT;
  const double affinityeps = POINTNORMAL ? nprm.affinityeps : eprm.affinityeps;

  // this lane's two columns (fp32 copies for the prefilter)
  bool validc[2];
  int32_t a0c[2], a1c[2];
  float p1c[2][D], p2c[2][D];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t g = c0 + 2 * lane + q;
    validc[q] = g < m;
    const int64_t gi = validc[q] ? g : (m - 1);
    a0c[q] = A0[gi];
    a1c[q] = A1[gi];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      p1c[q][k] = P1f[k * pstride + gi];
      p2c[q][k] = P2f[k * pstride + gi];
    }
  }
  // zero this wave's rows of the image
#pragma unroll 4
  for (int rr = 0; rr < AT_ROWS_PER_WAVE; ++rr) {
    float* row = img + (wave * AT_ROWS_PER_WAVE + rr) * AT_PITCH;
    row[2 * lane] = 0.f;
    row[2 * lane + 1] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // exact fp64 score of queue entries [head, head + n), scattered into the image
  auto drain = [&](uint32_t head, uint32_t n) {
    if (lane < n) {
      const uint32_t code = queue[(head + lane) & (AT_QUEUE - 1)];
      const int rl = static_cast<int>(code >> 8);
      const int cl = static_cast<int>(code & 0xffu);
      double scr;
      if (POINTNORMAL) scr = exact_pointnormal_score<float>(P1, P2, pstride, r0 + rl, c0 + cl, nprm);
      else scr = exact_euclid_score<float, D>(P1, P2, pstride, r0 + rl, c0 + cl, eprm);
      img[rl * AT_PITCH + cl] = store_score<float>(scr, affinityeps);
    }
  };

  // The survivors of the fp32 prefilter go into a per-wave ring; whenever 64 are waiting they are
  // evaluated by a full wave (the exact score is ~150 fp64-rate instructions: no idle lanes).
  uint32_t head = 0, tail = 0;  // wave-uniform
  for (int rr = 0; rr < AT_ROWS_PER_WAVE; ++rr) {
    const int rl = wave * AT_ROWS_PER_WAVE + rr;  // tile row
    const int64_t r = r0 + rl;
    if (r < m) {  // uniform
      const int32_t a0r = A0[r], a1r = A1[r];
      float p1r[D], p2r[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        p1r[k] = P1f[k * pstride + r];
        p2r[k] = P2f[k * pstride + r];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float t1 = p1r[k] - p1c[q][k];
          const float t2 = p2r[k] - p2c[q][k];
          s1 = fmaf(t1, t1, s1);
          s2 = fmaf(t2, t2, s2);
        }
        const float t = (s1 + s2) - E2;
        const bool close = (t <= 0.f) || (t * t < (4.0f * 1.0000038147f) * (s1 * s2));
        // clipper.cpp:35-38 distinctness (also removes the diagonal) + conservative c < eps
        const bool cand = validc[q] && (a0r != a0c[q]) && (a1r != a1c[q]) && close;
        const uint64_t mask = __ballot(cand);
        if (mask != 0) {  // uniform
          if (cand) queue[(tail + lane_prefix(mask)) & (AT_QUEUE - 1)] =
              (static_cast<uint32_t>(rl) << 8) | static_cast<uint32_t>(2 * lane + q);
          tail += static_cast<uint32_t>(__popcll(mask));
        }
      }
      if (tail - head >= 64) {  // uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (tail - head >= 64) {
          drain(head, 64);
          head += 64;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  while (head < tail) {
    const uint32_t n = (tail - head < 64) ? tail - head : 64;
    drain(head, n);
    head += n;
  }
  __syncthreads();

  // ---- the compressed copy: row blocks 2I, 2I+1 of strip J, and of the mirror image row blocks
  // 2J, 2J+1 of strip I, one column per thread straight from the image (csc_emit) ---------------
  if (O.Lc != nullptr) {
    static_assert(AT_WAVES == 8 && AT == CSC_CW && AT == 2 * CSC_RB, "four groups per tile");
    int* red = reinterpret_cast<int*>(sym_smem + AT_SYM_IMG_BYTES);  // the queues are drained
    unsigned long long* base_s = reinterpret_cast<unsigned long long*>(red + 8);
    const int t = threadIdx.x, gi = t >> 7, cl = t & 127;
    const int rb = gi & 1;
    const bool mirror = gi >= 2;
    // element q of the thread's column: tile (rb*64 + q, cl), or (cl, rb*64 + q) of the mirror
    const float* col = mirror ? img + cl * AT_PITCH + rb * CSC_RB : img + rb * CSC_RB * AT_PITCH + cl;
    const int strip = mirror ? I : J;
    const int b = 2 * (mirror ? J : I) + rb;
    const int64_t g = (b < O.nblocks && !(mirror && I == J))
                          ? static_cast<int64_t>(strip) * O.nblocks + b : -1;
    csc_emit_lds<4>(col, mirror ? 1 : AT_PITCH, g, O, red, base_s);
  }

  // ---- the tile as it stands: this wave's rows, 512-byte segments -----------------------------
  if (S != nullptr && c0 + 2 * lane < ld) {
    for (int rr = 0; rr < AT_ROWS_PER_WAVE; ++rr) {
      const int rl = wave * AT_ROWS_PER_WAVE + rr;
      const int64_t r = r0 + rl;
      if (r < m) {
        const float* row = img + rl * AT_PITCH + 2 * lane;
        *reinterpret_cast<float2*>(S + r * ld + c0 + 2 * lane) = make_float2(row[0], row[1]);
      }
    }
  }
  // ---- and transposed: rows c0 + ... receive the tile's columns ------------------------------
  if (S != nullptr && I != J && r0 + 2 * lane < ld) {
    for (int cc = 0; cc < AT_ROWS_PER_WAVE; ++cc) {
      const int cl = wave * AT_ROWS_PER_WAVE + cc;
      const int64_t c = c0 + cl;
      if (c < m) {
        const float v0 = img[(2 * lane) * AT_PITCH + cl];
        const float v1 = img[(2 * lane + 1) * AT_PITCH + cl];
        *reinterpret_cast<float2*>(S + c * ld + r0 + 2 * lane) = make_float2(v0, v1);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// matrix upload (setMatrixData / setSparseMatrixData) — clipper.cpp:149-166
// ------------------------------------------------------------------------------------------

// S[j][c] = Mdense(min(j,g), max(j,g)) for g = c0+c != j, 0 on the diagonal / padding.
// Mdense is column-major m x m fp64 in device memory (only its strict upper triangle is
// read). `mismatch` is raised when Cdense's upper triangle differs from pattern(Mdense).
template <typename T>
__global__ __launch_bounds__(256) void k_from_dense_upper(T* __restrict__ S, int64_t ld,
                                                           int64_t m, int64_t c0,
                                                           const double* __restrict__ Md,
                                                           const double* __restrict__ Cd,
                                                           T* __restrict__ Cs,
                                                           int* __restrict__ mismatch) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (c >= ld) return;
  const int64_t g = c0 + c;
  double mv = 0.0, cv = 0.0;
  if (g < m && g != j) {
    const int64_t lo = (j < g) ? j : g, hi = (j < g) ? g : j;
    mv = Md[lo + hi * m];
    cv = Cd[lo + hi * m];
    if (mismatch != nullptr) {
      const double want = (mv != 0.0) ? 1.0 : 0.0;
      if (cv != want) *mismatch = 1;
    }
  }
  T sv = static_cast<T>(mv);
  if (mv != 0.0 && sv == T(0)) sv = (mv > 0) ? static_cast<T>(1.17549435e-38)
                                             : static_cast<T>(-1.17549435e-38);
  S[j * ld + c] = sv;
  if (Cs != nullptr) Cs[j * ld + c] = static_cast<T>(cv);
}

// out[a*k + b] = M(idx[a], idx[b]) for the columns idx[b] this slice owns (others untouched):
// the sub-matrix induced by the non-zero entries of u, for the exact DSD rounding on the host
template <typename T>
__global__ __launch_bounds__(256) void k_gather_sub(const T* __restrict__ S, int64_t ld,
                                                     int64_t c0, int64_t W,
                                                     const int32_t* __restrict__ idx, int k,
                                                     double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= static_cast<int64_t>(k) * k) return;
  const int a = static_cast<int>(e / k), b = static_cast<int>(e - static_cast<int64_t>(a) * k);
  const int64_t col = idx[b];
  if (col >= c0 && col < c0 + W) out[e] = static_cast<double>(S[static_cast<int64_t>(idx[a]) * ld + (col - c0)]);
}

// scatter of strictly-upper CSC entries (both mirror images) into a zeroed slice
template <typename T>
__global__ __launch_bounds__(256) void k_from_csc(T* __restrict__ S, int64_t ld, int64_t m,
                                                   int64_t c0, int64_t W,
                                                   const int64_t* __restrict__ colptr,
                                                   const int32_t* __restrict__ row,
                                                   const double* __restrict__ val) {
  const int64_t j = blockIdx.x;  // CSC column
  for (int64_t p = colptr[j] + threadIdx.x; p < colptr[j + 1]; p += 256) {
    const int64_t i = row[p];
    if (i == j) continue;  // the solver treats the diagonal as implicit identity
    const T v = static_cast<T>(val[p]);
    // element (i,j): lives at S[i][j-c0] if j is an owned column, and at S[j][i-c0] if i is
    if (j >= c0 && j < c0 + W) S[i * ld + (j - c0)] = v;
    if (i >= c0 && i < c0 + W) S[j * ld + (i - c0)] = v;
  }
}

}  // namespace clipper_hip
