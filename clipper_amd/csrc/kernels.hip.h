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
//   k_gemv        : every `M_.selfadjointView<Upper>() * v` / `C_...* v` in
//                   src/clipper.cpp:194,202,205,219,240-241,268,271 (one fused pass gives both)
//   k_tail+k_decide: the O(m) algebra and control flow of findDenseClique,
//                   src/clipper.cpp:193-209 (init), :219-220 (gradient), :226-262 (step,
//                   projection, line search), :268-280 (penalty update)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// solver state that lives in device memory for the whole solve (the host only watches the
// pinned HostMirror)
// ------------------------------------------------------------------------------------------

enum Phase : int32_t {
  PH_NORMALIZE = 0,  // no rescale: u = u0/||u0||, no pass consumed       (clipper.cpp:196-198)
  PH_RESCALE = 1,    // pass was on x = u0: u = M_off u0 + u0, normalise  (clipper.cpp:193-198)
  PH_INIT = 2,       // pass was on x = u: initial d, first gradient      (clipper.cpp:200-220)
  PH_TRIAL = 3       // pass was on x = unew: line-search bookkeeping     (clipper.cpp:234-262)
};

// The vector a pass runs on is kept UN-normalised: x = Tin[sel] / nrm. The mat-vec multiplies
// M by Tin[sel]; the tail divides the two sums by nrm (one division per column instead of one
// per element of x before the pass) and materialises x where it is needed.
struct SolverState {
  double d;       // penalty
  double F;       // objective at u
  double alpha;   // current step size
  double s;       // sum(u)
  double sx;      // sum(x) of the pending trial vector
  double nrm;     // ||Tin[sel]|| (1 when the pending vector is already normalised / raw u0)
  int32_t sel;    // which of Tin[0], Tin[1] holds the pending trial vector
  int32_t ub;     // which of U[0]/G[0], U[1]/G[1] holds the current (u, gradF)
  int32_t phase;
  int32_t i, j, k;  // outer / inner / line-search counters (clipper.cpp:217)
  int32_t done;
  int32_t ifinal;
  int64_t n_passes;
  int64_t n_trials;
  int64_t n_iters;  // decisions taken so far (= solver iterations the device has retired)
};

// Host-visible progress record in pinned, coherent host memory. The deciding workgroup writes
// it with system-scope stores; the host spins on `iters` / `done` instead of issuing
// memcpy + event round trips, and keeps only a few iterations queued ahead of the device.
struct HostMirror {
  double F, d;
  int64_t n_passes, n_trials;
  int64_t iters;
  int32_t ifinal, ub;
  int32_t done;
  int32_t pad_;
};

struct SolverParams {
  double tol_u, tol_F, beta, eps;
  int32_t maxiniters, maxoliters, maxlsiters;
};

constexpr int TAIL_THREADS = 256;
constexpr int NSCAL = 6;  // per-workgroup partial scalars written by the tail

struct SolveArgs {
  SolverState* st;
  HostMirror* host;  // device address of the pinned progress record (may be null)
  SolverParams prm;
  int64_t m;   // problem size
  int64_t W;   // shard pitch: element i lives in block p = i / W of `ab`
  const double* u0;
  double* U[2];   // current point u (double-buffered: accepted x becomes u by flipping `ub`)
  double* G[2];   // gradF at u / at the trial vector
  // un-normalised trial vectors, [0] = "if accepted", [1] = "if rejected". A launch READS the
  // pending vector from Tin and WRITES the two candidates of the next pass to Tout; the host
  // swaps the pairs from launch to launch, so that a workgroup that finishes its columns early
  // can never overwrite an x another workgroup of the same launch is still multiplying by.
  const double* Tin[2];
  double* Tout[2];
  double* ab;     // [P][2][W]: a = M_off x, b = C_off x (raw sums in, normalised out)
  double* part;   // [ntiles][2][W] row-tile partials of this shard
  int ntiles;
  int slot;       // this shard's block of `ab`
  double* scal;   // [nwg][NSCAL] partial scalars of the tail
  int nwg;        // workgroups of the tail = ceil(m / TAIL_THREADS)
  int* cnt;       // arrival counters: [0, nstrips) one per column strip, [nstrips] the tail's
  int nstrips;
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void ab_at(const double* ab, int64_t W, int64_t i, double& a,
                                      double& b) {
  // block p = i / W of the gathered [P][2][W] layout; 32-bit division (m < 2^31)
  const uint32_t p = static_cast<uint32_t>(i) / static_cast<uint32_t>(W);
  const int64_t off = i - static_cast<int64_t>(p) * W;
  const double* blk = ab + static_cast<int64_t>(p) * 2 * W;
  a = blk[off];
  b = blk[W + off];
}

// Last-arriver hand-off inside one launch (CDNA guide, section 6 guideline 16, counter form;
// the split-K recipe): every workgroup publishes what it stored — each wave drains its own
// stores, one lane issues the agent-scope release and draws a ticket — and the workgroup that
// draws the last ticket acquires at agent scope and continues with plain loads. Correct for
// any placement of the workgroups over the 8 XCDs (their L2s are not coherent with each other).
// The counter is zeroed before the first launch of a solve (k_init) and re-armed by the last
// arriver. Returns true in every thread of the last workgroup. `flag` is one int of LDS.
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

// ------------------------------------------------------------------------------------------
// The O(m) part of one solver iteration, split so that it parallelises:
//
//   tail    (one workgroup per 256 elements, one thread per element) — everything element-wise
//           that follows a pass on the trial vector x (clipper.cpp:238-242, 253): a = M_off x,
//           b = C_off x (sum of the row-tile partials, divided by nrm), gradFnew, and the
//           per-workgroup partial sums of Fnew = x.gradFnew and ||x-u||^2. It also prepares
//           BOTH possible next trial vectors (clipper.cpp:235-236) speculatively —
//           Tout[0] = max(x + gradFnew, 0) if the step is accepted (alpha resets to 1),
//           Tout[1] = max(u + alpha*beta*gradF, 0) if it is rejected — with the partial sums
//           of their squared norms and of their entries, so that the only serial work left is
//   decide  (one workgroup) — adds the partial scalars in a fixed order, takes the
//           reference's decisions (clipper.cpp:244-251, 261) and updates the state. Only
//           the rare transitions (initialisation :193-220, penalty update :268-280, a new
//           outer iteration :219-220) sweep over the m-vectors here.
//
// Both are device functions shared by three launch shapes:
//   k_gemv -> k_tail<.., true>            two launches per pass: the last tail workgroup decides
//   k_pass<.., PASS_FUSED>                one launch per pass: the last workgroup of a column
//                                         strip runs that strip's tail, the last strip decides
//   k_pass<.., PASS_REDUCE> -> exchange -> k_tail<false, true>    column-sharded M
// All sums have a fixed shape (256 data-carrying threads, partials in workgroup order), so the
// results are bit-identical across the three shapes, from run to run and from rank to rank.
// ------------------------------------------------------------------------------------------

// Sum over the first NWAVES waves of the workgroup; every thread of the workgroup must call it
// (waves beyond NWAVES only take part in the barriers) and every thread gets the totals.
template <int N, int NWAVES>
__device__ __forceinline__ void block_reduce(double (&v)[N], double* lds /* [NWAVES*N] */) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] += shfl_xor_f64(v[q], off);
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && wave < NWAVES) {
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

constexpr int TAIL_WAVES = TAIL_THREADS / 64;
constexpr int RED_DOUBLES = TAIL_WAVES * NSCAL + 2;  // reduction scratch + the arrival flag

struct TailLoads {  // everything element i needs that does not depend on the state
  double t0, t1, u0v, u1v, g0v, g1v;
};

__device__ __forceinline__ TailLoads tail_loads(const SolveArgs& A, int64_t i, bool valid) {
  TailLoads L = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (valid) {
    L.t0 = A.Tin[0][i];
    L.t1 = A.Tin[1][i];
    L.u0v = A.U[0][i];
    L.u1v = A.U[1][i];
    L.g0v = A.G[0][i];
    L.g1v = A.G[1][i];
  }
  return L;
}

// partial sums of element i over the row tiles, in tile order (single shard: W >= m, block 0);
// 8 tiles of loads in flight at a time
__device__ __forceinline__ void sum_partials(const SolveArgs& A, int64_t i, double& a, double& b) {
  const double* p = A.part + i;
  const int64_t ts = 2 * A.W;
  int t = 0;
  for (; t + 8 <= A.ntiles; t += 8) {
    double va[8], vb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      va[q] = p[static_cast<int64_t>(t + q) * ts];
      vb[q] = p[static_cast<int64_t>(t + q) * ts + A.W];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a += va[q];
      b += vb[q];
    }
  }
  for (; t < A.ntiles; ++t) {
    a += p[static_cast<int64_t>(t) * ts];
    b += p[static_cast<int64_t>(t) * ts + A.W];
  }
}

// Element-wise tail of element i (raw sums a, b in). `valid` = this thread carries an element
// (one of the first 256 threads of the workgroup and i < m). Writes scal[slot][0..5].
// Returns false when the state says there is nothing to do after the normalisation.
__device__ __forceinline__ void tail_elements(const SolveArgs& A, const SolverState& stv,
                                              int64_t i, bool valid, double a, double b,
                                              const TailLoads& L, double* red, int64_t slot) {
  const double nrm = stv.nrm;
  // (a, b) for the pending vector x = Tin[sel]/nrm
  if (valid) {
    a = a / nrm;
    b = b / nrm;
    // normalised pair back into the gathered layout (read again only by decide's rare sweeps)
    const uint32_t pblk = static_cast<uint32_t>(i) / static_cast<uint32_t>(A.W);
    const int64_t off = i - static_cast<int64_t>(pblk) * A.W;
    A.ab[static_cast<int64_t>(pblk) * 2 * A.W + off] = a;
    A.ab[static_cast<int64_t>(pblk) * 2 * A.W + A.W + off] = b;
  }
  if (stv.phase != PH_TRIAL) return;  // initialisation phases are handled by decide alone

  const int ub = stv.ub;
  const double d = stv.d, sx = stv.sx, alpha = stv.alpha;
  double r[NSCAL] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (valid) {
    const double xi = (stv.sel ? L.t1 : L.t0) / nrm;  // clipper.cpp:237
    const double ui = ub ? L.u1v : L.u0v;
    const double gi = ub ? L.g1v : L.g0v;
    const double gn = (1 + d) * xi - d * sx + a + b * d;  // :238-241
    A.U[ub ^ 1][i] = xi;                                   // becomes u if accepted
    A.G[ub ^ 1][i] = gn;
    r[0] = xi * gn;  // :242
    const double du = xi - ui;
    r[1] = du * du;  // :253
    double ta = xi + gn;  // next trial if accepted: alpha = 1 (:227, :235)
    ta = (ta > 0.0) ? ta : 0.0;  // :236
    double tr = ui + (alpha * A.prm.beta) * gi;  // next trial if rejected: alpha*beta (:248)
    tr = (tr > 0.0) ? tr : 0.0;
    A.Tout[0][i] = ta;
    A.Tout[1][i] = tr;
    r[2] = ta * ta;
    r[3] = ta;
    r[4] = tr * tr;
    r[5] = tr;
  }
  block_reduce<NSCAL, TAIL_WAVES>(r, red);
  if (threadIdx.x < NSCAL) A.scal[slot * NSCAL + threadIdx.x] = r[threadIdx.x];
}

constexpr int VU = 4;  // elements per thread per sweep step (register budget of the fused epilogue)
constexpr int DECIDE_THREADS = 256;
constexpr int DECIDE_WAVES = DECIDE_THREADS / 64;
// only the first DECIDE_THREADS threads of the workgroup carry elements
#define VEC_CHUNKS(base) \
  for (int64_t base = (tid < DECIDE_THREADS) ? tid : m; base < m; base += DECIDE_THREADS * VU)
#define VEC_EACH(k, i, base)            \
  _Pragma("unroll") for (int k = 0; k < VU; ++k) \
    if (const int64_t i = base + static_cast<int64_t>(k) * DECIDE_THREADS; i < m)

// One decision of the solver's state machine, by ONE workgroup (all of its threads must call
// this; the first 256 carry data). `stv` is the state as read at the start of the launch.
__device__ __forceinline__ void decide_body(const SolveArgs& A, const SolverState& stv,
                                            double* red) {
  SolverState* st = A.st;
  const int tid = threadIdx.x;
  const int64_t m = A.m;
  const SolverParams P = A.prm;

  // the tail's partial scalars, one workgroup-strided sweep
  double r[NSCAL] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (tid < DECIDE_THREADS) {
    for (int w = tid; w < A.nwg; w += DECIDE_THREADS) {
#pragma unroll
      for (int q = 0; q < NSCAL; ++q) r[q] += A.scal[static_cast<int64_t>(w) * NSCAL + q];
    }
  }

  const int phase = stv.phase;
  double d = stv.d, F = stv.F, alpha = stv.alpha, s = stv.s, sx = stv.sx, nrm = stv.nrm;
  int i_ = stv.i, j_ = stv.j, k_ = stv.k, ub = stv.ub, sel = stv.sel;
  int64_t n_passes = stv.n_passes, n_trials = stv.n_trials;
  const int64_t n_iters = stv.n_iters + 1;
  if (phase != PH_NORMALIZE) ++n_passes;

  if (phase == PH_NORMALIZE || phase == PH_RESCALE) {
    // clipper.cpp:193-198 — u = M_off*u0 + u0 (or u0), then u /= u.norm()
    double* u = A.U[ub];
    double z[1] = {0.0};
    VEC_CHUNKS(base) {
      double uv[VU], av[VU];
      VEC_EACH(k, i, base) {
        uv[k] = A.u0[i];
        double b;
        if (phase == PH_RESCALE) ab_at(A.ab, A.W, i, av[k], b);
      }
      VEC_EACH(k, i, base) {
        const double ui = (phase == PH_RESCALE) ? av[k] + uv[k] : uv[k];
        u[i] = ui;
        z[0] += ui * ui;
      }
    }
    block_reduce<1, DECIDE_WAVES>(z, red);
    const double n0 = sqrt(z[0]);
    VEC_CHUNKS(base) {
      double uv[VU];
      VEC_EACH(k, i, base) uv[k] = u[i];
      VEC_EACH(k, i, base) {
        const double ui = uv[k] / n0;
        u[i] = ui;
        A.Tout[0][i] = ui;  // next pass runs on x = u (already normalised: nrm = 1)
      }
    }
    if (tid == 0) {
      st->phase = PH_INIT;
      st->sel = 0;
      st->nrm = 1.0;
      st->n_passes = n_passes;
      st->n_iters = n_iters;
      if (A.host != nullptr)
        __hip_atomic_store(&A.host->iters, n_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }

  bool begin_outer = false, end_inner = false, finished = false;
  bool need_trial_vector = false;  // a sweep must build Tout[0] from (u, g): after a transition

  if (phase == PH_INIT) {
    // clipper.cpp:200-209 — initial d from the pass on u
    const double* u = A.U[ub];
    double sv[1] = {0.0};
    VEC_CHUNKS(base) {
      double uv[VU];
      VEC_EACH(k, i, base) uv[k] = u[i];
      VEC_EACH(k, i, base) sv[0] += uv[k];
    }
    block_reduce<1, DECIDE_WAVES>(sv, red);
    s = sv[0];
    double ca[2] = {0.0, 0.0};  // count, sum of ratios
    VEC_CHUNKS(base) {
      double uv[VU], av[VU], bv[VU];
      VEC_EACH(k, i, base) {
        uv[k] = u[i];
        ab_at(A.ab, A.W, i, av[k], bv[k]);
      }
      VEC_EACH(k, i, base) {
        const double cbu = s - bv[k] - uv[k];  // :202
        if (cbu > P.eps && uv[k] > P.eps) {    // :203
          ca[0] += 1.0;
          ca[1] += (av[k] + uv[k]) / cbu;  // :205-208
        }
      }
    }
    block_reduce<2, DECIDE_WAVES>(ca, red);
    d = (ca[0] > 0.0) ? ca[1] / ca[0] : 0.0;
    i_ = 0;
    begin_outer = true;
  } else {  // PH_TRIAL — the decisions of clipper.cpp:244-262 from the tail's partial scalars
    ++n_trials;
    block_reduce<NSCAL, DECIDE_WAVES>(r, red);
    const double Fnew = r[0];
    const double deltaF = Fnew - F;  // :244
    bool accept = true;
    if (deltaF < -P.eps) {  // :246-248
      alpha = alpha * P.beta;
      ++k_;
      if (k_ < P.maxlsiters) accept = false;  // :234 loop bound; the last trial is kept
    }
    if (!accept) {
      sel = 1;  // the tail already built max(u + alpha*beta*g, 0) in Tout[1]
      nrm = (r[4] > 0.0) ? sqrt(r[4]) : 1.0;  // Eigen normalize(): only if squaredNorm > 0
      sx = r[5] / nrm;
    } else {
      const double deltau = sqrt(r[1]);
      F = Fnew;  // :256-258 — u <- x, gradF <- gradFnew by flipping the buffer index
      ub ^= 1;
      s = sx;
      ++j_;
      if (deltau < P.tol_u || fabs(deltaF) < P.tol_F || j_ >= P.maxiniters) {  // :261, :226
        end_inner = true;
      } else {
        alpha = 1.0;  // :227
        k_ = 0;
        sel = 0;  // the tail already built max(x + gradFnew, 0) in Tout[0]
        nrm = (r[2] > 0.0) ? sqrt(r[2]) : 1.0;
        sx = r[3] / nrm;
      }
    }
  }

  // Transitions that need no pass over M: penalty update (:268-280) and the gradient at the
  // start of the next outer iteration (:219-220) reuse (a, b) of the accepted vector.
  while (true) {
    if (end_inner) {
      const double* u = A.U[ub];
      double ca[2] = {0.0, 0.0};
      VEC_CHUNKS(base) {
        double uv[VU], av[VU], bv[VU];
        VEC_EACH(k, i, base) {
          uv[k] = u[i];
          ab_at(A.ab, A.W, i, av[k], bv[k]);
        }
        VEC_EACH(k, i, base) {
          const double cbu = s - bv[k] - uv[k];  // :268
          if (cbu > P.eps && uv[k] > P.eps) {    // :269
            ca[0] += 1.0;
            ca[1] += fabs((av[k] + uv[k]) / cbu);  // :271-274
          }
        }
      }
      block_reduce<2, DECIDE_WAVES>(ca, red);
      end_inner = false;
      if (ca[0] > 0.0) {
        d += ca[1] / ca[0];  // :276
        ++i_;                // :218 loop increment
        begin_outer = true;
      } else {
        finished = true;  // :278-280 break
        break;
      }
    }
    if (begin_outer) {
      begin_outer = false;
      if (i_ >= P.maxoliters) {  // :218 loop bound
        finished = true;
        break;
      }
      const double* u = A.U[ub];
      double* g = A.G[ub];
      double f[1] = {0.0};
      VEC_CHUNKS(base) {
        double uv[VU], av[VU], bv[VU];
        VEC_EACH(k, i, base) {
          uv[k] = u[i];
          ab_at(A.ab, A.W, i, av[k], bv[k]);
        }
        VEC_EACH(k, i, base) {
          const double gi = (1 + d) * uv[k] - d * s + av[k] + bv[k] * d;  // :219
          g[i] = gi;
          f[0] += uv[k] * gi;  // :220
        }
      }
      block_reduce<1, DECIDE_WAVES>(f, red);
      F = f[0];
      j_ = 0;
      if (P.maxiniters <= 0) {
        end_inner = true;
        continue;
      }
      alpha = 1.0;
      k_ = 0;
      need_trial_vector = true;
    }
    break;
  }

  if (!finished && need_trial_vector) {
    // :235-236 with alpha = 1 — gradient step and projection; normalisation is deferred (nrm)
    const double* u = A.U[ub];
    const double* g = A.G[ub];
    double zs[2] = {0.0, 0.0};
    VEC_CHUNKS(base) {
      double uv[VU], gv[VU];
      VEC_EACH(k, i, base) {
        uv[k] = u[i];
        gv[k] = g[i];
      }
      VEC_EACH(k, i, base) {
        double t = uv[k] + alpha * gv[k];
        t = (t > 0.0) ? t : 0.0;
        A.Tout[0][i] = t;
        zs[0] += t * t;
        zs[1] += t;
      }
    }
    block_reduce<2, DECIDE_WAVES>(zs, red);
    sel = 0;
    nrm = (zs[0] > 0.0) ? sqrt(zs[0]) : 1.0;
    sx = zs[1] / nrm;
  }

  if (tid == 0) {
    st->d = d;
    st->F = F;
    st->alpha = alpha;
    st->s = s;
    st->sx = sx;
    st->nrm = nrm;
    st->sel = sel;
    st->ub = ub;
    st->phase = PH_TRIAL;
    st->i = i_;
    st->j = j_;
    st->k = k_;
    st->n_passes = n_passes;
    st->n_trials = n_trials;
    st->n_iters = n_iters;
    if (finished) {
      st->ifinal = i_;
      st->done = 1;
    }
    if (A.host != nullptr) {
      HostMirror* hm = A.host;
      if (finished) {
        __hip_atomic_store(&hm->F, F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->d, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->n_passes, n_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->n_trials, n_trials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ifinal, i_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ub, ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // every store above (and the vectors this workgroup wrote) before the flag
        __hip_atomic_store(&hm->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(&hm->iters, n_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
#undef VEC_CHUNKS
#undef VEC_EACH

// Solve prologue, one launch: pending vector = u0 (un-normalised, nrm = 1), initial state,
// arrival counters zeroed.
__global__ __launch_bounds__(256) void k_init(SolveArgs A, SolverState init, double* T0) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < A.m) T0[i] = A.u0[i];
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c <= A.nstrips; c += 256) A.cnt[c] = 0;
    if (threadIdx.x == 0) *A.st = init;
  }
}

// The tail as its own launch (one workgroup per 256 elements).
//   FUSED_REDUCE: sum the row-tile partials of the single shard here (else `ab` holds the
//                 gathered raw sums of all shards);
//   FUSED_DECIDE: the last workgroup to arrive takes the decision (else k_decide follows).
template <bool FUSED_REDUCE, bool FUSED_DECIDE>
__global__ __launch_bounds__(TAIL_THREADS) void k_tail(SolveArgs A) {
  __shared__ double red[RED_DOUBLES];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * TAIL_THREADS + threadIdx.x;
  const bool valid = i < A.m;

  // Everything that does not depend on the solver state is loaded first, so that the state,
  // the partials and both buffer candidates share ONE memory round trip instead of chaining.
  const SolverState stv = *A.st;  // one 128-byte read
  const TailLoads L = tail_loads(A, i, valid);
  double a = 0.0, b = 0.0;
  if (valid) {
    if (FUSED_REDUCE) sum_partials(A, i, a, b);
    else ab_at(A.ab, A.W, i, a, b);
  }
  if (stv.done) return;
  tail_elements(A, stv, i, valid, a, b, L, red, blockIdx.x);
  if (FUSED_DECIDE) {
    if (!arrive_last(A.cnt + A.nstrips, gridDim.x, reinterpret_cast<int*>(red + RED_DOUBLES - 1)))
      return;
    decide_body(A, stv, red);
  }
}

__global__ __launch_bounds__(DECIDE_THREADS) void k_decide(SolveArgs A) {
  __shared__ double red[RED_DOUBLES];
  const SolverState stv = *A.st;
  if (stv.done) return;
  decide_body(A, stv, red);
}

// ------------------------------------------------------------------------------------------
// k_gemv — the fused symmetric mat-vec pair  a = M_off x,  b = C_off x  in ONE pass over M.
//
// grid = (strips of 256 columns, row tiles). A workgroup of NW waves shares one column
// strip; wave w takes rows r0 + w*UNR + k*NW*UNR ... of its tile, UNR rows per iteration
// so UNR independent 16-byte loads per lane are in flight. x[row] is wave-uniform: the
// compiler turns it into scalar loads. Per-wave partials are combined through LDS in wave
// order and written to part[tile][2][ld]; k_reduce adds the tiles in tile order. Nothing
// is atomic: the result is bit-reproducible from run to run and from rank to rank.
//
// HBM-bound: s*m*W bytes per launch (s = sizeof(T)); per element one cvt, one fma, one
// compare/select, one add — far below the fp64 vector rate, MFMA has nothing to offer a
// rank-1 product.
// ------------------------------------------------------------------------------------------

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

// The streaming part: this workgroup's (strip, row tile) partial sums -> part[tile][2][ld].
template <typename T, bool HASC, int NW, int UNR>
__device__ __forceinline__ void gemv_core(const T* __restrict__ S, const T* __restrict__ Cs,
                                          int64_t ld, int64_t m, int rows_per_tile,
                                          const double* __restrict__ x,
                                          double* __restrict__ part, double* lds) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_tile;
  const int64_t r1 = (r0 + rows_per_tile < m) ? r0 + rows_per_tile : m;

  double aa[4] = {0.0, 0.0, 0.0, 0.0};
  double bb[4] = {0.0, 0.0, 0.0, 0.0};

  if (col < ld) {
    const T* p = S + col;
    const T* pc = HASC ? Cs + col : nullptr;
    int64_t r = r0 + static_cast<int64_t>(wave) * UNR;
    for (; r + UNR <= r1; r += static_cast<int64_t>(NW) * UNR) {
      typename Vec4<T>::type v[UNR];
      typename Vec4<T>::type c[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        v[q] = load4(p + (r + q) * ld);
        if (HASC) c[q] = load4(pc + (r + q) * ld);
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const double xr = x[r + q];
        const double m0 = static_cast<double>(v[q].x), m1 = static_cast<double>(v[q].y),
                     m2 = static_cast<double>(v[q].z), m3 = static_cast<double>(v[q].w);
        aa[0] = fma(m0, xr, aa[0]);
        aa[1] = fma(m1, xr, aa[1]);
        aa[2] = fma(m2, xr, aa[2]);
        aa[3] = fma(m3, xr, aa[3]);
        if (HASC) {
          bb[0] = fma(static_cast<double>(c[q].x), xr, bb[0]);
          bb[1] = fma(static_cast<double>(c[q].y), xr, bb[1]);
          bb[2] = fma(static_cast<double>(c[q].z), xr, bb[2]);
          bb[3] = fma(static_cast<double>(c[q].w), xr, bb[3]);
        } else {
          bb[0] += (v[q].x != T(0)) ? xr : 0.0;
          bb[1] += (v[q].y != T(0)) ? xr : 0.0;
          bb[2] += (v[q].z != T(0)) ? xr : 0.0;
          bb[3] += (v[q].w != T(0)) ? xr : 0.0;
        }
      }
    }
    // tail rows of this wave's last chunk
    for (int q = 0; q < UNR; ++q) {
      const int64_t rr = r + q;
      if (rr < r1) {
        const typename Vec4<T>::type v = load4(p + rr * ld);
        const double xr = x[rr];
        const double mm[4] = {static_cast<double>(v.x), static_cast<double>(v.y),
                              static_cast<double>(v.z), static_cast<double>(v.w)};
        double cc[4];
        if (HASC) {
          const typename Vec4<T>::type c = load4(pc + rr * ld);
          cc[0] = static_cast<double>(c.x);
          cc[1] = static_cast<double>(c.y);
          cc[2] = static_cast<double>(c.z);
          cc[3] = static_cast<double>(c.w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          aa[e] = fma(mm[e], xr, aa[e]);
          if (HASC) {
            bb[e] = fma(cc[e], xr, bb[e]);
          } else {
            bb[e] += (mm[e] != 0.0) ? xr : 0.0;
          }
        }
      }
    }
  }

  // cross-wave combine in wave order (fixed summation tree)
  double* mine = lds + wave * 512 + lane * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mine[e] = aa[e];
    mine[256 + e] = bb[e];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 512; t += NW * 64) {
    double acc = lds[t];
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += lds[w * 512 + t];
    const int which = t >> 8;  // 0 = a, 1 = b
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + (t & 255);
    if (c < ld) part[(static_cast<int64_t>(blockIdx.y) * 2 + which) * ld + c] = acc;
  }
}


constexpr int GEMV_LDS_DOUBLES(int NW) { return NW * 2 * 256 + 2; }  // + the arrival flag

template <typename T, bool HASC, int NW, int UNR>
__global__ __launch_bounds__(NW * 64) void k_gemv(const T* __restrict__ S,
                                                   const T* __restrict__ Cs, int64_t ld,
                                                   int64_t m, int rows_per_tile,
                                                   const double* __restrict__ x0,
                                                   const double* __restrict__ x1,
                                                   double* __restrict__ part,
                                                   const SolverState* __restrict__ st) {
  if (st != nullptr && st->done) return;
  // driven by the solver: the pending trial vector is Tin[sel] (un-normalised, see SolverState)
  const double* __restrict__ x = (st != nullptr && st->sel) ? x1 : x0;
  __shared__ double lds[GEMV_LDS_DOUBLES(NW)];
  gemv_core<T, HASC, NW, UNR>(S, Cs, ld, m, rows_per_tile, x, part, lds);
}

// k_pass — the mat-vec with the rest of the solver iteration folded into its epilogue.
//   PASS_FUSED  (one shard): the LAST row-tile workgroup of a column strip (arrival counter per
//               strip) sums that strip's partials in tile order and runs the element-wise tail
//               for its 256 columns while the other strips are still streaming; the last strip
//               to finish (second counter) takes the decision. One launch per solver iteration.
//   PASS_REDUCE (column-sharded M): the last workgroup of a strip writes the strip's raw sums
//               into this shard's [a | b] block (what k_reduce did in a launch of its own); the
//               exchange and k_tail<false, true> follow.
// The strip tail writes the NEXT pass's candidates to Tout, never to the Tin other workgroups
// of this launch are still reading.
enum PassMode : int { PASS_FUSED = 1, PASS_REDUCE = 2 };

// two workgroups per CU (NW*64/256 * 2 waves per SIMD): caps the epilogue's register appetite
template <typename T, bool HASC, int NW, int UNR, int MODE>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_pass(const T* __restrict__ S,
                                                   const T* __restrict__ Cs,
                                                   int rows_per_tile, SolveArgs A) {
  static_assert(NW * 64 >= TAIL_THREADS, "the strip tail needs 256 threads");
  __shared__ double lds[GEMV_LDS_DOUBLES(NW)];
  const SolverState stv = *A.st;
  if (stv.done) return;
  const int64_t ld = A.W;
  gemv_core<T, HASC, NW, UNR>(S, Cs, ld, A.m, rows_per_tile, stv.sel ? A.Tin[1] : A.Tin[0],
                              A.part, lds);
  int* flag = reinterpret_cast<int*>(lds + GEMV_LDS_DOUBLES(NW) - 1);
  if (!arrive_last(A.cnt + blockIdx.x, gridDim.y, flag)) return;

  // ---- last workgroup of this column strip ------------------------------------------------
  if (MODE == PASS_REDUCE) {
    double* ab_block = A.ab + static_cast<int64_t>(A.slot) * 2 * ld;
    for (int t = threadIdx.x; t < 512; t += NW * 64) {
      const int which = t >> 8;
      const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + (t & 255);
      if (c < ld) {
        const double* p = A.part + which * ld + c;
        double acc = 0.0;
        int tt = 0;
        for (; tt + 8 <= A.ntiles; tt += 8) {
          double v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = p[static_cast<int64_t>(tt + q) * 2 * ld];
#pragma unroll
          for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; tt < A.ntiles; ++tt) acc += p[static_cast<int64_t>(tt) * 2 * ld];
        ab_block[which * ld + c] = acc;
      }
    }
    return;
  }
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const bool valid = (threadIdx.x < TAIL_THREADS) && (i < A.m);
  const TailLoads L = tail_loads(A, i, valid);
  double a = 0.0, b = 0.0;
  if (valid) sum_partials(A, i, a, b);
  tail_elements(A, stv, i, valid, a, b, L, lds, blockIdx.x);
  if (!arrive_last(A.cnt + A.nstrips, gridDim.x, flag)) return;
  // ---- last strip: the decision -------------------------------------------------------------
  decide_body(A, stv, lds);
}

// k_reduce — adds the row-tile partials in tile order and writes this shard's block of the
// gathered vector pair: ab_block = [a (W) | b (W)]. One thread per output element; the
// loads of 8 tiles are issued before they are summed (the partials sit in L2 / MALL).
__global__ __launch_bounds__(256) void k_reduce(const double* __restrict__ part, int ntiles,
                                                 int64_t ld, double* __restrict__ ab_block,
                                                 const SolverState* __restrict__ st) {
  if (st != nullptr && st->done) return;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;  // in [0, 2*ld)
  if (e >= 2 * ld) return;
  const int64_t which = e / ld, c = e - which * ld;
  const double* p = part + which * ld + c;
  const int64_t tstride = 2 * ld;
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
  ab_block[e] = acc;  // e = which*ld + c: exactly the [a | b] block layout
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
