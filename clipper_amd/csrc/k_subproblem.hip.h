// k_subproblem.hip.h — the LIVE SUB-PROBLEM: once the live rows are few AND the penalty is large, the columns
// outside a small set S cannot come back to life, and the solve continues on the associations of S alone.
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
//
// Why it is exact. A row view (k_solver.hip.h, LIVE ROWS) leaves out the ROWS whose candidates are zero; every
// pass still forms the gradient of every COLUMN, because the reference's next step max(u + alpha gradF, 0)
// (clipper.cpp:235-236) asks of a column c with u[c] = 0 whether gradF[c] > 0. For such a column
//     gradF[c] = -d sum(u) + sum_{r in nz(c)} (M[r,c] + d) u[r]                            (clipper.cpp:238-241)
//             <= -d sum(u) + (1 + d) sqrt(N_c) ||u||        (0 <= M <= 1 for the built-in invariants, Cauchy-Schwarz)
// with N_c = stored entries of column c among the rows where u is not zero. So with s = sum(x), z = ||x||^2 of an
// un-normalised candidate x:   d^2 s^2 > (1 + d)^2 N_c z   ==>   gradF[c] < 0 at x / ||x||,
// (N_c is counted over the rows of the view, where every live row lies when the solve is handed over; the rows of S
// outside the view may come back to life later — they are inside S — and add at most |S \ view| entries to a column:
// the host adds that to N.) Then
// column c stays at zero whatever the line search accepts, and NOTHING the reference computes from here on reads
// gradF[c] other than through that sign (Fnew = unew . gradFnew, the norms, the penalty sums :268-274 all carry
// unew[c] = 0). The penalty d grows by orders of magnitude from the second outer iteration on (m = 10k: 0.73 ->
// 1167; m = 30k: 0.79 -> 778 -> 3119) while the outliers' columns hold a tenth of the live rows: measured with the
// oracle's own loop, the inequality holds with N = 0.4 s^2 for every window of 306 of the 320 trials at m = 30k.
//
// What is built. When a row view that the resident solver does not take has just been built, the columns' entry
// counts among its rows are read off the view's slice headers (k_sub_colcount), S = view rows + every column with
// more than N0 = 0.4 sum(u)^2 entries (k_sub_flags; its list by k_rv_scan / k_rv_scatter), and the associations of
// S become a CLIPPER problem of their own: their points gathered (k_sub_gather_points), M[S,S] scored by the same
// symmetric fill, its own vectors. The decision (k_solver.hip.h: decide) then checks the inequality with
// N = the largest count outside S for every candidate of every window it plans:
//   * on the full problem, when it holds with a margin the solve goes on HOLD (hold = 2) and the host hands the
//     solve over: a decide-only iteration leaves the pending pass prepared, k_sub_enter gathers the point and the
//     state, and the same launches (k_gemv_slices, k_tail) continue on the sub-problem's arrays;
//   * on the sub-problem, when it fails the pending pass is left prepared and the solve goes on hold (hold = 3):
//     k_sub_leave scatters the point back (u = 0, gradF = -1 — any negative number — outside S) and the launches
//     on the full problem carry on from the prepared pass, on the row view.
// A sum over S in blocks of 256 associates differently from the sum over all m elements with zeros in between:
// the same situation as between any two work splits of this solver. The selected set, ifinal and the objective
// are the oracle's (tests/test_gpu_subproblem.py).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_rowview.hip.h"
#include "k_slices.hip.h"
#include "k_solver.hip.h"

namespace clipper_hip {

constexpr double SUB_THETA = 0.4;          // N0 = SUB_THETA * sum(u)^2: columns with more entries among the view's rows join S
constexpr double SUB_ENTER_MARGIN = 1.10;  // the inequality with this factor on N: hand the solve over
constexpr double SUB_STAY_MARGIN = 1.01;   // ... and with this one: stay (any factor > 1 is exact; the gap is hysteresis)

// cnt[c] = 4 x the quads column c holds in the view's slices: at least its stored entries among the view's rows.
// One wave per column group, a lane per column, eight headers in flight.
__global__ __launch_bounds__(256) void k_sub_colcount(SliceView RV, int32_t* __restrict__ cnt, int64_t mp) {
  const int lane = threadIdx.x & 63;
  const int cg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cg >= RV.ncg) return;  // whole wave
  const uint64_t* pre = RV.Pre + static_cast<int64_t>(cg) * RV.nchunks;
  int tot = 0;
  for (int k0 = 0; k0 < RV.nchunks; k0 += 8) {
    uint8_t nq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = (k0 + j < RV.nchunks) ? k0 + j : RV.nchunks - 1;
      nq[j] = RV.data[16 * pre[k] + 16 + lane];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k0 + j < RV.nchunks) tot += nq[j];
  }
  const int64_t c = static_cast<int64_t>(cg) * SL_W + lane;
  if (c < mp) cnt[c] = 4 * tot;
}

// what the selection leaves behind for the host (mapped, pinned) and for k_sub_flags' successors
struct SubRecord {
  int32_t nS;      // associations of the sub-problem (written by k_rv_scan through its count_out)
  int32_t ncol;    // the largest count of a column outside S (k_sub_publish)
  int32_t n0;      // the threshold the selection used
  int32_t pad;
  uint32_t entries_lo, entries_hi;  // the counts of the columns in S, summed: how dense M[S,S] will be (k_sub_publish)
};

// flags[c] = c is in S: a row of the view, or a column with more than N0 entries among the view's rows;
// blk[b] = members of block b (the input of k_rv_scan / k_rv_scatter); rec->ncol = max count outside S
template <int V>
__global__ __launch_bounds__(256) void k_sub_flags(const SolverState* __restrict__ st, const int32_t* __restrict__ cnt,
                                                    const uint8_t* __restrict__ in_view, int64_t m, int64_t mp,
                                                    uint8_t* __restrict__ flags, uint32_t* __restrict__ blk,
                                                    int32_t* __restrict__ acc /* device: [1] = max count outside S, [2] = N0,
                                                                                 [4..5] = the counts of the columns IN S, summed (u64) */) {
  __shared__ uint32_t wsum[4];
  __shared__ int wmax[4];
  __shared__ unsigned long long wcnt[4];
  unsigned long long insum = 0;
  const double s = st->s;  // sum(u) of the current (normalised) point
  const double n0d = floor(SUB_THETA * s * s);
  const int n0 = n0d < 2.0e9 ? static_cast<int>(n0d) : 2000000000;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * RV_BLK + threadIdx.x * 4;
  uint32_t n = 0;
  int mx = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k;
    bool in = false;
    if (i < m) {
      const int c = cnt[i];
      in = in_view[i] != 0 || c > n0;
      if (!in) mx = c > mx ? c : mx;
      else insum += static_cast<unsigned long long>(c);
    }
    if (i < mp) flags[i] = in ? 1 : 0;
    n += in ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o);
    const int t = __shfl_xor(mx, o);
    mx = t > mx ? t : mx;
    insum += __shfl_xor(insum, o);
  }
  if ((threadIdx.x & 63) == 0) {
    wsum[threadIdx.x >> 6] = n;
    wmax[threadIdx.x >> 6] = mx;
    wcnt[threadIdx.x >> 6] = insum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int b = wmax[0];
    for (int w = 1; w < 4; ++w) b = wmax[w] > b ? wmax[w] : b;
    atomicMax(&acc[1], b);  // (zeroed by k_sub_begin)
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + 4), wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
    if (blockIdx.x == 0) acc[2] = n0;
  }
}
// the counters of a selection, zeroed in front of it / handed to the host (mapped memory) behind it
__global__ void k_sub_begin(int32_t* acc) {
  if (threadIdx.x < 8) acc[threadIdx.x] = 0;
}
__global__ void k_sub_publish(const int32_t* __restrict__ acc, SubRecord* __restrict__ rec) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(&rec->ncol, acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&rec->entries_lo, static_cast<uint32_t>(acc[4]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&rec->entries_hi, static_cast<uint32_t>(acc[5]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&rec->n0, acc[2], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the sub-problem's point tables: row i = association colmap[i] of the full problem's (k_gather_points' layout:
// [d][pstride], padded with zeros; the fp32 copies for the prefilter; the association pairs for the
// distinctness test clipper.cpp:35-38)
__global__ __launch_bounds__(256) void k_sub_gather_points(const double* __restrict__ P, const float* __restrict__ Pf,
                                                            int d, int64_t pstride, const int32_t* __restrict__ A,
                                                            int64_t m, const int32_t* __restrict__ colmap, int64_t nS,
                                                            int64_t qstride, double* __restrict__ Q,
                                                            float* __restrict__ Qf, int32_t* __restrict__ B,
                                                            int which) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= qstride) return;
  const int64_t c = (i < nS) ? colmap[i] : 0;
  for (int k = 0; k < d; ++k) {
    Q[k * qstride + i] = (i < nS) ? P[k * pstride + c] : 0.0;
    Qf[k * qstride + i] = (i < nS) ? Pf[k * pstride + c] : 0.f;
  }
  if (i < nS) B[which * nS + i] = A[which * m + c];
}

// The hand-over: the full problem's state (a prepared pass: SolverState::resume) and its current point (u, gradF)
// and (a, b) at the associations of S become the sub-problem's. Grid over the sub-problem's padded length.
// `ctab` (a sub-problem kept as a DENSE store: its pass reads the pending window from a candidate table, k_gemv.hip.h):
// table `sel` of the set the first launch reads — row i = max(u + alpha beta^l g, 0), l < V, the tail's own expression
// and chain of multiplications (k_tail) —, written when the prepared pass is a window.
__global__ __launch_bounds__(256) void k_sub_enter(const SolverState* __restrict__ pst, const double* __restrict__ ppt,
                                                    const double* __restrict__ pcab, int64_t pmp, int V,
                                                    const int32_t* __restrict__ colmap, int64_t nS,
                                                    SolverState* __restrict__ cst, SolveShared* __restrict__ cshared,
                                                    double* __restrict__ cpt, double* __restrict__ ccab, int64_t cmp,
                                                    double* __restrict__ ctab, double beta) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t slot = static_cast<int64_t>(pst->ubp) * V + pst->ubv;
  if (i < cmp) {
    const bool in = i < nS;
    const int64_t c = in ? colmap[i] : 0;
    const double ui = in ? ppt[(slot * 2 + 0) * pmp + c] : 0.0;
    const double gi = in ? ppt[(slot * 2 + 1) * pmp + c] : 0.0;
    cpt[(slot * 2 + 0) * cmp + i] = ui;
    cpt[(slot * 2 + 1) * cmp + i] = gi;
    ccab[i] = in ? pcab[c] : 0.0;
    ccab[cmp + i] = in ? pcab[pmp + c] : 0.0;
    if (ctab != nullptr && pst->resume == 1) {
      double row[VS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      double al = pst->alpha;
      for (int l = 0; l < V; ++l) {
        const double t = ui + al * gi;  // clipper.cpp:235-236
        row[l] = (t > 0.0) ? t : 0.0;
        al = al * beta;
      }
      store_row(ctab + (static_cast<int64_t>(pst->sel) * cmp + i) * VS, row);
    }
  }
  if (blockIdx.x == 0) {
    copy_state(cst, pst, threadIdx.x, 256);
    __syncthreads();
    if (threadIdx.x == 0) {
      cst->hold = 0;
      cst->nout = 0;   // (no view inside the sub-problem)
      cst->view = 0;
      cshared->done = 0;
      cshared->hold = 0;
    }
  }
}

// The way back. Grid over the full problem's padded length: the point slot the sub-problem ended on and (a, b),
// at S from the sub-problem, elsewhere u = 0, gradF = -1 (the bound has held for every pass: negative), a = b = 0;
// the live rows of S outside the ROW view are counted (nout: whether the prepared pass may stream the view).
__global__ __launch_bounds__(256) void k_sub_leave(const SolverState* __restrict__ cst, const double* __restrict__ cpt,
                                                    const double* __restrict__ ccab, int64_t cmp, int V,
                                                    const int32_t* __restrict__ pos, const uint8_t* __restrict__ in_view,
                                                    int64_t m, double* __restrict__ ppt, double* __restrict__ pcab,
                                                    int64_t pmp, int32_t* __restrict__ nout_acc) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t slot = static_cast<int64_t>(cst->ubp) * V + cst->ubv;
  int out = 0;
  if (c < pmp) {
    const int64_t i = (c < m) ? pos[c] : -1;
    double u = 0.0, g = (c < m) ? -1.0 : 0.0, a = 0.0, b = 0.0;
    if (i >= 0) {
      u = cpt[(slot * 2 + 0) * cmp + i];
      g = cpt[(slot * 2 + 1) * cmp + i];
      a = ccab[i];
      b = ccab[cmp + i];
      out = ((u > 0.0 || g > 0.0) && in_view[c] == 0) ? 1 : 0;
    }
    ppt[(slot * 2 + 0) * pmp + c] = u;
    ppt[(slot * 2 + 1) * pmp + c] = g;
    pcab[c] = a;
    pcab[pmp + c] = b;
  }
  const unsigned long long any = __ballot(out != 0);
  if ((threadIdx.x & 63) == 0 && any != 0) atomicAdd(nout_acc, static_cast<int32_t>(__popcll(any)));
}
// ... and the state (one workgroup, after k_sub_leave): the sub-problem's, with the hold lifted and the live rows
// outside the row view as just counted
__global__ __launch_bounds__(256) void k_sub_leave_state(const SolverState* __restrict__ cst, SolverState* __restrict__ pst,
                                                          SolveShared* __restrict__ pshared, int32_t* __restrict__ nout_acc) {
  copy_state(pst, cst, threadIdx.x, 256);
  __syncthreads();
  if (threadIdx.x == 0) {
    pst->hold = 0;
    pst->nout = *nout_acc;
    pst->view = 0;
    pshared->hold = 0;
    *nout_acc = 0;
  }
}

// lifts a hold that asked for the hand-over (hold = 2) without touching the view policy's counters
__global__ void k_sub_resume(SolverState* st, SolveShared* shared) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    shared->hold = 0;
    st->hold = 0;
  }
}

}  // namespace clipper_hip
