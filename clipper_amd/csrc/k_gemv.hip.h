// k_gemv.hip.h — the dense pass over M: k_gemv, k_gemv_plain, k_reduce_pass (column shards), k_reduce, k_spread
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_solver.hip.h"

namespace clipper_hip {

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

constexpr int GEMV_LDS_DOUBLES(int NW) { return NW * GEMV_SLR * 256 + 2; }

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

// k_reduce — adds the row-tile partials in tile order and writes this shard's block of the
// gathered layout (matvec API only: the solver uses k_reduce_pass or folds it into k_tail). One thread per
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

// k_reduce_pass — column shards, between the pass and the exchange: the row-tile partials of the
// pass THIS iteration ran (none: nothing to do) added in tile order into this shard's block of
// the gathered layout ab[P][NSLOT][W]. A launch of its own: folding it into the pass (the last
// workgroup of a strip, found with an arrival counter and agent-scope release/acquire fences)
// cost 16 us per pass — every workgroup's release writes the L2 back.
__global__ __launch_bounds__(256) void k_reduce_pass(SolveArgs A, int nslots) {
  if (A.shared->done) return;
  if (A.st_next->n_passes == A.st_cur->n_passes) return;  // no pass in this iteration
  const int64_t ld = A.W;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t tstride = static_cast<int64_t>(nslots) * ld;
  if (e >= tstride) return;
  const double* p = A.part + e;
  const int ntiles = A.st_next->view ? A.rv_nslots : A.ntiles;  // a pass on the row view wrote its own (fewer) slots
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
  A.ab[static_cast<int64_t>(A.slot) * tstride + e] = acc;
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

}  // namespace clipper_hip
