// k_slices.hip.h — the compressed storage of M that the solver's passes stream: the layout
// ("slices"), the pass (slice_core, k_gemv_slices), the packers (k_slice_count, k_slice_scan*,
// k_slice_pack), k_slice_expand and k_slice_gather_sub.
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_solver.hip.h"

namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// Slices: the un-padded column lists of M (CLIPPER_HIP_STORE_F32_CSC / _F64_CSC)
// ------------------------------------------------------------------------------------------
// The reference keeps M_ as Eigen::SparseMatrix<double> (include/clipper/types.h:15) and every
// product of findDenseClique walks its stored entries only (src/clipper.cpp:194-271). Here the
// stored entries of BOTH triangles (zero diagonal) are kept as per-column lists, cut into
//   slice (cg, k) = columns [64 cg, 64 cg + 64) x rows [R k, R k + R),  R = 128 H  (a "chunk")
// — one column per lane of a wave, rows addressed by one byte inside a 128-row sub-block (the
// height of a tile of k_affinity_sym, which therefore writes finished slices itself). A
// lane's list in a slice is the concatenation of its H sub-block lists, each rounded up to
// whole QUADS of 4 entries (padding: value 0, row 0 — adds exact zeros). Nothing is padded to a
// neighbour's length: step q of a slice stores the quads of only those lanes that still have a
// q-th quad ("compact ELL"), lane-ordered, so the wave's load of a step is one contiguous run:
//
//   slice := head[16 B] = {u32 nquads, u32 maxq, u32 bytes, u32 0}
//            nq[H][64]  u8   quads of lane l in sub-block h
//            so[ceil(maxq / 16)] u32, padded to 16 B: byte offset (from the slice's start) of
//                            steps 0, 16, 32, ... — where a workgroup that takes only a range
//                            of the steps of a heavy slice starts
//            step 0, step 1, ... step maxq-1
//   step q := the value quads (4 x VT each) of the active lanes (q < tot_l = sum_h nq[h][l]),
//            in lane order, then their row quads (4 x u8 each), the step padded to 16 bytes.
//
// A lane finds its quad of step q at  stepbase + rank * sizeof(quad)  with rank = number of
// active lanes below it (ballot + mbcnt), and stepbase advances by wave-uniform arithmetic on
// the ballot's population count. Bytes held: (4 sizeof(VT) + 4) per quad + ~100 per slice:
// 5.35 B per stored entry at the headline problem (m = 10k, 11 % dense) against the 12.3 B of
// column lists padded to the longest of 128 neighbours that this replaces (round 1). Lanes whose
// list is shorter than the slice's longest idle inside a slice (issue slots), they cost no bytes.
//
// Where a slice lies in the arena is arbitrary (the fill kernel claims the space of a slice with
// one atomic; the packers of the other sources lay them out by a scan): its CONTENT is a pure
// function of the matrix, and with it every sum. Directory:
//   Pre[cg * nchunks + k]  byte offset / 16 of the slice
//   Lq [cg * nchunks + k]  maxq | entries << 8 (the cost of the slice in lock-step steps and
//                          its stored entries; planning and reporting only)
//   work[wg]               what workgroup wg does, most expensive first (SliceWork)
constexpr int SL_W = 64;      // columns per slice
#ifndef CLIPPER_SL_SUB
#define CLIPPER_SL_SUB 128
#endif
constexpr int SL_SUB = CLIPPER_SL_SUB;   // rows per sub-block (<= 256: one row byte)
constexpr int SL_SO = 16;     // steps between two recorded step offsets
constexpr int SL_TAILPAD = 4096;  // bytes behind the last slice a load front may touch

// One workgroup = NW adjacent column groups ("strip", one per wave) x chunks [t0, t1) — or, for
// the slices of a dense block (the inliers of a registration problem: a chain of dependent
// steps ten times as long as the average), ONE chunk and the step range [q0, q1) of it. Its
// partial sums go to part[slot][.][columns of the strip]; every (strip, slot < nslots) is
// written by exactly one workgroup (empty ones write zeros), the tail adds the slots in order.
struct SliceWork {
  int strip, slot, t0, t1;
  int q0, q1, pad0, pad1;
};

struct SliceView {
  const uint8_t* data;
  const uint64_t* Pre;
  const SliceWork* work;
  int nchunks;  // chunks of R rows
  int ncg;      // column groups
  int nwork;    // workgroups of a pass (entries of `work`)
  // A ROW VIEW (k_solver.hip.h, LIVE ROWS): the slices of M[rows, :] for a row list — row r' of the
  // view is row rowmap[r'] of M, the x rows of a chunk are gathered through the map when they are
  // staged. The matrix itself: rowmap == null, nrows == m.
  const int32_t* rowmap;
  int64_t nrows;
  int64_t pad;  // (64 bytes: the row view's descriptor is copied to the device in 16-byte words)
};
static_assert(sizeof(SliceView) == 64, "SliceView is copied in 16-byte words");

// The same descriptor as the device code uses it: every pointer typed as a GLOBAL-memory pointer. A
// descriptor that a kernel LOADS (the row view's) carries generic pointers as far as the compiler
// can tell, and every load of the streaming loop through one becomes a flat load, which waits on the
// LDS counter too: +35 % per pass at every size, measured (profiles/r03_variants.txt). The type says
// what inference cannot.
#define CLIPPER_GLOBAL __attribute__((address_space(1)))
typedef const CLIPPER_GLOBAL uint8_t* gbytes_t;
struct SliceViewG {
  gbytes_t data;
  const CLIPPER_GLOBAL uint64_t* Pre;
  const CLIPPER_GLOBAL SliceWork* work;
  int nchunks, ncg, nwork;
  const CLIPPER_GLOBAL int32_t* rowmap;
  int64_t nrows;
};
__device__ __forceinline__ SliceViewG to_global(const SliceView& M) {
  SliceViewG G;
  G.data = (gbytes_t)M.data;
  G.Pre = (const CLIPPER_GLOBAL uint64_t*)M.Pre;
  G.work = (const CLIPPER_GLOBAL SliceWork*)M.work;
  G.nchunks = M.nchunks;
  G.ncg = M.ncg;
  G.nwork = M.nwork;
  G.rowmap = (const CLIPPER_GLOBAL int32_t*)M.rowmap;
  G.nrows = M.nrows;
  return G;
}

// Window mode stages candidates 0 .. sl_xload(V)-1 of a table row at a pitch of sl_xpitch(V)
// doubles: an ODD number of 16-byte units (1, 3, 5), so that the rows of a sub-block spread over
// all 16 slots a ds_read_b128 lane group can serve in one LDS cycle (pitch 32 B would use 8
// of them, 64 B only 4).
// (Forming the candidates in registers per entry, and the linear window, were measured and not adopted: the
// harness keeps those variants — tools/slice_xmode.hip.h, DESIGN.md section 7 "tried".)
constexpr int sl_xload(int V) { return V <= 2 ? 2 : (V <= 4 ? 4 : (V <= 6 ? 6 : 8)); }
constexpr int sl_xpitch(int V) { return V <= 2 ? 2 : (V <= 6 ? 6 : 10); }
constexpr int sl_lds_doubles(int V, int H, int NW) {
  const int a = 2 * SL_SUB * H * sl_xpitch(V);  // two x buffers
  const int b = NW * 64 + NW * 2 * V + 8;       // the decision's scratch
  return a > b ? a : b;
}
__host__ __device__ constexpr int sl_so_bytes(int maxq) {
  return ((maxq + SL_SO - 1) / SL_SO * 4 + 15) & ~15;
}

template <typename VT>
struct SliceQuad;
template <>
struct SliceQuad<float> {
  float v[4];
  __device__ __forceinline__ void load(const uint8_t* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void load(const __attribute__((address_space(1))) uint8_t* p) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 t = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(uint8_t* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct SliceQuad<double> {
  double v[4];
  __device__ __forceinline__ void load(const uint8_t* p) {
    const double2 t0 = reinterpret_cast<const double2*>(p)[0];
    const double2 t1 = reinterpret_cast<const double2*>(p)[1];
    v[0] = t0.x; v[1] = t0.y; v[2] = t1.x; v[3] = t1.y;
  }
  __device__ __forceinline__ void load(const __attribute__((address_space(1))) uint8_t* p) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const f64x2 t0 = reinterpret_cast<const __attribute__((address_space(1))) f64x2*>(p)[0];
    const f64x2 t1 = reinterpret_cast<const __attribute__((address_space(1))) f64x2*>(p)[1];
    v[0] = t0.x; v[1] = t0.y; v[2] = t1.x; v[3] = t1.y;
  }
  __device__ __forceinline__ void store(uint8_t* p) const {
    reinterpret_cast<double2*>(p)[0] = make_double2(v[0], v[1]);
    reinterpret_cast<double2*>(p)[1] = make_double2(v[2], v[3]);
  }
};

__device__ __forceinline__ uint32_t sl_lane_rank(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Stage the x rows of chunk k ([R][XP] doubles, row pitch XP). Window mode: the candidates are
// BUILT here — row r, candidate l = max(u'[r] + alpha0 beta^l g'[r], 0) (clipper.cpp:235-236) from
// the point slot the window starts from, 16 bytes read per row instead of a 64-byte table row that
// the tail would have had to write for every outcome it speculates on. The expression, the chain of
// multiplications behind alpha0 beta^l and hence the bits are those of the tail (k_solver.hip.h).
// Pair mode: X[r * xstride]. Pieces of 16 bytes (two candidates), thread-linear.
struct WindowSource {
  const double* U;   // u' [mp]
  const double* G;   // g' [mp]
  double alpha0, beta;
};
template <bool WINDOW, int XL, int XP, int R, int NT>
struct SliceXStage {
  static constexpr int PIECES = R;  // one row per piece (window: all its candidates; pair: its x)
  static constexpr int PER = (PIECES + NT - 1) / NT;
  double wu[WINDOW ? PER : 1], wg[WINDOW ? PER : 1];
  double s[WINDOW ? 1 : PER];
  int32_t ridx[PER];  // rows of M the pieces of the chunk AFTER the one being loaded stand for
  // A row view gathers its x rows through the row list: the list entries of a chunk are requested
  // one chunk ahead of the values (index(k + 2) beside values(k + 1)), so that no load of the
  // prefetch waits for another one in front of the streaming loop.
  __device__ __forceinline__ void index(int64_t r0, int64_t nrows, const CLIPPER_GLOBAL int32_t* rowmap) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = threadIdx.x + i * NT;
      const int64_t rv = r0 + p;
      const bool in = p < PIECES && rv < nrows;
      ridx[i] = (in && rowmap != nullptr) ? rowmap[rv] : static_cast<int32_t>(rv);  // (uniform branch: one view per launch)
    }
  }
  __device__ __forceinline__ void load(const WindowSource& W, const double* __restrict__ X, int xstride,
                                       int64_t r0, int64_t nrows) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = threadIdx.x + i * NT;
      const int64_t rv = r0 + p;
      const bool in = p < PIECES && rv < nrows;
      const int64_t r = ridx[i];
      if constexpr (WINDOW) {
        wu[i] = in ? W.U[r] : 0.0;
        wg[i] = in ? W.G[r] : 0.0;
      } else {
        s[i] = in ? X[r * xstride] : 0.0;
      }
    }
  }
  __device__ __forceinline__ void store(const WindowSource& W, double* xs) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = threadIdx.x + i * NT;
      if constexpr (WINDOW) {
        double al = W.alpha0;
#pragma unroll
        for (int l = 0; l < XL; l += 2) {
          double t0 = wu[i] + al * wg[i];
          t0 = (t0 > 0.0) ? t0 : 0.0;
          al = al * W.beta;
          double t1 = wu[i] + al * wg[i];
          t1 = (t1 > 0.0) ? t1 : 0.0;
          al = al * W.beta;
          if (p < PIECES) *reinterpret_cast<double2*>(xs + p * XP + l) = make_double2(t0, t1);
        }
      } else {
        if (p < PIECES) xs[p] = s[i];
      }
    }
  }
};

// what a wave needs of a slice before it can address the steps
template <int H>
struct SliceHead {
  gbytes_t sp;
  int maxq;
  int nq[H];
  __device__ __forceinline__ void load(gbytes_t p, int lane) {
    sp = p;
    maxq = static_cast<int>(reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(p)[1]);
#pragma unroll
    for (int h = 0; h < H; ++h) nq[h] = p[16 + h * 64 + lane];
  }
};

// slice_begin() is everything of a workgroup's job that does not depend on the decision at the
// head of the launch — issued before it, so that the decision hides its latency
template <int H, int NW>
struct SliceJob {
  int strip, slot, cg, t0, t1, q0, q1;
  int live;  // the workgroup has an item of this list (its partial sums are written)
  SliceHead<H> first;
  uint64_t pre1;  // Pre of the slice after the first one (its header is requested one chunk ahead)
};

template <int H, int NW>
__device__ __forceinline__ void slice_begin(const SliceViewG& M, SliceJob<H, NW>& J, int item) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (a struct cannot be copied out of an address space: the work item as its six ints)
  const CLIPPER_GLOBAL int* w = reinterpret_cast<const CLIPPER_GLOBAL int*>(M.work + (item < M.nwork ? item : 0));
  J.strip = w[0];
  J.slot = w[1];
  J.cg = J.strip * NW + wave;
  J.t0 = w[2];
  J.t1 = w[3];
  J.q0 = w[4];
  J.q1 = w[5];
  J.live = item < M.nwork ? 1 : 0;
  J.first.maxq = 0;
  J.first.sp = M.data;
  J.pre1 = 0;
#pragma unroll
  for (int h = 0; h < H; ++h) J.first.nq[h] = 0;
  if (item >= M.nwork) {  // no item of this list for this workgroup
    J.strip = J.slot = J.t0 = J.t1 = J.q0 = J.q1 = 0;
    J.cg = M.ncg;
    J.first.maxq = 0;
    J.first.sp = M.data;
    J.pre1 = 0;
#pragma unroll
    for (int h = 0; h < H; ++h) J.first.nq[h] = 0;
    return;
  }
  if (J.cg < M.ncg && J.t0 < J.t1) {
    const CLIPPER_GLOBAL uint64_t* pre = M.Pre + static_cast<int64_t>(J.cg) * M.nchunks + J.t0;
    J.first.load(M.data + 16 * pre[0], threadIdx.x & 63);
    if (J.t0 + 1 < J.t1) J.pre1 = pre[1];
  }
}

// The streaming part of a pass on the slices: this workgroup's partial sums ->
// part[slot][.][ld]. Wave w of the workgroup owns column group strip * NW + w (a lane = a
// column: no cross-lane or cross-wave combine), the x rows of a chunk are staged once per
// workgroup (double buffered). WINDOW / pair mode and the slots as in gemv_core (k_gemv.hip.h).
// D = steps of a slice kept in flight per lane.
template <typename VT, int H, bool WINDOW, int V, int NSLOT, int NW, int D>
__device__ __forceinline__ void slice_core(const SliceViewG& M, const SliceJob<H, NW>& J, int64_t ld,
                                           int64_t m, double d, const WindowSource& WS,
                                           const double* __restrict__ X, int xstride,
                                           double* __restrict__ part, double* lds) {
  constexpr int NS = WINDOW ? V + 1 : 2;
  constexpr int XP = WINDOW ? sl_xpitch(V) : 1;
  constexpr int XL = WINDOW ? sl_xload(V) : 1;
  constexpr int R = SL_SUB * H;
  constexpr int NT = NW * 64;
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));  // bytes of a value quad
  const int lane = threadIdx.x & 63;
  const int cg = J.cg, t0 = J.t0, t1 = J.t1;
  const bool mine = cg < M.ncg;

  double acc[NS];
#pragma unroll
  for (int v = 0; v < NS; ++v) acc[v] = 0.0;

  SliceXStage<WINDOW, XL, XP, R, NT> xst;
  __syncthreads();  // the decision at the head of the launch used the same LDS
  if (t0 < t1) {
    xst.index(static_cast<int64_t>(t0) * R, M.nrows, M.rowmap);
    xst.load(WS, X, xstride, static_cast<int64_t>(t0) * R, M.nrows);
    xst.store(WS, lds);
    xst.index(static_cast<int64_t>(t0 + 1) * R, M.nrows, M.rowmap);
  }
  __syncthreads();
  SliceHead<H> cur = J.first;
  uint64_t pre_next = J.pre1;  // Pre of slice k + 1, requested a chunk ago
  const CLIPPER_GLOBAL uint64_t* pre_row = M.Pre + static_cast<int64_t>(mine ? cg : 0) * M.nchunks;
  for (int k = t0; k < t1; ++k) {
    const double* xs = lds + ((k - t0) & 1) * (R * XP);
    double* xnext = lds + (((k - t0) & 1) ^ 1) * (R * XP);
    const bool more = k + 1 < t1;
    SliceHead<H> nxt = cur;
    uint64_t pre_next2 = 0;
    if (more) {
      xst.load(WS, X, xstride, static_cast<int64_t>(k + 1) * R, M.nrows);
      xst.index(static_cast<int64_t>(k + 2) * R, M.nrows, M.rowmap);
      if (mine) nxt.load(M.data + 16 * pre_next, lane);
      if (k + 2 < t1) pre_next2 = pre_row[k + 2];
    }
    if (mine) {
      const int maxq = __builtin_amdgcn_readfirstlane(cur.maxq);
      const int qend = maxq < J.q1 ? maxq : J.q1;
      int tot = 0;
#pragma unroll
      for (int h = 0; h < H; ++h) tot += cur.nq[h];
      // load front (wave-uniform): step q0 of the slice
      gbytes_t fbase = cur.sp + 16 + H * 64 + sl_so_bytes(maxq);
      if (J.q0 > 0 && J.q0 < maxq)
        fbase = cur.sp + reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(cur.sp + 16 + H * 64)[J.q0 / SL_SO];
      SliceQuad<VT> mv[D];
      uint32_t rw[D];
      // Every lane issues every load of every step (an idle lane re-reads the step's first quad,
      // a step past the end the bytes behind the slice): the number of loads in flight is then
      // the same on every path, and the compiler's s_waitcnt bookkeeping keeps D steps in
      // flight instead of draining the queue at every divergent join.
      auto issue = [&](int q, SliceQuad<VT>& vq, uint32_t& rq) {
        const bool active = q < tot && q < qend;
        const uint64_t mask = __ballot(active);
        const int cnt = __builtin_amdgcn_readfirstlane(__popcll(mask));
        const uint32_t rank = active ? sl_lane_rank(mask) : 0u;
        vq.load(fbase + rank * QB);
        rq = *reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(fbase + cnt * QB + rank * 4);
        fbase += cnt * QB + ((cnt * 4 + 15) & ~15);
      };
#pragma unroll
      for (int j = 0; j < D; ++j) issue(J.q0 + j, mv[j], rw[j]);
      for (int qb = J.q0; qb < qend; qb += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const int q = qb + j;
          if (q < tot && q < qend) {
            int rowbase = 0;
            if constexpr (H > 1) {
              int edge = cur.nq[0];
#pragma unroll
              for (int h = 1; h < H; ++h) {
                rowbase = (q >= edge) ? h * SL_SUB : rowbase;
                edge += cur.nq[h];
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const VT mf = mv[j].v[e];
              const double mm = static_cast<double>(mf);
              const double ii = mf != VT(0) ? 1.0 : 0.0;
              const uint32_t row = rowbase + ((rw[j] >> (8 * e)) & 255u);
              if constexpr (WINDOW) {
                const double* xr = xs + row * XP;
                double xv[XP];
#pragma unroll
                for (int v = 0; v < XP; v += 2) {
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
          issue(q + D, mv[j], rw[j]);
        }
      }
    }
    if (more) xst.store(WS, xnext);
    cur = nxt;
    pre_next = pre_next2;
    __syncthreads();
  }

  const int64_t c = static_cast<int64_t>(cg) * SL_W + lane;
  if (c < ld && J.live != 0) {
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      const int slot = (v == NS - 1) ? NSLOT - 1 : v;
      // write-through (sc1): the partial sums leave the XCD's L2 while the launch still runs,
      // instead of as ~14 MB of dirty lines the kernel boundary has to write back before the tail
      __hip_atomic_store(&part[(static_cast<int64_t>(J.slot) * NSLOT + slot) * ld + c], acc[v],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// window or pair mode by the plan of this iteration
template <typename VT, int H, int V, int NW, int D>
__device__ __forceinline__ void slices_by_plan(const SliceViewG& M, const SliceJob<H, NW>& J,
                                               const SolveArgs& A, const PassPlan& plan,
                                               double* lds) {
  if (plan.phase == PH_TRIAL) {
    const WindowSource WS{A.pt + static_cast<int64_t>(plan.src) * 2 * A.mp,
                          A.pt + (static_cast<int64_t>(plan.src) * 2 + 1) * A.mp, plan.alpha0, A.prm.beta};
    // (SolverState::weff) candidate 0 alone — a and b apart, the same two fmas per entry in the same order as in the
    // full window: the same bits — or the whole window
    if (V > 1 && plan.weff == 1) slice_core<VT, H, true, 1, nslot(V), NW, D>(M, J, A.W, A.m, plan.d, WS, nullptr, 0, A.part, lds);
    else slice_core<VT, H, true, V, nslot(V), NW, D>(M, J, A.W, A.m, plan.d, WS, nullptr, 0, A.part, lds);
  } else {  // pair mode: straight on the u array of a point slot, or on candidate 0 of a table
    const bool fu = plan.from_u >= 0;
    const double* X = fu ? A.pt + static_cast<int64_t>(plan.from_u) * 2 * A.mp
                         : A.Xin + static_cast<int64_t>(plan.sel) * A.mp * VS;
    slice_core<VT, H, false, V, nslot(V), NW, D>(M, J, A.W, A.m, 0.0, WindowSource{}, X, fu ? 1 : VS,
                                                 A.part, lds);
  }
}

#ifndef CLIPPER_SL_NW
#define CLIPPER_SL_NW 4
#endif
constexpr int SL_NW = CLIPPER_SL_NW;  // waves (= column groups) per workgroup
#ifndef CLIPPER_SL_D
#define CLIPPER_SL_D 3
#endif
constexpr int SL_D = CLIPPER_SL_D;   // steps in flight per lane (2 / 3 / 4 / 6 / 8 measured: profiles/r02e_window_sweep.txt)
#ifndef CLIPPER_SL_OCC
#define CLIPPER_SL_OCC 6
#endif
constexpr int SL_OCC = CLIPPER_SL_OCC;  // waves per SIMD the pass kernel is compiled for (= workgroups per CU)

// G of a solver iteration on the slices (one shard): decision, then the pass
// `RV`: the descriptor of the row view of M, in device memory (read only by a launch whose decision
// chose the view: PassPlan::view — as a second by-value argument it cost 130 registers spilled to
// scratch, some of them inside the streaming loop). The grid covers the larger of the two work lists: one
// workgroup per item, also where the list is 3 - 8 times the workgroups the chip holds (m >= 30k). A PERSISTENT
// grid for those lists — each workgroup decides once and claims further items with an atomic — was measured in
// round 4 and is 4 - 9 % slower: with six workgroups per CU a new workgroup's decision runs while the others
// stream, and the dispatcher hands out the items in the same greedy order (profiles/r04_persistent_grid.txt).
template <typename VT, int H, int V>
__global__ __launch_bounds__(SL_NW * 64, SL_OCC) void k_gemv_slices(SliceView M, const SliceView* __restrict__ RV, SolveArgs A) {
  __shared__ __attribute__((aligned(16))) double lds[sl_lds_doubles(V, H, SL_NW)];
  __shared__ __attribute__((aligned(16))) SolverState stash;
  const long long c0 = A.stamps ? wall_clock64() : 0;
  SliceViewG G = to_global(M);
  SliceJob<H, SL_NW> J, JV;
  slice_begin<H, SL_NW>(G, J, blockIdx.x);
  // Both jobs are requested ahead of the decision, which hides them: a pass on a small view is a chain
  // of latencies (work item -> directory -> header), not a stream. (Also requesting this thread's
  // row-list entry for the first chunk here tips the register allocation into scratch: not done.)
  SliceViewG GV = G;
  if (A.in_view != nullptr) {
    GV = to_global(*RV);
    slice_begin<H, SL_NW>(GV, JV, blockIdx.x);
  }
  PassPlan plan;
  if (!iteration_head<V, SL_NW * 64>(A, lds, &stash, plan)) return;
  const long long c1 = A.stamps ? wall_clock64() : 0;
  if (plan.view) {  // (one call site of the streaming loop for both)
    G = GV;
    J = JV;
  }
  slices_by_plan<VT, H, V, SL_NW, SL_D>(G, J, A, plan, lds);
  flush_state(A, &stash);
  // (wide: window passes only, so that what is left behind is the last WINDOW pass of the solve)
  if (A.stamps && threadIdx.x == 0 && blockIdx.x < (A.stamps_wide ? 16384u : 1536u) && (!A.stamps_wide || plan.phase == PH_TRIAL)) {
    A.stamps[blockIdx.x * 4 + 0] = c0;
    A.stamps[blockIdx.x * 4 + 1] = c1;
    A.stamps[blockIdx.x * 4 + 2] = wall_clock64();
    // chunks of the item | (wide) HW_ID[15:0] and the XCD the workgroup ran on | phase
    const unsigned where = A.stamps_wide ? ((__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu) << 12) |
                                               ((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu) << 8)
                                         : 0u;
    A.stamps[blockIdx.x * 4 + 3] = (static_cast<long long>(J.t1 - J.t0) << 32) | where | static_cast<unsigned>(plan.phase);
  }
}

// the pair-mode product alone on table X (matvec API, micro-benchmark): a -> slot 0, b -> slot 1
template <typename VT, int H>
__global__ __launch_bounds__(SL_NW * 64, 2) void k_gemv_slices_plain(SliceView M, int64_t ld,
                                                                      int64_t m,
                                                                      const double* __restrict__ X,
                                                                      double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) double lds[2 * SL_SUB * H];
  const SliceViewG G = to_global(M);
  SliceJob<H, SL_NW> J;
  slice_begin<H, SL_NW>(G, J, blockIdx.x);
  slice_core<VT, H, false, 1, 2, SL_NW, SL_D>(G, J, ld, m, 0.0, WindowSource{}, X, VS, part, lds);
}

// ------------------------------------------------------------------------------------------
// Packers: (source of column lists) -> slices, in three launches
//   k_slice_count   one wave per slice: its size in 16-byte units and its maxq
//   k_slice_scan*   exclusive scan of the sizes (column group major, chunk minor) -> Pre
//   k_slice_pack    one wave per slice: header, lengths, step offsets, steps
// A source hands every lane (= column) the sub-block lists of its slice: prepare_sub()
// positions it on one 256-row sub-block, count() = its entries, fetch(i) = entry i (rows
// ascending).
// ------------------------------------------------------------------------------------------

// Source 1: "groups" — what the fill kernels emit (k_csc.hip.h): per (128-column strip s,
// 64-row block b) group g = s * nblocks + b the lists of its columns back to back,
//   Goff[g * GR_OFFS + cl]  (u16) where column cl's list starts, relative to the group's start;
//                           Goff[g * GR_OFFS + 128] = the group's entry count
//   Gpre[g]                 the group's start in vals / rows, in units of 4 entries
constexpr int GR_CW = 128;    // columns per group
constexpr int GR_RB = 64;     // rows per group
constexpr int GR_OFFS = 130;  // u16 per group in Goff (129 used)

// arena cursors of a group build (k_csc.hip.h): the pack kernel only looks at `overflow`
constexpr int CSC_ARENAS = 64;
struct alignas(128) CscArena {
  unsigned long long cursor;    // units of 4 entries claimed so far in this arena
  unsigned long long capacity;  // units available to it
  unsigned long long origin;    // where the arena starts, same units
  int overflow;
};
typedef CscArena CscBuildCtl;  // [CSC_ARENAS]

template <typename VT>
struct GroupSource {
  const uint16_t* Goff;
  const uint64_t* Gpre;
  const VT* vals;
  const uint8_t* rows;
  int nblocks;   // 64-row blocks of the matrix
  int64_t ncols; // columns the groups cover (local)
  // per-lane state: the SL_SUB / 64 blocks of one sub-block at a time (unused ones: empty)
  int64_t start[4];
  int cnt[4];
  __device__ __forceinline__ void prepare_sub(int64_t c, int64_t r0) {
    const int64_t s = c / GR_CW;
    const int cl = static_cast<int>(c - s * GR_CW);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t b = r0 / GR_RB + j;
      start[j] = 0;
      cnt[j] = 0;
      if (j < SL_SUB / GR_RB && c < ncols && b < nblocks) {
        const int64_t g = s * nblocks + b;
        const uint32_t o0 = Goff[g * GR_OFFS + cl], o1 = Goff[g * GR_OFFS + cl + 1];
        start[j] = static_cast<int64_t>(Gpre[g]) * 4 + o0;
        cnt[j] = static_cast<int>(o1 - o0);
      }
    }
  }
  __device__ __forceinline__ int count() const { return cnt[0] + cnt[1] + cnt[2] + cnt[3]; }
  // entry i of the prepared sub-block: value and row byte (row inside the sub-block); i >=
  // count(): value 0, row 0. Branch-free, so that the loads of a quad are issued together.
  __device__ __forceinline__ void fetch(int i, VT& v, uint32_t& row) const {
    const int n = count();
    const bool valid = i < n;
    const int ii = valid ? i : 0;
    const int c0 = cnt[0], c01 = c0 + cnt[1], c012 = c01 + cnt[2];
    const int j = (ii >= c0 ? 1 : 0) + (ii >= c01 ? 1 : 0) + (ii >= c012 ? 1 : 0);
    const int before = j == 0 ? 0 : (j == 1 ? c0 : (j == 2 ? c01 : c012));
    const int64_t st = j == 0 ? start[0] : (j == 1 ? start[1] : (j == 2 ? start[2] : start[3]));
    const int64_t a = valid ? st + (ii - before) : 0;
    const VT lv = vals[a];
    const uint32_t lr = rows[a];
    v = valid ? lv : VT(0);
    row = valid ? static_cast<uint32_t>(j * GR_RB) + lr : 0u;
  }
};

// Source 2: full symmetric CSC (both triangles, rows ascending per column) of this shard's
// columns — clipper_hip_set_sparse (the reference's setSparseMatrixData, clipper.cpp:162-166)
template <typename VT>
struct CscSource {
  const int64_t* colptr;
  const int32_t* rowidx;
  const double* values;
  int64_t ncols;
  int64_t start;
  int cnt;
  int64_t r0_;
  __device__ __forceinline__ int64_t lower_bound(int64_t lo, int64_t hi, int64_t key) const {
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowidx[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  }
  __device__ __forceinline__ void prepare_sub(int64_t c, int64_t r0) {
    start = 0;
    cnt = 0;
    r0_ = r0;
    if (c < ncols) {
      const int64_t a = colptr[c], b = colptr[c + 1];
      start = lower_bound(a, b, r0);
      cnt = static_cast<int>(lower_bound(start, b, r0 + SL_SUB) - start);
    }
  }
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ void fetch(int i, VT& v, uint32_t& row) const {
    const bool valid = i < cnt;
    const int64_t a = valid ? start + i : 0;
    const double x = values[a];
    const int32_t r = rowidx[a];
    VT t = static_cast<VT>(x);
    if (t == VT(0) && x != 0.0) t = static_cast<VT>(1.17549435e-38);  // an underflow keeps the pattern
    v = valid ? t : VT(0);
    row = valid ? static_cast<uint32_t>(r - r0_) : 0u;
  }
};

// wave-level integer helpers (all 64 lanes active)
__device__ __forceinline__ int sl_wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o);
    v = v > t ? v : t;
  }
  return v;
}
__device__ __forceinline__ int sl_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// size (bytes) of the steps of a slice whose lanes have `tot` quads: wave-uniform
__device__ __forceinline__ uint32_t sl_steps_bytes(int tot, int maxq, int QB) {
  uint32_t bytes = 0;
  for (int q = 0; q < maxq; ++q) {
    const int cnt = __popcll(__ballot(q < tot));
    bytes += cnt * QB + ((cnt * 4 + 15) & ~15);
  }
  return bytes;
}

// also folds the 64 arenas' overflow marks of the group build into ONE word (`ovf`), which is
// all the pack kernels look at (a strided read of the 64 marks by each of their waves cost more
// than the packing)
template <typename VT, int H, typename Source>
__global__ __launch_bounds__(256) void k_slice_count(Source S, int ncg, int nchunks,
                                                      uint32_t* __restrict__ sizes,
                                                      uint32_t* __restrict__ Lq,
                                                      const CscBuildCtl* __restrict__ ctl,
                                                      uint64_t* __restrict__ ovf) {
  const int lane = threadIdx.x & 63;
  const int64_t s = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (s >= static_cast<int64_t>(ncg) * nchunks) return;  // whole wave
  if (s == 0) {
    const bool o = ctl != nullptr && __ballot(ctl[lane].overflow != 0) != 0;
    if (lane == 0) ovf[0] = o ? 1 : 0;
  }
  const int cg = static_cast<int>(s / nchunks), k = static_cast<int>(s - static_cast<int64_t>(cg) * nchunks);
  const int64_t c = static_cast<int64_t>(cg) * SL_W + lane;
  int tot = 0, ent = 0;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    S.prepare_sub(c, (static_cast<int64_t>(k) * H + h) * SL_SUB);
    ent += S.count();
    tot += (S.count() + 3) >> 2;
  }
  const int maxq = sl_wave_max(tot);
  const int entries = sl_wave_sum(ent);
  const uint32_t bytes = 16 + H * 64 + sl_so_bytes(maxq) + sl_steps_bytes(tot, maxq, 4 * sizeof(VT));
  if (lane == 0) {
    sizes[s] = bytes >> 4;
    Lq[s] = static_cast<uint32_t>(maxq) | (static_cast<uint32_t>(entries) << 8);  // maxq <= 64 H
  }
}

// exclusive scan of n u32 sizes -> u64 offsets, in two levels of 1024-element blocks
constexpr int SCAN_BLK = 1024;
__global__ __launch_bounds__(256) void k_slice_scan_local(const uint32_t* __restrict__ sizes,
                                                           int64_t n, uint64_t* __restrict__ Pre,
                                                           uint64_t* __restrict__ blocksum) {
  __shared__ uint64_t wsum[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * SCAN_BLK + threadIdx.x * 4;
  uint64_t v[4], run = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (base + j < n) ? sizes[base + j] : 0;
    run += v[j];
  }
  // inclusive scan of `run` across the wave, then across the 4 waves
  uint64_t inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t t = __shfl_up(inc, o);
    if ((threadIdx.x & 63) >= o) inc += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint64_t off = 0;
  for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
  uint64_t excl = off + inc - run;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) Pre[base + j] = excl;
    excl += v[j];
  }
  if (threadIdx.x == 255) blocksum[blockIdx.x] = off + inc;
}
// one workgroup: exclusive scan of the block sums in place; total -> blocksum[nb]
__global__ __launch_bounds__(256) void k_slice_scan_blocks(uint64_t* __restrict__ blocksum, int64_t nb) {
  __shared__ uint64_t carry;
  __shared__ uint64_t wsum[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const uint64_t v = (i < nb) ? blocksum[i] : 0;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t t = __shfl_up(inc, o);
      if ((threadIdx.x & 63) >= o) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint64_t off = carry;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < nb) blocksum[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) blocksum[nb] = carry;
}
__global__ __launch_bounds__(256) void k_slice_scan_add(uint64_t* __restrict__ Pre, int64_t n,
                                                         const uint64_t* __restrict__ blocksum) {
  const int64_t base = static_cast<int64_t>(blockIdx.x) * SCAN_BLK + threadIdx.x * 4;
  const uint64_t off = blocksum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) Pre[base + j] += off;
}

// One workgroup of SL_PACKW waves per slice: every wave knows where every step starts (the
// lanes' lengths: ALU only), wave w writes steps w, w + SL_PACKW, ... — the slices of a dense
// block are 64 dependent load->store steps long, one wave alone would be the build's critical
// path. Returns at once if there is nothing to pack into (slice arena too small) or from (the
// groups overflowed): the host grows the buffers and repeats.
constexpr int SL_PACKW = 8;
constexpr int SL_STAGE_CAP = 2048;  // entries of a slice k_slice_pack_staged stages in LDS
template <typename VT, int H, typename Source>
__global__ __launch_bounds__(SL_PACKW * 64) void k_slice_pack(Source S, int ncg, int nchunks,
                                                               const uint64_t* __restrict__ Pre,
                                                               uint8_t* __restrict__ data,
                                                               const uint64_t* __restrict__ total_units,
                                                               uint64_t cap_units,
                                                               const uint32_t* __restrict__ Lq,
                                                               int heavy_only) {
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (total_units[0] > cap_units || total_units[1] != 0) return;  // [1]: the groups overflowed
  // (a dispatch holds at most 2^32 work-items: the slices of a large sparse matrix — 11 M at
  // m = 300k — are walked with a grid stride)
  for (int64_t s = blockIdx.x; s < static_cast<int64_t>(ncg) * nchunks; s += gridDim.x) {
  // heavy_only: slices of at most SL_STAGE_CAP entries were packed by k_slice_pack_staged
  if (heavy_only && (Lq[s] >> 8) <= static_cast<uint32_t>(SL_STAGE_CAP)) continue;
  if ((Lq[s] & 255u) == 0) {  // an empty slice: the header and the lengths only
    if (wave == 0) {
      uint8_t* sp0 = data + 16 * Pre[s];
#pragma unroll
      for (int h = 0; h < H; ++h) sp0[16 + h * 64 + lane] = 0;
      if (lane < 4) reinterpret_cast<uint32_t*>(sp0)[lane] = (lane == 2) ? 16 + H * 64 : 0;
    }
    continue;
  }
  const int cg = static_cast<int>(s / nchunks), k = static_cast<int>(s - static_cast<int64_t>(cg) * nchunks);
  const int64_t c = static_cast<int64_t>(cg) * SL_W + lane;
  int nq[H];
  int tot = 0;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    S.prepare_sub(c, (static_cast<int64_t>(k) * H + h) * SL_SUB);
    nq[h] = (S.count() + 3) >> 2;
    tot += nq[h];
  }
  const int maxq = sl_wave_max(tot);
  uint8_t* sp = data + 16 * Pre[s];
  uint32_t* so = reinterpret_cast<uint32_t*>(sp + 16 + H * 64);
  const int sob = sl_so_bytes(maxq);
  if (wave == 0) {
#pragma unroll
    for (int h = 0; h < H; ++h) sp[16 + h * 64 + lane] = static_cast<uint8_t>(nq[h]);
    if (lane * 4 < sob && lane >= (maxq + SL_SO - 1) / SL_SO) so[lane] = 0;  // the table's padding
  }
  uint32_t off = 16 + H * 64 + sob;  // wave-uniform
  int hcur = H == 1 ? 0 : -1;  // sub-block the source is positioned on (per lane)
  int edge = H == 1 ? nq[0] : 0;    // first step behind that sub-block
  int qbase = 0;                    // its first step
  for (int q = 0; q < maxq; ++q) {
    const bool active = q < tot;
    const uint64_t mask = __ballot(active);
    const int cnt = __popcll(mask);
    if ((q % SL_PACKW) == wave) {  // uniform
      if ((q % SL_SO) == 0 && lane == 0) so[q / SL_SO] = off;
      if (active) {
        if constexpr (H > 1) {
          while (q >= edge) {  // this lane's steps of sub-block hcur are used up: next one
            ++hcur;
            qbase = edge;
            int nn = 0;
#pragma unroll
            for (int h = 0; h < H; ++h) nn = (h == hcur) ? nq[h] : nn;
            edge += nn;
            if (nn > 0) S.prepare_sub(c, (static_cast<int64_t>(k) * H + hcur) * SL_SUB);
          }
        }
        const uint32_t rank = sl_lane_rank(mask);
        SliceQuad<VT> vq;
        uint32_t rq = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t row;
          S.fetch((q - qbase) * 4 + e, vq.v[e], row);
          rq |= row << (8 * e);
        }
        vq.store(sp + off + rank * QB);
        *reinterpret_cast<uint32_t*>(sp + off + cnt * QB + rank * 4) = rq;
      }
      // zero the step's padding (the row quads are padded to 16 bytes)
      const int padw = (((cnt * 4 + 15) & ~15) - cnt * 4) >> 2;  // dwords
      if (lane < padw) *reinterpret_cast<uint32_t*>(sp + off + cnt * QB + cnt * 4 + lane * 4) = 0;
    }
    off += cnt * QB + ((cnt * 4 + 15) & ~15);
  }
  const int nquads = sl_wave_sum(tot);
  if (wave == 0 && lane == 0) {
    uint32_t* hd = reinterpret_cast<uint32_t*>(sp);
    hd[0] = static_cast<uint32_t>(nquads);
    hd[1] = static_cast<uint32_t>(maxq);
    hd[2] = off;
    hd[3] = 0;
  }
  }  // slices of this workgroup
}

// k_slice_pack_staged — the packer of the common case (source = groups, H = 1, a slice of at most
// SL_STAGE_CAP entries): one wave per slice. The lists of the slice's 64 columns in one 64-row
// block are ONE contiguous run of the group (columns are stored back to back), so the wave
// copies its four runs into LDS with coalesced loads and the lanes pick their entries from
// there; k_slice_pack's per-lane gathers from global memory touch ~40 cache lines per load
// instruction and cost 10x as much. Heavier slices (dense blocks) are left to k_slice_pack
// (`heavy_only`), whose lanes then read long, nearly contiguous lists anyway.
template <typename VT>
__device__ __forceinline__ int sl_slice_entries(const GroupSource<VT>& S, int cg, int k, int lane) {
  // entries of slice (cg, k): the four runs' lengths (wave-uniform)
  const int64_t s = cg >> 1;
  const int cl0 = (cg & 1) * 64;
  int total = 0;
#pragma unroll
  for (int j = 0; j < SL_SUB / GR_RB; ++j) {
    const int64_t b = static_cast<int64_t>(k) * (SL_SUB / GR_RB) + j;
    if (b < S.nblocks) {
      const int64_t g = s * S.nblocks + b;
      total += static_cast<int>(S.Goff[g * GR_OFFS + cl0 + 64]) - static_cast<int>(S.Goff[g * GR_OFFS + cl0]);
    }
  }
  return total;
}

template <typename VT>
__global__ __launch_bounds__(256) void k_slice_pack_staged(GroupSource<VT> S, int ncg, int nchunks,
                                                            const uint64_t* __restrict__ Pre,
                                                            uint8_t* __restrict__ data,
                                                            const uint64_t* __restrict__ total_units,
                                                            uint64_t cap_units) {
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  __shared__ VT valsL[4][SL_STAGE_CAP];
  __shared__ uint8_t rowsL[4][SL_STAGE_CAP];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t s = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (s >= static_cast<int64_t>(ncg) * nchunks) return;  // whole wave
  if (total_units[0] > cap_units || total_units[1] != 0) return;  // [1]: the groups overflowed
  const int cg = static_cast<int>(s / nchunks), k = static_cast<int>(s - static_cast<int64_t>(cg) * nchunks);
  if (sl_slice_entries(S, cg, k, lane) > SL_STAGE_CAP) return;  // k_slice_pack's
  const int64_t strip = cg >> 1;
  const int cl0 = (cg & 1) * 64, cl = cl0 + lane;
  // the four runs -> LDS; this lane's list of block j starts at lst[j], cnt[j] entries
  int lst[4], cnt[4];
  int roff = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t b = static_cast<int64_t>(k) * (SL_SUB / GR_RB) + j;
    lst[j] = 0;
    cnt[j] = 0;
    if (j < SL_SUB / GR_RB && b < S.nblocks) {  // uniform
      const int64_t g = strip * S.nblocks + b;
      const uint16_t* go = S.Goff + g * GR_OFFS;
      const int r0 = go[cl0], r1 = go[cl0 + 64];
      const int o0 = go[cl], o1 = go[cl + 1];
      const int64_t base = static_cast<int64_t>(S.Gpre[g]) * 4 + r0;
      const int len = r1 - r0;
      // 512 entries per trip, every load issued before the first store (idle lanes re-read the
      // run's last entry: no divergent branch around a load, see slice_core)
      for (int i0 = 0; i0 < len; i0 += 512) {
        VT tv[8];
        uint8_t tr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 64 + lane;
          const int ic = i < len ? i : len - 1;
          tv[u] = S.vals[base + ic];
          tr[u] = S.rows[base + ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 64 + lane;
          if (i < len) {
            valsL[wave][roff + i] = tv[u];
            rowsL[wave][roff + i] = tr[u];
          }
        }
      }
      lst[j] = roff + (o0 - r0);
      cnt[j] = o1 - o0;
      roff += len;
    }
  }
  const int n = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  const int tot = (n + 3) >> 2;
  const int maxq = sl_wave_max(tot);
  const int nquads = sl_wave_sum(tot);
  uint8_t* sp = data + 16 * Pre[s];
  sp[16 + lane] = static_cast<uint8_t>(tot);
  uint32_t* so = reinterpret_cast<uint32_t*>(sp + 16 + 64);
  const int sob = sl_so_bytes(maxq);
  if (lane * 4 < sob && lane >= (maxq + SL_SO - 1) / SL_SO) so[lane] = 0;
  uint32_t off = 16 + 64 + sob;
  const int c0 = cnt[0], c01 = c0 + cnt[1], c012 = c01 + cnt[2];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int q = 0; q < maxq; ++q) {
    const bool active = q < tot;
    const uint64_t mask = __ballot(active);
    const int cntq = __popcll(mask);
    if ((q % SL_SO) == 0 && lane == 0) so[q / SL_SO] = off;
    if (active) {
      const uint32_t rank = sl_lane_rank(mask);
      SliceQuad<VT> vq;
      uint32_t rq = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = q * 4 + e;
        const bool valid = i < n;
        const int ii = valid ? i : 0;
        const int j = (ii >= c0 ? 1 : 0) + (ii >= c01 ? 1 : 0) + (ii >= c012 ? 1 : 0);
        const int before = j == 0 ? 0 : (j == 1 ? c0 : (j == 2 ? c01 : c012));
        const int st = j == 0 ? lst[0] : (j == 1 ? lst[1] : (j == 2 ? lst[2] : lst[3]));
        const int a = valid ? st + (ii - before) : 0;
        const VT lv = valsL[wave][a];
        const uint32_t lr = rowsL[wave][a];
        vq.v[e] = valid ? lv : VT(0);
        rq |= (valid ? static_cast<uint32_t>(j * GR_RB) + lr : 0u) << (8 * e);
      }
      vq.store(sp + off + rank * QB);
      *reinterpret_cast<uint32_t*>(sp + off + cntq * QB + rank * 4) = rq;
    }
    const int padw = (((cntq * 4 + 15) & ~15) - cntq * 4) >> 2;  // dwords of the step's padding
    if (lane < padw) *reinterpret_cast<uint32_t*>(sp + off + cntq * QB + cntq * 4 + lane * 4) = 0;
    off += cntq * QB + ((cntq * 4 + 15) & ~15);
  }
  if (lane == 0) {
    uint32_t* hd = reinterpret_cast<uint32_t*>(sp);
    hd[0] = static_cast<uint32_t>(nquads);
    hd[1] = static_cast<uint32_t>(maxq);
    hd[2] = off;
    hd[3] = 0;
  }
}

// exclusive scan by ONE workgroup (n up to a few 10^4: three launches cost more than the scan)
__global__ __launch_bounds__(1024) void k_slice_scan_small(const uint32_t* __restrict__ sizes,
                                                            int64_t n, uint64_t* __restrict__ Pre,
                                                            uint64_t* __restrict__ total) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < n; b0 += 1024) {
    const int64_t i = b0 + threadIdx.x;
    const uint64_t v = (i < n) ? sizes[i] : 0;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t t = __shfl_up(inc, o);
      if ((threadIdx.x & 63) >= o) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint64_t off = carry_s;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < n) Pre[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[0] = carry_s;
}

// k_slice_expand — the dense store S[j][c] (row pitch ld, element ST) back from the slices of
// this shard's columns (getters, the exact DSD rounding's gather, the matvec API's dense path):
// one wave per slice, a lane zeroes its column's rows of the chunk and scatters its entries.
template <typename VT, int H, typename ST>
__global__ __launch_bounds__(256) void k_slice_expand(SliceView M, ST* __restrict__ S, int64_t ld,
                                                       int64_t m) {
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  constexpr int R = SL_SUB * H;
  const int lane = threadIdx.x & 63;
  const int64_t s = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (s >= static_cast<int64_t>(M.ncg) * M.nchunks) return;
  const int cg = static_cast<int>(s / M.nchunks), k = static_cast<int>(s - static_cast<int64_t>(cg) * M.nchunks);
  const int64_t c = static_cast<int64_t>(cg) * SL_W + lane;
  const int64_t r0 = static_cast<int64_t>(k) * R;
  if (c < ld)
    for (int q = 0; q < R; ++q)
      if (r0 + q < m) S[(r0 + q) * ld + c] = ST(0);
  SliceHead<H> hd;
  hd.load(static_cast<gbytes_t>((gbytes_t)M.data + 16 * M.Pre[s]), lane);
  const int maxq = __builtin_amdgcn_readfirstlane(hd.maxq);
  int tot = 0;
#pragma unroll
  for (int h = 0; h < H; ++h) tot += hd.nq[h];
  gbytes_t fbase = hd.sp + 16 + H * 64 + sl_so_bytes(maxq);
  for (int q = 0; q < maxq; ++q) {
    const bool active = q < tot;
    const uint64_t mask = __ballot(active);
    const int cnt = __popcll(mask);
    if (active) {
      const uint32_t rank = sl_lane_rank(mask);
      SliceQuad<VT> vq;
      vq.load(fbase + rank * QB);
      const uint32_t rq = *reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(fbase + cnt * QB + rank * 4);
      int rowbase = 0, edge = hd.nq[0];
#pragma unroll
      for (int h = 1; h < H; ++h) {
        rowbase = (q >= edge) ? h * SL_SUB : rowbase;
        edge += hd.nq[h];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (vq.v[e] != VT(0) && c < ld)
          S[(r0 + rowbase + ((rq >> (8 * e)) & 255u)) * ld + c] = static_cast<ST>(vq.v[e]);
    }
    fbase += cnt * QB + ((cnt * 4 + 15) & ~15);
  }
}

// k_slice_gather_sub — the sub-matrix of M induced by a node list (the exact DSD rounding's input,
// dsd.cpp:274-320) straight from the slices: pos[g] = position of node g in the list or -1; one wave per
// slice of this shard's columns (c0 = its first column), a lane whose column is listed walks its quads
// and writes out[pos[row] * k + pos[column]] for the listed rows. `out` is zeroed by the caller; every
// stored entry is written by exactly one lane. No dense copy of M exists for this (m = 300 000: there
// could not be one).
template <typename VT, int H>
__global__ __launch_bounds__(256) void k_slice_gather_sub(SliceView M, const int32_t* __restrict__ pos,
                                                           int64_t c0, int64_t m, int k,
                                                           double* __restrict__ out) {
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  constexpr int R = SL_SUB * H;
  const int lane = threadIdx.x & 63;
  const int64_t s = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (s >= static_cast<int64_t>(M.ncg) * M.nchunks) return;
  const int cg = static_cast<int>(s / M.nchunks), kc = static_cast<int>(s - static_cast<int64_t>(cg) * M.nchunks);
  const int64_t c = c0 + static_cast<int64_t>(cg) * SL_W + lane;
  const int pc = c < m ? pos[c] : -1;
  if (__ballot(pc >= 0) == 0) return;  // none of the 64 columns is listed
  const int64_t r0 = static_cast<int64_t>(kc) * R;
  SliceHead<H> hd;
  hd.load(static_cast<gbytes_t>((gbytes_t)M.data + 16 * M.Pre[s]), lane);
  const int maxq = __builtin_amdgcn_readfirstlane(hd.maxq);
  int tot = 0;
#pragma unroll
  for (int h = 0; h < H; ++h) tot += hd.nq[h];
  gbytes_t fbase = hd.sp + 16 + H * 64 + sl_so_bytes(maxq);
  for (int q = 0; q < maxq; ++q) {
    const bool active = q < tot;
    const uint64_t mask = __ballot(active);
    const int cnt = __popcll(mask);
    if (active && pc >= 0) {
      const uint32_t rank = sl_lane_rank(mask);
      SliceQuad<VT> vq;
      vq.load(fbase + rank * QB);
      const uint32_t rq = *reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(fbase + cnt * QB + rank * 4);
      int rowbase = 0, edge = hd.nq[0];
#pragma unroll
      for (int h = 1; h < H; ++h) {
        rowbase = (q >= edge) ? h * SL_SUB : rowbase;
        edge += hd.nq[h];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (vq.v[e] != VT(0)) {
          const int64_t row = r0 + rowbase + ((rq >> (8 * e)) & 255u);
          const int pr = pos[row];
          if (pr >= 0) out[static_cast<int64_t>(pr) * k + pc] = static_cast<double>(vq.v[e]);
        }
    }
    fbase += cnt * QB + ((cnt * 4 + 15) & ~15);
  }
}

}  // namespace clipper_hip
