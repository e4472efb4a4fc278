// slice_xmode.hip.h — HARNESS ONLY (tools/slice_tune.hip): the streaming loop of a pass on the slices with the
// staging variants that were measured and NOT adopted (DESIGN.md section 7 "tried"): CLIPPER_SL_XMODE = 1, 2
// (candidates formed in registers per entry) and 3 (the linear window). A snapshot of slice_core as it stood
// when they were measured (round 3: profiles/r03_slice_tune_xmode.txt, r03_linear_window.txt); the product's
// k_slices.hip.h carries mode 0 only. Include after k_slices.hip.h; everything lives in clipper_hip::xmode.
#pragma once

#include "../clipper_amd/csrc/k_slices.hip.h"

namespace clipper_hip {
namespace xmode {

// Window mode stages candidates 0 .. sl_xload(V)-1 of a table row at a pitch of sl_xpitch(V)
// doubles: an ODD number of 16-byte units (1, 3, 5), so that the rows of a sub-block spread over
// all 16 slots a ds_read_b128 lane group can serve in one LDS cycle (pitch 32 B would use 8
// of them, 64 B only 4).
// CLIPPER_SL_XMODE (tools/slice_tune.hip only; the product is mode 0): what a window pass stages per x row
// and what it computes per entry instead —
//   0  all V candidates staged (48 bytes at V = 6: three ds_read_b128 per entry)
//   1  (u', g') staged (16 bytes: one ds_read_b128 per entry), the V candidates formed in registers per
//      entry with the tail's own expression
//   2  (u', g', candidate 0, candidate 1) staged (two ds_read_b128), candidates 2 .. V-1 in registers
//   3  LINEAR WINDOW: (u', g') staged, and no candidate is formed at all where none is clamped — candidate
//      l of row r is max(t, 0) with t = u'[r] + alpha_l g'[r], and once the iteration has settled (from the
//      third or fourth line search on: measured, m = 10k) t > 0 for every candidate of every live row, i.e.
//      the V products are LINEAR in two:  M c_l = M u' + alpha_l M g'.  A pass then accumulates M u', M g',
//      C u', C g' (one ds_read_b128 and four fmas per entry instead of three and seven) and combines them
//      per lane at its end. A row none of whose candidates is positive is staged as (0, 0); a row with
//      some clamped and some not ("mixed": a bit in a 128-bit mask per chunk) adds, entry by entry,
//      max(t, 0) - t per candidate to correction sums — taken only by the chunks that have such a row.
//      The sums differ from mode 0's by roundings (t is never rounded to a double before it is
//      multiplied), as they do between two orders of the partial sums.
//   4  SYMMETRIC HALF (round 4; VERDICT r01-r03 "upper-triangle pass"): only the entries ABOVE the diagonal are
//      stored (row < column: what selfadjointView<Upper> reads, clipper.cpp:195). A stored entry (r, c) then owes
//      two contributions: the column sum acc_c += w x[r] (the lane's own registers, as in mode 0) and the
//      MIRRORED one y[r] += w x[c] — c is the lane's column, r is anybody's row: V + 1 ds_add_f64 per entry
//      into a [128][V + 1] accumulator of the chunk in LDS (double buffered), flushed with global fp64 atomics
//      when the workgroup leaves the chunk (128 x (V + 1) per chunk and workgroup; their order is not fixed:
//      bit-reproducibility is gone). Half the bytes, 3 gathers + 7 atomic adds per entry through the LDS pipe.
// Measured: profiles/r03_slice_tune_xmode.txt (modes 1, 2), profiles/r04_symmetric_half.txt (mode 4).
#ifndef CLIPPER_SL_XMODE
#define CLIPPER_SL_XMODE 0
#endif
constexpr int SL_XMODE = CLIPPER_SL_XMODE;
constexpr int sl_xload(int V) { return V <= 2 ? 2 : (V <= 4 ? 4 : (V <= 6 ? 6 : 8)); }
constexpr int sl_xpitch(int V) { return (SL_XMODE == 1 || SL_XMODE == 3) ? 2 : (V <= 2 ? 2 : (V <= 6 ? 6 : 10)); }
constexpr int sl_lds_doubles(int V, int H, int NW) {
  // two x buffers (+ mode 3: their masks of mixed rows, and every lane's V + 1 correction sums)
  const int a = 2 * SL_SUB * H * sl_xpitch(V) + (SL_XMODE == 3 ? 4 + (V + 1) * NW * 64 : 0) +
                (SL_XMODE == 4 ? 2 * SL_SUB * H * (V + 1) : 0);
  const int b = NW * 64 + NW * 2 * V + 8;       // the decision's scratch
  return a > b ? a : b;
}
__device__ double* g_ypart = nullptr;  // (mode 4) [V + 1][ld]: where the mirrored contributions are added
__host__ __device__ constexpr int sl_so_bytes(int maxq) {
  return ((maxq + SL_SO - 1) / SL_SO * 4 + 15) & ~15;
}

// Stage the x rows of chunk k ([R][XP] doubles, row pitch XP). Window mode: the candidates are
// BUILT here — row r, candidate l = max(u'[r] + alpha0 beta^l g'[r], 0) (clipper.cpp:235-236) from
// the point slot the window starts from, 16 bytes read per row instead of a 64-byte table row that
// the tail would have had to write for every outcome it speculates on. The expression, the chain of
// multiplications behind alpha0 beta^l and hence the bits are those of the tail (k_solver.hip.h).
// Pair mode: X[r * xstride]. Pieces of 16 bytes (two candidates), thread-linear.
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
  // (mode 3) the 128-bit mask of mixed rows of the buffer `xs`: two words behind the two x buffers, found
  // from the buffer's own address (the buffers are R * XP doubles apart, the first one 2 R XP-aligned in LDS
  // only by construction of the caller: it passes the base)
  double* base = nullptr;
  __device__ __forceinline__ uint64_t* mask_of(const double* xs) const {
    return reinterpret_cast<uint64_t*>(base + 2 * (R * XP)) + ((xs == base) ? 0 : 2);
  }
  __device__ __forceinline__ void store(const WindowSource& W, double* xs) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = threadIdx.x + i * NT;
      if constexpr (WINDOW && SL_XMODE == 3) {
        static_assert(!WINDOW || SL_XMODE != 3 || (PER == 1 && R == 128), "a row per thread of the first two waves");
        // first and last candidate bound the others (the step sizes fall monotonically, and so do the
        // rounded sums): both positive = no candidate clamped, neither = the row adds nothing
        double al = W.alpha0;
        const double t0 = wu[i] + al * wg[i];
#pragma unroll
        for (int l = 1; l < XL; ++l) al = al * W.beta;
        const double t1 = wu[i] + al * wg[i];
        const bool pos0 = t0 > 0.0, pos1 = t1 > 0.0;
        const bool mixed = p < PIECES && (pos0 != pos1);
        const bool dead = !pos0 && !pos1;
        if (p < PIECES) *reinterpret_cast<double2*>(xs + p * XP) = dead ? make_double2(0.0, 0.0) : make_double2(wu[i], wg[i]);
        const uint64_t mk = __ballot(mixed);
        if ((threadIdx.x & 63) == 0 && threadIdx.x < R) mask_of(xs)[threadIdx.x >> 6] = mk;
      } else if constexpr (WINDOW && (SL_XMODE == 1 || SL_XMODE == 2)) {
        if (p < PIECES) {
          *reinterpret_cast<double2*>(xs + p * XP) = make_double2(wu[i], wg[i]);
          if constexpr (SL_XMODE == 2) {
            double t0 = wu[i] + W.alpha0 * wg[i];
            t0 = (t0 > 0.0) ? t0 : 0.0;
            const double a1 = W.alpha0 * W.beta;
            double t1 = wu[i] + a1 * wg[i];
            t1 = (t1 > 0.0) ? t1 : 0.0;
            *reinterpret_cast<double2*>(xs + p * XP + 2) = make_double2(t0, t1);
          }
        }
      } else if constexpr (WINDOW) {
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
  xst.base = lds;
  // (mode 3) the linear sums M u', M g', C u', C g' and the corrections of the mixed rows: candidate 0
  // against M and against C, candidates 1 .. V-1 against M + d C
  constexpr bool LIN = WINDOW && SL_XMODE == 3;
  // (the corrections live in LDS, one column per lane: they are touched by the chunks with a mixed row only,
  // and seven more doubles of registers through the streaming loop would cost a workgroup per CU)
  double lin[LIN ? 4 : 1];
#pragma unroll
  for (int v = 0; v < (LIN ? 4 : 1); ++v) lin[v] = 0.0;
  double* corr = lds + 2 * (R * XP) + 4 + threadIdx.x;  // corr[v * NT]
  // (mode 4) the mirrored contributions of a chunk: Y[2][R][NS] behind the x buffers; the lane's own column as a
  // multiplier: its V candidates, formed like a staged row's
  constexpr bool SYM = WINDOW && SL_XMODE == 4;
  double* const Ybase = lds + 2 * (R * XP);
  double xc[SYM ? V : 1];
  double* const ypart = g_ypart;
  __syncthreads();  // the decision at the head of the launch used the same LDS
  if constexpr (SYM) {
    for (int i = threadIdx.x; i < 2 * R * NS; i += NT) Ybase[i] = 0.0;
    const int64_t cc = static_cast<int64_t>(cg) * SL_W + lane;
    const bool has = mine && cc < m;
    const double uc = has ? WS.U[cc] : 0.0, gc = has ? WS.G[cc] : 0.0;
    double al = WS.alpha0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const double t = uc + al * gc;
      xc[v] = (t > 0.0) ? t : 0.0;
      al = al * WS.beta;
    }
  }
  auto flush_y = [&](int k) {  // chunk k's accumulator -> global, and zero again
    double* Yb = Ybase + ((k - t0) & 1) * (R * NS);
    for (int i = threadIdx.x; i < R * NS; i += NT) {
      const double val = Yb[i];
      Yb[i] = 0.0;
      const int64_t r = static_cast<int64_t>(k) * R + i / NS;
      if (val != 0.0 && r < m) atomicAdd(&ypart[static_cast<int64_t>(i % NS) * ld + r], val);
    }
  };
  if constexpr (LIN) {
#pragma unroll
    for (int v = 0; v <= V; ++v) corr[v * NT] = 0.0;
  }
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
    double* const Ycur = Ybase + ((k - t0) & 1) * (R * NS);
    (void)Ycur;
    const bool more = k + 1 < t1;
    SliceHead<H> nxt = cur;
    uint64_t pre_next2 = 0;
    if (more) {
      xst.load(WS, X, xstride, static_cast<int64_t>(k + 1) * R, M.nrows);
      xst.index(static_cast<int64_t>(k + 2) * R, M.nrows, M.rowmap);
      if (mine) nxt.load(M.data + 16 * pre_next, lane);
      if (k + 2 < t1) pre_next2 = pre_row[k + 2];
    }
    uint64_t mixlo = 0, mixhi = 0;  // (mode 3) the mixed rows of this chunk, wave-uniform
    if constexpr (LIN) {
      const uint64_t* mp = xst.mask_of(xs);
      const uint64_t a = mp[0], b = mp[1];
      mixlo = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32))) << 32) |
              static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a)));
      mixhi = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32))) << 32) |
              static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b)));
    }
    const bool any_mixed = (mixlo | mixhi) != 0;  // uniform
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
              if constexpr (LIN) {
                const double2 ug = *reinterpret_cast<const double2*>(xs + row * XP);
                lin[0] = fma(mm, ug.x, lin[0]);
                lin[1] = fma(mm, ug.y, lin[1]);
                lin[2] = fma(ii, ug.x, lin[2]);
                lin[3] = fma(ii, ug.y, lin[3]);
                if (any_mixed) {  // (uniform) a chunk with a mixed row: is this entry in one?
                  const uint64_t word = row < 64 ? mixlo : mixhi;
                  if ((word >> (row & 63)) & 1) {
                    const double w = fma(d, ii, mm);
                    double al = WS.alpha0;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                      const double t = ug.x + al * ug.y;
                      const double nl = (t > 0.0) ? 0.0 : -t;  // max(t, 0) - t
                      if (v == 0) {
                        corr[0] = fma(mm, nl, corr[0]);
                        corr[V * NT] = fma(ii, nl, corr[V * NT]);
                      } else {
                        corr[v * NT] = fma(w, nl, corr[v * NT]);
                      }
                      al = al * WS.beta;
                    }
                  }
                }
              } else if constexpr (WINDOW && (SL_XMODE == 1 || SL_XMODE == 2)) {
                const double* xr = xs + row * XP;
                const double2 ug = *reinterpret_cast<const double2*>(xr);
                double xv[V > 2 ? V : 2];
                double al = WS.alpha0;
                int l0 = 0;
                if constexpr (SL_XMODE == 2) {
                  const double2 t2 = *reinterpret_cast<const double2*>(xr + 2);
                  xv[0] = t2.x;
                  xv[1] = t2.y;
                  al = (al * WS.beta) * WS.beta;
                  l0 = 2;
                }
#pragma unroll
                for (int v = 0; v < V; ++v) {
                  if (v >= l0) {
                    double t = ug.x + al * ug.y;
                    xv[v] = (t > 0.0) ? t : 0.0;
                    al = al * WS.beta;
                  }
                }
                acc[0] = fma(mm, xv[0], acc[0]);
                acc[V] = fma(ii, xv[0], acc[V]);
                if (V > 1) {
                  const double w = fma(d, ii, mm);
#pragma unroll
                  for (int v = 1; v < V; ++v) acc[v] = fma(w, xv[v], acc[v]);
                }
              } else if constexpr (WINDOW) {
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
                const double w = fma(d, ii, mm);
                if (V > 1) {
#pragma unroll
                  for (int v = 1; v < V; ++v) acc[v] = fma(w, xv[v], acc[v]);
                }
                if constexpr (SYM) {
                  {  // (a padded entry adds zeros to row 0: a branch around the adds costs registers — the compiler
                     // then forms the products of all four entries ahead of the branches — and so occupancy)
                    double* yr = Ycur + row * NS;
                    atomicAdd(yr, mm * xc[0]);
                    atomicAdd(yr + V, ii * xc[0]);
#pragma unroll
                    for (int v = 1; v < V; ++v) atomicAdd(yr + v, w * xc[v]);
                  }
                  // (entry by entry: with the four entries' gathers and products in flight at once the kernel
                  // spills at every occupancy the LDS allows)
                  __builtin_amdgcn_sched_barrier(0);
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
    if constexpr (SYM) flush_y(k);  // (beside the other waves' next chunk: that one adds into the other buffer)
  }

  if constexpr (LIN) {  // the V + 1 sums a window pass hands to the tail, from the linear ones
    double al = WS.alpha0;
    acc[0] = fma(al, lin[1], lin[0]) + corr[0];
    acc[V] = fma(al, lin[3], lin[2]) + corr[V * NT];
    const double wu_ = fma(d, lin[2], lin[0]), wg_ = fma(d, lin[3], lin[1]);  // (M + d C) u', (M + d C) g'
#pragma unroll
    for (int v = 1; v < V; ++v) {
      al = al * WS.beta;
      acc[v] = fma(al, wg_, wu_) + corr[v * NT];
    }
  }
  const int64_t c = static_cast<int64_t>(cg) * SL_W + lane;
  if (c < ld && static_cast<int>(blockIdx.x) < M.nwork) {
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


}  // namespace xmode
}  // namespace clipper_hip
