// k_csc.hip.h — the compressed storage of M: layout, emission (csc_emit*), k_csc_build, k_csc_expand, k_gemv_csc
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_solver.hip.h"

namespace clipper_hip {

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

// window or pair mode by the plan of this iteration
template <int V, int NW>
__device__ __forceinline__ void csc_by_plan(const CscView& M, const SolveArgs& A,
                                            const PassPlan& plan, double* lds) {
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
  csc_by_plan<V, NW>(M, A, plan, lds);
  flush_state(A, &stash);
}

}  // namespace clipper_hip
