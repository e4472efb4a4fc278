// k_rowview.hip.h — the row list of a row view (k_solver.hip.h, LIVE ROWS): the rows that can be non-zero
// in the window the next pass multiplies, in ascending order.
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_affinity.hip.h"
#include "k_csc.hip.h"
#include "k_solver.hip.h"

namespace clipper_hip {

// The view is built while the solve is on HOLD (k_solver.hip.h): the decision that asked for it has
// worked out — deterministically, it will work it out again when the hold is lifted — which point the
// solve continues from (hold_slot: the accepted candidate's point slot, or the unchanged point's). The
// window the next pass multiplies is built from that point, so its rows are exactly the live rows of
// that point: u > 0 or g > 0.
constexpr int RV_BLK = 1024;  // rows per workgroup (256 threads x 4)

template <int V>
__device__ __forceinline__ bool rv_live(const SolverState* st, const double* pt, int64_t mp, int64_t i) {
  const int64_t slot = st->hold_slot;
  const double* u = pt + (slot * 2 + 0) * mp;
  const double* g = pt + (slot * 2 + 1) * mp;
  return u[i] > 0.0 || g[i] > 0.0;
}

// flags[i] = live(i); blk[b] = live rows of block b
template <int V>
__global__ __launch_bounds__(256) void k_rv_flags(const SolverState* __restrict__ st,
                                                   const double* __restrict__ pt, int64_t mp, int64_t m,
                                                   uint8_t* __restrict__ flags,
                                                   uint32_t* __restrict__ blk) {
  __shared__ uint32_t wsum[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * RV_BLK + threadIdx.x * 4;
  uint32_t n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k;
    const bool live = i < m && rv_live<V>(st, pt, mp, i);
    if (i < mp) flags[i] = live ? 1 : 0;
    n += live ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the nb block counts in place (one workgroup), total -> blk[nb] and `count_out`
// (mapped host memory: the host sizes the fill from it)
__global__ __launch_bounds__(1024) void k_rv_scan(uint32_t* __restrict__ blk, int nb,
                                                   int32_t* __restrict__ count_out) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const uint32_t v = (i < nb) ? blk[i] : 0;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if ((threadIdx.x & 63) >= o) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < nb) blk[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    blk[nb] = carry_s;
    __hip_atomic_store(count_out, static_cast<int32_t>(carry_s), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// rowmap[rank of i among the live rows] = i (ascending: the order of the additions of a pass on
// the view is a function of the row SET alone)
// viewpos[i] = that rank, or -1 (what k_slice_filter_rows asks of every entry's row)
// The three launches above as ONE workgroup, for mp <= RV_ONE_MAX (round 4: each of them runs at the floor of a
// launch, ~4.7 us for a microsecond of work; the headline's view is listed in one launch instead of three). The same
// flags, the same ascending list: coalesced loads of (u', g') -> a bit per row in LDS -> one count per 64-row word,
// scanned -> every thread writes 16 rows of its word.
constexpr int RV_ONE_E = 16;
constexpr int64_t RV_ONE_MAX = 1024 * RV_ONE_E;
template <int V>
__global__ __launch_bounds__(1024) void k_rv_list_one(const SolverState* __restrict__ st, const double* __restrict__ pt,
                                                       int64_t mp, int64_t m, uint8_t* __restrict__ flags,
                                                       int32_t* __restrict__ rowmap, int64_t cap,
                                                       int32_t* __restrict__ viewpos, int32_t* __restrict__ count_out) {
  __shared__ unsigned long long word[RV_ONE_E * 16];  // bit r of word w: row 64 w + r is live
  __shared__ uint32_t pre[RV_ONE_E * 16];             // live rows in front of word w
  __shared__ uint32_t wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int E = static_cast<int>((mp + 1023) / 1024);
  const int64_t slot = st->hold_slot;
  const double* u = pt + (slot * 2 + 0) * mp;
  const double* g = pt + (slot * 2 + 1) * mp;
  double uu[RV_ONE_E], gg[RV_ONE_E];
#pragma unroll
  for (int k = 0; k < RV_ONE_E; ++k) {  // (all loads in flight at once)
    const int64_t i = static_cast<int64_t>(k) * 1024 + t;
    const bool in = k < E && i < m;
    uu[k] = in ? u[i] : 0.0;
    gg[k] = in ? g[i] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < RV_ONE_E; ++k) {
    if (k < E) {  // (uniform)
      const int64_t i = static_cast<int64_t>(k) * 1024 + t;
      const bool live = uu[k] > 0.0 || gg[k] > 0.0;
      if (i < mp) flags[i] = live ? 1 : 0;
      const unsigned long long mk = __ballot(live);
      if (lane == 0) word[k * 16 + wave] = mk;
    }
  }
  __syncthreads();
  // counts of the words -> exclusive scan (the first 256 threads: one word each)
  const int nwords = E * 16;
  uint32_t n = 0, inc = 0;
  if (t < 256) {
    n = t < nwords ? static_cast<uint32_t>(__popcll(word[t])) : 0u;
    inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t x = __shfl_up(inc, o);
      if (lane >= o) inc += x;
    }
    if (lane == 63) wsum[wave] = inc;
  }
  __syncthreads();
  if (t < 256) {
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    pre[t] = off + inc - n;
  }
  __syncthreads();
  // thread t: rows [16 t, 16 t + 16) = quarter t & 3 of word t >> 2
  if ((t >> 2) < nwords) {
    const unsigned long long wd = word[t >> 2];
    const int q = t & 3;
    uint32_t at = pre[t >> 2] + static_cast<uint32_t>(__popcll(wd & ((1ull << (16 * q)) - 1ull)));
    const uint32_t bits = static_cast<uint32_t>(wd >> (16 * q)) & 0xffffu;
    const int64_t r0 = static_cast<int64_t>(t) * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t i = r0 + k;
      const bool f = (bits >> k) & 1u;
      if (i < mp) viewpos[i] = f ? static_cast<int32_t>(at) : -1;
      if (f) {
        if (static_cast<int64_t>(at) < cap) rowmap[at] = static_cast<int32_t>(i);
        ++at;
      }
    }
  }
  if (t == 0)
    __hip_atomic_store(count_out, static_cast<int32_t>(wsum[0] + wsum[1] + wsum[2] + wsum[3]), __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_rv_scatter(const uint8_t* __restrict__ flags, int64_t m,
                                                     const uint32_t* __restrict__ blk,
                                                     int32_t* __restrict__ rowmap, int64_t cap,
                                                     int32_t* __restrict__ viewpos, int64_t mp) {
  __shared__ uint32_t wsum[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * RV_BLK + threadIdx.x * 4;
  uint32_t f[4], n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[k] = (base + k < m && flags[base + k]) ? 1u : 0u;
    n += f[k];
  }
  uint32_t inc = n;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if ((threadIdx.x & 63) >= o) inc += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t off = blk[blockIdx.x];
  for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
  uint32_t at = off + inc - n;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < mp) viewpos[base + k] = f[k] ? static_cast<int32_t>(at) : -1;
    if (f[k]) {
      if (static_cast<int64_t>(at) < cap) rowmap[at] = static_cast<int32_t>(base + k);
      ++at;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_slice_filter_rows — the slices of M[rows, :] from the slices of M itself. A view keeps the entries
// of M whose row is in its list: they are already scored and stored, so a view is a FILTER of the
// matrix's own column lists (one read of M at streaming rate) rather than a second evaluation of
// nrows x m pairs from the points (k_affinity_rect: 2-5 x the time at m >= 100k, and only for matrices
// that were scored from staged points). Geometry and emission as k_affinity_rect: a workgroup owns a
// tile of 128 view rows x TW columns, fills an LDS image of it and writes the tile's slices from the
// image (slice_emit_lds) — the same bytes the rectangular fill writes, values bit for bit (both are
// the stored roundings of the same scores). The 128 rows of view chunk I are rows rowmap[128 I ..] of
// M, ascending: they lie in M's chunks k0 .. k1 = those rows / 128, and wave w walks the slices
// (column group of the tile, chunk k0 + w, k0 + w + 8, ...): an entry whose row has a position in this
// view chunk (viewpos) goes into the image. Different chunks are different rows: no two waves write
// the same element; a column's row mask is or-ed with LDS atomics.
// ------------------------------------------------------------------------------------------
struct FilterGeom {
  int64_t nrows;           // rows of the view
  const int32_t* rowmap;   // [nrows]
  const int32_t* viewpos;  // [m padded] position of a row of M in the view, or -1
  int nTc;                 // column tiles
  int64_t tile0;           // first tile of this launch (as RectGeom::tile0)
};

// One slice per tile (64 columns: a 33 KB image of fp32 values, four workgroups per CU); the walk keeps
// FILT_D steps of a slice and the header of the wave's next slice in flight (every lane issues every load,
// as in the pass: k_slices.hip.h).
constexpr int FILT_TW = 64;
constexpr int FILT_D = 4;
template <typename VT>
constexpr int filt_img_bytes() { return (AT * (FILT_TW + 1) * static_cast<int>(sizeof(VT)) + 15) / 16 * 16; }
template <typename VT>
constexpr int filt_lds_bytes() { return filt_img_bytes<VT>() + 64 + FILT_TW * 16; }

template <typename VT>
__global__ __launch_bounds__(AT_WAVES * 64, AT_WAVES / 2) void k_slice_filter_rows(SliceView M, FilterGeom G,
                                                                                     SliceOut O) {
  constexpr int PITCH = FILT_TW + 1;
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  static_assert(AT == SL_SUB, "a tile is as tall as a slice");
  extern __shared__ __attribute__((aligned(16))) char filt_smem[];
  VT* img = reinterpret_cast<VT*>(filt_smem);
  unsigned long long* base_s = reinterpret_cast<unsigned long long*>(filt_smem + filt_img_bytes<VT>());
  uint32_t* colmask = reinterpret_cast<uint32_t*>(filt_smem + filt_img_bytes<VT>() + 64);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = G.tile0 + blockIdx.x;
  const int I = static_cast<int>(tile / G.nTc);
  const int cg = static_cast<int>(tile % G.nTc);
  const int64_t v0 = static_cast<int64_t>(I) * AT;  // first view row of the tile
  for (int t = threadIdx.x; t < FILT_TW * 4; t += AT_WAVES * 64) colmask[t] = 0;
  const int64_t vlast = (v0 + AT - 1 < G.nrows) ? v0 + AT - 1 : G.nrows - 1;
  const int k0 = G.rowmap[v0] / SL_SUB, k1 = G.rowmap[vlast] / SL_SUB;
  __syncthreads();
  if (cg < M.ncg) {
    const gbytes_t data = (gbytes_t)M.data;
    const CLIPPER_GLOBAL uint64_t* pre = (const CLIPPER_GLOBAL uint64_t*)M.Pre + static_cast<int64_t>(cg) * M.nchunks;
    SliceHead<1> cur;
    int k = k0 + wave;
    if (k <= k1) cur.load(data + 16 * pre[k], lane);
    for (; k <= k1; k += AT_WAVES) {  // (wave-uniform)
      SliceHead<1> nxt = cur;
      if (k + AT_WAVES <= k1) nxt.load(data + 16 * pre[k + AT_WAVES], lane);
      const int64_t r0 = static_cast<int64_t>(k) * SL_SUB;
      const int maxq = __builtin_amdgcn_readfirstlane(cur.maxq);
      const int tot = cur.nq[0];
      gbytes_t fbase = cur.sp + 16 + 64 + sl_so_bytes(maxq);
      for (int qb = 0; qb < maxq; qb += FILT_D) {
        SliceQuad<VT> vq[FILT_D];
        uint32_t rq[FILT_D];
#pragma unroll
        for (int j = 0; j < FILT_D; ++j) {
          const bool active = qb + j < tot;
          const uint64_t mask = __ballot(active);
          const int cnt = __builtin_amdgcn_readfirstlane(__popcll(mask));
          const uint32_t rank = active ? sl_lane_rank(mask) : 0u;
          vq[j].load(fbase + rank * QB);
          rq[j] = *reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(fbase + cnt * QB + rank * 4);
          fbase += cnt * QB + ((cnt * 4 + 15) & ~15);
        }
#pragma unroll
        for (int j = 0; j < FILT_D; ++j) {
          if (qb + j < tot) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (vq[j].v[e] != VT(0)) {
                const int64_t vr = static_cast<int64_t>(G.viewpos[r0 + ((rq[j] >> (8 * e)) & 255u)]) - v0;
                if (vr >= 0 && vr < AT) {  // (-1 - v0 < 0: a row outside the view)
                  img[vr * PITCH + lane] = vq[j].v[e];
                  atomicOr(&colmask[lane * 4 + (static_cast<int>(vr) >> 5)], 1u << (static_cast<int>(vr) & 31));
                }
              }
          }
        }
      }
      cur = nxt;
    }
  }
  __syncthreads();
  // ---- the tile's slice (column group cg, chunk I), written by waves 0 and 4 as the two halves --------
  const int sl = wave & 3, half = wave >> 2;
  const VT* col = img + lane;
  const uint4 mk = *reinterpret_cast<const uint4*>(colmask + lane * 4);
  const uint64_t mlo = static_cast<uint64_t>(mk.x) | (static_cast<uint64_t>(mk.y) << 32);
  const uint64_t mhi = static_cast<uint64_t>(mk.z) | (static_cast<uint64_t>(mk.w) << 32);
  const int64_t s = (sl == 0 && cg < O.ncg && I < O.nchunks) ? static_cast<int64_t>(cg) * O.nchunks + I : -1;
  slice_emit_lds<VT>(col, PITCH, mlo, mhi, s, sl, half, O, base_s, nullptr);
}

// the host has built the view the state asked for (or refused: too many rows once the pending
// candidates were counted in): the hold is lifted, the policy's counters move on
__global__ void k_rv_resume(SolverState* st, SolveShared* shared, int refused_rows, uint32_t* ctl16 = nullptr) {
  // (the control words of the resident launch that may follow this build — error word, arrivals, unit 0's start —
  // zeroed here, on the way: no memset in front of the launch, none at the start of a solve)
  if (ctl16 != nullptr && blockIdx.x == 0 && threadIdx.x < 16) ctl16[threadIdx.x] = 0u;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    shared->hold = 0;
    st->hold = 0;
    if (refused_rows == 0) st->rv_builds += 1;  // a refusal does not use up one of the solve's views
    st->rv_last = static_cast<int32_t>(st->n_iters) - st->n_redo;  // (the policy's clock: SolverState::n_redo)
    st->rv_backoff = refused_rows;
  }
}

}  // namespace clipper_hip
