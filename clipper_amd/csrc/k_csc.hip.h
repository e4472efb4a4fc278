// k_csc.hip.h — producers of the slices: slice_emit_lds (k_affinity_sym writes them itself) and
// "groups" (gr_emit, k_groups_from_dense), the intermediate the slice packers read (GroupSource).
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_slices.hip.h"

namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// M of a registration problem is sparse by construction (the consistent pairs: ~11 % at the
// headline configuration); the solver streams it as slices (k_slices.hip.h). The
// symmetric fill kernel k_affinity_sym writes finished slices itself (slice_emit_lds below: a
// slice is as tall as its tile). Every other producer — the strip fill kernels, setMatrixData,
// fp64 values, column shards — goes through a dense store and GROUPS: per (128-column strip s,
// 64-row block b) the nonzeros of every column, un-padded, column after column, as (value,
// row-in-block u8), written by k_groups_from_dense and repacked by three small launches
// (k_slice_count / scan / k_slice_pack).
//   Goff[g * GR_OFFS + cl]   u16: where column cl's list starts inside group g = s * nblocks + b;
//                            entry 128 = the group's entry count
//   Gpre[g]                  where the group starts in vals / rows, in units of 4 entries
// The space of a group is claimed with one atomic on the cursor of one of CSC_ARENAS arenas
// (same-address atomics serialise at ~25-50 ns each — thousands of groups on ONE cursor cost more
// than the build itself): the ORDER of the groups in memory varies from build to build; the
// content of a group does not, and the slices packed from them are a pure function of M. A
// build that does not fit an arena (always: the first one of a problem size, capacity 0) writes
// only the offsets and the totals; the host grows the buffers and builds again.
// ------------------------------------------------------------------------------------------
template <typename VT>
struct GroupOut {
  uint16_t* Goff;
  uint64_t* Gpre;
  VT* vals;
  uint8_t* rows;
  CscBuildCtl* ctl;
  int nblocks;
};

// What k_affinity_sym takes to write finished slices itself (Pre == null: not in use): a slice is
// as tall as its tile and half as wide, so a tile and its mirror image are four slices.
struct SliceOut {
  uint64_t* Pre;     // [ncg * nchunks] where the slice starts in `data`, in units of 16 bytes
  uint32_t* Lq;      // [ncg * nchunks] maxq | entries << 8
  uint8_t* data;
  CscBuildCtl* ctl;  // arena cursors in units of 16 bytes
  int nchunks, ncg;
  long long* stamps; // measurement only (may be null): per tile {start, scores done, masks, space
                     // claimed, written} on the 100 MHz wall clock
};
typedef SliceOut CscOut;

// Emission of NG groups by one workgroup of NG*128 threads: thread t owns column (t & 127) of
// group (t >> 7); g = its group id or -1 (nothing to emit: the whole group, uniformly).
// gr_claim: the column's offset inside the group (prefix over the 128 columns) and the group's
// space, claimed with one atomic. `red` [2*NG] ints and `base_s` [NG] are LDS scratch. Contains
// barriers. Returns false for a thread that has nothing to write.
template <int NG, typename VT>
__device__ __forceinline__ bool gr_claim(int cnt, int64_t g, const GroupOut<VT>& O, int* red,
                                         unsigned long long* base_s, int& off,
                                         unsigned long long& base) {
  const int t = threadIdx.x, gi = t >> 7, cl = t & 127;
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o);
    if ((t & 63) >= o) inc += v;
  }
  if ((t & 63) == 63) red[t >> 6] = inc;
  __syncthreads();
  const int first = red[2 * gi], total = first + red[2 * gi + 1];
  off = inc - cnt + ((cl >= 64) ? first : 0);
  if (g >= 0) {
    O.Goff[g * GR_OFFS + cl] = static_cast<uint16_t>(off);
    if (cl == 127) O.Goff[g * GR_OFFS + 128] = static_cast<uint16_t>(total);
  }
  if (cl == 0 && g >= 0) {
    const unsigned long long U = static_cast<unsigned long long>((total + 3) >> 2);
    CscArena* ar = O.ctl + static_cast<int>((g * 11 + (g >> 6)) & (CSC_ARENAS - 1));
    unsigned long long bb = atomicAdd(&ar->cursor, U);
    if (bb + U > ar->capacity) {
      ar->overflow = 1;
      bb = ~0ull;
    } else {
      bb += ar->origin;
    }
    O.Gpre[g] = bb;
    base_s[gi] = bb;
  }
  __syncthreads();
  if (g < 0) return false;
  base = base_s[gi];
  return base != ~0ull;
}

// from 64 values held in registers (k_groups_from_dense)
template <int NG, typename VT>
__device__ __forceinline__ void gr_emit(const VT (&v)[GR_RB], int64_t g, const GroupOut<VT>& O,
                                        int* red, unsigned long long* base_s) {
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < GR_RB; ++q) cnt += (v[q] != VT(0)) ? 1 : 0;
  int off;
  unsigned long long base;
  if (!gr_claim<NG>(cnt, g, O, red, base_s, off, base)) return;
  VT* vp = O.vals + base * 4 + off;
  uint8_t* rp = O.rows + base * 4 + off;
  int k = 0;
#pragma unroll
  for (int q = 0; q < GR_RB; ++q)
    if (v[q] != VT(0)) {
      vp[k] = v[q];
      rp[k] = static_cast<uint8_t>(q);
      ++k;
    }
}

// Emission of one slice (k_slices.hip.h) by TWO waves of a workgroup straight from an LDS image:
// lane = column, col[q * stride] = its entry of row q of the tile (q < SL_SUB = 128), (mlo, mhi) =
// the bit mask of its nonzero rows (set with LDS atomics while the image was filled: sweeping the
// 128 rows of every column for them cost more than the writing). Both waves
// know where every step starts (the lanes' lengths: ALU only); wave `half` writes the steps of
// its parity. `s` = slice id cg * nchunks + k, or -1 (nothing to emit: both waves, uniformly).
// Space is claimed with one atomic per slice on one of the arena cursors; a slice that does not fit
// only records its size and maxq (the host grows the arena and repeats the fill).
// Contains ONE __syncthreads (every wave of the workgroup must call it). `base_s`: LDS, one
// unsigned long long per slice of the workgroup, this slice's at index `sl`.
template <typename VT>
__device__ __forceinline__ void slice_emit_lds(const VT* col, int stride, uint64_t mlo,
                                               uint64_t mhi, int64_t s, int sl, int half,
                                               const SliceOut& O, unsigned long long* base_s,
                                               long long* tstamp) {
  static_assert(SL_SUB == 128, "a slice is as tall as a tile of the fill kernels");
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  const int lane = threadIdx.x & 63;
  const int n = __popcll(mlo) + __popcll(mhi);
  const int tot = (n + 3) >> 2;
  const int maxq = sl_wave_max(tot);
  const uint32_t bytes = 16 + 64 + sl_so_bytes(maxq) + sl_steps_bytes(tot, maxq, QB);
  if (tstamp) tstamp[0] = wall_clock64() + (bytes == 1 ? 1 : 0);
  if (half == 0 && s >= 0) {
    const int nquads = sl_wave_sum(tot), entries = sl_wave_sum(n);
    if (lane == 0) {
      const unsigned long long U = bytes >> 4;
      CscArena* ar = O.ctl + static_cast<int>((s * 11 + (s >> 6)) & (CSC_ARENAS - 1));
      unsigned long long bb = atomicAdd(&ar->cursor, U);
      if (bb + U > ar->capacity) {
        ar->overflow = 1;
        bb = ~0ull;
      } else {
        bb += ar->origin;
      }
      O.Pre[s] = bb;
      O.Lq[s] = static_cast<uint32_t>(maxq) | (static_cast<uint32_t>(entries) << 8);
      base_s[sl] = bb;
      if (bb != ~0ull) {
        uint32_t* hd = reinterpret_cast<uint32_t*>(O.data + 16 * bb);
        hd[0] = static_cast<uint32_t>(nquads);
        hd[1] = static_cast<uint32_t>(maxq);
        hd[2] = bytes;
        hd[3] = 0;
      }
    }
  }
  __syncthreads();
  if (tstamp) tstamp[1] = wall_clock64();
  if (s < 0) return;
  const unsigned long long base = base_s[sl];
  if (base == ~0ull) return;
  uint8_t* sp = O.data + 16 * base;
  uint32_t* so = reinterpret_cast<uint32_t*>(sp + 16 + 64);
  const int sob = sl_so_bytes(maxq);
  if (half == 0) {
    sp[16 + lane] = static_cast<uint8_t>(tot);
    if (lane * 4 < sob && lane >= (maxq + SL_SO - 1) / SL_SO) so[lane] = 0;  // the table's padding
  }
  uint32_t off = 16 + 64 + sob;
  for (int q = 0; q < maxq; ++q) {
    const bool active = q < tot;
    const uint64_t mask = __ballot(active);
    const int cnt = __popcll(mask);
    const bool own = (q & 1) == half;  // uniform
    if (own && (q % SL_SO) == 0 && lane == 0) so[q / SL_SO] = off;
    if (active) {
      SliceQuad<VT> vq;
      vq.v[0] = vq.v[1] = vq.v[2] = vq.v[3] = VT(0);
      uint32_t rq = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int row = -1;
        if (mlo) {
          row = __ffsll(static_cast<unsigned long long>(mlo)) - 1;
          mlo &= mlo - 1;
        } else if (mhi) {
          row = 64 + __ffsll(static_cast<unsigned long long>(mhi)) - 1;
          mhi &= mhi - 1;
        }
        if (own && row >= 0) {
          vq.v[e] = col[row * stride];
          rq |= static_cast<uint32_t>(row) << (8 * e);
        }
      }
      if (own) {
        const uint32_t rank = sl_lane_rank(mask);
        vq.store(sp + off + rank * QB);
        *reinterpret_cast<uint32_t*>(sp + off + cnt * QB + rank * 4) = rq;
      }
    }
    if (own) {  // zero the step's padding (the row quads are padded to 16 bytes)
      const int padw = (((cnt * 4 + 15) & ~15) - cnt * 4) >> 2;
      if (lane < padw) *reinterpret_cast<uint32_t*>(sp + off + cnt * QB + cnt * 4 + lane * 4) = 0;
    }
    off += cnt * QB + ((cnt * 4 + 15) & ~15);
  }
}

// k_groups_from_dense — from a dense store: two groups (row blocks 2y, 2y+1 of strip x) per
// workgroup, one column per thread, the 64 rows of the block loaded at once.
template <typename VT>
__global__ __launch_bounds__(256) void k_groups_from_dense(const VT* __restrict__ S, int64_t ld,
                                                            int64_t m, GroupOut<VT> O) {
  __shared__ int red[4];
  __shared__ unsigned long long base_s[2];
  const int s = blockIdx.x, t = threadIdx.x;
  const int b = 2 * blockIdx.y + (t >> 7);
  const int64_t c = static_cast<int64_t>(s) * GR_CW + (t & 127);
  const int64_t r0 = static_cast<int64_t>(b) * GR_RB;
  VT v[GR_RB];
#pragma unroll
  for (int q = 0; q < GR_RB; ++q) {
    const int64_t r = r0 + q;
    v[q] = (c < ld && r < m) ? S[r * ld + c] : VT(0);
  }
  const int64_t g = (b < O.nblocks) ? static_cast<int64_t>(s) * O.nblocks + b : -1;
  gr_emit<2>(v, g, O, red, base_s);
}

}  // namespace clipper_hip
