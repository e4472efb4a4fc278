// k_csc.hip.h — "groups": the column lists of M as the fill kernels emit them (csc_emit*,
// k_groups_from_dense), the intermediate the slice packers read (k_slices.hip.h: GroupSource).
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_slices.hip.h"

namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// M of a registration problem is sparse by construction (the consistent pairs: ~11 % at the
// headline configuration); the solver streams it as slices (k_slices.hip.h). A fill kernel
// cannot write slices directly: a slice spans 256 rows, a tile of k_affinity_sym 128, and where
// a lane's quad lands depends on the lengths of all 64 columns over the whole chunk. So the
// fill kernels emit GROUPS — per (128-column strip s, 64-row block b) the nonzeros of every
// column, un-padded, column after column, as (value, row-in-block u8) — and three small
// launches repack them (k_slice_count / scan / k_slice_pack). A group is as wide as a tile of
// k_affinity_sym, which emits the groups of the tiles it computes (and of their mirror images)
// straight from its LDS image; k_groups_from_dense does the same from a dense store (the other
// fill kernels, setMatrixData).
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
typedef GroupOut<float> CscOut;  // what k_affinity_sym takes (Goff == null: not in use)

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
  int off;
  unsigned long long base;
  if (!gr_claim<NG>(cnt, g, O, red, base_s, off, base)) return;
  float* vp = O.vals + base * 4 + off;
  uint8_t* rp = O.rows + base * 4 + off;
  int k = 0;
  while (mlo) {
    const int q = __ffs(mlo) - 1;
    mlo &= mlo - 1;
    vp[k] = col[q * stride];
    rp[k] = static_cast<uint8_t>(q);
    ++k;
  }
  while (mhi) {
    const int q = __ffs(mhi) - 1 + 32;
    mhi &= mhi - 1;
    vp[k] = col[q * stride];
    rp[k] = static_cast<uint8_t>(q);
    ++k;
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
