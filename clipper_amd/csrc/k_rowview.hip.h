// k_rowview.hip.h — the row list of a row view (k_solver.hip.h, LIVE ROWS): the rows that can be non-zero
// in the window the next pass multiplies, in ascending order.
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
__global__ __launch_bounds__(256) void k_rv_scatter(const uint8_t* __restrict__ flags, int64_t m,
                                                     const uint32_t* __restrict__ blk,
                                                     int32_t* __restrict__ rowmap, int64_t cap) {
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
  for (int k = 0; k < 4; ++k)
    if (f[k]) {
      if (static_cast<int64_t>(at) < cap) rowmap[at] = static_cast<int32_t>(base + k);
      ++at;
    }
}

// the host has built the view the state asked for (or refused: too many rows once the pending
// candidates were counted in): the hold is lifted, the policy's counters move on
__global__ void k_rv_resume(SolverState* st, SolveShared* shared, int refused_rows) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    shared->hold = 0;
    st->hold = 0;
    st->rv_builds += 1;
    st->rv_last = static_cast<int32_t>(st->n_iters);
    st->rv_backoff = refused_rows;
  }
}

}  // namespace clipper_hip
