// k_knn.hip.h — brute-force k-nearest-neighbour search between two small point clouds (SURVEY 8f
// rank 1: the putative-association generator BEFORE the hot path).
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
//
// Reference site: benchmarks/bm_utils.cpp:147-176 — a nanoflann kd-tree over pcd1, one
// KNNResultSet(knn) query per point of pcd0 (squared L2 distances, ascending). Here: no tree.
// n0 * n1 distance evaluations are a few hundred microseconds of fp64 VALU work at the sizes of
// the reference benchmark (10k x 10k) and there is nothing to build or to traverse:
//   k_knn_partial  grid (ceil(n0/256), S): thread = one query point, workgroup = one chunk of
//                  pcd1, read with SCALAR loads (all lanes look at the same point: no LDS tile,
//                  no barriers; as fast as an LDS-tiled version at K = 1, 1.4x at K = 8, 0.87x at
//                  K = 16); every thread keeps its K best of the chunk
//                  in registers (sorted insertion, strict '<': among equal distances the LOWER
//                  index of pcd1 stays in front — a kd-tree's order among exact ties is its
//                  traversal order, which nothing downstream may rely on)
//   k_knn_merge    thread = one query point: merges its S sorted partial lists in chunk order
// Distances are fp64, (q0-p0)^2 + (q1-p1)^2 + ... added in coordinate order, no fma.
// Measured (MI355X, rocprof): 4096 x 4096, K = 1: 52-70 us; 10k x 10k: 75 us (K = 1), 222 us
// (K = 8); 100k x 100k: 5.7 ms (K = 1) = 1.8e12 pairs/s, ~40 % of the fp64 VALU issue rate for
// the ~9 instructions a pair costs; 10.8 ms at K = 16.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clipper_hip {

constexpr int KNN_TILE = 1024;   // granularity of the chunks pcd1 is split into
constexpr int KNN_UNR = 8;       // points of pcd1 fetched per batch of scalar loads
constexpr int KNN_DMAX = 8;      // coordinates per point
constexpr double KNN_INF = 1.0e300;

template <int K, int D>
__global__ __launch_bounds__(256) void k_knn_partial(const double* __restrict__ P0, int64_t n0,
                                                      const double* __restrict__ P1, int64_t n1,
                                                      int64_t chunk, double* __restrict__ pd,
                                                      int32_t* __restrict__ pi) {
  // every lane of a wave looks at the SAME point of pcd1 at the same time: its coordinates come
  // through the constant address space as scalar loads (wave-uniform address), KNN_UNR points
  // per batch — no LDS tile, no vector memory traffic in the loop
  typedef const __attribute__((address_space(4))) double* cptr;
  const cptr Q1 = (cptr)P1;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * chunk;
  const int64_t j1 = (j0 + chunk < n1) ? j0 + chunk : n1;
  double q[D];
#pragma unroll
  for (int k = 0; k < D; ++k) q[k] = (i < n0) ? P0[i * D + k] : 0.0;
  double bd[K];
  int32_t bi[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bd[k] = KNN_INF;
    bi[k] = -1;
  }
  auto visit = [&](int64_t t, const double (&p)[D]) {
    double dist = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double df = q[k] - p[k];
      dist = dist + df * df;
    }
    if (dist < bd[K - 1]) {  // sorted insertion; equal distances keep the earlier (lower) index
      const int32_t jj = static_cast<int32_t>(t);
#pragma unroll
      for (int k = K - 1; k >= 0; --k) {
        const bool here = (k == 0) || !(dist < bd[k - 1]);
        if (dist < bd[k]) {
          if (here) {
            bd[k] = dist;
            bi[k] = jj;
          } else {
            bd[k] = bd[k - 1];
            bi[k] = bi[k - 1];
          }
        }
      }
    }
  };
  int64_t t = j0;
  for (; t + KNN_UNR <= j1; t += KNN_UNR) {
    double p[KNN_UNR][D];
#pragma unroll
    for (int u = 0; u < KNN_UNR; ++u)
#pragma unroll
      for (int k = 0; k < D; ++k) p[u][k] = Q1[(t + u) * D + k];
#pragma unroll
    for (int u = 0; u < KNN_UNR; ++u) visit(t + u, p[u]);
  }
  for (; t < j1; ++t) {
    double p[D];
#pragma unroll
    for (int k = 0; k < D; ++k) p[k] = Q1[t * D + k];
    visit(t, p);
  }
  if (i < n0) {
    const int64_t o = (static_cast<int64_t>(blockIdx.y) * n0 + i) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pd[o + k] = bd[k];
      pi[o + k] = bi[k];
    }
  }
}

// out[i][0..K) = the K best of the S partial lists of query i, chunk by chunk (ascending indices)
template <int K>
__global__ __launch_bounds__(256) void k_knn_merge(const double* __restrict__ pd,
                                                    const int32_t* __restrict__ pi, int64_t n0,
                                                    int S, double* __restrict__ od,
                                                    int32_t* __restrict__ oi) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n0) return;
  double bd[K];
  int32_t bi[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bd[k] = KNN_INF;
    bi[k] = -1;
  }
  for (int s = 0; s < S; ++s) {
    const int64_t o = (static_cast<int64_t>(s) * n0 + i) * K;
    for (int c = 0; c < K; ++c) {
      const double dist = pd[o + c];
      const int32_t jj = pi[o + c];
      if (jj < 0 || !(dist < bd[K - 1])) break;  // the list is sorted: nothing better follows
#pragma unroll
      for (int k = K - 1; k >= 0; --k) {
        const bool here = (k == 0) || !(dist < bd[k - 1]);
        if (dist < bd[k]) {
          if (here) {
            bd[k] = dist;
            bi[k] = jj;
          } else {
            bd[k] = bd[k - 1];
            bi[k] = bi[k - 1];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    od[i * K + k] = bd[k];
    oi[i * K + k] = bi[k];
  }
}

}  // namespace clipper_hip
