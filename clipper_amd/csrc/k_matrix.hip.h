// k_matrix.hip.h — matrix upload and gather: k_from_dense_upper, k_from_csc, k_gather_sub
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>


namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// matrix upload (setMatrixData / setSparseMatrixData) — clipper.cpp:149-166
// ------------------------------------------------------------------------------------------

// S[j][c] = Mdense(min(j,g), max(j,g)) for g = c0+c != j, 0 on the diagonal / padding.
// Mdense is column-major m x m fp64 in device memory (only its strict upper triangle is
// read). `mismatch` is raised when Cdense's upper triangle differs from pattern(Mdense).
template <typename T>
__global__ __launch_bounds__(256) void k_from_dense_upper(T* __restrict__ S, int64_t ld,
                                                           int64_t m, int64_t c0,
                                                           const double* __restrict__ Md,
                                                           const double* __restrict__ Cd,
                                                           T* __restrict__ Cs,
                                                           int* __restrict__ mismatch) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= ld) return;
  const int64_t g = c0 + c;
  for (int64_t j = blockIdx.y; j < m; j += gridDim.y) {  // grid.y is capped at 65535 rows
    double mv = 0.0, cv = 0.0;
    if (g < m && g != j) {
      const int64_t lo = (j < g) ? j : g, hi = (j < g) ? g : j;
      mv = Md[lo + hi * m];
      cv = Cd[lo + hi * m];
      if (mismatch != nullptr) {
        const double want = (mv != 0.0) ? 1.0 : 0.0;
        if (cv != want) *mismatch = 1;
      }
    }
    T sv = static_cast<T>(mv);
    if (mv != 0.0 && sv == T(0)) sv = (mv > 0) ? static_cast<T>(1.17549435e-38)
                                               : static_cast<T>(-1.17549435e-38);
    S[j * ld + c] = sv;
    if (Cs != nullptr) Cs[j * ld + c] = static_cast<T>(cv);
  }
}

// out[a*k + b] = M(idx[a], idx[b]) for the columns idx[b] this slice owns (others untouched):
// the sub-matrix induced by the non-zero entries of u, for the exact DSD rounding on the host
template <typename T>
__global__ __launch_bounds__(256) void k_gather_sub(const T* __restrict__ S, int64_t ld,
                                                     int64_t c0, int64_t W,
                                                     const int32_t* __restrict__ idx, int k,
                                                     double* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= static_cast<int64_t>(k) * k) return;
  const int a = static_cast<int>(e / k), b = static_cast<int>(e - static_cast<int64_t>(a) * k);
  const int64_t col = idx[b];
  if (col >= c0 && col < c0 + W) out[e] = static_cast<double>(S[static_cast<int64_t>(idx[a]) * ld + (col - c0)]);
}

// scatter of strictly-upper CSC entries (both mirror images) into a zeroed slice
template <typename T>
__global__ __launch_bounds__(256) void k_from_csc(T* __restrict__ S, int64_t ld, int64_t m,
                                                   int64_t c0, int64_t W,
                                                   const int64_t* __restrict__ colptr,
                                                   const int32_t* __restrict__ row,
                                                   const double* __restrict__ val) {
  for (int64_t j = blockIdx.x; j < m; j += gridDim.x)  // CSC columns
    for (int64_t p = colptr[j] + threadIdx.x; p < colptr[j + 1]; p += 256) {
      const int64_t i = row[p];
      if (i == j) continue;  // the solver treats the diagonal as implicit identity
      const T v = static_cast<T>(val[p]);
      // element (i,j): lives at S[i][j-c0] if j is an owned column, and at S[j][i-c0] if i is
      if (j >= c0 && j < c0 + W) S[i * ld + (j - c0)] = v;
      if (i >= c0 && i < c0 + W) S[j * ld + (i - c0)] = v;
    }
}

// k_copy_words — a few KB of device memory into mapped host memory by the CUs themselves: the
// read-back of a build's directory. A DMA copy of the same bytes costs ~15 us of engine start-up
// behind the fill kernel; this launch ~4.
__global__ __launch_bounds__(256) void k_copy_words(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                    int64_t n16) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n16;
       i += static_cast<int64_t>(gridDim.x) * 256)
    dst[i] = src[i];
}

// k_debug_occupy — test infrastructure (clipper_hip_debug_occupy): every workgroup holds its dynamic LDS and one wave
// slot for `ticks` of the 100 MHz wall clock and does nothing else — "another tenant on the device" for the tests of
// the resident solvers' time-outs (their units need a CU's whole LDS each and wait for each other).
__global__ __launch_bounds__(64) void k_debug_occupy(long long ticks) {
  extern __shared__ __attribute__((aligned(16))) uint8_t occ_lds[];
  if (threadIdx.x == 0) occ_lds[0] = 1;  // (the allocation is what counts)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace clipper_hip
