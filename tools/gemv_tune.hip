// gemv_tune.hip — stand-alone tuning harness for the fused mat-vec kernel (k_gemv) on MI355X.
// Not part of the product; it includes the product kernels and times geometry variants so
// that the constants in clipper_hip.hip are chosen from measurements.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gemv_tune.hip -o gemv_tune
//   ./gemv_tune [m]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../clipper_amd/csrc/kernels.hip.h"

using namespace clipper_hip;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void k_fill(float* S, int64_t n, float density) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    float u = (h & 0xFFFFFF) / 16777216.0f;
    float v = ((h >> 24) & 0xFFFFFF) / 16777216.0f;
    S[i] = (u < density) ? v : 0.0f;
  }
}

// streaming ceiling: every lane sums float4s of one contiguous chunk (pure HBM read)
__global__ __launch_bounds__(256) void k_stream_sum(const float4* __restrict__ p, int64_t n4,
                                                     float* out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y +
           d.z + d.w;
  }
  for (; i < n4; i += stride) {
    float4 a = p[i];
    acc += a.x + a.y + a.z + a.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p, bool nt) {
  const v4f* q = reinterpret_cast<const v4f*>(p);
  v4f v = nt ? __builtin_nontemporal_load(q) : *q;
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---- experimental variant: NT loads / wider lane footprint ---------------------------------
template <int NW, int UNR, int CPL /*float4 per lane: 1,2,4*/, bool NT>
__global__ __launch_bounds__(NW * 64) void k_gemv_x(const float* __restrict__ S, int64_t ld,
                                                     int64_t m, int rows_per_tile,
                                                     const double* __restrict__ x,
                                                     double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int SW = 256 * CPL;  // strip width
  const int64_t col0 = static_cast<int64_t>(blockIdx.x) * SW + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_tile;
  const int64_t r1 = (r0 + rows_per_tile < m) ? r0 + rows_per_tile : m;
  double aa[CPL][4], bb[CPL][4];
#pragma unroll
  for (int c = 0; c < CPL; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) aa[c][e] = bb[c][e] = 0.0;

  int64_t r = r0 + static_cast<int64_t>(wave) * UNR;
  for (; r + UNR <= r1; r += static_cast<int64_t>(NW) * UNR) {
    float4 v[UNR][CPL];
#pragma unroll
    for (int q = 0; q < UNR; ++q)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int64_t col = col0 + c * 256;
        if (col < ld) {
          v[q][c] = ld4(S + (r + q) * ld + col, NT);
        } else {
          v[q][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const double xr = x[r + q];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const float f[4] = {v[q][c].x, v[q][c].y, v[q][c].z, v[q][c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          aa[c][e] = fma(static_cast<double>(f[e]), xr, aa[c][e]);
          bb[c][e] += (f[e] != 0.f) ? xr : 0.0;
        }
      }
    }
  }
  for (int q = 0; q < UNR; ++q) {
    const int64_t rr = r + q;
    if (rr < r1) {
      const double xr = x[rr];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int64_t col = col0 + c * 256;
        if (col < ld) {
          const float4 v = *reinterpret_cast<const float4*>(S + rr * ld + col);
          const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            aa[c][e] = fma(static_cast<double>(f[e]), xr, aa[c][e]);
            bb[c][e] += (f[e] != 0.f) ? xr : 0.0;
          }
        }
      }
    }
  }
  // LDS combine: layout [wave][2][SW]
  double* mine = lds + static_cast<int64_t>(wave) * 2 * SW;
#pragma unroll
  for (int c = 0; c < CPL; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mine[c * 256 + lane * 4 + e] = aa[c][e];
      mine[SW + c * 256 + lane * 4 + e] = bb[c][e];
    }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * SW; t += NW * 64) {
    double acc = lds[t];
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += lds[static_cast<int64_t>(w) * 2 * SW + t];
    const int which = t / SW;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * SW + (t % SW);
    if (c < ld) part[(static_cast<int64_t>(blockIdx.y) * 2 + which) * ld + c] = acc;
  }
}

struct Plan {
  int nstrips, ntiles, rows_per_tile;
};
static int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

Plan plan(int64_t m, int64_t ld, int sw, int nw, int unr, int wg_per_cu, int cus = 256) {
  Plan p;
  p.nstrips = (int)cdiv(ld, sw);
  int64_t chunk = (int64_t)nw * unr;
  int64_t target = (int64_t)cus * wg_per_cu;
  int64_t nt = std::max<int64_t>(1, cdiv(target, p.nstrips));
  nt = std::min<int64_t>(nt, std::max<int64_t>(1, cdiv(m, chunk)));
  int64_t rpt = rup(cdiv(m, nt), chunk);
  p.rows_per_tile = (int)rpt;
  p.ntiles = (int)cdiv(m, rpt);
  return p;
}

template <typename F>
void timeit(const char* name, double bytes, hipStream_t st, int reps, F launch, int ntiles) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(st));
  float best = 1e30f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, st));
    launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
    sum += ms;
  }
  CK(hipGetLastError());
  printf("%-44s ntiles %4d  avg %8.2f us  min %8.2f us  %7.1f GB/s (min: %7.1f)\n", name, ntiles,
         sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
}

template <int NW, int UNR, int CPL, bool NT>
void run_x(const char* tag, const float* S, int64_t ld, int64_t m, const double* x, double* part,
           hipStream_t st, int wgpcu, double bytes) {
  Plan p = plan(m, ld, 256 * CPL, NW, UNR, wgpcu);
  char name[128];
  snprintf(name, sizeof name, "%s NW%d UNR%d CPL%d NT%d wg/cu %d", tag, NW, UNR, CPL, (int)NT, wgpcu);
  size_t lds = (size_t)NW * 2 * 256 * CPL * sizeof(double);
  if (hipFuncSetAttribute((const void*)k_gemv_x<NW, UNR, CPL, NT>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    printf("%-44s skipped (LDS %zu B not accepted)\n", name, lds);
    return;
  }
  timeit(name, bytes, st, 20, [&] {
    hipLaunchKernelGGL((k_gemv_x<NW, UNR, CPL, NT>), dim3(p.nstrips, p.ntiles), dim3(NW * 64), lds,
                       st, S, ld, m, p.rows_per_tile, x, part);
  }, p.ntiles);
}

int main(int argc, char** argv) {
  int64_t m = (argc > 1) ? atoll(argv[1]) : 10000;
  int64_t ld = rup(m, 64);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float* S;
  double *x, *part, *ab;
  CK(hipMalloc(&S, (size_t)m * ld * 4));
  CK(hipMalloc(&x, (size_t)ld * 8));
  CK(hipMalloc(&part, (size_t)4096 * 2 * ld * 8));
  CK(hipMalloc(&ab, (size_t)2 * ld * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, S, m * ld, 0.13f);
  std::vector<double> hx(ld, 0.0);
  for (int64_t i = 0; i < m; ++i) hx[i] = (double)rand() / RAND_MAX;
  CK(hipMemcpy(x, hx.data(), ld * 8, hipMemcpyHostToDevice));
  CK(hipStreamSynchronize(st));
  const double bytes = 4.0 * m * m;
  printf("m=%lld ld=%lld bytes/pass=%.1f MB\n", (long long)m, (long long)ld, bytes / 1e6);

  float* dummy;
  CK(hipMalloc(&dummy, 64));
  for (int g : {1024, 2048, 4096, 8192})
    timeit("stream_sum (pure read ceiling)", bytes, st, 20, [&] {
      hipLaunchKernelGGL(k_stream_sum, dim3(g), dim3(256), 0, st, (const float4*)S, m * ld / 4, dummy);
    }, g);

  // product kernel at its current geometry
  {
    Plan p = plan(m, ld, 256, 4, 8, 8);
    timeit("product k_gemv<float,false,4,8> wg/cu 8", bytes, st, 20, [&] {
      hipLaunchKernelGGL((k_gemv<float, false, 4, 8>), dim3(p.nstrips, p.ntiles), dim3(256), 0, st,
                         S, (const float*)nullptr, ld, m, p.rows_per_tile, x, part,
                         (const SolverState*)nullptr);
    }, p.ntiles);
    timeit("product k_reduce", (double)p.ntiles * 2 * ld * 8, st, 20, [&] {
      hipLaunchKernelGGL(k_reduce, dim3((unsigned)cdiv(ld, 256)), dim3(256), 0, st, part, p.ntiles, ld,
                         ab, (const SolverState*)nullptr);
    }, p.ntiles);
  }

  for (int wg : {1, 2}) {
    run_x<16, 8, 1, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<16, 8, 1, true>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<16, 4, 1, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<16, 4, 2, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<16, 4, 2, true>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<16, 2, 4, false>("x", S, ld, m, x, part, st, wg, bytes);
  }
  for (int wg : {2, 4}) {
    run_x<8, 8, 1, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<8, 8, 1, true>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<8, 4, 2, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<8, 4, 2, true>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<8, 16, 1, false>("x", S, ld, m, x, part, st, wg, bytes);
  }
  for (int wg : {4, 8, 16}) {
    run_x<4, 8, 1, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<4, 8, 1, true>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<4, 16, 1, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<4, 4, 2, false>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<4, 4, 2, true>("x", S, ld, m, x, part, st, wg, bytes);
    run_x<4, 4, 4, false>("x", S, ld, m, x, part, st, wg, bytes);
  }
  return 0;
}
