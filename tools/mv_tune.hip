// mv_tune.hip — stand-alone feasibility harness for the V-vector ("window") mat-vec: ONE pass
// over M multiplies V candidate vectors at once (a_v = M x_v, b_v = pattern(M) x_v), trading
// idle fp64 VALU for HBM passes. Times V x geometry variants on MI355X. Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mv_tune.hip -o /tmp/mv_tune
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

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

constexpr int VS = 8;  // doubles per row of the interleaved candidate table X[row][VS]

template <int V, int NW, int UNR, int WPS /*waves per SIMD the register budget is cut for*/>
__global__ __launch_bounds__(NW * 64, WPS) void k_mv(const float* __restrict__ S, int64_t ld,
                                                      int64_t m, int rows_per_tile,
                                                      const double* __restrict__ X,
                                                      double* __restrict__ part) {
  __shared__ double lds[NW * 512];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_tile;
  const int64_t r1 = (r0 + rows_per_tile < m) ? r0 + rows_per_tile : m;
  double aa[V][4], bb[V][4];
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) aa[v][e] = bb[v][e] = 0.0;
  if (col < ld) {
    const float* p = S + col;
    for (int64_t r = r0 + static_cast<int64_t>(wave) * UNR; r + UNR <= r1;
         r += static_cast<int64_t>(NW) * UNR) {
      float4 t[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) t[q] = *reinterpret_cast<const float4*>(p + (r + q) * ld);
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const double* xr = X + (r + q) * VS;
        const double mm[4] = {(double)t[q].x, (double)t[q].y, (double)t[q].z, (double)t[q].w};
        const double ii[4] = {t[q].x != 0.f ? 1.0 : 0.0, t[q].y != 0.f ? 1.0 : 0.0,
                              t[q].z != 0.f ? 1.0 : 0.0, t[q].w != 0.f ? 1.0 : 0.0};
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const double xv = xr[v];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            aa[v][e] = fma(mm[e], xv, aa[v][e]);
            bb[v][e] = fma(ii[e], xv, bb[v][e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    double* mine = lds + wave * 512 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mine[e] = aa[v][e];
      mine[256 + e] = bb[v][e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 512; t += NW * 64) {
      double acc = lds[t];
#pragma unroll
      for (int w = 1; w < NW; ++w) acc += lds[w * 512 + t];
      const int which = t >> 8;
      const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + (t & 255);
      if (c < ld) part[((static_cast<int64_t>(blockIdx.y) * V + v) * 2 + which) * ld + c] = acc;
    }
    __syncthreads();
  }
}


// variant B: 2 columns per lane (8-byte loads), strip = 128 columns
template <int V, int NW, int UNR, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void k_mv2(const float* __restrict__ S, int64_t ld,
                                                       int64_t m, int rows_per_tile,
                                                       const double* __restrict__ X,
                                                       double* __restrict__ part) {
  __shared__ double lds[NW * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 128 + lane * 2;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_tile;
  const int64_t r1 = (r0 + rows_per_tile < m) ? r0 + rows_per_tile : m;
  double aa[V][2], bb[V][2];
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int e = 0; e < 2; ++e) aa[v][e] = bb[v][e] = 0.0;
  if (col < ld) {
    const float* p = S + col;
    for (int64_t r = r0 + static_cast<int64_t>(wave) * UNR; r + UNR <= r1;
         r += static_cast<int64_t>(NW) * UNR) {
      float2 t[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) t[q] = *reinterpret_cast<const float2*>(p + (r + q) * ld);
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const double* xr = X + (r + q) * VS;
        const double mm[2] = {(double)t[q].x, (double)t[q].y};
        const double ii[2] = {t[q].x != 0.f ? 1.0 : 0.0, t[q].y != 0.f ? 1.0 : 0.0};
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const double xv = xr[v];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            aa[v][e] = fma(mm[e], xv, aa[v][e]);
            bb[v][e] = fma(ii[e], xv, bb[v][e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    double* mine = lds + wave * 256 + lane * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      mine[e] = aa[v][e];
      mine[128 + e] = bb[v][e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 256; t += NW * 64) {
      double acc = lds[t];
#pragma unroll
      for (int w = 1; w < NW; ++w) acc += lds[w * 256 + t];
      const int which = t >> 7;
      const int64_t c = static_cast<int64_t>(blockIdx.x) * 128 + (t & 127);
      if (c < ld) part[((static_cast<int64_t>(blockIdx.y) * V + v) * 2 + which) * ld + c] = acc;
    }
    __syncthreads();
  }
}

// variant C: the V vectors are split over G groups of waves; every group streams ALL rows of
// the tile (the second read of a row segment is an L1/L2 hit) and carries V/G vectors
template <int V, int G, int NW, int UNR, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void k_mvs(const float* __restrict__ S, int64_t ld,
                                                       int64_t m, int rows_per_tile,
                                                       const double* __restrict__ X,
                                                       double* __restrict__ part) {
  constexpr int VG = V / G;       // vectors per group
  constexpr int WG_ = NW / G;     // waves per group
  __shared__ double lds[NW * 512];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave / WG_, wig = wave % WG_;
  const int64_t col = static_cast<int64_t>(blockIdx.x) * 256 + lane * 4;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_tile;
  const int64_t r1 = (r0 + rows_per_tile < m) ? r0 + rows_per_tile : m;
  double aa[VG][4], bb[VG][4];
#pragma unroll
  for (int v = 0; v < VG; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) aa[v][e] = bb[v][e] = 0.0;
  if (col < ld) {
    const float* p = S + col;
    for (int64_t r = r0 + static_cast<int64_t>(wig) * UNR; r + UNR <= r1;
         r += static_cast<int64_t>(WG_) * UNR) {
      float4 t[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) t[q] = *reinterpret_cast<const float4*>(p + (r + q) * ld);
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const double* xr = X + (r + q) * VS + grp * VG;
        const double mm[4] = {(double)t[q].x, (double)t[q].y, (double)t[q].z, (double)t[q].w};
        const double ii[4] = {t[q].x != 0.f ? 1.0 : 0.0, t[q].y != 0.f ? 1.0 : 0.0,
                              t[q].z != 0.f ? 1.0 : 0.0, t[q].w != 0.f ? 1.0 : 0.0};
#pragma unroll
        for (int v = 0; v < VG; ++v) {
          const double xv = xr[v];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            aa[v][e] = fma(mm[e], xv, aa[v][e]);
            bb[v][e] = fma(ii[e], xv, bb[v][e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VG; ++v) {
    double* mine = lds + wave * 512 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mine[e] = aa[v][e];
      mine[256 + e] = bb[v][e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 512 * G; t += NW * 64) {
      const int g = t / 512, tt = t % 512;
      double acc = lds[g * WG_ * 512 + tt];
#pragma unroll
      for (int w = 1; w < WG_; ++w) acc += lds[(g * WG_ + w) * 512 + tt];
      const int which = tt >> 8;
      const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + (tt & 255);
      if (c < ld)
        part[((static_cast<int64_t>(blockIdx.y) * V + g * VG + v) * 2 + which) * ld + c] = acc;
    }
    __syncthreads();
  }
}

struct Ctx {
  float* S;
  double* X;
  double* part;
  int64_t m, ld;
  int cus;
};

typedef void (*kern_t)(const float*, int64_t, int64_t, int, const double*, double*);
void run_k(const Ctx& c, kern_t k, int V, int NW, int UNR, int WPS, int strip, int rowwaves,
           int wg_per_cu, const char* note);
template <int V, int NW, int UNR, int WPS>
void run(const Ctx& c, int wg_per_cu, const char* note) {
  run_k(c, k_mv<V, NW, UNR, WPS>, V, NW, UNR, WPS, 256, NW, wg_per_cu, note);
}
void run_k(const Ctx& c, kern_t k, int V, int NW, int UNR, int WPS, int strip, int rowwaves,
           int wg_per_cu, const char* note) {
  const int64_t chunk = (int64_t)rowwaves * UNR;
  const int nstrips = (int)((c.ld + strip - 1) / strip);
  int64_t target = (int64_t)c.cus * wg_per_cu;
  int64_t nt = (target + nstrips - 1) / nstrips;
  if (nt < 1) nt = 1;
  int64_t rpt = ((c.m + nt - 1) / nt + chunk - 1) / chunk * chunk;
  int ntiles = (int)((c.m + rpt - 1) / rpt);
  dim3 grid(nstrips, ntiles), block(NW * 64);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL(k, grid, block, 0, 0, c.S, c.ld, c.m, (int)rpt, c.X, c.part);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, grid, block, 0, 0, c.S, c.ld, c.m, (int)rpt, c.X, c.part);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = 4.0 * c.m * c.m;
  printf("V%d NW%d UNR%d wps%d wg/cu %d ntiles %3d  avg %8.2f us  min %8.2f us  %7.1f GB/s  %s\n", V,
         NW, UNR, WPS, wg_per_cu, ntiles, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e9,
         note);
  fflush(stdout);
}

int main(int argc, char** argv) {
  Ctx c;
  c.m = argc > 1 ? atoll(argv[1]) : 10000;
  c.ld = (c.m + 63) / 64 * 64;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  c.cus = prop.multiProcessorCount;
  CK(hipMalloc(&c.S, (size_t)c.m * c.ld * 4));
  CK(hipMalloc(&c.X, (size_t)c.ld * VS * 8 + 4096));
  CK(hipMalloc(&c.part, (size_t)64 * 8 * 2 * c.ld * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.S, c.m * c.ld, 0.11f);
  std::vector<double> x((size_t)c.ld * VS);
  for (size_t i = 0; i < x.size(); ++i) x[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
  CK(hipMemcpy(c.X, x.data(), x.size() * 8, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  printf("m = %lld, %d CUs\n", (long long)c.m, c.cus);
  run<1, 8, 8, 4>(c, 2, "baseline shape");
  run<5, 8, 4, 4>(c, 2, "");
  run<5, 8, 8, 4>(c, 2, "");
  run<6, 8, 2, 4>(c, 2, "");
  run_k(c, k_mv2<6, 8, 8, 4>, 6, 8, 8, 4, 128, 8, 2, "2 col/lane");
  run_k(c, k_mv2<6, 8, 16, 4>, 6, 8, 16, 4, 128, 8, 2, "2 col/lane");
  run_k(c, k_mv2<6, 8, 8, 4>, 6, 8, 8, 4, 128, 8, 4, "2 col/lane");
  run_k(c, k_mv2<8, 8, 8, 4>, 8, 8, 8, 4, 128, 8, 2, "2 col/lane");
  run_k(c, k_mv2<4, 8, 16, 4>, 4, 8, 16, 4, 128, 8, 2, "2 col/lane");
  run_k(c, k_mvs<6, 2, 8, 8, 4>, 6, 8, 8, 4, 256, 4, 2, "split 2 groups");
  run_k(c, k_mvs<6, 2, 16, 8, 4>, 6, 16, 8, 4, 256, 8, 1, "split 2 groups NW16");
  run_k(c, k_mvs<8, 2, 8, 8, 4>, 8, 8, 8, 4, 256, 4, 2, "split 2 groups");
  run_k(c, k_mvs<6, 3, 12, 8, 4>, 6, 12, 8, 4, 256, 4, 1, "split 3 groups NW12");
  run<2, 8, 8, 4>(c, 2, "");
  run<3, 8, 8, 4>(c, 2, "");
  run<4, 8, 8, 4>(c, 2, "");
  run<4, 8, 8, 2>(c, 1, "");
  run<4, 8, 16, 2>(c, 1, "");
  run<6, 8, 8, 2>(c, 1, "");
  run<6, 8, 16, 2>(c, 1, "");
  run<6, 8, 4, 4>(c, 2, "");
  run<6, 16, 8, 4>(c, 1, "");
  run<6, 4, 16, 2>(c, 2, "");
  run<8, 8, 8, 2>(c, 1, "");
  run<8, 8, 16, 2>(c, 1, "");
  run<3, 8, 16, 2>(c, 1, "");
  run<2, 8, 16, 4>(c, 2, "");
  return 0;
}
