// csc_tune.hip — stand-alone feasibility harness for a column-compressed copy of M for the
// window pass: per (256-column strip, 64-row block) group every column's nonzeros are stored as
// (row-in-block u8, value fp32), all 256 columns padded to the group's longest list (rounded up
// to 4), laid out [column-of-lane e][quad kq][lane][4] so a wave reads 1 KiB per instruction. A
// lane multiplies only ITS columns' nonzeros; the x rows of the block are staged in LDS.
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/csc_tune.hip -o /tmp/csc_tune
//   /tmp/csc_tune <m> [density | file.f32] [tile_blocks]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int VS = 8;
#ifndef CSC_RB
#define CSC_RB 64
#endif
constexpr int RB = CSC_RB;  // rows per block
#ifndef CSC_CW
#define CSC_CW 128
#endif
constexpr int CW = CSC_CW;     // columns per strip
constexpr int CPL = CW / 64;   // columns per lane

__global__ void k_rand(float* S, int64_t ld, int64_t m, float density) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < m * ld; i += stride) {
    int64_t r = i / ld, c = i % ld;
    if (c >= m || r == c) { S[i] = 0.f; continue; }
    int64_t lo = r < c ? r : c, hi = r < c ? c : r;
    uint64_t h = (uint64_t)(lo * m + hi) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    float u = (h & 0xFFFFFF) / 16777216.0f;
    float v = (((h >> 24) & 0xFFFFFF) + 1) / 16777216.0f;
    S[i] = (u < density) ? v : 0.0f;
  }
}

// ---- build ----------------------------------------------------------------------------------
__global__ __launch_bounds__(CW) void k_csc_count(const float* __restrict__ S, int64_t ld,
                                                    int64_t m, int nblocks,
                                                    uint32_t* __restrict__ Lc) {
  __shared__ int red[4];
  const int s = blockIdx.x, b = blockIdx.y;
  const int64_t c = (int64_t)s * CW + threadIdx.x;
  const int64_t r0 = (int64_t)b * RB;
  int cnt = 0;
  if (c < ld) {
#pragma unroll 8
    for (int q = 0; q < RB; ++q) {
      const int64_t r = r0 + q;
      if (r < m) cnt += (S[r * ld + c] != 0.f) ? 1 : 0;
    }
  }
  // wave max, block max
  for (int o = 32; o > 0; o >>= 1) {
    int other = __shfl_xor(cnt, o);
    cnt = cnt > other ? cnt : other;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int mx = red[0];
    for (int w = 1; w < CW / 64; ++w) mx = mx > red[w] ? mx : red[w];
    Lc[(int64_t)s * nblocks + b] = (uint32_t)((mx + 3) & ~3);
  }
}

__global__ __launch_bounds__(CW) void k_csc_fill(const float* __restrict__ S, int64_t ld,
                                                   int64_t m, int nblocks,
                                                   const uint32_t* __restrict__ Lc,
                                                   const uint64_t* __restrict__ Pre,
                                                   float* __restrict__ vals,
                                                   uint8_t* __restrict__ rows) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int64_t g = (int64_t)s * nblocks + b;
  const int L = (int)Lc[g];
  const int LQ = L >> 2;
  const int64_t base = (int64_t)Pre[g] * CW;  // entries before this group
  const int t = threadIdx.x, lane = t / CPL, e = t % CPL;
  const int64_t c = (int64_t)s * CW + t;
  const int64_t r0 = (int64_t)b * RB;
  float4* vq = reinterpret_cast<float4*>(vals + base);
  uint32_t* rq = reinterpret_cast<uint32_t*>(rows + base);
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t r4 = 0;
  int k = 0;
  if (c < ld) {
    for (int q = 0; q < RB; ++q) {
      const int64_t r = r0 + q;
      if (r >= m) break;
      const float v = S[r * ld + c];
      if (v != 0.f) {
        const int j = k & 3;
        v4[0] = j == 0 ? v : v4[0];
        v4[1] = j == 1 ? v : v4[1];
        v4[2] = j == 2 ? v : v4[2];
        v4[3] = j == 3 ? v : v4[3];
        r4 |= (uint32_t)q << (8 * j);
        ++k;
        if (j == 3) {
          const int kq = (k >> 2) - 1;
          vq[(e * LQ + kq) * 64 + lane] = make_float4(v4[0], v4[1], v4[2], v4[3]);
          rq[(e * LQ + kq) * 64 + lane] = r4;
          v4[0] = v4[1] = v4[2] = v4[3] = 0.f;
          r4 = 0;
        }
      }
    }
  }
  // the open quad and the padding quads
  for (int kq = k >> 2; kq < LQ; ++kq) {
    vq[(e * LQ + kq) * 64 + lane] = make_float4(v4[0], v4[1], v4[2], v4[3]);
    rq[(e * LQ + kq) * 64 + lane] = r4;
    v4[0] = v4[1] = v4[2] = v4[3] = 0.f;
    r4 = 0;
  }
}

// ---- the pass ---------------------------------------------------------------------------------
constexpr int xpitch(int V) { return V <= 1 ? 2 : (V <= 6 ? 6 : 10); }  // doubles, 16-B aligned rows

// wave (e, h): column e of every lane's 4, blocks b0 + h, b0 + h + NH, ... of the tile. The 4
// column phases of a block cost the same by construction (padded to one length).
template <int V, int NW, int MAXQ>
__global__ __launch_bounds__(NW * 64) void k_gemv_csc(
    const float* __restrict__ vals, const uint8_t* __restrict__ rows,
    const uint32_t* __restrict__ Lc, const uint64_t* __restrict__ Pre, int nblocks,
    const int* __restrict__ tb, int ntmax, int64_t ld, int64_t m, double d,
    const double* __restrict__ X, double* __restrict__ part) {
  constexpr int XP = xpitch(V);
  constexpr int NS = V + 1;
  constexpr int NH = NW / CPL;
  constexpr int XT = RB * XP;
  constexpr int LDSD = (NW * XT > NW * NS * 64) ? NW * XT : NW * NS * 64;
  __shared__ double lds[LDSD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int e = wave % CPL, h = wave / CPL;
  const int s = blockIdx.x;
  const int b0 = tb[s * (ntmax + 1) + blockIdx.y];
  const int b1 = tb[s * (ntmax + 1) + blockIdx.y + 1];
  double* xs = lds + wave * XT;

  double acc[NS];
#pragma unroll
  for (int v = 0; v < NS; ++v) acc[v] = 0.0;

#ifdef CSC_PREFETCH
  // cross-block software pipeline: the directory, the first quads and the x rows of the NEXT block
  // are in flight while the current one is multiplied
  int LQ = 0;
  const float4* vq = nullptr;
  const uint32_t* rq = nullptr;
  float4 mv[MAXQ];
  uint32_t rw[MAXQ];
  double xr[VS];
  auto fetch = [&](int bb, int& LQ_, const float4*& vq_, const uint32_t*& rq_, float4 (&mv_)[MAXQ],
                   uint32_t (&rw_)[MAXQ], double (&xr_)[VS]) {
    const int64_t g = (int64_t)s * nblocks + bb;
    LQ_ = __builtin_amdgcn_readfirstlane((int)(Lc[g] >> 2));
    const int64_t base = (int64_t)Pre[g] * CW;
    vq_ = reinterpret_cast<const float4*>(vals + base) + (int64_t)e * LQ_ * 64 + lane;
    rq_ = reinterpret_cast<const uint32_t*>(rows + base) + (int64_t)e * LQ_ * 64 + lane;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      if (q < LQ_) {
        mv_[q] = vq_[q * 64];
        rw_[q] = rq_[q * 64];
      }
    }
    const int64_t r = (int64_t)bb * RB + lane;
#pragma unroll
    for (int v = 0; v < VS; ++v) xr_[v] = 0.0;
    if (r < m) {
      const double2* xp = reinterpret_cast<const double2*>(X + r * VS);
#pragma unroll
      for (int v = 0; v < ((V + 1) & ~1); v += 2) {
        const double2 t2 = xp[v >> 1];
        xr_[v] = t2.x;
        xr_[v + 1] = t2.y;
      }
    }
  };
  if (b0 + h < b1) fetch(b0 + h, LQ, vq, rq, mv, rw, xr);
  for (int b = b0 + h; b < b1; b += NH) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int v = 0; v < ((V + 1) & ~1); v += 2)
      *reinterpret_cast<double2*>(xs + lane * XP + v) = make_double2(xr[v], xr[v + 1]);
    __builtin_amdgcn_wave_barrier();
    int LQn = 0;
    const float4* vqn = nullptr;
    const uint32_t* rqn = nullptr;
    float4 mvn[MAXQ];
    uint32_t rwn[MAXQ];
    if (b + NH < b1) fetch(b + NH, LQn, vqn, rqn, mvn, rwn, xr);
#else
  for (int b = b0 + h; b < b1; b += NH) {
    const int64_t g = (int64_t)s * nblocks + b;
    const int LQ = __builtin_amdgcn_readfirstlane((int)(Lc[g] >> 2));
    const int64_t base = (int64_t)Pre[g] * CW;
    const float4* vq = reinterpret_cast<const float4*>(vals + base) + (int64_t)e * LQ * 64 + lane;
    const uint32_t* rq = reinterpret_cast<const uint32_t*>(rows + base) + (int64_t)e * LQ * 64 + lane;
    float4 mv[MAXQ];
    uint32_t rw[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      if (q < LQ) {
        mv[q] = vq[q * 64];
        rw[q] = rq[q * 64];
      }
    }
    // stage the block's x rows
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rr = 0; rr < RB; rr += 64) {
      const int64_t r = (int64_t)b * RB + rr + lane;
      double xr[VS];
#pragma unroll
      for (int v = 0; v < VS; ++v) xr[v] = 0.0;
      if (r < m) {
        const double4* xp = reinterpret_cast<const double4*>(X + r * VS);
        double4 a = xp[0];
        xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w;
        if (V > 4) {
          double4 c2 = xp[1];
          xr[4] = c2.x; xr[5] = c2.y; xr[6] = c2.z; xr[7] = c2.w;
        }
      }
#pragma unroll
      for (int v = 0; v < ((V + 1) & ~1); v += 2)
        *reinterpret_cast<double2*>(xs + (rr + lane) * XP + v) = make_double2(xr[v], xr[v + 1]);
    }
    __builtin_amdgcn_wave_barrier();
#endif
    for (int k0 = 0; k0 < LQ; k0 += MAXQ) {
      if (k0 > 0) {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
          if (k0 + q < LQ) {
            mv[q] = vq[(k0 + q) * 64];
            rw[q] = rq[(k0 + q) * 64];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        if (k0 + q < LQ) {
          const float mf[4] = {mv[q].x, mv[q].y, mv[q].z, mv[q].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double mm = (double)mf[j];
            const double ii = mf[j] != 0.f ? 1.0 : 0.0;
            const uint32_t row = (rw[q] >> (8 * j)) & 255u;
            const double* xr = xs + row * XP;
            double xv[((V + 1) & ~1)];
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
          }
        }
      }
    }
  #ifdef CSC_PREFETCH
    LQ = LQn;
    vq = vqn;
    rq = rqn;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      mv[q] = mvn[q];
      rw[q] = rwn[q];
    }
#endif
  }

  __syncthreads();
#pragma unroll
  for (int v = 0; v < NS; ++v) lds[(wave * NS + v) * 64 + lane] = acc[v];
  __syncthreads();
  for (int t = threadIdx.x; t < NS * CW; t += NW * 64) {
    const int v = t / CW, cl = t % CW;
    const int ee = cl % CPL, ln = cl / CPL;
    double sum = lds[(ee * NS + v) * 64 + ln];
#pragma unroll
    for (int hh = 1; hh < NH; ++hh) sum += lds[((hh * CPL + ee) * NS + v) * 64 + ln];
    const int64_t c = (int64_t)blockIdx.x * CW + cl;
    if (c < ld) part[((int64_t)blockIdx.y * NS + v) * ld + c] = sum;
  }
}

// reference: one thread per column
template <int V>
__global__ void k_ref(const float* __restrict__ S, int64_t ld, int64_t m, double d,
                      const double* __restrict__ X, double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  double acc[V + 1];
  for (int v = 0; v <= V; ++v) acc[v] = 0.0;
  for (int64_t r = 0; r < m; ++r) {
    const float mf = S[r * ld + c];
    if (mf == 0.f) continue;
    const double mm = mf, w = fma(d, 1.0, mm);
    acc[0] = fma(mm, X[r * VS], acc[0]);
    acc[V] = fma(1.0, X[r * VS], acc[V]);
    for (int v = 1; v < V; ++v) acc[v] = fma(w, X[r * VS + v], acc[v]);
  }
  for (int v = 0; v <= V; ++v) out[v * ld + c] = acc[v];
}

template <int V, int NW, int MAXQ>
static void run(const char* tag, int64_t ld, int64_t m, const float* vals, const uint8_t* rows,
                const uint32_t* Lc, const uint64_t* Pre, const std::vector<uint32_t>& hL,
                int nblocks, int target_wgs, const double* X, const std::vector<double>& ref,
                double bytes) {
  const int nstrips = (int)(ld / CW);
  // cost-balanced tiles: cost of a block = Lc + 2
  std::vector<double> tot(nstrips, 0.0);
  double total = 0;
  for (int s = 0; s < nstrips; ++s) {
    for (int b = 0; b < nblocks; ++b) tot[s] += hL[(size_t)s * nblocks + b] + 2.0;
    total += tot[s];
  }
  const double Q = total / target_wgs;
  std::vector<int> nts(nstrips);
  int ntmax = 1;
  for (int s = 0; s < nstrips; ++s) {
    nts[s] = (int)fmax(1.0, floor(tot[s] / Q + 0.5));
    if (nts[s] > nblocks) nts[s] = nblocks;
    if (nts[s] > ntmax) ntmax = nts[s];
  }
  std::vector<int> htb((size_t)nstrips * (ntmax + 1));
  int real_wgs = 0;
  for (int s = 0; s < nstrips; ++s) {
    int* t = &htb[(size_t)s * (ntmax + 1)];
    double run_cost = 0;
    int k = 1;
    t[0] = 0;
    for (int b = 0; b < nblocks; ++b) {
      run_cost += hL[(size_t)s * nblocks + b] + 2.0;
      while (k < nts[s] && run_cost >= tot[s] * k / nts[s]) t[k++] = b + 1;
    }
    for (; k <= ntmax; ++k) t[k] = nblocks;
    real_wgs += nts[s];
  }
  int* tb;
  CK(hipMalloc(&tb, htb.size() * 4));
  CK(hipMemcpy(tb, htb.data(), htb.size() * 4, hipMemcpyHostToDevice));
  const int nt = ntmax;
  double* part;
  CK(hipMalloc(&part, sizeof(double) * nt * (V + 1) * ld));
  dim3 grid(nstrips, nt);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    k_gemv_csc<V, NW, MAXQ><<<grid, NW * 64>>>(vals, rows, Lc, Pre, nblocks, tb, ntmax, ld, m, 0.37, X, part);
  CK(hipDeviceSynchronize());
  const int reps = 50;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i)
    k_gemv_csc<V, NW, MAXQ><<<grid, NW * 64>>>(vals, rows, Lc, Pre, nblocks, tb, ntmax, ld, m, 0.37, X, part);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<double> hp((size_t)nt * (V + 1) * ld);
  CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int v = 0; v <= V; ++v)
    for (int64_t c = 0; c < m; ++c) {
      double sum = 0;
      for (int t = 0; t < nt; ++t) sum += hp[((size_t)t * (V + 1) + v) * ld + c];
      const double r = ref[(size_t)(v == V ? 6 : v) * ld + c];
      const double err = fabs(sum - r) / (fabs(r) + 1e-300);
      if (err > worst) worst = err;
    }
  const double us = ms * 1e3 / reps;
  printf("%-10s RB=%d V=%d NW=%d MAXQ=%d target=%d grid=%dx%d (%d real)  %.1f us  %.2f TB/s compressed, %.2f TB/s dense-equivalent  maxrel %.2e\n",
         tag, RB, V, NW, MAXQ, target_wgs, nstrips, nt, real_wgs, us, bytes / us * 1e-6,
         4.0 * m * m / us * 1e-6, worst);
  CK(hipFree(part));
  CK(hipFree(tb));
}

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 10000;
  const char* src = argc > 2 ? argv[2] : "0.1125";
  const int64_t ld = (m + 255) / 256 * 256;
  float* S;
  CK(hipMalloc(&S, sizeof(float) * m * ld));
  FILE* f = fopen(src, "rb");
  if (f) {  // dense m x m fp32, row-major
    std::vector<float> row(m);
    std::vector<float> host((size_t)m * ld, 0.f);
    for (int64_t r = 0; r < m; ++r) {
      if (fread(row.data(), 4, m, f) != (size_t)m) { printf("short file\n"); return 1; }
      memcpy(&host[(size_t)r * ld], row.data(), 4 * m);
    }
    fclose(f);
    CK(hipMemcpy(S, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    printf("matrix from %s\n", src);
  } else {
    k_rand<<<4096, 256>>>(S, ld, m, (float)atof(src));
  }
  std::vector<double> hx((size_t)m * VS);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.25 + (double)((i * 2654435761u) % 1000) / 1000.0;
  double* X;
  CK(hipMalloc(&X, hx.size() * 8));
  CK(hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));

  const int nstrips = (int)(ld / CW), nblocks = (int)((m + RB - 1) / RB);
  const int64_t G = (int64_t)nstrips * nblocks;
  uint32_t* Lc;
  uint64_t* Pre;
  CK(hipMalloc(&Lc, G * 4));
  CK(hipMalloc(&Pre, G * 8));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  k_csc_count<<<dim3(nstrips, nblocks), CW>>>(S, ld, m, nblocks, Lc);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_csc_count<<<dim3(nstrips, nblocks), CW>>>(S, ld, m, nblocks, Lc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> hL(G);
  CK(hipMemcpy(hL.data(), Lc, G * 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> hP(G);
  uint64_t tot = 0;
  uint32_t mx = 0;
  for (int64_t g = 0; g < G; ++g) {
    hP[g] = tot;
    tot += hL[g];
    if (hL[g] > mx) mx = hL[g];
  }
  CK(hipMemcpy(Pre, hP.data(), G * 8, hipMemcpyHostToDevice));
  float* vals;
  uint8_t* rows;
  CK(hipMalloc(&vals, tot * CW * 4));
  CK(hipMalloc(&rows, tot * CW));
  k_csc_fill<<<dim3(nstrips, nblocks), CW>>>(S, ld, m, nblocks, Lc, Pre, vals, rows);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e1));
  k_csc_fill<<<dim3(nstrips, nblocks), CW>>>(S, ld, m, nblocks, Lc, Pre, vals, rows);
  CK(hipEventRecord(e2));
  CK(hipDeviceSynchronize());
  float msf;
  CK(hipEventElapsedTime(&msf, e1, e2));
  // nnz for the padding ratio
  std::vector<float> hs;
  double bytes = (double)tot * CW * 5;
  printf("m=%lld groups=%lld  sum(Lc)=%llu  mean Lc=%.2f  max Lc=%u  compressed %.1f MB (dense %.1f MB)  fill %.1f us\n",
         (long long)m, (long long)G, (unsigned long long)tot, (double)tot / G, mx, bytes * 1e-6,
         4.0 * m * ld * 1e-6, msf * 1e3);

  constexpr int V = 6;
  double* out;
  CK(hipMalloc(&out, sizeof(double) * (V + 1) * ld));
  k_ref<V><<<(unsigned)((ld + 63) / 64), 64>>>(S, ld, m, 0.37, X, out);
  CK(hipDeviceSynchronize());
  std::vector<double> ref((size_t)(V + 1) * ld);
  CK(hipMemcpy(ref.data(), out, ref.size() * 8, hipMemcpyDeviceToHost));

  const int targets[] = {512, 768, 1024, 1536, 2048};
  for (int tg : targets) {
    run<V, 8, 6>("csc", ld, m, vals, rows, Lc, Pre, hL, nblocks, tg, X, ref, bytes);
  }
  run<V, 8, 4>("csc", ld, m, vals, rows, Lc, Pre, hL, nblocks, 1024, X, ref, bytes);
  run<V, 8, 8>("csc", ld, m, vals, rows, Lc, Pre, hL, nblocks, 1024, X, ref, bytes);
  run<4, 8, 6>("csc", ld, m, vals, rows, Lc, Pre, hL, nblocks, 1024, X, ref, bytes);
  run<1, 8, 6>("csc", ld, m, vals, rows, Lc, Pre, hL, nblocks, 1024, X, ref, bytes);
  return 0;
}
