// slice_tune.hip — stand-alone harness for the slice storage of M (clipper_amd/csrc/k_slices.hip.h):
// builds a synthetic symmetric sparse matrix with the statistics of the headline problem (uniform
// density + a fully dense "inlier" block at the end), packs it on the HOST into slices, runs the
// streaming part of a pass (slice_core) in several geometries, checks every result against an
// fp64 host product and prints time / bytes / lock-step efficiency. Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/slice_tune.hip -o tools/_bin/slice_tune
//   tools/_bin/slice_tune <m> [density=0.105] [inlier_fraction=0.05] [reps=200]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../clipper_amd/csrc/k_slices.hip.h"

using namespace clipper_hip;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Entry {
  int row;
  float val;
};

static inline uint64_t mix(uint64_t h) {
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return h;
}

// host packer: the specification of the layout (k_slices.hip.h), mirrored by the device packers
template <typename VT>
struct Packed {
  std::vector<uint8_t> data;
  std::vector<uint64_t> Pre;
  std::vector<uint32_t> Lq;
  int nchunks = 0, ncg = 0;
  size_t quads = 0, entries = 0, steps = 0;
};

template <typename VT>
Packed<VT> pack(const std::vector<std::vector<Entry>>& cols, int64_t m, int H) {
  const int R = SL_SUB * H;
  Packed<VT> P;
  P.nchunks = static_cast<int>((m + R - 1) / R);
  P.ncg = static_cast<int>((m + SL_W - 1) / SL_W);
  P.Pre.resize(static_cast<size_t>(P.ncg) * P.nchunks);
  P.Lq.resize(P.Pre.size());
  constexpr int QB = 4 * sizeof(VT);
  std::vector<size_t> cursor(static_cast<size_t>(m), 0);
  for (int cg = 0; cg < P.ncg; ++cg) {
    std::fill(cursor.begin() + cg * SL_W, cursor.begin() + std::min<int64_t>(m, (cg + 1) * SL_W), 0);
    for (int k = 0; k < P.nchunks; ++k) {
      // per lane, per sub-block lists
      std::vector<std::vector<Entry>> sub(static_cast<size_t>(64) * H);
      int nq[8][64];
      int tot[64];
      int maxq = 0;
      size_t nquads = 0;
      for (int l = 0; l < 64; ++l) {
        tot[l] = 0;
        const int64_t c = static_cast<int64_t>(cg) * SL_W + l;
        for (int h = 0; h < H; ++h) {
          nq[h][l] = 0;
          if (c >= m) continue;
          const int64_t r0 = static_cast<int64_t>(k) * R + h * SL_SUB, r1 = r0 + SL_SUB;
          auto& lst = sub[static_cast<size_t>(l) * H + h];
          size_t& cur = cursor[static_cast<size_t>(c)];
          const auto& col = cols[static_cast<size_t>(c)];
          while (cur < col.size() && col[cur].row < r1) {
            lst.push_back({static_cast<int>(col[cur].row - r0), col[cur].val});
            ++cur;
          }
          P.entries += lst.size();
          nq[h][l] = static_cast<int>((lst.size() + 3) / 4);
          tot[l] += nq[h][l];
        }
        maxq = std::max(maxq, tot[l]);
        nquads += tot[l];
      }
      while (P.data.size() % 16) P.data.push_back(0);
      const size_t start = P.data.size();
      P.Pre[static_cast<size_t>(cg) * P.nchunks + k] = start / 16;
      P.Lq[static_cast<size_t>(cg) * P.nchunks + k] = static_cast<uint32_t>(maxq);
      P.quads += nquads;
      P.steps += maxq;
      uint32_t head[4] = {static_cast<uint32_t>(nquads), static_cast<uint32_t>(maxq), 0, 0};
      P.data.insert(P.data.end(), reinterpret_cast<uint8_t*>(head), reinterpret_cast<uint8_t*>(head) + 16);
      for (int h = 0; h < H; ++h)
        for (int l = 0; l < 64; ++l) P.data.push_back(static_cast<uint8_t>(nq[h][l]));
      for (int q = 0; q < maxq; ++q) {
        std::vector<VT> vv;
        std::vector<uint8_t> rr;
        for (int l = 0; l < 64; ++l) {
          if (q >= tot[l]) continue;
          int h = 0, qq = q;
          while (qq >= nq[h][l]) {
            qq -= nq[h][l];
            ++h;
          }
          const auto& lst = sub[static_cast<size_t>(l) * H + h];
          for (int e = 0; e < 4; ++e) {
            const size_t idx = static_cast<size_t>(qq) * 4 + e;
            vv.push_back(idx < lst.size() ? static_cast<VT>(lst[idx].val) : VT(0));
            rr.push_back(idx < lst.size() ? static_cast<uint8_t>(lst[idx].row) : 0);
          }
        }
        const uint8_t* vb = reinterpret_cast<const uint8_t*>(vv.data());
        P.data.insert(P.data.end(), vb, vb + vv.size() * sizeof(VT));
        P.data.insert(P.data.end(), rr.begin(), rr.end());
        while (P.data.size() % 16) P.data.push_back(0);
      }
      uint32_t bytes = static_cast<uint32_t>(P.data.size() - start);
      std::memcpy(P.data.data() + start + 8, &bytes, 4);
      (void)QB;
    }
  }
  P.data.resize(P.data.size() + 4096, 0);  // a load front may run one step past the last slice
  return P;
}

// tiles of equal cost per strip of NW column groups; cost of a chunk = the slowest of its NW
// slices (the waves of a workgroup meet at every chunk) + a constant for the staging
std::vector<int> plan(const std::vector<uint32_t>& Lq, int ncg, int nchunks, int NW, double target,
                      int& nstrips, int& ntmax, std::vector<int2>& work, bool heavy_first) {
  nstrips = (ncg + NW - 1) / NW;
  std::vector<std::vector<double>> cost(static_cast<size_t>(nstrips), std::vector<double>(static_cast<size_t>(nchunks)));
  std::vector<double> tot(static_cast<size_t>(nstrips), 0.0);
  double total = 0.0;
  for (int s = 0; s < nstrips; ++s)
    for (int k = 0; k < nchunks; ++k) {
      double c = 0.0;
      for (int w = 0; w < NW; ++w) {
        const int cg = s * NW + w;
        if (cg < ncg) c = std::max(c, static_cast<double>(Lq[static_cast<size_t>(cg) * nchunks + k]));
      }
      c += 2.0;
      cost[s][k] = c;
      tot[s] += c;
      total += c;
    }
  const double Q = total / target;
  std::vector<int> nts(static_cast<size_t>(nstrips));
  ntmax = 1;
  for (int s = 0; s < nstrips; ++s) {
    int n = static_cast<int>(std::max(1.0, std::floor(tot[s] / Q + 0.5)));
    n = std::min(n, nchunks);
    nts[s] = n;
    ntmax = std::max(ntmax, n);
  }
  std::vector<int> tb(static_cast<size_t>(nstrips) * (ntmax + 1));
  for (int s = 0; s < nstrips; ++s) {
    int* t = tb.data() + static_cast<size_t>(s) * (ntmax + 1);
    const int n = nts[s];
    double run = 0.0;
    int kk = 1;
    t[0] = 0;
    for (int k = 0; k < nchunks; ++k) {
      run += cost[s][k];
      while (kk < n && run >= tot[s] * kk / n) t[kk++] = k + 1;
    }
    for (; kk <= ntmax; ++kk) t[kk] = nchunks;
  }
  std::vector<std::pair<double, int2>> wl;
  for (int s = 0; s < nstrips; ++s)
    for (int t = 0; t < ntmax; ++t) {
      const int a = tb[static_cast<size_t>(s) * (ntmax + 1) + t], b = tb[static_cast<size_t>(s) * (ntmax + 1) + t + 1];
      if (a >= b) continue;
      double c = 0.0;
      for (int k = a; k < b; ++k) c += cost[s][k];
      wl.push_back({c, make_int2(s, t)});
    }
  if (heavy_first)
    std::stable_sort(wl.begin(), wl.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
  work.clear();
  for (auto& w : wl) work.push_back(w.second);
  return tb;
}

template <typename VT, int H, bool WINDOW, int V, int NW, int D, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void k_pass(SliceView M, int64_t ld, int64_t m, double d,
                                                       const double* X, double* part,
                                                       long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const long long c0 = stamps ? wall_clock64() : 0;
  SliceJob<H, NW> J;
  slice_begin<H, NW>(M, J);
  slice_core<VT, H, WINDOW, V, nslot(V), NW, D>(M, J, ld, m, d, X, VS, part, lds);
  if (stamps && (threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * NW + (threadIdx.x >> 6)) * 2] = c0;
    stamps[(blockIdx.x * NW + (threadIdx.x >> 6)) * 2 + 1] = wall_clock64();
  }
}

struct Ctx {
  int64_t m, mp;
  std::vector<std::vector<Entry>> cols;
  std::vector<double> X;      // [mp][VS]
  double d = 0.37;
  double* dX = nullptr;
  int cus = 256;
  bool timeline = false;
};

// fp64 host product for (window V): a, g_v, b
void host_ref(const Ctx& c, int V, std::vector<double>& out /* [V+1][m] */) {
  out.assign(static_cast<size_t>(V + 1) * c.m, 0.0);
  for (int64_t col = 0; col < c.m; ++col) {
    double acc[9] = {0};
    for (const Entry& e : c.cols[static_cast<size_t>(col)]) {
      const double mm = e.val, ii = 1.0;
      const double* xr = c.X.data() + static_cast<size_t>(e.row) * VS;
      acc[0] = fma(mm, xr[0], acc[0]);
      acc[V] = fma(ii, xr[0], acc[V]);
      const double w = fma(c.d, ii, mm);
      for (int v = 1; v < V; ++v) acc[v] = fma(w, xr[v], acc[v]);
    }
    for (int v = 0; v <= V; ++v) out[static_cast<size_t>(v) * c.m + col] = acc[v];
  }
}

template <typename VT, int H, int V, int NW, int D, int OCC>
void run_variant(Ctx& c, const Packed<VT>& P, double wg_target, int reps, const std::vector<double>& ref,
                 bool heavy_first = true) {
  int nstrips, ntmax;
  std::vector<int2> work;
  std::vector<int> tb = plan(P.Lq, P.ncg, P.nchunks, NW, wg_target, nstrips, ntmax, work, heavy_first);
  int2* dwork;
  CK(hipMalloc(&dwork, work.size() * sizeof(int2)));
  CK(hipMemcpy(dwork, work.data(), work.size() * sizeof(int2), hipMemcpyHostToDevice));
  uint8_t* ddata;
  uint64_t* dPre;
  int* dtb;
  double* dpart;
  const int64_t ld = (c.m + 63) / 64 * 64;
  CK(hipMalloc(&ddata, P.data.size()));
  CK(hipMalloc(&dPre, P.Pre.size() * 8));
  CK(hipMalloc(&dtb, tb.size() * 4));
  constexpr int VV = V > 0 ? V : 1;
  const size_t npart = static_cast<size_t>(ntmax) * (VV + 1) * ld;
  CK(hipMalloc(&dpart, npart * 8));
  CK(hipMemset(dpart, 0, npart * 8));
  CK(hipMemcpy(ddata, P.data.data(), P.data.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dPre, P.Pre.data(), P.Pre.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtb, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
  SliceView M{ddata, dPre, dtb, dwork, P.nchunks, P.ncg, ntmax};
  auto kern = k_pass<VT, H, (V > 0), (V > 0 ? V : 1), NW, D, OCC>;
  const size_t lds_bytes = static_cast<size_t>(2) * SL_SUB * H * ((V > 0) ? sl_xpitch(VV) : 1) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                         static_cast<int>(lds_bytes)));
  dim3 grid(static_cast<unsigned>(work.size())), block(NW * 64);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, grid, block, lds_bytes, 0, M, ld, c.m, c.d, c.dX, dpart, (long long*)nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, block, lds_bytes, 0, M, ld, c.m, c.d, c.dX, dpart, (long long*)nullptr);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  // validate: sum tiles in order
  std::vector<double> hp(npart);
  CK(hipMemcpy(hp.data(), dpart, npart * 8, hipMemcpyDeviceToHost));
  double maxerr = 0.0, maxref = 0.0;
  const int NSL = VV + 1;
  const int nout = (V > 0) ? V + 1 : 2;
  for (int v = 0; v < nout; ++v) {
    const int slot = (v == nout - 1) ? NSL - 1 : v;
    const int rv = (V > 0) ? v : (v == 0 ? 0 : 1);  // pair mode compares a and b of candidate 0
    for (int64_t col = 0; col < c.m; ++col) {
      double s = 0.0;
      for (int t = 0; t < ntmax; ++t) s += hp[(static_cast<size_t>(t) * NSL + slot) * ld + col];
      const double r = ref[static_cast<size_t>((V > 0) ? v : (rv == 0 ? 0 : VV)) * c.m + col];
      maxerr = std::max(maxerr, std::fabs(s - r));
      maxref = std::max(maxref, std::fabs(r));
    }
  }
  const int wgs = static_cast<int>(work.size());
  if (c.timeline) {  // one more launch with per-wave time stamps (100 MHz wall clock)
    long long* dst;
    const size_t ns = work.size() * NW * 2;
    CK(hipMalloc(&dst, ns * 8));
    CK(hipMemset(dst, 0, ns * 8));
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, 0, M, ld, c.m, c.d, c.dX, dpart, dst);
    CK(hipDeviceSynchronize());
    std::vector<long long> st(ns);
    CK(hipMemcpy(st.data(), dst, ns * 8, hipMemcpyDeviceToHost));
    long long lo = st[0], hi = 0;
    for (size_t i = 0; i < ns; i += 2) { lo = std::min(lo, st[i]); hi = std::max(hi, st[i + 1]); }
    std::vector<double> dur, endt, begt;
    for (size_t i = 0; i < ns; i += 2) { dur.push_back((st[i + 1] - st[i]) * 0.01); endt.push_back((st[i + 1] - lo) * 0.01); begt.push_back((st[i] - lo) * 0.01); }
    std::vector<double> sd = dur, se = endt, sb = begt;
    std::sort(sd.begin(), sd.end()); std::sort(se.begin(), se.end()); std::sort(sb.begin(), sb.end());
    auto pc = [](const std::vector<double>& v, double f) { return v[static_cast<size_t>(f * (v.size() - 1))]; };
    printf("    timeline: span %.2f us; wave start p50 %.2f p99 %.2f max %.2f; wave duration p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f; wave end p50 %.2f p90 %.2f p99 %.2f\n",
           (hi - lo) * 0.01, pc(sb, .5), pc(sb, .99), sb.back(), pc(sd, .1), pc(sd, .5), pc(sd, .9), pc(sd, .99), sd.back(), pc(se, .5), pc(se, .9), pc(se, .99));
    // the five waves that end last
    std::vector<size_t> idx(dur.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::partial_sort(idx.begin(), idx.begin() + 5, idx.end(), [&](size_t a, size_t b) { return endt[a] > endt[b]; });
    for (int q = 0; q < 5; ++q) {
      const size_t w = idx[q];
      const int2 wk = work[w / NW];
      const int a = tb[static_cast<size_t>(wk.x) * (ntmax + 1) + wk.y], b = tb[static_cast<size_t>(wk.x) * (ntmax + 1) + wk.y + 1];
      unsigned steps = 0;
      const int cg = wk.x * NW + static_cast<int>(w % NW);
      for (int k = a; k < b; ++k) if (cg < P.ncg) steps += P.Lq[static_cast<size_t>(cg) * P.nchunks + k];
      printf("      last: wg %zu wave %zu strip %d chunks [%d,%d) steps %u  start %.2f end %.2f\n", w / NW, w % NW, wk.x, a, b, steps, begt[w], endt[w]);
    }
    (void)hipFree(dst);
  }
  const double bytes = static_cast<double>(P.data.size()) + P.Pre.size() * 8.0;
  printf("  VT=%zu H=%d V=%d NW=%d D=%d occ=%d %s strips=%d ntmax=%d (%d WGs) : %8.2f us  %7.1f GB/s  relerr %.2e %s\n",
         sizeof(VT), H, V, NW, D, OCC, heavy_first ? "heavy-first" : "grid-order", nstrips, ntmax, wgs, us, bytes / us * 1e-3, maxerr / (maxref + 1e-300),
         (maxerr <= 1e-9 * (maxref + 1.0)) ? "ok" : "MISMATCH");
  fflush(stdout);
  (void)hipFree(ddata);
  (void)hipFree(dPre);
  (void)hipFree(dtb);
  (void)hipFree(dwork);
  (void)hipFree(dpart);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  Ctx c;
  c.m = argc > 1 ? atoll(argv[1]) : 10000;
  const double density = argc > 2 ? atof(argv[2]) : 0.105;
  const double inl = argc > 3 ? atof(argv[3]) : 0.05;
  const int reps = argc > 4 ? atoi(argv[4]) : 200;
  c.timeline = argc > 5 && atoi(argv[5]) != 0;
  const int64_t m = c.m;
  c.mp = (m + 63) / 64 * 64;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  c.cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; m=%lld density=%.3f inliers=%.3f\n", prop.gcnArchName, c.cus, (long long)m, density, inl);
  // matrix
  c.cols.assign(static_cast<size_t>(m), {});
  const int64_t i0 = m - static_cast<int64_t>(inl * m);
  size_t nnz = 0;
  for (int64_t hi = 1; hi < m; ++hi)
    for (int64_t lo = 0; lo < hi; ++lo) {
      const uint64_t h = mix(static_cast<uint64_t>(lo) * m + hi);
      const double u = (h & 0xFFFFFF) / 16777216.0;
      const bool on = (lo >= i0) || u < density;
      if (!on) continue;
      const float v = (((h >> 24) & 0xFFFFFF) + 1) / 16777216.0f;
      c.cols[static_cast<size_t>(hi)].push_back({static_cast<int>(lo), v});
      c.cols[static_cast<size_t>(lo)].push_back({static_cast<int>(hi), v});
      nnz += 2;
    }
  for (auto& col : c.cols) std::sort(col.begin(), col.end(), [](const Entry& a, const Entry& b) { return a.row < b.row; });
  c.X.assign(static_cast<size_t>(c.mp) * VS, 0.0);
  for (int64_t r = 0; r < m; ++r)
    for (int v = 0; v < VS; ++v) c.X[static_cast<size_t>(r) * VS + v] = (mix(r * 8 + v + 12345) & 0xFFFFF) / 1048576.0 / (1 + v);
  CK(hipMalloc(&c.dX, c.X.size() * 8));
  CK(hipMemcpy(c.dX, c.X.data(), c.X.size() * 8, hipMemcpyHostToDevice));
  printf("stored entries (both triangles) %zu = %.2f %% ; 5 B/entry = %.1f MB\n", nnz, 100.0 * nnz / (double(m) * m), nnz * 5e-6);

  std::vector<double> ref6, ref4, ref1, ref8;
  host_ref(c, 6, ref6);
  host_ref(c, 4, ref4);
  host_ref(c, 1, ref1);
  host_ref(c, 8, ref8);
  const double slots = c.cus * 2.0;
  const bool tl = c.timeline;
  for (int H : {1, 2}) {
    Packed<float> P = pack<float>(c.cols, m, H);
    printf("H=%d: %zu quads (%.3f padded entries per entry), %zu lock-step steps, lane efficiency %.3f, %.2f MB (%.2f B/entry)\n",
           H, P.quads, P.quads * 4.0 / P.entries, P.steps, P.quads / (64.0 * P.steps), P.data.size() * 1e-6,
           double(P.data.size()) / P.entries);
    for (double tgt : {slots, 1.5 * slots, 2.0 * slots, 3.0 * slots}) {
      printf(" target %.0f workgroups\n", tgt);
      c.timeline = tl && tgt == 2.0 * slots;
      if (H == 1) {
        run_variant<float, 1, 6, 4, 4, 2>(c, P, tgt, reps, ref6, false);
        run_variant<float, 1, 6, 4, 4, 2>(c, P, tgt, reps, ref6);
        c.timeline = false;
        run_variant<float, 1, 6, 4, 6, 2>(c, P, tgt, reps, ref6);
        run_variant<float, 1, 6, 4, 8, 2>(c, P, tgt, reps, ref6);
        run_variant<float, 1, 6, 8, 4, 2>(c, P, tgt, reps, ref6);
        run_variant<float, 1, 6, 2, 4, 2>(c, P, tgt, reps, ref6);
        c.timeline = tl && tgt == 2.0 * slots;
        run_variant<float, 1, 0, 4, 8, 2>(c, P, tgt, reps, ref1);
        c.timeline = false;
        run_variant<float, 1, 0, 4, 4, 2>(c, P, tgt, reps, ref1);
      } else {
        run_variant<float, 2, 6, 4, 4, 2>(c, P, tgt, reps, ref6);
        run_variant<float, 2, 6, 4, 8, 2>(c, P, tgt, reps, ref6);
        run_variant<float, 2, 6, 8, 4, 2>(c, P, tgt, reps, ref6);
        run_variant<float, 2, 0, 4, 8, 2>(c, P, tgt, reps, ref1);
      }
    }
    c.timeline = false;
    if (H == 1) {
      printf(" other windows (target %.0f)\n", 2.0 * slots);
      run_variant<float, 1, 1, 4, 8, 2>(c, P, 2.0 * slots, reps, ref1);
      run_variant<float, 1, 4, 4, 8, 2>(c, P, 2.0 * slots, reps, ref4);
      run_variant<float, 1, 8, 4, 4, 2>(c, P, 2.0 * slots, reps, ref8);
    }
  }
  {
    Packed<double> P = pack<double>(c.cols, m, 1);
    printf("fp64 values, H=1: %.2f MB (%.2f B/entry)\n", P.data.size() * 1e-6, double(P.data.size()) / P.entries);
    run_variant<double, 1, 6, 4, 4, 2>(c, P, 2.0 * slots, reps, ref6);
    run_variant<double, 1, 6, 4, 8, 2>(c, P, 2.0 * slots, reps, ref6);
  }
  return 0;
}
