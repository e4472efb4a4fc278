// k_resident.hip.h — the RESIDENT solver: findDenseClique as ONE launch for problems whose slices
// fit the LDS of the workgroups that share them (m up to a few thousand).
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_slices.hip.h"

namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// Below m ~ 2000 a pass over M moves less than 3 MB: the two launches per iteration of the
// streaming solver (k_gemv_slices, k_tail) cost more than the work in them. Here the whole of
// findDenseClique (clipper.cpp:172-323) is one launch of P workgroups that never leave the chip:
//
//   * M-stationary: workgroup p copies the slices of ITS unit — column groups [cg0, cg0+ncgs) x
//     chunks [k0, k1) — from the arena into LDS once, and every pass reads them there.
//   * Every workgroup holds the whole vectors (u, gradF, ... in registers: thread t owns elements
//     t, t + 512, ...) and repeats the O(m) algebra and every decision of the line search
//     redundantly: same code on the same bits => the same decisions everywhere, no scalar
//     ever crosses a workgroup. (The column-shard protocol of the multi-GPU driver, with
//     workgroups as ranks.)
//   * The only exchange is the pass's all-gather, and the data IS the flag: a workgroup publishes
//     every raw sum of its columns as two 8-byte granules {epoch, half of the fp64 value} with
//     write-through (sc1) atomic stores into xb[parity][slot]; every thread of every workgroup
//     sweeps the granules of ITS elements (relaxed agent-scope loads) until both carry the pass's
//     epoch. No drain, no flag, no fence, no barrier on the hand-off, and nothing that depends
//     on where a workgroup runs. Spins are bounded by the wall clock; a time-out sets `err`
//     and every workgroup leaves (the host then runs the streaming solver).
//   * One workgroup (P = 1: everything fits one LDS) exchanges nothing at all.
//   * Inside a unit the steps of a column group's slices are dealt out to its waves in equal
//     PIECES (slice, step range): the chains of the dense inlier slices end with the others.
//   * One-XCD mode (speed only): see ResidentArgs::xcd_mode.
//
// The line-search WINDOW is kept (V candidates per pass, walked in the reference's order): here it
// saves exchanges instead of bytes. Trial sequence, per-trial arithmetic and results are those of
// the streaming solver up to the order of the additions inside a sum.
// ------------------------------------------------------------------------------------------
constexpr int RS_NT = 512;           // threads per workgroup (8 waves, two per SIMD: 256 VGPRs each)
constexpr int RS_NWV = RS_NT / 64;   // waves
constexpr int RS_TMAX = 64;          // slices a unit holds at most
constexpr int RS_PMAX = 12;          // pieces (slice, step range) a wave works on at most
constexpr int RS_MAXE = 4;           // elements per thread at most (m <= 2048)

struct ResidentUnit {
  int cg0, ncgs;  // column groups [cg0, cg0 + ncgs)
  int k0, k1;     // chunks [k0, k1)
  int slot;       // which of the partial-sum slots of its columns this unit fills
  int pad0, pad1, pad2;
};

struct ResidentArgs {
  SliceView M;
  const ResidentUnit* units;
  int nunits;
  const uint8_t* nslots_of_cg;  // [ncg] slots to add for a column of the group
  // what every wave works on: pieces[(unit * RS_NWV + wave) * RS_PMAX + j] = chunk - k0 | q0 << 8 | q1 << 16
  // (steps [q0, q1) of that slice of the wave's column group), npieces[unit * RS_NWV + wave] of them —
  // the planner cuts the long step chains of dense slices so that the waves of a group end together
  const uint32_t* pieces;
  const uint8_t* npieces;
  const uint8_t* wave_cg;       // [unit * RS_NWV + wave] which column group of the unit the wave works for (255: none)
  int maxslots;
  int64_t m, mp;
  SolverParams prm;
  int rescale;
  const double* u0;
  unsigned long long* xb;        // [2][maxslots][V+1][mp][2] granules {epoch << 32 | half of a sum}, zero at allocation
  unsigned long long epoch0;     // this solve's epochs are epoch0 + 1, epoch0 + 2, ... (low 32 bits = the tags)
  uint32_t* err;                 // 0 | RS_ERR_*
  uint32_t lds_slices;           // bytes of LDS the slices of a unit may take
  double* u_dev;                 // [mp] final u (device)
  double* host_u;                // pinned (may be null)
  HostMirror* host;              // pinned (may be null)
  SolveShared* shared;
  // one-XCD mode (speed only, verified at run time): the launch is 8 x nunits workgroups, the ones
  // the hardware put on XCD `home` claim the units (ctrs[0]) and exchange through THEIR L2 — plain
  // stores, L1-bypassing loads, no trip to memory; ctrs[1] counts arrivals, the last one checks that
  // every unit was claimed (else RS_ERR_PLAN: the host launches again in the placement-free mode)
  int xcd_mode, home;
  unsigned long long* ctrs;      // [2], zero between launches
  long long* stamps;             // measurement only (may be null): unit 0, [pass][8] wall-clock stamps
  long long timeout_ticks;       // longest wait for the other workgroups' flags, 100 MHz wall clock
};
enum : uint32_t { RS_ERR_LDS = 1, RS_ERR_TIMEOUT = 2, RS_ERR_PLAN = 3 };

// LDS carve (bytes, all multiples of 16): [X table | reduce scratch] [y (P = 1 only)] [block_reduce
// scratch] [slice offsets] [slices]
__host__ __device__ constexpr uint32_t rs_xt_bytes(int V, int64_t mp) {
  const uint32_t xt = static_cast<uint32_t>(mp) * V * 8u;
  const uint32_t sc = RS_NWV * (V + 1) * 64u * 8u;
  return xt > sc ? xt : sc;
}
__host__ __device__ constexpr uint32_t rs_y_bytes(int V, int64_t mp, bool single) {
  return single ? static_cast<uint32_t>(mp) * (V + 1) * 8u : 0u;
}
constexpr uint32_t RS_RED_BYTES = 2 * RS_NWV * 16 * 8;       // block_reduce scratch (N <= 16), twice
constexpr uint32_t RS_TAB_BYTES = RS_TMAX * 4 + 64;          // slice offsets + a few words
constexpr uint32_t RS_SLICE_PAD = 2048;                       // a load front may run past the last slice

// a granule as every reader must see it: past this CU's L1 (agent scope, relaxed)
__device__ __forceinline__ unsigned long long rs_ld_granule(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The pieces of this wave against the X table: acc[0] = a, acc[1..V-1] = g_v, acc[V] = b
// (clipper.cpp:238-241 with w = M + d C folded per entry, see k_gemv.hip.h). `toff` = LDS offsets of
// the slices of the wave's column group (by chunk - k0), `pc` = this lane's copy of piece `lane`.
// (kscale, lrow: a PACKED slice of the resident solver on a view, k_rv_resident.hip.h — slice kl stands for the
// chunks kscale * kl ..., one per lane group, and this lane's rows start lrow rows into them; 1 and 0 otherwise)
template <typename VT, int V>
__device__ __forceinline__ void rs_wave_pass(const uint8_t* sl, const uint32_t* toff, uint32_t pc, int np,
                                             int k0, const double* Xt, double d, double (&acc)[V + 1],
                                             int kscale = 1, int lrow = 0) {
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v <= V; ++v) acc[v] = 0.0;
  for (int j = 0; j < np; ++j) {
    const uint32_t piece = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pc), j));
    const int kl = static_cast<int>(piece & 255u), q0 = static_cast<int>((piece >> 8) & 255u);
    const uint8_t* sp = sl + toff[kl];
    const int maxq = __builtin_amdgcn_readfirstlane(
        static_cast<int>(reinterpret_cast<const uint32_t*>(sp)[1]));
    const int q1 = min(static_cast<int>((piece >> 16) & 255u), maxq);
    const int tot = sp[16 + lane];
    const uint8_t* fb = sp + 16 + 64 + sl_so_bytes(maxq);
    for (int q = 0; q < q0; ++q) {  // where step q0 starts: the steps before it, by their lane counts
      const int cnt = __builtin_amdgcn_readfirstlane(__popcll(__ballot(q < tot)));
      fb += cnt * QB + ((cnt * 4 + 15) & ~15);
    }
    const double* xs = Xt + (static_cast<int64_t>(k0 + kl * kscale) * SL_SUB + lrow) * V;
    // one step ahead: every lane loads (an idle lane the step's first quad, a step past the end the
    // bytes behind the slice), so that the loads of step q + 1 fly while step q is multiplied
    SliceQuad<VT> cur, nxt;
    uint32_t rcur = 0, rnxt = 0;
    auto issue = [&](int q, SliceQuad<VT>& vq, uint32_t& rq) {
      const bool active = q < tot;
      const uint64_t mask = __ballot(active);
      const int cnt = __builtin_amdgcn_readfirstlane(__popcll(mask));
      const uint32_t rank = active ? sl_lane_rank(mask) : 0u;
      vq.load(fb + rank * QB);
      rq = *reinterpret_cast<const uint32_t*>(fb + cnt * QB + rank * 4);
      fb += cnt * QB + ((cnt * 4 + 15) & ~15);
    };
    auto mult = [&](int q, const SliceQuad<VT>& vq, uint32_t rq) {
      if (q < tot && q < q1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const VT mf = vq.v[e];
          const double mm = static_cast<double>(mf);
          const double ii = mf != VT(0) ? 1.0 : 0.0;
          const uint32_t row = (rq >> (8 * e)) & 255u;
          const double* xr = xs + row * V;
          double xv[V];
          if constexpr (V == 1) {
            xv[0] = xr[0];
          } else {
#pragma unroll
            for (int v = 0; v < V; v += 2) {
              const double2 t2 = *reinterpret_cast<const double2*>(xr + v);
              xv[v] = t2.x;
              xv[v + 1] = t2.y;
            }
          }
          acc[0] = fma(mm, xv[0], acc[0]);
          acc[V] = fma(ii, xv[0], acc[V]);
          if constexpr (V > 1) {
            const double w = fma(d, ii, mm);
#pragma unroll
            for (int v = 1; v < V; ++v) acc[v] = fma(w, xv[v], acc[v]);
          }
        }
      }
    };
    // two steps per turn, the quads of the next two requested before these two are multiplied
    SliceQuad<VT> nx2, nx3;
    uint32_t r2 = 0, r3 = 0;
    issue(q0, cur, rcur);
    issue(q0 + 1, nxt, rnxt);
    for (int q = q0; q < q1; q += 2) {
      issue(q + 2, nx2, r2);
      issue(q + 3, nx3, r3);
      mult(q, cur, rcur);
      mult(q + 1, nxt, rnxt);
      cur = nx2;
      rcur = r2;
      nxt = nx3;
      rnxt = r3;
    }
  }
}

template <typename VT, int V, int E>
__global__ __launch_bounds__(RS_NT) void k_solve_resident(ResidentArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rs_lds[];
  constexpr int NS = V + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m = A.m, mp = A.mp;
  const bool single = A.nunits == 1;
  const SolverParams P = A.prm;

  // ---- carve -------------------------------------------------------------------------------
  uint32_t off = 0;
  double* Xt = reinterpret_cast<double*>(rs_lds + off);   // [mp][V]; reused as reduce scratch
  double* scr = Xt;                                       // [16 waves][NS][64]
  off += rs_xt_bytes(V, mp);
  double* ylds = reinterpret_cast<double*>(rs_lds + off); // [NS][mp], single only
  off += rs_y_bytes(V, mp, single);
  double* red = reinterpret_cast<double*>(rs_lds + off);
  double* red2 = red + RS_NWV * 16;  // the window's norm sums (read after the pass)
  off += RS_RED_BYTES;
  uint32_t* tab = reinterpret_cast<uint32_t*>(rs_lds + off);  // [RS_TMAX] slice offsets, then words
  uint32_t* words = tab + RS_TMAX;                            // [0..15] a few words
  off += RS_TAB_BYTES;
  uint8_t* sl = rs_lds + off;

  // ---- this workgroup's unit; its slices -> LDS ----------------------------------------------
  const long long ts0 = A.stamps ? wall_clock64() : 0;
  int unit = blockIdx.x;
  if (A.xcd_mode) {
    if (tid == 0) {
      // HW_REG_XCC_ID (id 20), bits [3:0]: which XCD this workgroup runs on
      const int xcc = static_cast<int>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & 15;
      long long idx = -1;
      if (xcc == A.home)
        idx = static_cast<long long>(__hip_atomic_fetch_add(A.ctrs, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the claim before the arrival
      const unsigned long long arrived =
          __hip_atomic_fetch_add(A.ctrs + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
      if (arrived == gridDim.x) {  // everybody has claimed or passed: were all units taken?
        const unsigned long long claimed = __hip_atomic_load(A.ctrs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (claimed < static_cast<unsigned long long>(A.nunits))
          __hip_atomic_store(A.err, static_cast<uint32_t>(RS_ERR_PLAN), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.ctrs, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.ctrs + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      words[RS_NWV] = (idx >= 0 && idx < A.nunits) ? static_cast<uint32_t>(idx) : 0xffffffffu;
    }
    __syncthreads();
    const uint32_t got = words[RS_NWV];
    if (got == 0xffffffffu) return;  // not on the home XCD, or more workgroups there than units
    unit = static_cast<int>(got);
    __syncthreads();
  }
  const ResidentUnit U = A.units[unit];
  // the unit's wave -> column group map (8 bytes), kept by every thread for the sums over waves
  const unsigned long long wcg = *reinterpret_cast<const unsigned long long*>(A.wave_cg + static_cast<int64_t>(unit) * RS_NWV);
  const int cgl = static_cast<int>((wcg >> (8 * wave)) & 255ull);
  const bool has_cg = cgl < U.ncgs && U.cg0 + cgl < A.M.ncg;
  const int S = U.k1 - U.k0;           // slices per column group of the unit
  const int nslices = U.ncgs * S;      // local slice t = cgl * S + (chunk - k0)
  // sizes (thread t reads the header of slice t), offsets (one wave scans them), copy (wave w takes
  // the slices t = w, w + 8, ...)
  uint32_t bytes_t = 0;
  if (tid < nslices && nslices <= RS_TMAX) {
    const int tc = tid / S, tk = tid - tc * S;
    const uint8_t* src_t = A.M.data + 16 * A.M.Pre[static_cast<int64_t>(U.cg0 + tc) * A.M.nchunks + U.k0 + tk];
    bytes_t = reinterpret_cast<const uint32_t*>(src_t)[2];
  }
  if (wave == 0) {
    uint32_t inc = bytes_t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t2 = __shfl_up(inc, o);
      if (lane >= o) inc += t2;
    }
    tab[lane] = inc - bytes_t;
    if (lane == 63) words[0] = inc;
  }
  __syncthreads();
  const uint32_t total = words[0];
  const bool bad_plan = total + RS_SLICE_PAD > A.lds_slices || nslices > RS_TMAX;
  if (__syncthreads_or(bad_plan ? 1 : 0)) {
    if (tid == 0) __hip_atomic_store(A.err, static_cast<uint32_t>(RS_ERR_LDS), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  for (int t = wave; t < nslices; t += RS_NWV) {
    const int tc = t / S, tk = t - tc * S;
    const uint8_t* src = A.M.data + 16 * A.M.Pre[static_cast<int64_t>(U.cg0 + tc) * A.M.nchunks + U.k0 + tk];
    const uint32_t nb = reinterpret_cast<const uint32_t*>(src)[2];
    const uint32_t o = tab[t];
    for (uint32_t b = lane * 16; b < nb; b += 64 * 16)
      *reinterpret_cast<uint4*>(sl + o + b) = *reinterpret_cast<const uint4*>(src + b);
  }
  const uint32_t* toff = tab + (has_cg ? cgl * S : 0);
  // this wave's pieces: lane j keeps piece j
  const int np = has_cg ? A.npieces[unit * RS_NWV + wave] : 0;
  const uint32_t pc = (lane < RS_PMAX) ? A.pieces[(static_cast<int64_t>(unit) * RS_NWV + wave) * RS_PMAX + lane] : 0u;

  // ---- per-thread elements -----------------------------------------------------------------
  bool valid[E];
  int nslots_e[E];
  double u[E], g[E], a[E], b[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int64_t i = tid + static_cast<int64_t>(e) * RS_NT;
    valid[e] = i < m;
    nslots_e[e] = (valid[e] && !single) ? A.nslots_of_cg[i >> 6] : 0;
    u[e] = valid[e] ? A.u0[i] : 0.0;
    g[e] = a[e] = b[e] = 0.0;
  }
  __syncthreads();

  unsigned long long epoch = A.epoch0;
  int stamp_row = 0;
  auto stamp = [&](int col) {
    if (A.stamps && unit == 0 && tid == 0 && stamp_row < 500) A.stamps[stamp_row * 8 + col] = wall_clock64();
  };

  // One pass: the X table (raw candidates of this thread's elements) -> y[e][0..V] for them.
  // Returns false on a time-out (uniform over the workgroup).
  auto pass = [&](const double (&x)[E][V], double d, double (&y)[E][NS]) -> bool {
    stamp(1);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int64_t i = tid + static_cast<int64_t>(e) * RS_NT;
      if (i < mp) {
#pragma unroll
        for (int v = 0; v < V; ++v) Xt[i * V + v] = x[e][v];
      }
    }
    __syncthreads();
    double acc[NS];
    rs_wave_pass<VT, V>(sl, toff, pc, np, U.k0, Xt, d, acc);
    __syncthreads();  // the X table is dead: its memory becomes the reduce scratch
    stamp(2);
#pragma unroll
    for (int v = 0; v < NS; ++v) scr[(wave * NS + v) * 64 + lane] = acc[v];
    __syncthreads();
    ++epoch;
    const int par = static_cast<int>(epoch & 1ull);
    const unsigned long long tag = (epoch & 0xffffffffull) << 32;
    // sums over the waves of a column group, in wave order; one (column, v) per thread and step
    const int nout = U.ncgs * 64 * NS;
    for (int o = tid; o < nout; o += RS_NT) {
      const int c = o & 63, v = (o >> 6) % NS, gl = o / (64 * NS);
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < RS_NWV; ++w)
        if (static_cast<int>((wcg >> (8 * w)) & 255ull) == gl) sum += scr[(w * NS + v) * 64 + c];
      const int64_t col = static_cast<int64_t>(U.cg0 + gl) * 64 + c;
      if (col < mp) {
        if (single) {
          ylds[v * mp + col] = sum;
        } else {  // two self-describing granules {epoch, half of the value}: the data IS the flag
          const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(sum));
          unsigned long long* gq = A.xb + ((((static_cast<int64_t>(par) * A.maxslots + U.slot) * NS + v) * mp + col) << 1);
          if (A.xcd_mode) {  // every reader sits behind the same L2: the granules stay there
            __hip_atomic_store(gq, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(gq + 1, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {           // write-through: readers may sit behind another XCD's L2
            __hip_atomic_store(gq, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gq + 1, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    if (single) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int64_t i = tid + static_cast<int64_t>(e) * RS_NT;
#pragma unroll
        for (int v = 0; v < NS; ++v) y[e][v] = valid[e] ? ylds[v * mp + i] : 0.0;
      }
      __syncthreads();
      stamp(3);
      stamp(4);
      stamp(5);
      return true;
    }
    stamp(3);
    // gather: every thread sweeps the granules of ITS elements until they carry this pass's epoch —
    // no drain, no flag, no acquire (cdna_hip_programming.md 6 G16, form R2); slot by slot, in order
    int fail = 0;
    const unsigned long long* xb = A.xb + ((static_cast<int64_t>(par) * A.maxslots * NS * mp) << 1);
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int v = 0; v < NS; ++v) y[e][v] = 0.0;
    }
    const long long t_poll = wall_clock64();
    for (int sl_ = 0; sl_ < A.maxslots && !fail; ++sl_) {
      double tv[E][NS];
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int64_t i = tid + static_cast<int64_t>(e) * RS_NT;
          if (sl_ < nslots_e[e]) {
#pragma unroll
            for (int v = 0; v < NS; ++v) {
              const unsigned long long* gq = xb + (((static_cast<int64_t>(sl_) * NS + v) * mp + i) << 1);
              const unsigned long long g0 = rs_ld_granule(gq), g1 = rs_ld_granule(gq + 1);
              ok = ok && ((g0 ^ tag) >> 32) == 0 && ((g1 ^ tag) >> 32) == 0;
              tv[e][v] = __longlong_as_double(static_cast<long long>((g0 << 32) | (g1 & 0xffffffffull)));
            }
          }
        }
        if (__all(ok)) break;
        if ((spins & 63u) == 63u || A.timeout_ticks < 0) {  // (negative: a test's way to force the time-out)
          const bool late = wall_clock64() - t_poll > A.timeout_ticks;
          const uint32_t e2 = __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (late || e2 != 0) {
            fail = 1;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!fail) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (sl_ < nslots_e[e]) {
#pragma unroll
            for (int v = 0; v < NS; ++v) y[e][v] += tv[e][v];
          }
        }
      }
    }
    // (also the barrier between this pass's reads of the reduce scratch and the next X table)
    if (__syncthreads_or(fail)) {
      if (tid == 0) {
        uint32_t expect = 0;
        __hip_atomic_compare_exchange_strong(A.err, &expect, static_cast<uint32_t>(RS_ERR_TIMEOUT),
                                             __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      }
      return false;
    }
    stamp(4);
    if (A.stamps && unit == 0 && tid == 0 && stamp_row < 500)
      A.stamps[stamp_row * 8 + 5] = wall_clock64() + (y[0][0] > 1e300 ? 1 : 0);
    return true;
  };

  // a pair-mode pass: x in candidate 0, the others zero, d = 0 => y[.][0] = M_off x, y[.][V] = C_off x
  auto pair_pass = [&](const double (&xv)[E], double (&y)[E][NS]) -> bool {
    double x[E][V];
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int v = 0; v < V; ++v) x[e][v] = (v == 0) ? xv[e] : 0.0;
    }
    return pass(x, 0.0, y);
  };

  int64_t n_passes = 0, n_trials = 0;
  double y[E][NS];
  double d = 0.0, F = 0.0, s = 0.0;
  int i_ = 0;

  const long long ts1 = A.stamps ? wall_clock64() : 0;
  // ---- clipper.cpp:193-198 — u = M_off u0 + u0 (or u0), normalised --------------------------
  if (A.rescale) {
    if (!pair_pass(u, y)) return;
    ++n_passes;
#pragma unroll
    for (int e = 0; e < E; ++e) u[e] = valid[e] ? y[e][0] + u[e] : 0.0;
  }
  {
    double z[1] = {0.0};
#pragma unroll
    for (int e = 0; e < E; ++e) z[0] += u[e] * u[e];
    block_reduce<1, RS_NWV>(z, red);
    const double n0 = sqrt(z[0]);
#pragma unroll
    for (int e = 0; e < E; ++e) u[e] = valid[e] ? u[e] / n0 : 0.0;
  }
  // ---- :200-209 — initial penalty -------------------------------------------------------------
  {
    double sv[1] = {0.0};
#pragma unroll
    for (int e = 0; e < E; ++e) sv[0] += u[e];
    block_reduce<1, RS_NWV>(sv, red);
    s = sv[0];
    if (!pair_pass(u, y)) return;
    ++n_passes;
    double ca[2] = {0.0, 0.0};
#pragma unroll
    for (int e = 0; e < E; ++e) {
      a[e] = y[e][0];
      b[e] = y[e][V];
      const double cbu = s - b[e] - u[e];              // :202
      if (valid[e] && cbu > P.eps && u[e] > P.eps) {   // :203
        ca[0] += 1.0;
        ca[1] += (a[e] + u[e]) / cbu;                  // :205-208
      }
    }
    block_reduce<2, RS_NWV>(ca, red);
    d = (ca[0] > 0.0) ? ca[1] / ca[0] : 0.0;
  }

  const long long ts2 = A.stamps ? wall_clock64() : 0;
  // ---- :218 — outer iterations -----------------------------------------------------------------
  while (i_ < P.maxoliters) {
    // :219-220 — gradient and objective at u under the current penalty
    {
      double f[1] = {0.0};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        g[e] = valid[e] ? (1 + d) * u[e] - d * s + a[e] + b[e] * d : 0.0;
        f[0] += u[e] * g[e];
      }
      block_reduce<1, RS_NWV>(f, red);
      F = f[0];
    }
    int j_ = 0;
    bool pen_ready = false;  // the penalty sums of the inner loop's last candidate are at hand
    double pen_cnt = 0.0, pen_rs = 0.0;
    if (P.maxiniters > 0) {
      double alpha = 1.0;
      int k_ = 0;
      for (;;) {  // one window of V step sizes alpha, alpha beta, ... from the current (u, g)
        stamp(0);
        double x[E][V], r[2 * V];
#pragma unroll
        for (int q = 0; q < 2 * V; ++q) r[q] = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          double al = alpha;
#pragma unroll
          for (int l = 0; l < V; ++l) {
            double t = u[e] + al * g[e];        // :235
            t = (t > 0.0) ? t : 0.0;            // :236
            x[e][l] = t;
            r[2 * l] += t * t;
            r[2 * l + 1] += t;
            al = al * P.beta;
          }
        }
        // the norms are needed only after the pass: their wave sums wait in LDS while it runs
#pragma unroll
        for (int q = 0; q < 2 * V; ++q) r[q] = wave_sum_to_lane63(r[q]);
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < 2 * V; ++q) red2[wave * 2 * V + q] = r[q];
        }
        if (!pass(x, d, y)) return;
        ++n_passes;
        double nrm[V], sx[V];
#pragma unroll
        for (int l = 0; l < V; ++l) {
          double z = red2[2 * l], t1 = red2[2 * l + 1];
#pragma unroll
          for (int w = 1; w < RS_NWV; ++w) {
            z += red2[w * 2 * V + 2 * l];
            t1 += red2[w * 2 * V + 2 * l + 1];
          }
          nrm[l] = (z > 0.0) ? sqrt(z) : 1.0;  // :237 Eigen normalize()
          sx[l] = t1 / nrm[l];
        }
        // :238-242, :253, and the penalty terms :268-274 of candidate 0
        double gn[E][V], an[E], bn[E], q2[2 * V + 2];
#pragma unroll
        for (int q = 0; q < 2 * V + 2; ++q) q2[q] = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          an[e] = y[e][0] / nrm[0];
          bn[e] = y[e][V] / nrm[0];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const double xi = x[e][v] / nrm[v];
            double gv;
            if (v == 0) {
              gv = (1 + d) * xi - d * sx[0] + an[e] + bn[e] * d;
              const double cbu = sx[0] - bn[e] - xi;
              if (valid[e] && cbu > P.eps && xi > P.eps) {
                q2[2 * V] += 1.0;
                q2[2 * V + 1] += fabs((an[e] + xi) / cbu);
              }
            } else {
              const double gs = y[e][v] / nrm[v];
              gv = (1 + d) * xi - d * sx[v] + gs;
            }
            gv = valid[e] ? gv : 0.0;
            gn[e][v] = gv;
            x[e][v] = xi;  // from here on the normalised candidate
            q2[2 * v] += xi * gv;              // :242
            const double du = xi - u[e];
            q2[2 * v + 1] += du * du;          // :253
          }
        }
        block_reduce<2 * V + 2, RS_NWV>(q2, red);
        stamp(6);
        ++stamp_row;
        // :244-251 — walk the window in the reference's order
        int jstar = -1;
        double Fnew = 0.0, deltaF = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (jstar < 0) {
            ++n_trials;
            Fnew = q2[2 * v];
            deltaF = Fnew - F;        // :244
            bool accept = true;
            if (deltaF < -P.eps) {    // :246-248
              alpha = alpha * P.beta;
              ++k_;
              if (k_ < P.maxlsiters) accept = false;
            }
            if (accept) jstar = v;
          }
        }
        if (jstar < 0) continue;  // all V rejected: V more factors of beta are in alpha
        double du2 = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (v == jstar) {
            du2 = q2[2 * v + 1];
            s = sx[v];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              u[e] = x[e][v];   // :256-258
              g[e] = gn[e][v];
            }
          }
        }
        const double deltau = sqrt(du2);
        F = Fnew;
        ++j_;
        if (jstar == 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            a[e] = an[e];
            b[e] = bn[e];
          }
        }
        if (deltau < P.tol_u || fabs(deltaF) < P.tol_F || j_ >= P.maxiniters) {  // :261, :226
          if (jstar == 0) {
            pen_ready = true;
            pen_cnt = q2[2 * V];
            pen_rs = q2[2 * V + 1];
          } else {
            if (!pair_pass(u, y)) return;
            ++n_passes;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              a[e] = y[e][0];
              b[e] = y[e][V];
            }
          }
          break;
        }
        alpha = 1.0;  // :227
        k_ = 0;
      }
    }
    // ---- :268-280 — penalty update ----------------------------------------------------------------
    if (!pen_ready) {
      double ca[2] = {0.0, 0.0};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const double cbu = s - b[e] - u[e];            // :268
        if (valid[e] && cbu > P.eps && u[e] > P.eps) { // :269
          ca[0] += 1.0;
          ca[1] += fabs((a[e] + u[e]) / cbu);          // :271-274
        }
      }
      block_reduce<2, RS_NWV>(ca, red);
      pen_cnt = ca[0];
      pen_rs = ca[1];
    }
    if (pen_cnt > 0.0) {
      d += pen_rs / pen_cnt;  // :276
      ++i_;
    } else {
      break;  // :278-280
    }
  }

  // ---- the end: unit 0 hands the result over ---------------------------------------------------
  const long long ts3 = A.stamps ? wall_clock64() : 0;
  if (unit == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int64_t i = tid + static_cast<int64_t>(e) * RS_NT;
      if (i < m) {
        A.u_dev[i] = u[e];
        if (A.host_u) __hip_atomic_store(A.host_u + i, u[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      SolveShared* sh = A.shared;
      sh->F = F;
      sh->d = d;
      sh->n_passes = n_passes;
      sh->n_trials = n_trials;
      sh->ifinal = i_;
      sh->ubp = 0;
      sh->ubv = 0;
      sh->done = 1;
      if (A.host) {
        HostMirror* hm = A.host;
        __hip_atomic_store(&hm->F, F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->d, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->n_passes, n_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->n_trials, n_trials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->iters, static_cast<int64_t>(epoch - A.epoch0), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ifinal, i_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ubp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->ubv, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hm->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (A.stamps) {
        long long* row = A.stamps + 500 * 8;
        row[0] = ts0;
        row[1] = ts1;
        row[2] = ts2;
        row[3] = ts3;
        row[4] = wall_clock64();
      }
    }
  }
}

}  // namespace clipper_hip
