// k_rv_resident.hip.h — the resident solver ON A ROW VIEW: the iterations of findDenseClique that stream
// a row view of M (k_solver.hip.h, LIVE ROWS) as ONE launch, for views whose slices fit the LDS of the
// workgroups that share them (the headline problem: 524 rows x 10 000 columns, 4.8 MB).
// Part of kernels.hip.h (include that one): hand-written gfx950 device code of the CLIPPER hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_resident.hip.h"

namespace clipper_hip {

// ------------------------------------------------------------------------------------------
// Once a view covers the live rows, an iteration of the streaming solver at m = 10k is 16 us of pass
// (a chain of latencies over 4.8 MB), 5 us of tail and two kernel boundaries, 25 times over. Here
// those iterations (clipper.cpp:226-280: line-search windows, the penalty update, the start of the next
// outer iteration) run inside one launch of P workgroups that keep the view on chip:
//
//   * COLUMN OWNERS. A unit = column groups [cg0, cg0 + ncgs) of the view x ALL its chunks (every row
//     of the view), copied into LDS once. The sums of a unit's columns are therefore COMPLETE inside the
//     workgroup: no partial sums cross workgroups, and the elementwise tail of a column (k_tail's
//     expressions, clipper.cpp:237-242) is computed by the one thread that owns it. The slices of the
//     dense inlier block (524 rows x 64 columns, all stored: 168 KB) exceed one LDS: such a column group
//     is split by LANES — a unit takes lanes [l0, l1) of every step (the copy re-packs each step for the
//     lane subset; the pass code is the resident solver's, rs_wave_pass, unchanged).
//   * What everybody needs of everybody is SMALL, because x is zero outside the view's rows R: the
//     objective, the step norms, the penalty sums, the convergence tests (clipper.cpp:242-262, 268-276)
//     are sums over R only. Every workgroup keeps (u, gradF, a, b) on R (two rows per thread) and
//     computes all of these — and every decision of the line search — redundantly on the same bits,
//     as in the resident solver. The owners of the columns in R publish the candidates' gradients on R
//     (V + 2 numbers per row of the view); the columns outside R only ever answer one question — did a
//     row outside the view become live (gradient > 0)? — as one packed count per unit. One exchange per
//     iteration, and the data is the flag (self-describing granules {epoch, half a double}, write-
//     through stores, relaxed polls: k_resident.hip.h).
//   * It STARTS from a prepared pass (SolverState::resume, left by a decide-only launch after the view
//     was built) and LEAVES one behind the moment it cannot go on: a row outside the view became live,
//     the policy wants a smaller view, or its budget of exchanges is used up; or it finishes the solve.
//     Whatever it leaves is committed by the LAST unit to arrive, into a point slot the entry state does
//     not name: a launch that gives up (a time-out of the exchange) leaves the entry state untouched
//     and the streaming launches queued behind it carry on from there.
//
// Arithmetic: the streaming solver's expressions (k_tail, decide) on the same operands; what differs is
// the association of the sums over R (one fixed tree per workgroup here, 256-element blocks and chains
// there), as between any two work splits.
// ------------------------------------------------------------------------------------------
constexpr int RVR_NT = RS_NT;        // 512 threads: 8 waves, two per SIMD
constexpr int RVR_NWV = RS_NWV;
constexpr int RVR_TMAX = RS_TMAX;    // slices a unit holds at most
constexpr int RVR_PMAX = RS_PMAX;    // pieces a wave works on at most
constexpr int RVR_MAXROWS = 2 * RVR_NT;  // rows of the view: two per thread
constexpr int RVR_MAXUNITS = 256;
constexpr int RVR_NRED = 40;         // doubles per wave of the reductions' scratch

struct RvrUnit {
  int cg0, ncgs;  // column groups [cg0, cg0 + ncgs), ncgs <= 8 (one per wave in the tail)
  int l0, l1;     // lanes [l0, l1) of each (a proper subset only for a unit of one group)
  int pack;       // 0, or G = 64 / (l1 - l0): the unit's slices are PACKED — lane group g (lanes [g W, g W + W),
                  // W = l1 - l0) of packed slice s holds the columns [l0, l1) of chunk s G + g: every step keeps
                  // all 64 lanes busy, and the dense inlier block's chains of 32 steps per chunk are walked G
                  // chunks at a time. The unit's column c sits in lane c - l0 of EVERY lane group.
  int pad1, pad2, pad3;
};

struct RvrArgs {
  SliceView R;                  // the view: slices, directory, rowmap, nrows
  const RvrUnit* units;
  int nunits;
  const uint32_t* pieces;       // as ResidentArgs (k0 = 0)
  const uint8_t* npieces;
  const uint8_t* wave_cg;
  const int32_t* viewpos;       // [mp] position of association i among the view's rows, or -1
  SolverState* st;              // the state to resume from; overwritten only by a committed exit
  SolveShared* shared;
  HostMirror* host;             // pinned (may be null)
  double* host_u;               // pinned (may be null)
  SolverParams prm;
  int64_t m, mp;
  double* pt;                   // point slots [2][V][2][mp]
  unsigned long long* xb;       // [2][(V + 2) * RVR_MAXROWS + RVR_MAXUNITS][2] granules, zero at allocation
  unsigned long long epoch0;
  uint32_t* ctl;                // [0] error word, [1] arrivals at the exit (zero at launch), [4..5] unit 0's start (wall clock)
  uint32_t* giveup_host;        // pinned (may be null), 4 words of this launch for the host to read after the solve:
                                // [0] the error word of a launch that gave up (the host counts them and backs off:
                                // host_rv_resident.hpp); a launch that ran: [1] its iterations (exchanges), [2..3] its
                                // duration — unit 0's first instruction to the last unit's commit, 100 MHz wall-clock ticks
  uint32_t lds_slices;          // bytes of LDS the slices of a unit may take
  long long timeout_ticks;      // longest wait for the other units' granules, 100 MHz wall clock
  int max_exchanges;            // leave to the streaming launches after this many
  ViewPolicy rvp;
  int rv_rows;
  long long* stamps;            // measurement only (may be null): unit 0, [iteration][8] wall-clock stamps
  const uint8_t* marks;         // profiling only (may be null): the streaming iterations' marks (SolveArgs::marks) and
  uint8_t* kind;                // their pinned copy — a launch that ENDS the solve hands them over, as decide() does
  int64_t marks_n;              // the iterations queued before this launch
};
enum : uint32_t { RVR_ERR_LDS = 1, RVR_ERR_TIMEOUT = 2, RVR_ERR_STATE = 3, RVR_ERR_PEER = 4 /* host: another rank's launch gave up */ };

__host__ __device__ constexpr uint32_t rvr_xt_bytes(int V, int nrows) {
  const uint32_t xt = static_cast<uint32_t>((nrows + 127) / 128 * 128) * V * 8u;
  const uint32_t sc = RVR_NWV * (V + 1) * 64u * 8u;
  return xt > sc ? xt : sc;
}
constexpr uint32_t RVR_RED_BYTES = RVR_NWV * RVR_NRED * 8 + 64 * 8;  // the rounds' wave sums; totals, counts' scratch, norms
constexpr uint32_t RVR_TAB_BYTES = RVR_TMAX * 4 + 64 + RVR_TMAX * 4;  // slice offsets, a few words, slice sizes
__host__ __device__ constexpr int64_t rvr_xb_granules(int V) {
  return 2 * (static_cast<int64_t>(V + 2) * RVR_MAXROWS + RVR_MAXUNITS) * 2;
}

// Sums over the workgroup, LEFT IN LDS: tot[q] = the sum of v[q] over the 512 threads, in a fixed order (DPP
// wave sums, then the waves in order, by thread q). The totals are uniform: the code that follows reads the
// few it needs at a time (broadcast reads) instead of holding all of them in registers — the 26 sums of a
// window iteration as a register array cost 200 LDS reads per thread and spilled (tools/spill_report.py).
// Ends with a barrier: tot[] may be read at once.
template <int N>
__device__ __forceinline__ void rvr_reduce(double (&v)[N], double* part /* [RVR_NWV * N] */, double* tot /* [N] */) {
  static_assert(N <= RVR_NRED, "scratch");
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_sum_to_lane63(v[q]);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
#pragma unroll
    for (int q = 0; q < N; ++q) part[wave * N + q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double acc = part[threadIdx.x];
#pragma unroll
    for (int w = 1; w < RVR_NWV; ++w) acc += part[w * N + threadIdx.x];
    tot[threadIdx.x] = acc;
  }
  __syncthreads();
}
// a uniform double out of LDS, told to the compiler as such (scalar registers)
__device__ __forceinline__ double rvr_uni(const double* p) {
  const double x = *p;
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                          __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

__device__ __forceinline__ void rvr_publish(unsigned long long* gq, unsigned long long tag, double x) {
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(x));
  __hip_atomic_store(gq, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(gq + 1, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one value: true if both granules carry the tag
__device__ __forceinline__ bool rvr_poll(const unsigned long long* gq, unsigned long long tag, double& x) {
  const unsigned long long g0 = rs_ld_granule(gq), g1 = rs_ld_granule(gq + 1);
  x = __longlong_as_double(static_cast<long long>((g0 << 32) | (g1 & 0xffffffffull)));
  return ((g0 ^ tag) >> 32) == 0 && ((g1 ^ tag) >> 32) == 0;
}

// Both granules of a value with ONE 16-byte load past this CU's L1 and the XCD's L2 (sc1, as the compiler
// emits a relaxed agent-scope load): half the requests of a sweep. The load may tear between the two 8-byte
// granules — each carries its own epoch. The compiler does not track an asm load: rvr_wait_loads() before the
// first use, then rvr_tie() on every value so that no use is scheduled ahead of the wait.
typedef uint32_t rvr_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rvr_u4 rvr_ld16(const unsigned long long* gq) {
  rvr_u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(gq) : "memory");
  return v;
}
__device__ __forceinline__ void rvr_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void rvr_tie(rvr_u4& v) { asm volatile("" : "+v"(v)); }
// {half, tag, half, tag}: the value, and whether both granules carry the tag
__device__ __forceinline__ bool rvr_take(const rvr_u4& v, uint32_t tag32, double& x) {
  x = __hiloint2double(static_cast<int>(v.x), static_cast<int>(v.z));
  return v.y == tag32 && v.w == tag32;
}
// wave sum of a small non-negative integer with DPP moves: lane 63 holds the total
__device__ __forceinline__ int wave_isum_to_lane63(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

template <typename VT, int V>
__global__ __launch_bounds__(RVR_NT) void k_solve_view_resident(RvrArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rvr_lds[];
  constexpr int NS = V + 1;
  constexpr int QB = 4 * static_cast<int>(sizeof(VT));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  static_assert(V >= 2 && V <= 6 && (V % 2) == 0, "window of the streaming solver on slices");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m = A.m, mp = A.mp;
  const int nrows = static_cast<int>(A.R.nrows);
  const SolverParams P = A.prm;
  const int unit = blockIdx.x;
  const bool two = nrows > RVR_NT;  // a second row per thread (uniform): the work on it is skipped otherwise

  // ---- carve -------------------------------------------------------------------------------
  uint32_t off = 0;
  double* Xt = reinterpret_cast<double*>(rvr_lds + off);   // [rows][V]; reused as the waves' sums
  double* scr = Xt;                                        // [8 waves][NS][64]
  off += rvr_xt_bytes(V, nrows);
  double* red = reinterpret_cast<double*>(rvr_lds + off);  // block_reduce scratch
  double* tot = red + RVR_NWV * RVR_NRED;                  // [0 .. 31] totals of rvr_reduce / the counts' scratch, [32 .. 32 + V) nrm, [40 .. 40 + V) sx
  double* nrmL = tot + 32;
  double* sxL = tot + 40;
  off += RVR_RED_BYTES;
  uint32_t* tab = reinterpret_cast<uint32_t*>(rvr_lds + off);  // [RVR_TMAX] slice offsets, then a few words
  uint32_t* words = tab + RVR_TMAX;                            // [16]
  uint32_t* sizes = words + 16;                                // [RVR_TMAX] bytes of the slices as they lie in LDS
  off += RVR_TAB_BYTES;
  uint8_t* sl = rvr_lds + off;

  const long long ts0 = wall_clock64();
  if (unit == 0 && tid == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(A.ctl + 4), static_cast<unsigned long long>(ts0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- the state this launch starts from: a prepared pass, or nothing to do here --------------------
  const SolverState* st = A.st;
  const int e_done = A.shared->done, e_hold = st->hold, e_stage = st->stage, e_resume = st->resume;
  if (e_done || e_hold || e_stage != ST_PASS || (e_resume != 1 && e_resume != 2) || st->nout != 0 ||
      nrows > RVR_MAXROWS || nrows < 1 || P.maxiniters < 1)
    return;  // (uniform over the grid: every unit reads the same state; the streaming launches go on)

  // ---- this unit; its slices -> LDS (lanes [l0, l1) of every step) -------------------------------
  const RvrUnit U = A.units[unit];
  const unsigned long long wcg = *reinterpret_cast<const unsigned long long*>(A.wave_cg + static_cast<int64_t>(unit) * RVR_NWV);
  const int cgl = static_cast<int>((wcg >> (8 * wave)) & 255ull);
  const bool has_cg = cgl < U.ncgs && U.cg0 + cgl < A.R.ncg;
  const int G = U.pack > 1 ? U.pack : 1;          // chunks side by side in a slice as it lies in LDS
  const int Wd = U.l1 - U.l0;                     // lanes (= columns of a group) the unit owns
  const int S = (A.R.nchunks + G - 1) / G;        // slices per column group as they lie in LDS
  const int nslices = U.ncgs * S;                 // local slice t = cgl * S + (packed) chunk
  const bool whole = U.l0 <= 0 && U.l1 >= 64;
  const bool packed = G > 1;
  const int lgrp = packed ? lane / Wd : 0;        // this lane's group in a packed slice, its column's lane in M's slices
  const int lsrc = packed ? U.l0 + (lane - lgrp * Wd) : lane;
  const bool in_lanes = packed ? true : (lane >= U.l0 && lane < U.l1);
  const gbytes_t rdata = (gbytes_t)A.R.data;
  const CLIPPER_GLOBAL uint64_t* rpre = (const CLIPPER_GLOBAL uint64_t*)A.R.Pre;
  auto src_of = [&](int t) -> gbytes_t {  // (not packed) the view's slice behind local slice t
    const int tc = t / S, tk = t - tc * S;
    return rdata + 16 * rpre[static_cast<int64_t>(U.cg0 + tc) * A.R.nchunks + tk];
  };
  auto src_chunk = [&](int k) -> gbytes_t {  // (packed: one column group) the view's slice of chunk k
    return rdata + 16 * rpre[static_cast<int64_t>(U.cg0) * A.R.nchunks + k];
  };
  // sizes: wave w sizes the slices t = w, w + 8, ... as they will lie in LDS
  if (nslices <= RVR_TMAX) {
    for (int t = wave; t < nslices; t += RVR_NWV) {
      uint32_t bytes = 0;
      if (packed) {  // the lanes' lists: lane group g = the unit's columns in chunk t G + g
        const int k = t * G + lgrp;
        const bool have = k < A.R.nchunks;
        const gbytes_t sp = src_chunk(have ? k : 0);
        int mq = have ? static_cast<int>(reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(sp)[1]) : 0;
        const int tq = have ? static_cast<int>(sp[16 + lsrc]) : 0;
        mq = sl_wave_max(mq);
        bytes = 16 + 64 + sl_so_bytes(mq) + sl_steps_bytes(tq, mq, QB);
        if (lane == 0) sizes[t] = bytes;
        continue;
      }
      const gbytes_t sp = src_of(t);
      const uint32_t nb = reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(sp)[2];
      bytes = nb;
      if (!whole) {
        const int maxq = __builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(sp)[1]));
        const int tq = in_lanes ? static_cast<int>(sp[16 + lane]) : 0;
        bytes = 16 + 64 + sl_so_bytes(maxq) + sl_steps_bytes(tq, maxq, QB);
      }
      if (lane == 0) sizes[t] = bytes;
    }
  }
  __syncthreads();
  if (wave == 0) {
    const uint32_t bytes_t = (lane < nslices && nslices <= RVR_TMAX) ? sizes[lane] : 0u;
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
  const bool bad_plan = total + RS_SLICE_PAD > A.lds_slices || nslices > RVR_TMAX || U.ncgs > RVR_NWV;
  if (__syncthreads_or(bad_plan ? 1 : 0)) {
    if (tid == 0) {
      __hip_atomic_store(A.ctl, static_cast<uint32_t>(RVR_ERR_LDS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (A.giveup_host) __hip_atomic_store(A.giveup_host, static_cast<uint32_t>(RVR_ERR_LDS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;  // (the other units time out on this one's granules — or see the error word — and leave: nothing is committed)
  }
  for (int t = wave; t < nslices; t += RVR_NWV) {
    const uint32_t o = tab[t];
    if (packed) {
      // Packed slice t: lane L = g Wd + c holds column l0 + c of chunk t G + g. Every step of it is the quads
      // of its active lanes in lane order (the format's own rule); a lane finds its quad of step q in ITS
      // chunk's slice at the rank of its column among that slice's active lanes.
      const int k = t * G + lgrp;
      const bool have = k < A.R.nchunks;
      const gbytes_t src = src_chunk(have ? k : 0);
      const int mq_own = have ? static_cast<int>(reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(src)[1]) : 0;
      const int tot_dst = have ? static_cast<int>(src[16 + lsrc]) : 0;
      const int maxq = __builtin_amdgcn_readfirstlane(sl_wave_max(mq_own));
      uint8_t* dp = sl + o;
      if (lane < 4) reinterpret_cast<uint32_t*>(dp)[lane] = (lane == 1) ? static_cast<uint32_t>(maxq) : 0u;
      dp[16 + lane] = static_cast<uint8_t>(tot_dst);
      for (int b = lane * 4; b < sl_so_bytes(maxq); b += 256) *reinterpret_cast<uint32_t*>(dp + 16 + 64 + b) = 0u;
      uint8_t* dfb = dp + 16 + 64 + sl_so_bytes(maxq);
      // this lane's walk through its own chunk's slice: the lists of all 64 columns of that slice are needed
      // for the step sizes and the ranks — group by group (G <= 8), each group's wave-wide ballot
      gbytes_t sfb = src + 16 + 64 + sl_so_bytes(mq_own);  // (per lane)
      int tgs[8];  // lane l: the quads of column l of the slice of chunk t G + g
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int kg = t * G + g;
        tgs[g] = (g < G && kg < A.R.nchunks) ? static_cast<int>(src_chunk(kg)[16 + lane]) : 0;
      }
      for (int q = 0; q < maxq; ++q) {
        uint64_t ms_own = 0;  // the active columns of this lane's chunk at step q
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const uint64_t mg = __ballot(q < tgs[g]);
          if (g == lgrp) ms_own = mg;
        }
        const bool ad = q < tot_dst;
        const uint64_t md = __ballot(ad);
        const int cd = __builtin_amdgcn_readfirstlane(__popcll(md));
        const int cs = __popcll(ms_own);  // (per lane)
        if (ad) {
          const uint32_t rs_ = static_cast<uint32_t>(__popcll(ms_own & ((1ull << lsrc) - 1ull)));
          const uint32_t rd_ = sl_lane_rank(md);
          SliceQuad<VT> vq;
          vq.load(sfb + rs_ * QB);
          const uint32_t rq = *reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(sfb + cs * QB + rs_ * 4);
          vq.store(dfb + rd_ * QB);
          *reinterpret_cast<uint32_t*>(dfb + cd * QB + rd_ * 4) = rq;
        }
        sfb += cs * QB + ((cs * 4 + 15) & ~15);
        dfb += cd * QB + ((cd * 4 + 15) & ~15);
      }
      continue;
    }
    const gbytes_t src = src_of(t);
    if (whole) {
      const uint32_t nb = reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(src)[2];
      for (uint32_t b = lane * 16; b < nb; b += 64 * 16)
        *reinterpret_cast<u32x4*>(sl + o + b) = *reinterpret_cast<const CLIPPER_GLOBAL u32x4*>(src + b);
    } else {
      // the slice re-packed for the lane subset: same header, the other lanes' lists empty, every step
      // holds the quads of the subset's active lanes in lane order (the format's own rule)
      const int maxq = __builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(src)[1]));
      const int tot_src = static_cast<int>(src[16 + lane]);
      const int tot_dst = in_lanes ? tot_src : 0;
      uint8_t* dp = sl + o;
      if (lane < 4) reinterpret_cast<uint32_t*>(dp)[lane] = (lane == 1) ? static_cast<uint32_t>(maxq) : 0u;
      dp[16 + lane] = static_cast<uint8_t>(tot_dst);
      for (int b = lane * 4; b < sl_so_bytes(maxq); b += 256) *reinterpret_cast<uint32_t*>(dp + 16 + 64 + b) = 0u;
      gbytes_t sfb = src + 16 + 64 + sl_so_bytes(maxq);
      uint8_t* dfb = dp + 16 + 64 + sl_so_bytes(maxq);
      for (int q = 0; q < maxq; ++q) {
        const bool as = q < tot_src, ad = q < tot_dst;
        const uint64_t ms = __ballot(as), md = __ballot(ad);
        const int cs = __builtin_amdgcn_readfirstlane(__popcll(ms)), cd = __builtin_amdgcn_readfirstlane(__popcll(md));
        if (ad) {
          const uint32_t rs_ = sl_lane_rank(ms), rd_ = sl_lane_rank(md);
          SliceQuad<VT> vq;
          vq.load(sfb + rs_ * QB);
          const uint32_t rq = *reinterpret_cast<const CLIPPER_GLOBAL uint32_t*>(sfb + cs * QB + rs_ * 4);
          vq.store(dfb + rd_ * QB);
          *reinterpret_cast<uint32_t*>(dfb + cd * QB + rd_ * 4) = rq;
        }
        sfb += cs * QB + ((cs * 4 + 15) & ~15);
        dfb += cd * QB + ((cd * 4 + 15) & ~15);
      }
    }
  }
  const uint32_t* toff = tab + (has_cg ? cgl * S : 0);
  const int np = has_cg ? A.npieces[unit * RVR_NWV + wave] : 0;
  const uint32_t pc = (lane < RVR_PMAX) ? A.pieces[(static_cast<int64_t>(unit) * RVR_NWV + wave) * RVR_PMAX + lane] : 0u;

  // ---- the solver state (every thread, scalar loads) ---------------------------------------------
  double d = st->d, F = st->F, alpha = st->alpha, s = st->s;
  int i_ = st->i, j_ = st->j, k_ = st->k;
  const int e_ubp = st->ubp, e_ubv = st->ubv;
  int64_t n_passes = st->n_passes, n_trials = st->n_trials, n_iters = st->n_iters, n_view_passes = st->n_view_passes;
  int nlive = st->nlive, nout = 0;
  const int rv_builds = st->rv_builds, rv_last = st->rv_last, rv_backoff = st->rv_backoff;
  const int n_redo = st->n_redo;  // (the view policy's clock runs in iterations that were not repeats: k_solver.hip.h)

  // ---- row role: rows t and t + 512 of the view; column role: this thread's own column --------------
  const double* Ue = A.pt + ((static_cast<int64_t>(e_ubp) * V + e_ubv) * 2 + 0) * mp;
  const double* Ge = A.pt + ((static_cast<int64_t>(e_ubp) * V + e_ubv) * 2 + 1) * mp;
  double UR[2], GR[2], aR[2] = {0.0, 0.0}, bR[2] = {0.0, 0.0};
  bool rok[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = tid + e * RVR_NT;
    rok[e] = r < nrows;
    const int64_t i = rok[e] ? A.R.rowmap[r] : 0;
    UR[e] = rok[e] ? Ue[i] : 0.0;
    GR[e] = rok[e] ? Ge[i] : 0.0;
  }
  // (wave w owns column group cg0 + w in the tail; a packed unit: wave 0, lane c = column l0 + c)
  const int64_t col = static_cast<int64_t>(U.cg0 + wave) * 64 + (packed ? U.l0 + lane : lane);
  const bool cown = wave < U.ncgs && (packed ? lane < Wd : in_lanes) && col < m;
  const int rpos = cown ? A.viewpos[col] : -1;
  double u_c = cown ? Ue[col] : 0.0, g_c = cown ? Ge[col] : 0.0, a_c = 0.0, b_c = 0.0;
  __syncthreads();

  unsigned long long epoch = A.epoch0;
  int exchanges = 0;
  int stamp_row = 0;
  const long long ts1 = A.stamps ? wall_clock64() : 0;  // (the slices are in LDS)
  auto stamp_end = [&]() {  // row 500: launch start, slices in LDS, end, turns of the loop
    if (A.stamps && unit == 0 && tid == 0) {
      long long* row = A.stamps + 500 * 8;
      row[0] = ts0;
      row[1] = ts1;
      row[2] = wall_clock64();
      row[3] = stamp_row;
    }
  };
  auto stamp = [&](int c) {  // unit 0: every turn; every unit: turn 6 (rows behind the first 512 x 8 words)
    if (A.stamps && tid == 0) {
      if (unit == 0 && stamp_row < 500) A.stamps[stamp_row * 8 + c] = wall_clock64();
      if (stamp_row == 6 && c < 4 && unit < RVR_MAXUNITS) A.stamps[4096 + unit * 4 + c] = wall_clock64();
    }
  };
  unsigned long long* const xb0 = A.xb;
  const int64_t par_stride = rvr_xb_granules(V) / 2;
  auto sec_g = [&](int par, int v, int r) { return xb0 + par * par_stride + ((static_cast<int64_t>(v) * RVR_MAXROWS + r) << 1); };
  auto sec_c = [&](int par, int p) { return xb0 + par * par_stride + ((static_cast<int64_t>(V + 2) * RVR_MAXROWS + p) << 1); };

  // What is left behind: the point (u, gradF) of every column into slot (e_ubp ^ 1, 0) — the entry state
  // names (e_ubp, e_ubv), which stays as it is —, then the LAST unit to arrive commits the state.
  auto arrive_last = [&]() __attribute__((always_inline)) -> bool {  // true in every thread of the last unit to arrive
    __threadfence_system();  // (every thread: its stores of the point are out before the unit counts as arrived)
    __syncthreads();
    if (tid == 0) {
      const uint32_t k = __hip_atomic_fetch_add(A.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      words[1] = (k == static_cast<uint32_t>(A.nunits)) ? 1u : 0u;
    }
    __syncthreads();
    return words[1] != 0u;
  };
  auto report_run = [&]() __attribute__((always_inline)) {  // (the committing thread) iterations and duration of the launch, to the host
    if (A.giveup_host) {
      const unsigned long long t_begin = __hip_atomic_load(reinterpret_cast<unsigned long long*>(A.ctl + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long dt = static_cast<unsigned long long>(wall_clock64()) - t_begin;
      __hip_atomic_store(A.giveup_host + 1, static_cast<uint32_t>(exchanges), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(A.giveup_host + 2, static_cast<uint32_t>(dt & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(A.giveup_host + 3, static_cast<uint32_t>(dt >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  auto write_point = [&](bool to_host) __attribute__((always_inline)) {
    double* Ux = A.pt + ((static_cast<int64_t>(e_ubp ^ 1) * V + 0) * 2 + 0) * mp;
    double* Gx = A.pt + ((static_cast<int64_t>(e_ubp ^ 1) * V + 0) * 2 + 1) * mp;
    if (cown) {
      Ux[col] = u_c;
      Gx[col] = g_c;
      if (to_host && A.host_u) __hip_atomic_store(A.host_u + col, u_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };

  // ---- the loop: one turn = one iteration of the streaming solver (G + T) -----------------------------
  //   K_TRIAL  a window pass: V step sizes alpha, alpha beta, ... from the current (u, g)   (:234-262)
  //   K_PAIR   a pair-mode pass on x = u: M_off u and C_off u apart for the penalty update   (:268-271)
  //   K_BUILD  no pass: gradient and objective at u under the new penalty                   (:219-220)
  enum { K_TRIAL = 0, K_PAIR = 1, K_BUILD = 2 };
  int kind = (e_resume == 2) ? K_PAIR : K_TRIAL;
  bool want_out = false;  // a prepared pass is the next thing: leave it to the streaming launches
  for (;;) {
    stamp(0);
    // ---- leave? (only ever in front of a pass: what goes out is a PREPARED pass) ----------------------
    const bool out_now = kind != K_BUILD && (exchanges >= A.max_exchanges || (kind == K_TRIAL ? want_out : nout != 0));
    if (out_now) {
      // (a window whose point has a live row OUTSIDE the view — that is why it leaves — goes out with the RAW sums of
      // its norms over the view's rows, resume = 3: that row's candidates are not zero, the launch that runs the pass
      // adds the rows outside the view: k_solver.hip.h, iteration_head)
      const int resume = (kind == K_TRIAL) ? (nout != 0 ? 3 : 1) : 2;
      const bool window = kind == K_TRIAL;
      if (window) {  // the pending window's norms (:235-237) go out with the state
        double r2[2 * V];
#pragma unroll
        for (int q = 0; q < 2 * V; ++q) r2[q] = 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          double al = alpha;
#pragma unroll
          for (int l = 0; l < V; ++l) {
            double t = UR[e] + al * GR[e];
            t = (t > 0.0) ? t : 0.0;
            r2[2 * l] += t * t;
            r2[2 * l + 1] += t;
            al = al * P.beta;
          }
        }
        rvr_reduce<2 * V>(r2, red, tot);
      }
      write_point(false);
      if (arrive_last() && tid == 0) {
        SolverState* o = A.st;
        o->d = d;
        o->F = F;
        o->alpha = alpha;
        o->s = s;
#pragma unroll
        for (int l = 0; l < V; ++l) {
          const double nl = (window && tot[2 * l] > 0.0) ? sqrt(tot[2 * l]) : 1.0;
          o->nrm[l] = (resume == 3) ? tot[2 * l] : nl;
          o->sx[l] = (resume == 3) ? tot[2 * l + 1] : (window ? tot[2 * l + 1] / nl : 0.0);
        }
        o->sel = 0;
        o->ubp = e_ubp ^ 1;
        o->ubv = 0;
        o->phase = window ? static_cast<int>(PH_TRIAL) : static_cast<int>(PH_PENALTY);
        o->stage = ST_PASS;
        o->i = i_;
        o->j = j_;
        o->k = k_;
        o->n_passes = n_passes;
        o->n_trials = n_trials;
        o->n_iters = n_iters;
        o->nlive = nlive;
        o->nout = nout;
        o->view = 0;
        o->hold = 0;
        o->n_view_passes = n_view_passes;
        o->rv_builds = rv_builds;
        o->rv_last = rv_last;
        o->rv_backoff = rv_backoff;
        o->hold_slot = 0;
        o->hold_nlive = 0;
        o->resume = resume;
        o->weff = V;       // (the prepared pass multiplies the whole window; the streaming launches' policy starts over)
        o->zero_run = 0;
        report_run();
        if (A.host) {
          HostMirror* hm = A.host;
          __hip_atomic_store(&hm->nlive, nlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->nout, nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->n_view_passes, n_view_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->iters, n_iters, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      stamp_end();
      return;
    }

    // ---- the pass: X table (raw candidates of the view's rows) -> the COMPLETE sums y[0 .. V] of this
    // thread's own column (acc[0] = a, acc[1 .. V-1] = g_v, acc[V] = b: rs_wave_pass) ---------------------
    double y[NS];
#pragma unroll
    for (int v = 0; v < NS; ++v) y[v] = 0.0;
    if (kind != K_BUILD) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = tid + e * RVR_NT;
        if (r < nrows) {  // (rows behind the last one are never read: entries, padding included, name rows of the view)
          double al = alpha;
#pragma unroll
          for (int l = 0; l < V; ++l) {
            double t = UR[e] + al * GR[e];
            t = (t > 0.0) ? t : 0.0;
            if (kind == K_PAIR) t = (l == 0) ? UR[e] : 0.0;
            Xt[r * V + l] = t;
            al = al * P.beta;
          }
        }
      }
      __syncthreads();
      double acc[NS];
      rs_wave_pass<VT, V>(sl, toff, pc, np, 0, Xt, (kind == K_TRIAL) ? d : 0.0, acc, G, lgrp * SL_SUB);
      // the window's norms (:237): wave l sums candidate l over the rows of the X table (64 rows per step, one
      // DPP sum of two numbers: the thread-per-row sums of all 2 V numbers cost six times that in every wave)
      if (wave < V) {
        double z = 0.0, t1 = 0.0;
        if (kind == K_TRIAL) {
          for (int r = lane; r < nrows; r += 64) {
            const double c = Xt[r * V + wave];
            z += c * c;
            t1 += c;
          }
          z = wave_sum_to_lane63(z);
          t1 = wave_sum_to_lane63(t1);
        }
        if (lane == 63) {
          const double nl = (kind == K_TRIAL && z > 0.0) ? sqrt(z) : 1.0;  // Eigen normalize(): only if squaredNorm > 0
          nrmL[wave] = nl;
          sxL[wave] = (kind == K_TRIAL) ? t1 / nl : 0.0;
        }
      }
      __syncthreads();  // the X table is dead: its memory becomes the waves' sums
      if (packed) {  // a column sits in one lane of every lane group: their sums meet in a butterfly (fixed order)
        for (int o = Wd; o < 64; o <<= 1) {
#pragma unroll
          for (int v = 0; v < NS; ++v) acc[v] += __shfl_xor(acc[v], o);
        }
      }
#pragma unroll
      for (int v = 0; v < NS; ++v) scr[(wave * NS + v) * 64 + lane] = acc[v];
      __syncthreads();
#pragma unroll
      for (int v = 0; v < NS; ++v) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < RVR_NWV; ++w)  // the waves that worked for this thread's column group, in wave order
          if (static_cast<int>((wcg >> (8 * w)) & 255ull) == wave) sum += scr[(w * NS + v) * 64 + lane];
        y[v] = sum;
      }
      __syncthreads();  // (the next X table overwrites the sums)
    }
    stamp(1);
    // ---- the tail of this thread's own column (k_tail's expressions, :237-242), what it publishes -----------
    ++epoch;
    ++exchanges;
    const int par = static_cast<int>(epoch & 1ull);
    const unsigned long long tag = (epoch & 0xffffffffull) << 32;
    double gn_c[V], an_c = a_c, bn_c = b_c;
    uint32_t cbits = 0;  // bit v: this thread's column lies outside R and candidate v makes it live
    if (kind == K_TRIAL) {
      double al = alpha;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        double t = u_c + al * g_c;
        t = (t > 0.0) ? t : 0.0;
        const double nv = rvr_uni(nrmL + v), sv = rvr_uni(sxL + v);
        const double xi = t / nv;
        double gn;
        if (v == 0) {
          an_c = y[0] / nv;
          bn_c = y[V] / nv;
          gn = (1 + d) * xi - d * sv + an_c + bn_c * d;
        } else {
          const double gs = y[v] / nv;
          gn = (1 + d) * xi - d * sv + gs;
        }
        gn_c[v] = gn;
        if (cown && rpos < 0 && (xi > 0.0 || gn > 0.0)) cbits |= 1u << v;
        if (cown && rpos >= 0) rvr_publish(sec_g(par, v, rpos), tag, gn);
        al = al * P.beta;
      }
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) gn_c[v] = 0.0;
      if (kind == K_PAIR) {
        an_c = y[0];
        bn_c = y[V];
      } else if (cown) {  // K_BUILD (:219)
        gn_c[0] = (1 + d) * u_c - d * s + a_c + b_c * d;
        if (rpos < 0 && (u_c > 0.0 || gn_c[0] > 0.0)) cbits = 1u;
      }
    }
    if (kind != K_BUILD && cown && rpos >= 0) {
      rvr_publish(sec_g(par, V, rpos), tag, an_c);
      rvr_publish(sec_g(par, V + 1, rpos), tag, bn_c);
    }
    // (a, b) of candidate 0 replace the current ones whatever is accepted — as cab does in k_tail; they are
    // read only when they are the current point's (penalty update, start of an outer iteration)
    a_c = an_c;
    b_c = bn_c;
    {  // the unit's counts, 10 bits per candidate, as one value
      unsigned long long w = 0;
#pragma unroll
      for (int v = 0; v < V; ++v) w |= static_cast<unsigned long long>(__popcll(__ballot((cbits >> v) & 1u))) << (7 * v);
      if (lane == 0) reinterpret_cast<unsigned long long*>(tot)[wave] = w;
      __syncthreads();
      if (tid == 0) {
        unsigned long long pk = 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          unsigned long long tsum = 0;
          for (int w2 = 0; w2 < RVR_NWV; ++w2) tsum += (reinterpret_cast<unsigned long long*>(tot)[w2] >> (7 * v)) & 127ull;
          pk |= tsum << (10 * v);
        }
        rvr_publish(sec_c(par, unit), tag, __longlong_as_double(static_cast<long long>(pk)));
      }
      __syncthreads();
    }
    stamp(2);

    // ---- the exchange: every value of this thread's rows (and one unit's counts) until both its granules carry
    // the iteration's epoch — no drain, no flag, no barrier (k_resident.hip.h) ----------------------------------
    double gR[2][V], anR[2], bnR[2];
    unsigned long long cpk = 0;  // one unit's packed counts (threads < nunits)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      anR[e] = aR[e];
      bnR[e] = bR[e];
#pragma unroll
      for (int v = 0; v < V; ++v) gR[e][v] = 0.0;
    }
    {
      const uint32_t tag32 = static_cast<uint32_t>(epoch & 0xffffffffull);
      int fail = 0;
      const long long t_poll = wall_clock64();
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = tid + e * RVR_NT;
          if ((e == 0 || two) && rok[e] && kind != K_BUILD) {
            rvr_u4 w[V + 2];
            if (kind == K_TRIAL) {
#pragma unroll
              for (int v = 0; v < V; ++v) w[v] = rvr_ld16(sec_g(par, v, r));
            }
            w[V] = rvr_ld16(sec_g(par, V, r));
            w[V + 1] = rvr_ld16(sec_g(par, V + 1, r));
            rvr_wait_loads();
            if (kind == K_TRIAL) {
#pragma unroll
              for (int v = 0; v < V; ++v) {
                rvr_tie(w[v]);
                ok = rvr_take(w[v], tag32, gR[e][v]) && ok;
              }
            }
            rvr_tie(w[V]);
            rvr_tie(w[V + 1]);
            ok = rvr_take(w[V], tag32, anR[e]) && ok;
            ok = rvr_take(w[V + 1], tag32, bnR[e]) && ok;
          }
        }
        if (tid < A.nunits) {
          rvr_u4 w = rvr_ld16(sec_c(par, tid));
          rvr_wait_loads();
          rvr_tie(w);
          double cd;
          ok = rvr_take(w, tag32, cd) && ok;
          cpk = static_cast<unsigned long long>(__double_as_longlong(cd));
        }
        if (__all(ok)) break;
        if ((spins & 63u) == 63u || A.timeout_ticks < 0) {
          const bool late = wall_clock64() - t_poll > A.timeout_ticks;
          const uint32_t e2 = __hip_atomic_load(A.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (late || e2 != 0) {
            fail = 1;
            break;
          }
        }
      }
      if (__syncthreads_or(fail)) {  // nothing is committed: the streaming launches carry on from the entry state
        if (tid == 0) {
          uint32_t expect = 0;
          // (the one unit whose exchange ran out first reports; the others leave on its error word)
          if (__hip_atomic_compare_exchange_strong(A.ctl, &expect, static_cast<uint32_t>(RVR_ERR_TIMEOUT), __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
              A.giveup_host)
            __hip_atomic_store(A.giveup_host, static_cast<uint32_t>(RVR_ERR_TIMEOUT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
    }
    stamp(3);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      aR[e] = anR[e];
      bR[e] = bnR[e];
    }

    // ---- the sums of the iteration, from R (:242, :253, :268-274) and the units' counts, and what they mean
    // (decide()'s rules). One ROUND per quantity set: wave sums (DPP) -> LDS -> one barrier -> every thread adds
    // the eight wave sums in wave order. A window is walked LAZILY, candidate by candidate in the reference's order
    // (:244-251): the sums of candidate v + 1 are formed only if candidate v was rejected — 1.8 rounds per window
    // on the headline problem instead of the sums of all V.
    //   round buffers: rbuf[round][wave][4] doubles, ibuf[round][wave][4] ints (a round never reuses one within a turn)
    double* rbuf = red;
    int* ibuf = reinterpret_cast<int*>(red + V * RVR_NWV * 4);
    auto round_sums = [&](int rd, double (&dv)[3], int (&iv)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 3; ++q) dv[q] = wave_sum_to_lane63(dv[q]);
      if (wave * 64 < A.nunits) iv[2] = wave_isum_to_lane63(iv[2]);  // (the units' counts sit in the first threads)
      if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 3; ++q) rbuf[(rd * RVR_NWV + wave) * 4 + q] = dv[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) ibuf[(rd * RVR_NWV + wave) * 4 + q] = iv[q];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        double acc = rbuf[(rd * RVR_NWV) * 4 + q];
        int iacc = ibuf[(rd * RVR_NWV) * 4 + q];
#pragma unroll
        for (int w = 1; w < RVR_NWV; ++w) {
          acc += rbuf[(rd * RVR_NWV + w) * 4 + q];
          iacc += ibuf[(rd * RVR_NWV + w) * 4 + q];
        }
        dv[q] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(acc)),
                                 __builtin_amdgcn_readfirstlane(__double2loint(acc)));
        iv[q] = __builtin_amdgcn_readfirstlane(iacc);
      }
    };
    ++n_iters;
    if (kind != K_BUILD) {
      ++n_passes;
      ++n_view_passes;
    }
    ++stamp_row;
    bool penalty = false;
    int pen_cnt = 0;
    double pen_rs = 0.0;
    if (kind == K_TRIAL) {
      int jstar = -1;
      double Fnew = 0.0, deltaF = 0.0, du2 = 0.0;
      int lr = 0, no = 0;
      const double alpha_w = alpha;  // the step size of this window's candidate 0
      double al = alpha_w;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (jstar < 0) {  // (uniform) candidate v is evaluated only if every one before it was rejected
          // dv: F_v (:242), ||x_v - u||^2 (:253), [v = 0] the penalty's sum of ratios (:271-274)
          // iv: live rows of candidate v in R, [v = 0] the penalty's count, live columns outside R
          double dv[3] = {0.0, 0.0, 0.0};
          int iv[3] = {0, 0, 0};
          const double nv = rvr_uni(nrmL + v);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (e == 0 || two) {
              double t = UR[e] + al * GR[e];
              t = (t > 0.0) ? t : 0.0;
              const double xi = t / nv;
              const double gv = gR[e][v];
              bool live = false, pen = false;
              if (rok[e]) {
                dv[0] += xi * gv;
                const double du = xi - UR[e];
                dv[1] += du * du;
                live = xi > 0.0 || gv > 0.0;
                if (v == 0) {
                  const double cbu = rvr_uni(sxL) - bnR[e] - xi;
                  if (cbu > P.eps && xi > P.eps) {
                    pen = true;
                    dv[2] += fabs((anR[e] + xi) / cbu);
                  }
                }
              }
              iv[0] += __popcll(__ballot(live));
              if (v == 0) iv[1] += __popcll(__ballot(pen));
            }
          }
          iv[2] = (tid < A.nunits) ? static_cast<int>((cpk >> (10 * v)) & 1023ull) : 0;
          round_sums(v, dv, iv);
          if (v == 0) {
            pen_rs = dv[2];
            pen_cnt = iv[1];
          }
          ++n_trials;
          Fnew = dv[0];
          deltaF = Fnew - F;        // :244
          bool accept = true;
          if (deltaF < -P.eps) {    // :246-248
            alpha = alpha * P.beta;
            ++k_;
            if (k_ < P.maxlsiters) accept = false;
          }
          if (accept) {
            // :256-258 — the accepted candidate becomes the point (its raw value: the window's own chain of
            // step sizes from alpha_w, exactly what was staged and what k_tail forms)
            jstar = v;
            du2 = dv[1];
            lr = iv[0];
            no = iv[2];
            s = rvr_uni(sxL + v);
            double t = u_c + al * g_c;
            t = (t > 0.0) ? t : 0.0;
            u_c = t / nv;
            g_c = gn_c[v];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              double t2 = UR[e] + al * GR[e];
              t2 = (t2 > 0.0) ? t2 : 0.0;
              UR[e] = rok[e] ? t2 / nv : 0.0;
              GR[e] = rok[e] ? gR[e][v] : 0.0;
            }
          }
        }
        al = al * P.beta;
      }
      stamp(4);
      if (jstar < 0) continue;  // all V rejected: V more factors of beta are in alpha, the point is unchanged
      const double deltau = sqrt(du2);
      F = Fnew;
      ++j_;
      nout = no;
      nlive = lr + nout;
      if (deltau < P.tol_u || fabs(deltaF) < P.tol_F || j_ >= P.maxiniters) {  // :261, :226
        if (jstar == 0) {
          penalty = true;  // candidate 0 carries (a, b) and its penalty terms are summed
        } else {
          kind = K_PAIR;
          continue;
        }
      } else {
        alpha = 1.0;  // :227
        k_ = 0;
        want_out = nout != 0 || view_wanted_v(A.rvp, m, true, A.rv_rows, nlive, nout, n_iters + 1 - n_redo, rv_builds, rv_last, rv_backoff);
        continue;
      }
    } else if (kind == K_PAIR) {
      double dv[3] = {0.0, 0.0, 0.0};
      int iv[3] = {0, 0, 0};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double cbu = s - bR[e] - UR[e];             // :268
        bool pen = false;
        if (rok[e] && cbu > P.eps && UR[e] > P.eps) {    // :269
          pen = true;
          dv[2] += fabs((aR[e] + UR[e]) / cbu);           // :271-274
        }
        iv[1] += __popcll(__ballot(pen));
      }
      round_sums(0, dv, iv);
      pen_rs = dv[2];
      pen_cnt = iv[1];
      penalty = true;
      stamp(4);
    } else {  // K_BUILD (:219-220): the first window of the outer iteration is pending
      double dv[3] = {0.0, 0.0, 0.0};
      int iv[3] = {0, 0, 0};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        bool live = false;
        if (rok[e]) {
          const double gi = (1 + d) * UR[e] - d * s + aR[e] + bR[e] * d;  // :219
          GR[e] = gi;
          dv[0] += UR[e] * gi;  // :220
          live = UR[e] > 0.0 || gi > 0.0;
        }
        iv[0] += __popcll(__ballot(live));
      }
      iv[2] = (tid < A.nunits) ? static_cast<int>(cpk & 1023ull) : 0;
      round_sums(0, dv, iv);
      stamp(4);
      if (cown) g_c = gn_c[0];
      F = dv[0];
      nout = iv[2];
      nlive = iv[0] + nout;
      j_ = 0;
      alpha = 1.0;
      k_ = 0;
      kind = K_TRIAL;
      want_out = nout != 0 || view_wanted_v(A.rvp, m, true, A.rv_rows, nlive, nout, n_iters + 1 - n_redo, rv_builds, rv_last, rv_backoff);
      continue;
    }
    if (penalty) {  // :276-280
      bool done = true;
      if (pen_cnt > 0) {
        d += pen_rs / static_cast<double>(pen_cnt);
        ++i_;
        done = i_ >= P.maxoliters;
      }
      if (!done) {
        kind = K_BUILD;
        continue;
      }
      // ---- the end of the solve (decide(): ACT_DONE) ----------------------------------------------------------
      write_point(true);
      if (blockIdx.x == 0 && A.kind != nullptr && A.marks != nullptr) {
        const int64_t n = (A.marks_n < KIND_CAP) ? A.marks_n : KIND_CAP;
        for (int64_t i = tid; i < n; i += RVR_NT)
          __hip_atomic_store(A.kind + i, A.marks[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (arrive_last() && tid == 0) {
        SolverState* o = A.st;  // (the launches queued behind see `done`; the state only names the final point)
        o->ubp = e_ubp ^ 1;
        o->ubv = 0;
        SolveShared* sh = A.shared;
        sh->F = F;
        sh->d = d;
        sh->n_passes = n_passes;
        sh->n_trials = n_trials;
        sh->ifinal = i_;
        sh->ubp = e_ubp ^ 1;
        sh->ubv = 0;
        sh->done = 1;
        report_run();
        if (A.host) {
          HostMirror* hm = A.host;
          __hip_atomic_store(&hm->F, F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->d, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->n_passes, n_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->n_trials, n_trials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->n_view_passes, n_view_passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->nlive, nlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->nout, nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->iters, n_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->ifinal, i_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->ubp, e_ubp ^ 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->ubv, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&hm->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // (every unit's u went out before it arrived)
        }
      }
      stamp_end();
      return;
    }
  }
}

}  // namespace clipper_hip
