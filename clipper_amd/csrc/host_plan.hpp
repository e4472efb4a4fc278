// host_plan.hpp — the two planners that run on the host between a build and its first pass, as pure
// functions of the slice directory (no HIP, no context): plan_pass (the work list of the streaming
// pass on the slices) and plan_resident (units, pieces and wave map of the resident solver).
// Included by clipper_hip.hip (host_matrix.hpp / host_resident.hpp) and by tests/cpp/test_planners.cpp,
// which checks on the CPU that every plan covers every step of every slice exactly once.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace clipper_plan {

// what the streaming pass's workgroup does: layout of clipper_hip::SliceWork (k_slices.hip.h)
struct Work {
  int strip, slot, t0, t1;
  int q0, q1, pad0, pad1;
};

struct PassConsts {
  int nw;   // column groups (waves) per workgroup     SL_NW
  int so;   // steps between two recorded step offsets  SL_SO
  int occ = 6;        // workgroups per CU the pass kernel is compiled for  SL_OCC
  int per_strip = 12; // work items per strip a large matrix is cut into (12: M itself, 8: a row view)
};

struct PassPlan {
  std::vector<Work> work;  // most expensive first
  int nslots = 1;          // partial-sum slots per column the tail adds
  uint64_t entries = 0;    // stored entries of the matrix
};

// L[cg * nchunks + k] = maxq | entries << 8 of slice (cg, k). target <= 0: the default number of
// workgroups — four per CU; for small matrices a quarter of the slices (fewer partial-sum slots for
// the tail to add) and exactly one per CU when that is close (a second short workgroup on a few CUs
// doubles those CUs' time): profiles/r02e_window_sweep.txt. LARGE matrices (round 3,
// profiles/r03_wgs_sweep.txt) get whole ROUNDS of the workgroups the chip holds at once (occ per CU):
// about per_strip items per strip — 1024 workgroups left a third of the 1536 places empty and every
// CU's last item uneven; with 3 - 8 rounds the dispatcher evens the items out: m = 100k 1.83 -> 1.57 ms
// per pass on M, 300k 16.8 -> 13.8 ms, at 8 - 16 partial-sum slots per strip for the tail.
inline void plan_pass(const uint32_t* L, int ncg, int nchunks, const PassConsts& K, int cus, double target,
                      double C0 /* cost of a chunk besides its steps, in steps */, PassPlan& out) {
  const int nstrips = (ncg + K.nw - 1) / K.nw;
  if (target <= 0.0) {
    const double quarter = static_cast<double>(ncg) * nchunks / 4.0;
    target = quarter < 1.5 * cus ? cus : std::min<double>(quarter, 4.0 * cus);
    const double places = static_cast<double>(K.occ) * cus;  // workgroups the chip holds at once
    const double want = static_cast<double>(K.per_strip) * nstrips;
    if (want >= 0.8 * places && quarter >= places)
      target = places * std::min(8.0, std::max(1.0, std::floor(want / places + 0.5)));
  }
  // (this runs between the fill and the first pass of every build: buffers are kept, the order is a
  // counting sort)
  // What a chunk of a strip costs: the longest of its waves' chains of lock-step steps. (Blending in the entries
  // the slices hold — the bytes view — was measured worse at every weight: profiles/r03_plan_cost_model.txt.)
  constexpr double entry_weight = 0.0;
  static thread_local std::vector<int> cost;
  static thread_local std::vector<double> bcost;
  static thread_local std::vector<Work> items;
  static thread_local std::vector<double> key;
  static thread_local std::vector<int> nslot_of;
  cost.resize(static_cast<size_t>(nstrips) * nchunks);
  bcost.assign(static_cast<size_t>(nstrips) * nchunks, 0.0);
  double total = 0.0;
  uint64_t entries = 0;
  for (int st = 0; st < nstrips; ++st) {
    int* crow = cost.data() + static_cast<size_t>(st) * nchunks;
    const int w1 = std::min(K.nw, ncg - st * K.nw);
    for (int k = 0; k < nchunks; ++k) crow[k] = 0;
    for (int w = 0; w < w1; ++w) {
      const uint32_t* lrow = L + static_cast<size_t>(st * K.nw + w) * nchunks;
      for (int k = 0; k < nchunks; ++k) {
        const uint32_t v = lrow[k];
        crow[k] = std::max(crow[k], static_cast<int>(v & 255u));
        entries += v >> 8;
        bcost[static_cast<size_t>(st) * nchunks + k] += static_cast<double>(v >> 8);
      }
    }
    for (int k = 0; k < nchunks; ++k) {
      double& b = bcost[static_cast<size_t>(st) * nchunks + k];
      b = (1.0 - entry_weight) * crow[k] + entry_weight * (b / (256.0 * w1) * 2.0);  // the chunk's blended cost
      total += b + C0;
    }
  }
  out.entries = entries;
  const double T = std::max(8.0, total / target);
  auto push = [&](double c, const Work& w) {
    items.push_back(w);
    key.push_back(c);
  };
  items.clear();
  key.clear();
  nslot_of.assign(static_cast<size_t>(nstrips), 0);
  int nslots = 1;
  for (int st = 0; st < nstrips; ++st) {
    int slot = 0, start = 0;
    double acc = 0.0;
    auto flush = [&](int end) {
      if (end > start) push(acc, Work{st, slot++, start, end, 0, 1 << 30, 0, 0});
      start = end;
      acc = 0.0;
    };
    const int* crow = cost.data() + static_cast<size_t>(st) * nchunks;
    const double* brow = bcost.data() + static_cast<size_t>(st) * nchunks;
    for (int k = 0; k < nchunks; ++k) {
      const int mq = crow[k];
      const double c = brow[k] + C0;
      if (c > 1.5 * T && mq >= 2 * K.so) {
        // a dense block's chunk: a chain of steps several times the average — cut by step range
        flush(k);
        const int parts = std::min(static_cast<int>(std::ceil(c / T)), (mq + K.so - 1) / K.so);
        const int per = ((mq + parts - 1) / parts + K.so - 1) / K.so * K.so;
        for (int q0 = 0; q0 < mq; q0 += per)
          push(std::min(per, mq - q0) + C0, Work{st, slot++, k, k + 1, q0, std::min(q0 + per, mq), 0, 0});
        start = k + 1;
      } else {
        acc += c;
        if (acc >= T) flush(k + 1);
      }
    }
    flush(nchunks);
    nslot_of[static_cast<size_t>(st)] = slot;
    nslots = std::max(nslots, slot);
  }
  for (int st = 0; st < nstrips; ++st)  // every (strip, slot) is written by some workgroup
    for (int slot = nslot_of[static_cast<size_t>(st)]; slot < nslots; ++slot)
      push(0.0, Work{st, slot, 0, 0, 0, 0, 0, 0});
  out.nslots = nslots;
  // most expensive first, stable: counting sort by the cost relative to the most expensive item
  constexpr int KEYS = 1024;
  const size_t nw = items.size();
  out.work.resize(nw);
  double cmax = 1e-9;
  for (size_t i = 0; i < nw; ++i) cmax = std::max(cmax, key[i]);
  const double scale = (KEYS - 1) / cmax;
  int count[KEYS + 1] = {0};
  for (size_t i = 0; i < nw; ++i) ++count[KEYS - 1 - static_cast<int>(key[i] * scale)];
  int run = 0;
  for (int b = 0; b < KEYS; ++b) {
    const int c = count[b];
    count[b] = run;
    run += c;
  }
  for (size_t i = 0; i < nw; ++i) out.work[static_cast<size_t>(count[KEYS - 1 - static_cast<int>(key[i] * scale)]++)] = items[i];
}

// ---- the resident solver (k_resident.hip.h) -----------------------------------------------------

struct Unit {  // layout of clipper_hip::ResidentUnit
  int cg0, ncgs;  // column groups [cg0, cg0 + ncgs)
  int k0, k1;     // chunks [k0, k1)
  int slot;       // which of the partial-sum slots of its columns this unit fills
  int pad0, pad1, pad2;
};

struct ResidentConsts {
  int nt;              // threads per workgroup                         RS_NT
  int nwv;             // waves per workgroup                           RS_NWV
  int tmax;            // slices a unit holds at most                   RS_TMAX
  int pmax;            // pieces a wave works on at most                RS_PMAX
  int maxe;            // elements per thread at most                   RS_MAXE
  uint32_t lds_max;    // dynamic LDS of a workgroup                    RS_LDS_MAX
  uint32_t red_bytes, tab_bytes, slice_pad;  //                         RS_RED_BYTES, RS_TAB_BYTES, RS_SLICE_PAD
  int so;              //                                               SL_SO
};

constexpr uint32_t so_bytes(int maxq, int so) { return static_cast<uint32_t>(((maxq + so - 1) / so * 4 + 15) & ~15); }
inline uint32_t xt_bytes(int V, int64_t mp, int nwv) {
  const uint32_t xt = static_cast<uint32_t>(mp) * V * 8u, sc = static_cast<uint32_t>(nwv) * (V + 1) * 64u * 8u;
  return xt > sc ? xt : sc;
}
// upper bound of the bytes of a slice from its directory word (maxq | entries << 8)
inline uint32_t slice_bound(uint32_t lq, uint32_t quad_bytes, int so) {
  const uint32_t maxq = lq & 255u, entries = lq >> 8;
  if (maxq == 0) return 16 + 64 + so_bytes(0, so);
  const uint32_t nquads = std::min<uint32_t>((entries + 3u * 64u) / 4u, 64u * maxq);
  return 16 + 64 + so_bytes(static_cast<int>(maxq), so) + nquads * (quad_bytes + 4u) + maxq * 12u;
}

struct ResidentPlan {
  bool ok = false;
  int V = 0, E = 0, maxslots = 1;
  uint32_t lds_slices = 0;
  uint64_t total_bound = 0;
  std::vector<Unit> units;
  std::vector<uint8_t> nsl;       // [ncg] slots to add for a column of the group
  std::vector<uint8_t> npieces;   // [unit * nwv + wave]
  std::vector<uint8_t> wave_cg;   // [unit * nwv + wave] column group of the unit the wave works for (255: none)
  std::vector<uint32_t> pieces;   // [(unit * nwv + wave) * pmax + j] = chunk - k0 | q0 << 8 | q1 << 16
};

// Decides whether the slices fit the resident solver and lays out its units: a column group x a range
// of chunks sized by slice_bound() so that a unit fits the LDS beside the x table; one unit when
// everything fits one workgroup. Then the pieces: the steps of a column group's slices dealt out to
// its waves in equal shares (the dense slices of an inlier block are chains ten times as long as the
// others: cut, they end together), waves to column groups in proportion to their steps.
inline void plan_resident(const uint32_t* L, int ncg, int nchunks, int64_t m, int64_t mp, int esize,
                          int max_units, int v_forced, const ResidentConsts& K, ResidentPlan& out) {
  out = ResidentPlan{};
  if (m > static_cast<int64_t>(K.maxe) * K.nt || mp > static_cast<int64_t>(K.maxe) * K.nt) return;  // above: the streaming launches win
  const int E = (mp <= K.nt) ? 1 : (mp <= 2 * K.nt ? 2 : 4);
  const uint32_t QBY = 4u * static_cast<uint32_t>(esize);
  std::vector<uint32_t> ub(static_cast<size_t>(ncg) * nchunks);
  uint64_t total = 0;
  for (size_t i = 0; i < ub.size(); ++i) {
    ub[i] = slice_bound(L[i], QBY, K.so);
    total += ub[i];
  }
  out.total_bound = total;
  int vmax = 1;  // at these sizes the line search rarely rejects: a window only adds arithmetic
  if (v_forced) vmax = v_forced;
  std::vector<Unit>& units = out.units;
  out.nsl.assign(static_cast<size_t>(ncg), 0);
  int V = 0;
  for (int v = vmax; v >= 1; v = (v_forced ? 0 : v / 2)) {
    units.clear();
    const uint32_t fixed = xt_bytes(v, mp, K.nwv) + K.red_bytes + K.tab_bytes;
    // everything in ONE workgroup: no exchange at all
    if (ncg <= K.nwv) {
      const uint32_t fixed1 = fixed + static_cast<uint32_t>(mp) * (v + 1) * 8u;
      if (fixed1 + K.slice_pad < K.lds_max && total <= K.lds_max - fixed1 - K.slice_pad && ncg * nchunks <= K.tmax) {
        units.push_back(Unit{0, ncg, 0, nchunks, 0, 0, 0, 0});
        std::fill(out.nsl.begin(), out.nsl.end(), static_cast<uint8_t>(1));
        V = v;
        out.lds_slices = K.lds_max - fixed1;
        break;
      }
    }
    if (fixed + K.slice_pad + 4096 >= K.lds_max) continue;
    const uint32_t cap = K.lds_max - fixed - K.slice_pad;
    bool ok = true;
    for (int cg = 0; cg < ncg && ok; ++cg) {
      int slot = 0, k0 = 0;
      uint32_t acc = 0;
      for (int k = 0; k < nchunks; ++k) {
        const uint32_t b = ub[static_cast<size_t>(cg) * nchunks + k];
        if (b > cap) {
          ok = false;
          break;
        }
        if (acc + b > cap || k - k0 >= K.tmax) {
          units.push_back(Unit{cg, 1, k0, k, slot++, 0, 0, 0});
          k0 = k;
          acc = 0;
        }
        acc += b;
      }
      units.push_back(Unit{cg, 1, k0, nchunks, slot++, 0, 0, 0});
      out.nsl[static_cast<size_t>(cg)] = static_cast<uint8_t>(slot);
    }
    if (ok && static_cast<int>(units.size()) <= max_units) {
      V = v;
      out.lds_slices = K.lds_max - fixed;
      break;
    }
  }
  if (V == 0) return;
  out.V = V;
  out.E = E;
  out.maxslots = 1;
  for (uint8_t x : out.nsl) out.maxslots = std::max<int>(out.maxslots, x);

  const size_t NWV = static_cast<size_t>(K.nwv), PM = static_cast<size_t>(K.pmax);
  out.pieces.assign(units.size() * NWV * PM, 0u);
  out.npieces.assign(units.size() * NWV, 0);
  out.wave_cg.assign(units.size() * NWV, 255);
  std::vector<int> T(NWV), nw(NWV);
  for (size_t ui = 0; ui < units.size(); ++ui) {
    const Unit& U = units[ui];
    // waves to column groups in proportion to their steps (every group at least one)
    std::fill(T.begin(), T.end(), 0);
    std::fill(nw.begin(), nw.end(), 0);
    for (int cgl = 0; cgl < U.ncgs; ++cgl) {
      const uint32_t* lrow = L + static_cast<size_t>(U.cg0 + cgl) * nchunks;
      for (int k = U.k0; k < U.k1; ++k) T[cgl] += static_cast<int>(lrow[k] & 255u);
      nw[cgl] = 1;
    }
    for (int spare = K.nwv - U.ncgs; spare > 0; --spare) {
      int best = 0;
      for (int cgl = 1; cgl < U.ncgs; ++cgl)
        if (static_cast<int64_t>(T[cgl]) * nw[best] > static_cast<int64_t>(T[best]) * nw[cgl]) best = cgl;
      ++nw[best];
    }
    int wave = 0;
    for (int cgl = 0; cgl < U.ncgs; ++cgl) {
      const uint32_t* lrow = L + static_cast<size_t>(U.cg0 + cgl) * nchunks;
      const int target = std::max(1, (T[cgl] + nw[cgl] - 1) / nw[cgl]);
      int kcur = U.k0, qcur = 0;
      for (int sub = 0; sub < nw[cgl]; ++sub, ++wave) {
        const size_t wv = ui * NWV + static_cast<size_t>(wave);
        out.wave_cg[wv] = static_cast<uint8_t>(cgl);
        int rem = (sub == nw[cgl] - 1) ? (1 << 30) : target, n = 0;
        while (rem > 0 && kcur < U.k1) {
          const int mq = static_cast<int>(lrow[kcur] & 255u);
          if (qcur >= mq) {
            ++kcur;
            qcur = 0;
            continue;
          }
          if (n == K.pmax) {
            if (sub == nw[cgl] - 1) return;  // does not fit the piece lists: streaming launches (ok stays false)
            break;
          }
          const int take = std::min(rem, mq - qcur);
          out.pieces[wv * PM + static_cast<size_t>(n++)] = static_cast<uint32_t>(kcur - U.k0) |
                                                           (static_cast<uint32_t>(qcur) << 8) |
                                                           (static_cast<uint32_t>(qcur + take) << 16);
          qcur += take;
          rem -= take;
        }
        out.npieces[wv] = static_cast<uint8_t>(n);
      }
    }
  }
  out.ok = true;
}

// ---- the resident solver on a row view (k_rv_resident.hip.h) -------------------------------------------

struct ViewUnit {  // layout of clipper_hip::RvrUnit
  int cg0, ncgs;   // column groups [cg0, cg0 + ncgs) x every chunk of the view
  int l0, l1;      // lanes [l0, l1) of each
  int pack;        // 0, or G = 64 / (l1 - l0) >= 2: a unit of ONE group whose slices are packed G chunks side by
                   // side — lane group g of packed slice s holds the unit's columns of chunk s * G + g — so that a
                   // step keeps all 64 lanes busy and the chains of steps are G times shorter
  int pad1, pad2, pad3;
};

struct ViewResidentPlan {
  bool ok = false;
  uint32_t lds_slices = 0;        // bytes of LDS the slices of a unit may take
  uint64_t entries = 0;           // stored entries of the view
  std::vector<ViewUnit> units;
  std::vector<uint8_t> npieces;   // [unit * nwv + wave]
  std::vector<uint8_t> wave_cg;   // [unit * nwv + wave] column group of the unit the wave works for (255: none)
  std::vector<uint32_t> pieces;   // [(unit * nwv + wave) * pmax + j] = chunk | q0 << 8 | q1 << 16
  // scratch of the planner, kept from call to call (a view is planned between two iterations of a solve)
  std::vector<uint64_t> ent, byt;
  std::vector<int> stp, sorted_stp, T, nw;
};

// Units of COMPLETE columns: a unit holds every chunk of its column groups (so that a column's sums never
// leave the workgroup), as many groups as balance the ENTRIES (the LDS gathers of a pass go with them) up
// to `target_units` workgroups; a group whose slices exceed a unit's LDS or its share of the entries (the
// dense inlier block) is split by LANES into equal parts. Then the pieces, as plan_resident deals them.
// xt_fixed: bytes of LDS a unit needs besides its slices.
inline void plan_view_resident(const uint32_t* L, int ncg, int nchunks, int esize, int target_units, int max_units,
                               uint32_t fixed_bytes, const ResidentConsts& K, ViewResidentPlan& out) {
  out.ok = false;
  out.units.clear();
  out.entries = 0;
  if (ncg < 1 || nchunks < 1 || nchunks > K.tmax || fixed_bytes + K.slice_pad + 8192 >= K.lds_max) return;
  const uint32_t cap = K.lds_max - fixed_bytes - K.slice_pad;
  out.lds_slices = K.lds_max - fixed_bytes;
  const uint32_t QBY = 4u * static_cast<uint32_t>(esize);
  std::vector<uint64_t>&ent = out.ent, &byt = out.byt;
  ent.assign(static_cast<size_t>(ncg), 0);
  byt.assign(static_cast<size_t>(ncg), 0);
  uint64_t total = 0;
  for (int cg = 0; cg < ncg; ++cg) {
    uint64_t e = 0, b = 0;
    for (int k = 0; k < nchunks; ++k) {
      const uint32_t lq = L[static_cast<size_t>(cg) * nchunks + k];
      e += lq >> 8;
      b += slice_bound(lq, QBY, K.so);
    }
    ent[static_cast<size_t>(cg)] = e;
    byt[static_cast<size_t>(cg)] = b;
    total += e;
  }
  out.entries = total;
  target_units = std::max(1, std::min(target_units, max_units));
  const double Te0 = std::max(256.0, static_cast<double>(total) / target_units);
  const int max_cgs = std::min(K.nwv, K.tmax / nchunks);  // one wave per group in the tail; slices per unit
  if (max_cgs < 1) return;
  // A pass's time in a unit goes with its lock-step STEPS (the CU's LDS pipe issues per wave-step, however few
  // lanes are busy: measured 0.45 us per step and wave on top of 4 us), and the steps of a sparse group are those
  // of its longest column: 30 .. 125 per group at the headline view for 3 400 .. 4 600 entries. A group far above
  // the median is cut into two packed halves (its long column weighs on one of them only, and each half walks
  // two chunks per step) — while the units still fit the chip.
  std::vector<int>&stp = out.stp, &sorted_stp = out.sorted_stp;
  stp.assign(static_cast<size_t>(ncg), 0);
  for (int c = 0; c < ncg; ++c) {
    int t = 0;
    for (int k = 0; k < nchunks; ++k) t += static_cast<int>(L[static_cast<size_t>(c) * nchunks + k] & 255u);
    stp[static_cast<size_t>(c)] = t;
  }
  sorted_stp = stp;
  std::nth_element(sorted_stp.begin(), sorted_stp.begin() + ncg / 2, sorted_stp.end());
  const double Tmed = std::max(1, sorted_stp[static_cast<size_t>(ncg / 2)]);
  std::vector<ViewUnit>& units = out.units;
  // Attempts: with the heavy groups split, without, then with more entries per unit — a view of 600 rows over
  // 12 000 columns has more column groups than the chip has places for units of one group each (round 4: such views
  // stayed with the streaming launches for no better reason)
  const double scales[] = {1.0, 1.0, 1.35, 1.8, 2.4, 3.2};
  for (int attempt = (nchunks >= 2 ? 0 : 1); attempt < 6; ++attempt) {
  const int heavy_split = attempt == 0 ? 1 : 0;
  const double Te = Te0 * scales[attempt];
  units.clear();
  int cg = 0;
  while (cg < ncg) {
    const uint64_t e = ent[static_cast<size_t>(cg)], b = byt[static_cast<size_t>(cg)];
    // (a lane subset keeps every step's padding and the header: its bytes fall a little slower than its lanes)
    int nsplit = std::max(static_cast<int>((b + cap - 1) / cap), static_cast<int>(std::floor(static_cast<double>(e) / (1.5 * Te) + 0.5)));
    nsplit = std::max(1, nsplit);
    if (nsplit == 1 && heavy_split && static_cast<double>(stp[static_cast<size_t>(cg)]) > 1.35 * Tmed) {
      units.push_back(ViewUnit{cg, 1, 0, 32, 2, 0, 0, 0});
      units.push_back(ViewUnit{cg, 1, 32, 64, 2, 0, 0, 0});
      ++cg;
      continue;
    }
    if (nsplit > 1) {
      // (lane parts of 32, 16 or 8 lanes: as many chunks side by side in a packed slice)
      int parts = 2;
      while (parts < nsplit) parts *= 2;
      while (parts < 8 && parts < nchunks) parts *= 2;  // (as few packed slices as the lanes allow: the chains of steps end sooner)
      while (parts < 8 && (b / parts) + static_cast<uint64_t>(nchunks) * (16 + 64 + 64 * 16) > cap) parts *= 2;
      if (parts > 8 || (b / parts) + static_cast<uint64_t>(nchunks) * (16 + 64 + 64 * 16) > cap) return;
      for (int p = 0; p < parts; ++p)
        units.push_back(ViewUnit{cg, 1, 64 * p / parts, 64 * (p + 1) / parts, parts, 0, 0, 0});
      ++cg;
      continue;
    }
    int n = 0;
    uint64_t eacc = 0, bacc = 0;
    while (cg + n < ncg && n < max_cgs) {
      const uint64_t e2 = ent[static_cast<size_t>(cg + n)], b2 = byt[static_cast<size_t>(cg + n)];
      if (n > 0 && (bacc + b2 > cap || static_cast<double>(eacc + e2) > 1.15 * Te || static_cast<double>(e2) > 1.5 * Te ||
                    (heavy_split && static_cast<double>(stp[static_cast<size_t>(cg + n)]) > 1.35 * Tmed)))
        break;
      eacc += e2;
      bacc += b2;
      ++n;
    }
    units.push_back(ViewUnit{cg, n, 0, 64, 0, 0, 0, 0});
    cg += n;
  }
  if (static_cast<int>(units.size()) <= max_units) break;
  }
  if (static_cast<int>(units.size()) > max_units) return;

  const size_t NWV = static_cast<size_t>(K.nwv), PM = static_cast<size_t>(K.pmax);
  out.pieces.assign(units.size() * NWV * PM, 0u);
  out.npieces.assign(units.size() * NWV, 0);
  out.wave_cg.assign(units.size() * NWV, 255);
  std::vector<int>&T = out.T, &nw = out.nw;
  T.assign(NWV, 0);
  nw.assign(NWV, 0);
  for (size_t ui = 0; ui < units.size(); ++ui) {
    const ViewUnit& U = units[ui];
    std::fill(T.begin(), T.end(), 0);
    std::fill(nw.begin(), nw.end(), 0);
    // (a packed unit: its slices are the packed ones, as long as the longest of the chunks in each)
    const int G = U.pack > 1 ? U.pack : 1;
    const int nsl = (nchunks + G - 1) / G;
    auto steps_of = [&](int cgl, int sidx) {
      const uint32_t* lrow = L + static_cast<size_t>(U.cg0 + cgl) * nchunks;
      int mq = 0;
      for (int g = 0; g < G && sidx * G + g < nchunks; ++g) mq = std::max(mq, static_cast<int>(lrow[sidx * G + g] & 255u));
      return mq;
    };
    for (int cgl = 0; cgl < U.ncgs; ++cgl) {
      for (int k = 0; k < nsl; ++k) T[static_cast<size_t>(cgl)] += steps_of(cgl, k);
      nw[static_cast<size_t>(cgl)] = 1;
    }
    for (int spare = K.nwv - U.ncgs; spare > 0; --spare) {  // waves to groups in proportion to their steps
      int best = 0;
      for (int cgl = 1; cgl < U.ncgs; ++cgl)
        if (static_cast<int64_t>(T[static_cast<size_t>(cgl)]) * nw[static_cast<size_t>(best)] >
            static_cast<int64_t>(T[static_cast<size_t>(best)]) * nw[static_cast<size_t>(cgl)])
          best = cgl;
      ++nw[static_cast<size_t>(best)];
    }
    int wave = 0;
    for (int cgl = 0; cgl < U.ncgs; ++cgl) {
      const int nwc = nw[static_cast<size_t>(cgl)];
      const int target = std::max(1, (T[static_cast<size_t>(cgl)] + nwc - 1) / nwc);
      int kcur = 0, qcur = 0;
      for (int sub = 0; sub < nwc; ++sub, ++wave) {
        const size_t wv = ui * NWV + static_cast<size_t>(wave);
        out.wave_cg[wv] = static_cast<uint8_t>(cgl);
        int rem = (sub == nwc - 1) ? (1 << 30) : target, n = 0;
        while (rem > 0 && kcur < nsl) {
          const int mq = steps_of(cgl, kcur);
          if (qcur >= mq) {
            ++kcur;
            qcur = 0;
            continue;
          }
          if (n == K.pmax) {
            if (sub == nwc - 1) return;  // does not fit the piece lists: the streaming launches keep the view
            break;
          }
          const int take = std::min(rem, mq - qcur);
          out.pieces[wv * PM + static_cast<size_t>(n++)] = static_cast<uint32_t>(kcur) | (static_cast<uint32_t>(qcur) << 8) |
                                                           (static_cast<uint32_t>(qcur + take) << 16);
          qcur += take;
          rem -= take;
        }
        out.npieces[wv] = static_cast<uint8_t>(n);
      }
    }
  }
  out.ok = true;
}

}  // namespace clipper_plan
