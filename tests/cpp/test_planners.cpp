// CPU test of the two host planners (clipper_amd/csrc/host_plan.hpp): on random slice directories —
// sizes on and off the edges, empty slices, dense blocks whose chunks must be cut by step range —
// every plan must cover every step of every slice exactly once.
//   plan_pass      work list of the streaming pass: per strip, the chunks are covered once by whole-range
//                  items or by step-range items that partition [0, maxq); every (strip, slot < nslots) is
//                  written by exactly one item; the stored-entry count is the directory's
//   plan_resident  units cover every (column group, chunk) once, slots are numbered 0..n-1 per group, a
//                  unit's slices fit its LDS share by the planner's own bound, the pieces of a unit
//                  partition the steps of each of its slices among waves mapped to that column group
//   g++ -std=c++17 -O1 -I clipper_amd/csrc tests/cpp/test_planners.cpp -o /tmp/t && /tmp/t
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>

#include "host_plan.hpp"

using namespace clipper_plan;

#define REQUIRE(c)                                                         \
  do {                                                                     \
    if (!(c)) {                                                            \
      std::printf("FAILED %s at line %d (case %s)\n", #c, __LINE__, g_case); \
      std::exit(1);                                                        \
    }                                                                      \
  } while (0)
static char g_case[128];

static std::vector<uint32_t> directory(std::mt19937& rng, int ncg, int nchunks, double density, double dense_frac) {
  // column lengths binomial-ish around density * 128 per slice; the last dense_frac of the matrix is a dense block
  std::vector<uint32_t> L(static_cast<size_t>(ncg) * nchunks);
  std::binomial_distribution<int> bin(128, density);
  for (int cg = 0; cg < ncg; ++cg)
    for (int k = 0; k < nchunks; ++k) {
      const bool dense = cg >= ncg * (1 - dense_frac) && k * 2 >= nchunks * 2 * (1 - dense_frac);
      int maxlen = 0, entries = 0;
      for (int l = 0; l < 64; ++l) {
        const int n = dense ? 128 - (rng() % 3) : bin(rng);
        maxlen = std::max(maxlen, n);
        entries += n;
      }
      if (density == 0.0 && !dense) maxlen = entries = 0;
      L[static_cast<size_t>(cg) * nchunks + k] = static_cast<uint32_t>((maxlen + 3) / 4) | (static_cast<uint32_t>(entries) << 8);
    }
  return L;
}

static size_t check_pass(const std::vector<uint32_t>& L, int ncg, int nchunks, int cus, double target,
                         const PassConsts K = PassConsts{4, 16}) {
  PassPlan P;
  plan_pass(L.data(), ncg, nchunks, K, cus, target, 2.0, P);
  const int nstrips = (ncg + K.nw - 1) / K.nw;
  uint64_t entries = 0;
  for (uint32_t v : L) entries += v >> 8;
  REQUIRE(P.entries == entries);
  std::map<std::pair<int, int>, int> slot_seen;
  std::vector<std::vector<std::vector<std::pair<int, int>>>> cover(nstrips, std::vector<std::vector<std::pair<int, int>>>(nchunks));
  for (const Work& w : P.work) {
    REQUIRE(w.strip >= 0 && w.strip < nstrips && w.slot >= 0 && w.slot < P.nslots);
    const int times = ++slot_seen[std::make_pair(w.strip, w.slot)];
    REQUIRE(times == 1);
    REQUIRE(w.t0 >= 0 && w.t0 <= w.t1 && w.t1 <= nchunks);
    for (int k = w.t0; k < w.t1; ++k) cover[w.strip][k].push_back(std::make_pair(w.q0, w.q1));
    if (w.t0 == w.t1) continue;  // a filler: writes the zeros of a (strip, slot) nobody else fills
    if (w.q0 != 0 || w.q1 != (1 << 30)) REQUIRE(w.t1 - w.t0 == 1 && w.q0 < w.q1 && w.q0 % K.so == 0);
  }
  REQUIRE(static_cast<int>(slot_seen.size()) == nstrips * P.nslots);  // every (strip, slot) written once
  for (int st = 0; st < nstrips; ++st)
    for (int k = 0; k < nchunks; ++k) {
      int mq = 0;
      for (int w = 0; w < K.nw && st * K.nw + w < ncg; ++w)
        mq = std::max(mq, static_cast<int>(L[static_cast<size_t>(st * K.nw + w) * nchunks + k] & 255u));
      auto& c = cover[st][k];
      REQUIRE(!c.empty());
      if (c.size() == 1 && c[0].first == 0 && c[0].second == (1 << 30)) continue;  // one whole-range item
      std::sort(c.begin(), c.end());
      int at = 0;
      for (auto& pr : c) {
        REQUIRE(pr.first == at);
        at = pr.second;
      }
      REQUIRE(at == mq);  // the step ranges partition [0, maxq of the strip's chunk)
    }
  return P.work.size();
}

static bool check_resident(const std::vector<uint32_t>& L, int ncg, int nchunks, int64_t m, int esize, int max_units) {
  const ResidentConsts K{512, 8, 64, 12, 4, 159u * 1024u, 2 * 8 * 16 * 8, 64 * 4 + 64, 2048, 16};
  const int64_t mp = (m + 63) / 64 * 64;
  ResidentPlan P;
  plan_resident(L.data(), ncg, nchunks, m, mp, esize, max_units, 0, K, P);
  if (!P.ok) return false;
  REQUIRE(P.V == 1 && (P.E == 1 || P.E == 2 || P.E == 4) && static_cast<int64_t>(P.E) * K.nt >= mp);
  REQUIRE(static_cast<int>(P.units.size()) <= std::max(max_units, 1));
  // units cover every (cg, chunk) once; slots 0..n-1 per column group
  std::vector<std::vector<int>> seen(ncg, std::vector<int>(nchunks, 0));
  std::vector<std::set<int>> slots(ncg);
  for (size_t ui = 0; ui < P.units.size(); ++ui) {
    const Unit& U = P.units[ui];
    REQUIRE(U.ncgs >= 1 && U.ncgs <= K.nwv && U.cg0 >= 0 && U.cg0 + U.ncgs <= ncg && U.k0 >= 0 && U.k0 < U.k1 && U.k1 <= nchunks);
    REQUIRE(U.ncgs * (U.k1 - U.k0) <= K.tmax);
    uint64_t bytes = 0;
    for (int c = 0; c < U.ncgs; ++c) {
      const bool fresh = slots[U.cg0 + c].insert(U.slot).second;
      REQUIRE(fresh);
      for (int k = U.k0; k < U.k1; ++k) {
        ++seen[U.cg0 + c][k];
        bytes += slice_bound(L[static_cast<size_t>(U.cg0 + c) * nchunks + k], 4u * esize, K.so);
      }
    }
    REQUIRE(bytes + K.slice_pad <= P.lds_slices);
    // pieces: per column group of the unit, the steps of each slice are dealt out exactly once, to waves of that group
    std::map<std::pair<int, int>, std::vector<std::pair<int, int>>> steps;  // (cgl, kl) -> ranges
    std::vector<int> waves_of(U.ncgs, 0);
    for (int w = 0; w < K.nwv; ++w) {
      const size_t wv = ui * K.nwv + w;
      const int cgl = P.wave_cg[wv];
      if (cgl == 255) {
        REQUIRE(P.npieces[wv] == 0);
        continue;
      }
      REQUIRE(cgl < U.ncgs);
      ++waves_of[cgl];
      REQUIRE(P.npieces[wv] <= K.pmax);
      for (int j = 0; j < P.npieces[wv]; ++j) {
        const uint32_t pc = P.pieces[wv * K.pmax + j];
        const int kl = pc & 255, q0 = (pc >> 8) & 255, q1 = (pc >> 16) & 255;
        REQUIRE(kl < U.k1 - U.k0 && q0 < q1);
        steps[std::make_pair(cgl, kl)].push_back(std::make_pair(q0, q1));
      }
    }
    for (int c = 0; c < U.ncgs; ++c) {
      REQUIRE(waves_of[c] >= 1);
      for (int k = U.k0; k < U.k1; ++k) {
        const int mq = static_cast<int>(L[static_cast<size_t>(U.cg0 + c) * nchunks + k] & 255u);
        auto it = steps.find(std::make_pair(c, k - U.k0));
        if (mq == 0) {
          REQUIRE(it == steps.end());
          continue;
        }
        REQUIRE(it != steps.end());
        std::sort(it->second.begin(), it->second.end());
        int at = 0;
        for (auto& pr : it->second) {
          REQUIRE(pr.first == at);
          at = pr.second;
        }
        REQUIRE(at == mq);
      }
    }
  }
  for (int cg = 0; cg < ncg; ++cg) {
    REQUIRE(static_cast<int>(slots[cg].size()) == P.nsl[cg] && P.nsl[cg] <= P.maxslots);
    REQUIRE(*slots[cg].begin() == 0 && *slots[cg].rbegin() == P.nsl[cg] - 1);
    for (int k = 0; k < nchunks; ++k) REQUIRE(seen[cg][k] == 1);
  }
  return true;
}

// plan_view_resident: the units partition every (column group, lane) — a group whole in one unit, or split
// into lane ranges that tile [0, 64) — and hold ALL chunks of their groups; a unit fits its LDS share by the
// planner's own bound (lane subsets: their share of the quads + every step's padding); the pieces of a unit
// partition the steps of each of its slices among waves mapped to that column group.
static bool check_view_resident(const std::vector<uint32_t>& L, int ncg, int nchunks, int esize, int target, int max_units,
                                uint32_t fixed) {
  const ResidentConsts K{512, 8, 64, 12, 2, 159u * 1024u, 2 * 8 * 16 * 8 + 64 * 8, 64 * 8 + 64, 2048, 16};
  ViewResidentPlan P;
  plan_view_resident(L.data(), ncg, nchunks, esize, target, max_units, fixed, K, P);
  if (!P.ok) return false;
  REQUIRE(static_cast<int>(P.units.size()) <= max_units && !P.units.empty());
  std::vector<std::vector<int>> lanes(ncg, std::vector<int>(64, 0));
  uint64_t entries = 0;
  for (uint32_t v : L) entries += v >> 8;
  REQUIRE(P.entries == entries);
  for (size_t ui = 0; ui < P.units.size(); ++ui) {
    const ViewUnit& U = P.units[ui];
    REQUIRE(U.ncgs >= 1 && U.ncgs <= K.nwv && U.cg0 >= 0 && U.cg0 + U.ncgs <= ncg);
    REQUIRE(U.l0 >= 0 && U.l0 < U.l1 && U.l1 <= 64);
    REQUIRE(U.ncgs == 1 || (U.l0 == 0 && U.l1 == 64));
    REQUIRE(U.ncgs * nchunks <= K.tmax);
    // a lane part is a PACKED unit: G = 64 / width chunks side by side per slice
    const int G = U.pack > 1 ? U.pack : 1;
    REQUIRE((U.l0 == 0 && U.l1 == 64) ? U.pack == 0 : (U.pack >= 2 && U.pack <= 8 && (U.l1 - U.l0) * U.pack == 64 && U.l0 % (U.l1 - U.l0) == 0));
    const int nsl = (nchunks + G - 1) / G;
    auto steps_of = [&](int c, int sidx) {
      int mq = 0;
      for (int g = 0; g < G && sidx * G + g < nchunks; ++g)
        mq = std::max(mq, static_cast<int>(L[static_cast<size_t>(U.cg0 + c) * nchunks + sidx * G + g] & 255u));
      return mq;
    };
    uint64_t bytes = 0;
    for (int c = 0; c < U.ncgs; ++c) {
      for (int l = U.l0; l < U.l1; ++l) ++lanes[U.cg0 + c][l];
      for (int k = 0; k < nchunks; ++k) {
        const uint32_t lq = L[static_cast<size_t>(U.cg0 + c) * nchunks + k];
        const uint64_t full = slice_bound(lq, 4u * esize, K.so);
        bytes += (U.l0 == 0 && U.l1 == 64) ? full : full * (U.l1 - U.l0) / 64 + 16 + 64 + 64 * 16;
      }
    }
    REQUIRE(bytes + K.slice_pad <= P.lds_slices + 8192);  // (the kernel checks the real sizes; the plan must be close)
    std::map<std::pair<int, int>, std::vector<std::pair<int, int>>> steps;
    std::vector<int> waves_of(U.ncgs, 0);
    for (int w = 0; w < K.nwv; ++w) {
      const size_t wv = ui * K.nwv + w;
      const int cgl = P.wave_cg[wv];
      if (cgl == 255) {
        REQUIRE(P.npieces[wv] == 0);
        continue;
      }
      REQUIRE(cgl < U.ncgs);
      ++waves_of[cgl];
      REQUIRE(P.npieces[wv] <= K.pmax);
      for (int j = 0; j < P.npieces[wv]; ++j) {
        const uint32_t pc = P.pieces[wv * K.pmax + j];
        const int kl = pc & 255, q0 = (pc >> 8) & 255, q1 = (pc >> 16) & 255;
        REQUIRE(kl < nsl && q0 < q1);
        steps[std::make_pair(cgl, kl)].push_back(std::make_pair(q0, q1));
      }
    }
    for (int c = 0; c < U.ncgs; ++c) {
      REQUIRE(waves_of[c] >= 1);
      for (int k = 0; k < nsl; ++k) {
        const int mq = steps_of(c, k);
        auto it = steps.find(std::make_pair(c, k));
        if (mq == 0) {
          REQUIRE(it == steps.end());
          continue;
        }
        REQUIRE(it != steps.end());
        std::sort(it->second.begin(), it->second.end());
        int at = 0;
        for (auto& pr : it->second) {
          REQUIRE(pr.first == at);
          at = pr.second;
        }
        REQUIRE(at == mq);
      }
    }
  }
  for (int cg = 0; cg < ncg; ++cg)
    for (int l = 0; l < 64; ++l) REQUIRE(lanes[cg][l] == 1);
  return true;
}

int main() {
  std::mt19937 rng(20260926);
  int npass = 0, nres = 0, nres_ok = 0;
  const int64_t sizes[] = {1, 2, 63, 64, 65, 100, 129, 300, 511, 512, 513, 777, 1000, 1500, 2047, 2048, 2500, 4000, 10000, 30000};
  for (int64_t m : sizes)
    for (double density : {0.0, 0.01, 0.11, 0.4, 1.0})
      for (double dense_frac : {0.0, 0.05, 0.5}) {
        const int ncg = static_cast<int>((m + 63) / 64), nchunks = static_cast<int>((m + 127) / 128);
        std::snprintf(g_case, sizeof(g_case), "m=%lld density=%.2f dense=%.2f", static_cast<long long>(m), density, dense_frac);
        const auto L = directory(rng, ncg, nchunks, density, dense_frac);
        for (double target : {0.0, 64.0, 1024.0, 5000.0}) {
          check_pass(L, ncg, nchunks, 256, target);
          ++npass;
        }
        if (m <= 4096)
          for (int esize : {4, 8})
            for (int max_units : {64, 200}) {
              ++nres;
              nres_ok += check_resident(L, ncg, nchunks, m, esize, max_units) ? 1 : 0;
            }
      }
  // large matrices and their row views: whole rounds of the workgroups the chip holds (6 per CU)
  for (int nchunks : {782, 41}) {
    std::snprintf(g_case, sizeof(g_case), "m=100000 nchunks=%d", nchunks);
    const auto L = directory(rng, 1563, nchunks, 0.11, 0.05);
    for (int per_strip : {12, 8}) {
      const size_t nw = check_pass(L, 1563, nchunks, 256, 0.0, PassConsts{4, 16, 6, per_strip});
      ++npass;
      REQUIRE(nw >= 1536 && nw <= 8 * 1536 + 391 * 40);  // (+ the fillers of strips with fewer items than the fullest)
    }
  }
  // row views of a few hundred rows over thousands of columns (the headline: 524 rows x 10 000 columns, a dense
  // inlier block in the last column groups): complete columns per unit, dense groups split by lanes
  int nview = 0, nview_ok = 0;
  for (int ncg : {47, 79, 157, 235})
    for (int nchunks : {1, 3, 5, 8})
      for (double density : {0.02, 0.16, 0.5})
        for (int esize : {4, 8})
          for (int target : {40, 96, 170, 248}) {
            std::snprintf(g_case, sizeof(g_case), "view ncg=%d nchunks=%d density=%.2f esize=%d target=%d", ncg, nchunks, density, esize, target);
            auto L = directory(rng, ncg, nchunks, density, 0.0);
            for (int cg = ncg - std::max(1, ncg / 18); cg < ncg; ++cg)  // the inlier block: every row of the view is stored
              for (int k = 0; k < nchunks; ++k) L[static_cast<size_t>(cg) * nchunks + k] = 32u | (uint32_t(128 * 64) << 8);
            const uint32_t fixed = std::max<uint32_t>(nchunks * 128 * 6 * 8, 8 * 7 * 64 * 8) + (2 * 8 * 16 * 8 + 64 * 8) + (64 * 8 + 64);
            ++nview;
            nview_ok += check_view_resident(L, ncg, nchunks, esize, target, 248, fixed) ? 1 : 0;
          }
  // a view with more column groups than the chip has places for units of one group each (600 rows x 12 500 columns:
  // 196 groups, ten of them dense): the planner packs more entries per unit until the units fit
  for (int ncg : {196, 223})
    for (int esize : {4, 8}) {
      std::snprintf(g_case, sizeof(g_case), "wide view ncg=%d esize=%d", ncg, esize);
      auto L = directory(rng, ncg, 6, 0.16, 0.0);
      for (int cg = ncg - 10; cg < ncg; ++cg)
        for (int k = 0; k < 6; ++k) L[static_cast<size_t>(cg) * 6 + k] = 32u | (uint32_t(128 * 64) << 8);
      const uint32_t fixed = 6 * 128 * 6 * 8 + (2 * 8 * 16 * 8 + 64 * 8) + (64 * 8 + 64);
      ++nview;
      const bool ok = check_view_resident(L, ncg, 6, esize, 170, 248, fixed);
      nview_ok += ok ? 1 : 0;
      REQUIRE(ok || esize == 8);
    }
  REQUIRE(nview_ok > nview / 2);
  REQUIRE(nres_ok > nres / 4);  // (the small and the sparse ones fit)
  std::printf("view-resident plans: %d (%d fit the chip)\n", nview, nview_ok);
  std::printf("planners ok: %d pass plans, %d resident plans (%d fit the chip)\n", npass, nres, nres_ok);
  return 0;
}
