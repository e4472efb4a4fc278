// dsd_host.h — exact densest-subgraph rounding (`Rounding::DSD`) on the host.
//
// Reference: CLIPPER::findDenseClique, src/clipper.cpp:294-300 — the subgraph induced by the
// non-zero entries of u goes to dsd::solve (src/dsd.cpp:274-320), Goldberg's flow-based exact
// algorithm (dsd.cpp:171-245): bisection on the density g over [0, m/2] while n(n-1)(U-L) >= 1;
// for each g a minimum cut of the network  source -> v (capacity m/2),  v -> sink
// (m/2 + 2g - degree(v)),  u <-> v (w_uv)  decides: if only the source is on the source side,
// g is too large (U = g), else L = g and that source side is the answer so far.
//
// Written for what this path produces — a COMPLETE weighted graph on k nodes (every pair of S is
// an edge of the reference's list, zero weights included): no adjacency lists. The pair arcs are
// one k x k matrix of residual capacities (R[a][b] = W[a][b] - flow a->b, the flow antisymmetric);
// Dinic's BFS levels and DFS with current-arc pointers walk the dense rows, from a seeded feasible flow.
// The minimal source side of a minimum cut is unique, so the node set equals the reference's; the strict
// `flow < cap` tests are its own.
// The reference runs one maximum flow per bisection step (~45 at m = 10k); densest_subgraph() below
// replays the bisection against a candidate density and certifies it with the two flows at its end
// points (see there). Serial host code on |S| = nnz(u) nodes, O(k^2) memory, the same place the reference
// spends it; the gather of the sub-matrix is on the device (k_slice_gather_sub / k_gather_sub).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

namespace clipper_hip {
namespace dsd {

class DenseCut {
 public:
  // W: k x k symmetric, ZERO diagonal (Problem sees to it)
  DenseCut(const double* W, int k) : W_(W), k_(k), R_(new double[static_cast<size_t>(k) * k]),
                                     fs_(k, 0.0), ft_(k, 0.0), ct_(k, 0.0), dist_(k + 2, -1),
                                     it_(k + 2, 0) {}  // (R_ is set by every solve(), not here)

  // minimum cut for source capacities `cs` and sink capacities ct[a]; returns the source side
  // (reachable set in the residual graph) over the k inner nodes
  void solve(double cs, const std::vector<double>& ct, std::vector<char>& side) {
    cs_ = cs;
    ct_ = ct;
    std::memcpy(R_.get(), W_, static_cast<size_t>(k_) * k_ * sizeof(double));
    seed();
    while (bfs()) {
      std::fill(it_.begin(), it_.end(), 0);
      while (augment()) {
      }
    }
    // reachability from the source: the nodes whose source arc has room, then along residual arcs
    side.assign(static_cast<size_t>(k_), 0);
    q_.clear();
    un_.clear();
    for (int a = 0; a < k_; ++a) {
      if (fs_[a] < cs_) {
        side[a] = 1;
        q_.push_back(a);
      } else {
        un_.push_back(a);
      }
    }
    for (size_t h = 0; h < q_.size() && !un_.empty(); ++h) {
      const double* r = R_.get() + static_cast<size_t>(q_[h]) * k_;
      size_t keep = 0;
      for (int b : un_) {
        if (r[b] > 0.0) {
          side[b] = 1;
          q_.push_back(b);
        } else {
          un_[keep++] = b;
        }
      }
      un_.resize(keep);
    }
  }

 private:
  // A feasible flow to start from (any maximum flow has the same minimal source side). Every node
  // forwards what its own sink arc takes: source -> a -> sink, min(cs, ct[a]). What is left is a
  // transport problem — nodes whose source arc still has room (degree above 2g) against nodes whose sink
  // arc has (degree below 2g) — and most of it goes over the direct arcs a -> b, row by row; the
  // phases below find the rest (paths over more than one inner arc).
  void seed() {
    un_.clear();  // the nodes whose sink arc has room
    for (int a = 0; a < k_; ++a) {
      const double f = std::max(0.0, std::min(cs_, ct_[a]));
      fs_[a] = f;
      ft_[a] = f;
      if (ct_[a] - f > 0.0) un_.push_back(a);
    }
    for (int a = 0; a < k_ && !un_.empty(); ++a) {
      double room = cs_ - fs_[a];
      if (!(room > 0.0)) continue;
      double* r = R_.get() + static_cast<size_t>(a) * k_;
      size_t keep = 0, i = 0;
      for (; i < un_.size() && room > 0.0; ++i) {
        const int b = un_[i];
        const double want = ct_[b] - ft_[b];
        const double d = std::min(room, std::min(want, r[b]));
        if (d > 0.0) {
          r[b] -= d;
          R_[static_cast<size_t>(b) * k_ + a] += d;
          ft_[b] += d;
          room -= d;
        }
        if (ct_[b] - ft_[b] > 0.0) un_[keep++] = b;
      }
      for (; i < un_.size(); ++i) un_[keep++] = un_[i];
      un_.resize(keep);
      fs_[a] = cs_ - room;
    }
  }

  // node ids: 0 = source, 1..k = inner, k+1 = sink. Levels by breadth-first search over the residual
  // arcs; a row is compared against the nodes that have no level yet only (on these nearly complete
  // graphs the first few rows label everything).
  bool bfs() {
    std::fill(dist_.begin(), dist_.end(), -1);
    dist_[0] = 0;
    q_.clear();
    un_.clear();
    for (int a = 0; a < k_; ++a) {
      if (fs_[a] < cs_) {
        dist_[a + 1] = 1;
        q_.push_back(a);
      } else {
        un_.push_back(a);
      }
    }
    for (size_t h = 0; h < q_.size(); ++h) {
      const int a = q_[h];
      const int da = dist_[a + 1];
      if (dist_[k_ + 1] >= 0 && da >= dist_[k_ + 1]) break;  // (deeper levels cannot lie on a shortest path)
      const double* r = R_.get() + static_cast<size_t>(a) * k_;
      size_t keep = 0;
      for (int b : un_) {
        if (r[b] > 0.0) {
          dist_[b + 1] = da + 1;
          q_.push_back(b);
        } else {
          un_[keep++] = b;
        }
      }
      un_.resize(keep);
      if (dist_[k_ + 1] < 0 && ft_[a] < ct_[a]) dist_[k_ + 1] = da + 1;
    }
    return dist_[k_ + 1] >= 0;
  }

  // one augmenting path along level arcs (iterative DFS, current-arc pointers it_[node]:
  // target 0 = the sink, target b + 1 = inner node b)
  bool augment() {
    path_.clear();
    int cur = -1;  // -1 = at the source
    while (true) {
      if (cur < 0) {
        int& i = it_[0];
        bool adv = false;
        for (; i < k_; ++i)
          if (fs_[i] < cs_ && dist_[i + 1] == 1) {
            path_.push_back(i);
            cur = i;
            adv = true;
            break;
          }
        if (!adv) return false;
        continue;
      }
      const int a = cur;
      int& i = it_[a + 1];
      bool adv = false, at_sink = false;
      const double* r = R_.get() + static_cast<size_t>(a) * k_;
      const int next = dist_[a + 1] + 1;
      if (i == 0) {
        if (ft_[a] < ct_[a] && dist_[k_ + 1] == next) at_sink = true;
        else i = 1;
      }
      if (!at_sink) {
        const int* dist1 = dist_.data() + 1;
        for (; i <= k_; ++i) {
          const int b = i - 1;
          if (r[b] > 0.0 && dist1[b] == next) {
            path_.push_back(b);
            cur = b;
            adv = true;
            break;
          }
        }
      }
      if (!at_sink) {
        if (adv) continue;
        // dead end: retreat one node and skip the arc that led here
        path_.pop_back();
        if (path_.empty()) {
          cur = -1;
          ++it_[0];
        } else {
          cur = path_.back();
          ++it_[cur + 1];
        }
        continue;
      }
      break;
    }
    // bottleneck and push: source -> path_[0] -> ... -> path_.back() -> sink
    double df = cs_ - fs_[path_[0]];
    for (size_t h = 0; h + 1 < path_.size(); ++h) {
      const double r = R_[static_cast<size_t>(path_[h]) * k_ + path_[h + 1]];
      if (r < df) df = r;
    }
    const int last = path_.back();
    if (ct_[last] - ft_[last] < df) df = ct_[last] - ft_[last];
    fs_[path_[0]] += df;
    for (size_t h = 0; h + 1 < path_.size(); ++h) {
      const int a = path_[h], b = path_[h + 1];
      R_[static_cast<size_t>(a) * k_ + b] -= df;
      R_[static_cast<size_t>(b) * k_ + a] += df;
    }
    ft_[last] += df;
    return df > 0.0;
  }

  const double* W_;
  int k_;
  double cs_ = 0.0;
  std::unique_ptr<double[]> R_;  // residual capacity of the inner arcs: W - (antisymmetric flow)
  std::vector<double> fs_, ft_, ct_;
  std::vector<int> dist_, it_, path_, q_, un_;
};

// The reference's bisection (dsd.cpp:200-241) with the test "is the minimum cut at density g more than
// the source alone" left to the caller: returns the last g that passed (L), the last that failed (U)
// and whether either bound ever moved. n_total: nodes of the WHOLE graph (termination, dsd.cpp:219).
struct Bisection {
  double L, U;
  bool moved_L, moved_U;
};
template <typename NonTrivial>
inline Bisection bisect(double half, int64_t n_total, NonTrivial&& nontrivial) {
  Bisection b{0.0, half, false, false};  // dsd.cpp:200-201
  const double nn = static_cast<double>(n_total) * static_cast<double>(n_total - 1);
  while (nn * (b.U - b.L) >= 1.0) {  // dsd.cpp:219
    const double g = (b.U + b.L) / 2;
    if (nontrivial(g)) {
      b.L = g;  // dsd.cpp:232-234
      b.moved_L = true;
    } else {
      b.U = g;  // dsd.cpp:229-230
      b.moved_U = true;
    }
  }
  return b;
}

struct Problem {
  std::vector<double>& W;
  int k;
  int64_t n_total;
  double half;
  std::vector<double> degree, ct;
  DenseCut net;
  Problem(std::vector<double>& W_, int k_, int64_t n_)
      : W(W_), k(k_), n_total(n_), half(static_cast<double>((static_cast<int64_t>(k_) * k_ - k_) / 2)),  // dsd.cpp:286, :25
        degree(static_cast<size_t>(k_), 0.0), ct(static_cast<size_t>(k_)), net(W_.data(), k_) {
    for (int a = 0; a < k; ++a) {
      W[static_cast<size_t>(a) * k + a] = 0.0;
      double s = 0.0;
      for (int b = 0; b < k; ++b) s += W[static_cast<size_t>(a) * k + b];  // dsd.cpp:190-195
      degree[static_cast<size_t>(a)] = s;
    }
  }
  // The source side of the minimum cut at density g (returns false = only the source). The flow sees g
  // only through the sink capacities, and in double precision these stop changing with g once the
  // bisection's interval is below their resolution (half = k(k-1)/2 dominates the sum): a capacity vector
  // seen before gives the result computed then — the same flow would compute the same bits.
  bool cut(double g, std::vector<char>& side) {
    for (int a = 0; a < k; ++a) ct[static_cast<size_t>(a)] = half + 2 * g - degree[static_cast<size_t>(a)];  // dsd.cpp:33
    for (const Seen& m : seen)
      if (m.ct == ct) {
        side = m.side;
        return m.any;
      }
    ++flows;
    net.solve(half, ct, side);
    bool any = false;
    for (char c : side) any = any || c;
    if (seen.size() >= 8) seen.erase(seen.begin());
    seen.push_back(Seen{ct, side, any});
    return any;
  }
  struct Seen {
    std::vector<double> ct;
    std::vector<char> side;
    bool any;
  };
  std::vector<Seen> seen;
  int flows = 0;  // maximum flows computed
};

// The reference's procedure as it stands: one maximum flow per bisection step (about
// log2(k^2 n^2 / 2) of them). The fast path below falls back to it, and the tests hold it against it.
inline std::vector<char> side_by_bisection(Problem& P) {
  std::vector<char> side, final_side;
  bisect(P.half, P.n_total, [&](double g) {
    if (!P.cut(g, side)) return false;
    final_side = side;
    return true;
  });
  return final_side;
}

// Edge weight inside a node set over its size (each pair once), in extended precision.
inline void weight_and_size(const std::vector<double>& W, int k, const std::vector<char>& in, long double& e,
                            long double& n) {
  e = 0.0L;
  n = 0.0L;
  for (int a = 0; a < k; ++a) {
    if (!in[static_cast<size_t>(a)]) continue;
    n += 1.0L;
    const double* w = W.data() + static_cast<size_t>(a) * k;
    long double s = 0.0L;
    for (int b = a + 1; b < k; ++b)
      if (in[static_cast<size_t>(b)]) s += w[b];
    e += s;
  }
}

// Greedy peeling (remove the node of least weighted degree, keep the densest prefix seen): a node set
// whose density is a lower bound of the maximum, and on the graphs this path produces — one dense
// cluster, a sparse fringe — usually the maximum itself. O(k^2).
inline std::vector<char> peel(const Problem& P) {
  const int k = P.k;
  std::vector<double> deg(P.degree);
  std::vector<char> alive(static_cast<size_t>(k), 1);
  std::vector<int> order;
  order.reserve(static_cast<size_t>(k));
  long double e = 0.0L;
  for (double d : deg) e += d;
  e /= 2;
  long double best = -1.0L;
  int best_removed = 0;
  for (int left = k; left >= 1; --left) {
    const long double dens = e / left;
    if (dens > best) {
      best = dens;
      best_removed = k - left;
    }
    int v = -1;
    double dv = std::numeric_limits<double>::infinity();
    for (int a = 0; a < k; ++a)
      if (alive[static_cast<size_t>(a)] && deg[static_cast<size_t>(a)] < dv) {
        v = a;
        dv = deg[static_cast<size_t>(a)];
      }
    if (v < 0) break;  // (only NaN degrees left)
    alive[static_cast<size_t>(v)] = 0;
    order.push_back(v);
    e -= dv;
    const double* w = P.W.data() + static_cast<size_t>(v) * k;
    for (int b = 0; b < k; ++b) deg[static_cast<size_t>(b)] -= w[b];
  }
  std::vector<char> in(static_cast<size_t>(k), 1);
  for (int i = 0; i < best_removed; ++i) in[static_cast<size_t>(order[static_cast<size_t>(i)])] = 0;
  return in;
}

// W: k x k symmetric weights (row-major, diagonal ignored) of the complete graph on the
// sub-graph's nodes; n_total: number of nodes of the WHOLE graph (it enters the reference's
// termination test, dsd.cpp:219). Returns the local indices (ascending) of the densest subgraph.
//
// The minimum cut at density g is more than the source alone exactly when g < d* = the maximum density
// (Goldberg): every step of the reference's bisection only compares its midpoint with d*. Given a
// candidate set of density d the bisection is REPLAYED: a midpoint further than `zone` from d (a few
// units in the last place of the capacities — beyond that the flow's own rounding cannot change its
// answer) is decided by "g < d", a midpoint inside the zone by the flow itself, exactly as the reference
// would. The replay ends at some (L, U) and is then CERTIFIED by the flows the reference ran there, where
// they were not run already: the cut at U must be the source alone (no set is denser than U — every step
// that lowered U was right, by monotonicity), the cut at L must not be (every step that raised L was
// right), and that cut is the answer, the set the reference remembers from its last successful step.
// Whenever a flow finds a set denser than the candidate, it becomes the candidate (Dinkelbach's step) and
// the replay is repeated. Two flows instead of ~45 at m = 10k when the peeling finds the optimum, up to
// ~8 where n_total^2 k^2 exhausts double precision and the last steps of the reference's bisection
// are decided inside the zone; anything unexpected falls back to the plain procedure.
inline std::vector<int32_t> densest_subgraph(std::vector<double>& W, int k, int64_t n_total,
                                             int* flows_out = nullptr) {
  std::vector<int32_t> out;
  if (k < 2) return out;
  Problem P(W, k, n_total);
  const long double zone = 8.0L * (std::nextafter(P.half, std::numeric_limits<double>::infinity()) - P.half);
  std::vector<char> final_side, side, last_side, cand = peel(P);
  bool done = false;
  for (int round = 0; round < 16 && !done; ++round) {
    long double e, n;
    weight_and_size(W, k, cand, e, n);
    if (n < 1.0L) break;
    bool improved = false, L_by_flow = false, U_by_flow = false;
    auto denser_than_candidate = [&](const std::vector<char>& set) {
      long double e2, n2;
      weight_and_size(W, k, set, e2, n2);
      return n2 >= 1.0L && e2 * n > e * n2;
    };
    const Bisection b = bisect(P.half, n_total, [&](double g) {
      if (improved) return false;  // (this replay is abandoned)
      const long double gap = static_cast<long double>(g) - e / n;
      if (gap > zone || gap < -zone) {
        const bool below = gap < 0.0L;
        (below ? L_by_flow : U_by_flow) = false;
        return below;
      }
      const bool any = P.cut(g, side);
      if (any && denser_than_candidate(side)) {
        cand = side;
        improved = true;
        return false;
      }
      if (any) last_side = side;
      (any ? L_by_flow : U_by_flow) = true;
      return any;
    });
    if (improved) continue;
    if (b.moved_U && !U_by_flow && P.cut(b.U, side)) {
      if (!denser_than_candidate(side)) break;  // (a set no denser than the candidate beyond the zone: plain procedure)
      cand = side;
      continue;
    }
    if (!b.moved_L) {
      final_side.assign(static_cast<size_t>(k), 0);  // no step ever succeeded: the reference returns nothing
    } else if (L_by_flow) {
      final_side = last_side;
    } else if (!P.cut(b.L, final_side)) {
      break;  // (the candidate's own density refuted beyond the zone: plain procedure)
    }
    done = true;
  }
  int flows = P.flows;
  if (!done) {
    final_side = side_by_bisection(P);
    flows = -P.flows;
  }
  if (flows_out) *flows_out = flows;  // (negative: the plain procedure ran)
  for (int a = 0; a < static_cast<int>(final_side.size()); ++a)
    if (final_side[static_cast<size_t>(a)]) out.push_back(a);
  return out;
}

// the plain procedure (tests, fall-back)
inline std::vector<int32_t> densest_subgraph_by_bisection(std::vector<double>& W, int k, int64_t n_total) {
  std::vector<int32_t> out;
  if (k < 2) return out;
  Problem P(W, k, n_total);
  const std::vector<char> final_side = side_by_bisection(P);
  for (int a = 0; a < static_cast<int>(final_side.size()); ++a)
    if (final_side[static_cast<size_t>(a)]) out.push_back(a);
  return out;
}

}  // namespace dsd
}  // namespace clipper_hip
