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
// one antisymmetric k x k flow matrix X (residual a->b = W[a][b] - X[a][b]); Dinic's BFS levels
// and DFS with current-arc pointers walk the dense rows. The minimal source side of a minimum cut
// is unique, so the node set equals the reference's; the strict `flow < cap` tests are its own.
// This is a serial O(44 * k^2 * phases) host computation on |S| = nnz(u) nodes, by design the
// same place the reference spends it; the hot path (affinity, passes over M) is on the device.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace clipper_hip {
namespace dsd {

class DenseCut {
 public:
  DenseCut(const double* W, int k) : W_(W), k_(k), X_(static_cast<size_t>(k) * k, 0.0),
                                     fs_(k, 0.0), ft_(k, 0.0), ct_(k, 0.0), dist_(k + 2, -1),
                                     it_(k + 2, 0) {}

  // minimum cut for source capacities `cs` and sink capacities ct[a]; returns the source side
  // (reachable set in the residual graph) over the k inner nodes
  void solve(double cs, const std::vector<double>& ct, std::vector<char>& side) {
    cs_ = cs;
    ct_ = ct;
    std::fill(X_.begin(), X_.end(), 0.0);
    std::fill(fs_.begin(), fs_.end(), 0.0);
    std::fill(ft_.begin(), ft_.end(), 0.0);
    while (bfs()) {
      std::fill(it_.begin(), it_.end(), 0);
      while (augment()) {
      }
    }
    // reachability from the source (node 0); inner node a is index a + 1
    side.assign(static_cast<size_t>(k_), 0);
    std::vector<int> st;
    for (int a = 0; a < k_; ++a)
      if (fs_[a] < cs_) {
        side[a] = 1;
        st.push_back(a);
      }
    while (!st.empty()) {
      const int a = st.back();
      st.pop_back();
      const double* w = W_ + static_cast<size_t>(a) * k_;
      const double* x = X_.data() + static_cast<size_t>(a) * k_;
      for (int b = 0; b < k_; ++b)
        if (!side[b] && b != a && x[b] < w[b]) {
          side[b] = 1;
          st.push_back(b);
        }
    }
  }

 private:
  // node ids: 0 = source, 1..k = inner, k+1 = sink
  bool bfs() {
    std::fill(dist_.begin(), dist_.end(), -1);
    dist_[0] = 0;
    std::vector<int> q;
    q.reserve(static_cast<size_t>(k_) + 2);
    for (int a = 0; a < k_; ++a)
      if (fs_[a] < cs_) {
        dist_[a + 1] = 1;
        q.push_back(a);
      }
    for (size_t h = 0; h < q.size(); ++h) {
      const int a = q[h];
      const int da = dist_[a + 1];
      const double* w = W_ + static_cast<size_t>(a) * k_;
      const double* x = X_.data() + static_cast<size_t>(a) * k_;
      for (int b = 0; b < k_; ++b)
        if (dist_[b + 1] < 0 && b != a && x[b] < w[b]) {
          dist_[b + 1] = da + 1;
          q.push_back(b);
        }
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
      const double* w = W_ + static_cast<size_t>(a) * k_;
      const double* x = X_.data() + static_cast<size_t>(a) * k_;
      for (; i <= k_; ++i) {
        if (i == 0) {
          if (ft_[a] < ct_[a] && dist_[k_ + 1] == dist_[a + 1] + 1) {
            at_sink = true;
            break;
          }
        } else {
          const int b = i - 1;
          if (b != a && x[b] < w[b] && dist_[b + 1] == dist_[a + 1] + 1) {
            path_.push_back(b);
            cur = b;
            adv = true;
            break;
          }
        }
      }
      if (at_sink) break;
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
    }
    // bottleneck and push: source -> path_[0] -> ... -> path_.back() -> sink
    double df = cs_ - fs_[path_[0]];
    for (size_t h = 0; h + 1 < path_.size(); ++h) {
      const int a = path_[h], b = path_[h + 1];
      const double r = W_[static_cast<size_t>(a) * k_ + b] - X_[static_cast<size_t>(a) * k_ + b];
      if (r < df) df = r;
    }
    const int last = path_.back();
    if (ct_[last] - ft_[last] < df) df = ct_[last] - ft_[last];
    fs_[path_[0]] += df;
    for (size_t h = 0; h + 1 < path_.size(); ++h) {
      const int a = path_[h], b = path_[h + 1];
      X_[static_cast<size_t>(a) * k_ + b] += df;
      X_[static_cast<size_t>(b) * k_ + a] -= df;
    }
    ft_[last] += df;
    return df > 0.0;
  }

  const double* W_;
  int k_;
  double cs_ = 0.0;
  std::vector<double> X_, fs_, ft_, ct_;
  std::vector<int> dist_, it_, path_;
};

// W: k x k symmetric weights (row-major, diagonal ignored) of the complete graph on the
// sub-graph's nodes; n_total: number of nodes of the WHOLE graph (it enters the reference's
// termination test, dsd.cpp:219). Returns the local indices (ascending) of the densest subgraph.
inline std::vector<int32_t> densest_subgraph(std::vector<double>& W, int k, int64_t n_total) {
  std::vector<int32_t> out;
  if (k < 2) return out;
  for (int a = 0; a < k; ++a) W[static_cast<size_t>(a) * k + a] = 0.0;
  const int64_t m = static_cast<int64_t>(k) * k - k;  // dsd.cpp:286
  const double half = static_cast<double>(m / 2);     // dsd.cpp:25 (`m / 2` on int64)
  std::vector<double> degree(static_cast<size_t>(k), 0.0), ct(static_cast<size_t>(k));
  for (int a = 0; a < k; ++a) {
    double s = 0.0;
    for (int b = 0; b < k; ++b) s += W[static_cast<size_t>(a) * k + b];  // dsd.cpp:190-195
    degree[a] = s;
  }
  double L = 0.0, U = half;  // dsd.cpp:200-201
  std::vector<char> side, final_side;
  DenseCut net(W.data(), k);
  const double nn = static_cast<double>(n_total) * static_cast<double>(n_total - 1);
  while (nn * (U - L) >= 1.0) {  // dsd.cpp:219
    const double g = (U + L) / 2;
    for (int a = 0; a < k; ++a) ct[a] = half + 2 * g - degree[a];  // dsd.cpp:33
    net.solve(half, ct, side);
    bool any = false;
    for (char c : side) any = any || c;
    if (!any) {
      U = g;  // dsd.cpp:229-230
    } else {
      L = g;
      final_side = side;
    }
  }
  for (int a = 0; a < static_cast<int>(final_side.size()); ++a)
    if (final_side[a]) out.push_back(a);
  return out;
}

}  // namespace dsd
}  // namespace clipper_hip
