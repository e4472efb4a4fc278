/**
 * @file dsd.h
 * @brief clipper::dsd::solve — exact densest subgraph (Goldberg's flow algorithm), the facade of
 *        the reference's include/clipper/dsd.h:22-55 / src/dsd.cpp:274-327.
 *
 * Like the reference's function this is a host computation on a matrix the caller holds; it
 * shares the flow solver with the device path's `Rounding::DSD` (clipper_amd/csrc/dsd_host.h),
 * where the induced sub-matrix is gathered from HBM first.
 */
#pragma once

#include <vector>

#include "../../clipper_amd/csrc/dsd_host.h"
#include "clipper/types.h"

namespace clipper {
namespace dsd {

/// A: dense weighted adjacency (symmetric; the upper triangle is read, the diagonal ignored);
/// S: optional restriction to a node subset. Returns the nodes of the densest subgraph, ascending.
inline std::vector<int> solve(const MatrixXd& A, const std::vector<int>& S_in = {}) {
  const int n = static_cast<int>(A.rows());
  std::vector<int> S = S_in;
  if (S.empty()) {
    S.resize(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) S[static_cast<size_t>(i)] = i;
  }
  const int k = static_cast<int>(S.size());
  std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      const int i = S[static_cast<size_t>(a)], j = S[static_cast<size_t>(b)];
      if (i != j) W[static_cast<size_t>(a) * k + b] = (i < j) ? A(i, j) : A(j, i);  // dsd.cpp:306
    }
  std::vector<int> nodes;
  for (int32_t a : clipper_hip::dsd::densest_subgraph(W, k, n)) nodes.push_back(S[static_cast<size_t>(a)]);
  return nodes;
}

}  // namespace dsd
}  // namespace clipper
