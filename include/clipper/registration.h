/**
 * @file registration.h
 * @brief The host-side neighbours of the CLIPPER hot path that the reference keeps in its
 *        benchmark utilities (benchmarks/bm_utils.h: namespace utils): PLY vertex reader, synthetic
 *        putative associations, precision / recall — plus the closed-form rigid transform that
 *        consumes the selected associations (examples/python/ex4_bunny.ipynb). Thin C++ wrappers
 *        over the C ABI (include/clipper_hip.h); no device work.
 */
#pragma once

#include <stdexcept>
#include <string>
#include <utility>

#include "../clipper_hip.h"
#include <cstdio>

#include "clipper/invariants/abstract.h"
#include "clipper/types.h"

namespace clipper {
namespace registration {

/// utils::read_ply (bm_utils.cpp:24-79). pts: 3 x n, one point per column. False on failure.
inline bool read_ply(const std::string& plyfile, invariants::Data& pts, bool silent = true) {
  const int64_t n = clipper_hip_read_ply_xyz(plyfile.c_str(), nullptr, 0);
  if (n < 0) {
    if (!silent) std::fprintf(stderr, "read_ply: %s\n", clipper_hip_last_error());
    return false;
  }
  pts = invariants::Data(3, n);
  return n == 0 || clipper_hip_read_ply_xyz(plyfile.c_str(), pts.data(), n) == n;
}

/// utils::generate_synthetic_correspondences (bm_utils.cpp:277-341); `seed` instead of the
/// reference's random_device. Returns {A, Agt}; two empty matrices when Agood is too small.
inline std::pair<Association, Association> generate_synthetic_correspondences(
    int64_t n0, int64_t n1, const Association& Agood, size_t m, double rho, uint64_t seed) {
  Association A(static_cast<int64_t>(m), 2), Agt(static_cast<int64_t>(m), 2);
  int64_t ni = 0;
  const int rc = clipper_hip_generate_synthetic_correspondences(
      n0, n1, Agood.data(), Agood.rows(), static_cast<int64_t>(m), rho, seed, A.data(), Agt.data(), &ni);
  if (rc == CLIPPER_HIP_E_STATE) return {};
  if (rc) throw std::invalid_argument(clipper_hip_last_error());
  Association G(ni, 2);
  for (int64_t i = 0; i < ni; ++i) {
    G(i, 0) = Agt.data()[i];
    G(i, 1) = Agt.data()[ni + i];
  }
  return {A, G};
}

/// utils::get_precision_recall (bm_utils.cpp:345-371)
inline std::pair<double, double> get_precision_recall(const Association& A, const Association& Agt) {
  double p = 0, r = 0;
  if (clipper_hip_precision_recall(A.data(), A.rows(), Agt.data(), Agt.rows(), &p, &r))
    throw std::invalid_argument(clipper_hip_last_error());
  return {p, r};
}

/// Least-squares rigid transform (column-major 4 x 4 in T[16]) over the associations A
inline void estimate_rigid_transform(const invariants::Data& D1, const invariants::Data& D2,
                                     const Association& A, double (&T)[16]) {
  if (D1.rows() != 3 || D2.rows() != 3) throw std::invalid_argument("points must be 3 x n");
  if (clipper_hip_estimate_rigid_transform(D1.data(), D1.cols(), D2.data(), D2.cols(), A.data(), A.rows(), T))
    throw std::invalid_argument(clipper_hip_last_error());
}

}  // namespace registration
}  // namespace clipper
