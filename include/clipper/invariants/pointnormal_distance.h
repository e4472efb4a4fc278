/**
 * @file pointnormal_distance.h
 * @brief Point-normal (plane / surfel) invariant (mirror of the reference
 *        include/clipper/invariants/pointnormal_distance.h:22-53, .cpp:13-35).
 *        Datum: 6x1 — top 3 rows the point, bottom 3 rows the unit normal.
 */
#pragma once

#include "clipper/invariants/abstract.h"

namespace clipper {
namespace invariants {

class PointNormalDistance : public PairwiseInvariant {
 public:
  struct Params {
    double sigp = 0.5;   ///< point  - spread of exp kernel
    double epsp = 0.5;   ///< point  - bound on consistency score
    double sign = 0.10;  ///< normal - spread of exp kernel
    double epsn = 0.35;  ///< normal - bound on consistency score
  };

  PointNormalDistance(const Params& params) : params_(params) {}
  ~PointNormalDistance() = default;

  double operator()(const Datum& ai, const Datum& aj, const Datum& bi, const Datum& bj) override;

  const Params& params() const { return params_; }

 private:
  Params params_;
};
using PointNormalDistancePtr = std::shared_ptr<PointNormalDistance>;

}  // namespace invariants
}  // namespace clipper
