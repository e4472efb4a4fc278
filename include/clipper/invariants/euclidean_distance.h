/**
 * @file euclidean_distance.h
 * @brief Pairwise Euclidean distance invariant (mirror of the reference
 *        include/clipper/invariants/euclidean_distance.h:19-49, .cpp:13-31).
 */
#pragma once

#include "clipper/invariants/abstract.h"

namespace clipper {
namespace invariants {

class EuclideanDistance : public PairwiseInvariant {
 public:
  struct Params {
    double sigma = 0.01;    ///< spread of the exponential kernel
    double epsilon = 0.06;  ///< bound on the consistency residual (inlier / outlier)
    double mindist = 0;     ///< minimum allowed distance between points of one dataset
  };

  EuclideanDistance(const Params& params) : params_(params) {}
  ~EuclideanDistance() = default;

  /// Host evaluation of one pair (euclidean_distance.cpp:13-31). CLIPPER evaluates this
  /// invariant on the GPU; the functor exists for API parity and for direct calls.
  double operator()(const Datum& ai, const Datum& aj, const Datum& bi, const Datum& bj) override;

  /// The reference keeps params_ private with no getter; the GPU dispatcher needs them.
  const Params& params() const { return params_; }

 private:
  Params params_;
};
using EuclideanDistancePtr = std::shared_ptr<EuclideanDistance>;

}  // namespace invariants
}  // namespace clipper
