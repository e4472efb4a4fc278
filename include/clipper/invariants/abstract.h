/**
 * @file abstract.h
 * @brief Invariant plug-in API (mirror of the reference include/clipper/invariants/abstract.h).
 *
 * A pairwise invariant scores the consistency of two associations (ai,bi), (aj,bj).
 * Built-in invariants are evaluated on the GPU by the HIP library; user subclasses of
 * PairwiseInvariant are evaluated on the host exactly as the reference does (one virtual call
 * per pair, clipper.cpp:52) and the resulting matrix is uploaded.
 */
#pragma once

#include <memory>

#include "clipper/types.h"

namespace clipper {
namespace invariants {

using Data = MatrixXd;   ///< d x n, one datum per column (abstract.h:19)
using Datum = VectorXd;  ///< d x 1                       (abstract.h:20)

/// Base of all invariant types (abstract.h:37-40).
class Invariant {
 public:
  virtual ~Invariant() = default;
};
using InvariantPtr = std::shared_ptr<Invariant>;

/// f : A x A x A x A -> R, the pairwise consistency scoring function (abstract.h:56-72).
class PairwiseInvariant : public Invariant {
 public:
  virtual ~PairwiseInvariant() = default;
  /// @return consistency score of the associations (ai,bi) and (aj,bj)
  virtual double operator()(const Datum& ai, const Datum& aj, const Datum& bi, const Datum& bj) = 0;
};
using PairwiseInvariantPtr = std::shared_ptr<PairwiseInvariant>;

}  // namespace invariants
}  // namespace clipper
