/**
 * @file clipper.h
 * @brief clipper::CLIPPER — the reference's facade class (include/clipper/clipper.h:78-182)
 *        over the MI355X hot path. Same public names, argument order and defaults; the dense
 *        cluster path (scorePairwiseConsistency + solve) runs on the GPU through the C ABI of
 *        include/clipper_hip.h.
 *
 * Differences a maintainer should know (all additive):
 *   - the built-in invariants expose params() (the GPU dispatcher needs them);
 *   - setDevice()/setStorage() choose the GPU and the storage of M (slices with fp32 or fp64
 *     values — the default is fp32 —, or dense fp32 / fp64);
 *   - failures of the GPU path throw std::runtime_error (the reference has no error path;
 *     there is deliberately NO silent CPU fallback for the built-in invariants);
 *   - Rounding::DSD (clipper.cpp:294-300) is supported: the exact densest subgraph of the
 *     sub-matrix induced by nnz(u), gathered from the device (include/clipper/dsd.h);
 *   - solveAsMaximumClique / solveAsMSRCSDR are outside this build and report so, exactly
 *     like a reference build without PMC / SCS (maxclique.cpp:141-145, sdp.cpp:298-302).
 */
#pragma once

#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "clipper/invariants/abstract.h"
#include "clipper/invariants/builtins.h"
#include "clipper/types.h"

struct clipper_hip_ctx;  // opaque handle of the C ABI

namespace clipper {

namespace maxclique {
/// Mirror of the reference maxclique::Method / Params (maxclique.h:15-23); solver not built here.
enum class Method { EXACT, HEU, KCORE };
struct Params {
  Method method = Method::EXACT;
  size_t threads = 24;
  int time_limit = 3600;
  bool verbose = false;
};
}  // namespace maxclique

namespace sdp {
/// Mirror of the reference sdp::Params (sdp.h:40-52); solver not built here.
struct Params {
  bool verbose = false;
  int max_iters = 2000;
  int acceleration_interval = 10;
  int acceleration_lookback = 10;
  float eps_abs = 1e-3f;
  float eps_rel = 1e-3f;
  float eps_infeas = 1e-7f;
  float time_limit_secs = 0;
};
}  // namespace sdp

/// CLIPPER parameters (reference clipper.h:27-60)
struct Params {
  double tol_u = 1e-8;     ///< stop when change in u < tol
  double tol_F = 1e-9;     ///< stop when change in F < tol
  double tol_Fop = 1e-10;  ///< declared by the reference, never read
  int maxiniters = 200;    ///< max num of gradient ascent steps for each d
  int maxoliters = 1000;   ///< max num of outer loop iterations to find d
  double beta = 0.25;      ///< backtracking step size reduction, in (0, 1)
  int maxlsiters = 99;     ///< maximum number of line search iters per grad step
  double eps = 1e-9;       ///< numerical threshold around 0
  double affinityeps = 1e-4;  ///< sparsity-promoting threshold for affinities
  bool rescale_u0 = true;  ///< rescale u0 using one power iteration
  enum Rounding { NONZERO, DSD, DSD_HEU };
  Rounding rounding = Rounding::DSD_HEU;
};

/// Data associated with a CLIPPER dense clique solution (reference clipper.h:65-73)
struct Solution {
  double t = 0;            ///< duration spent solving [s]
  int ifinal = 0;          ///< number of outer iterations before convergence
  std::vector<int> nodes;  ///< indices of graph vertices in dense clique
  VectorXd u0;             ///< initial vector used for local solver
  VectorXd u;              ///< characteristic vector associated with graph
  double score = 0;        ///< value of objective function / largest eigenvalue
};

class CLIPPER {
 public:
  /// Element type of the dense M kept in HBM (vectors and accumulators are always fp64).
  enum class Storage { F32 = 0, F64 = 1, F32_CSC = 2, F64_CSC = 3 };  ///< see clipper_hip.h (CLIPPER_HIP_STORE_*)

  CLIPPER(const invariants::PairwiseInvariantPtr& invariant, const Params& params);
  ~CLIPPER();
  /// (NOT copyable, unlike the reference's class — clipper.h:82 there is copyable by default: an object
  /// owns a device context with M in HBM; share it through a pointer, or build a second one)
  CLIPPER(const CLIPPER&) = delete;
  CLIPPER& operator=(const CLIPPER&) = delete;

  /// Affinity matrix of the m associations in A (all-to-all when A is empty). clipper.cpp:21-65
  void scorePairwiseConsistency(const invariants::Data& D1, const invariants::Data& D2,
                                const Association& A = Association());

  /// Graduated projected gradient ascent (clipper.cpp:69-78, 172-323). Random u0 if empty.
  void solve(const VectorXd& u0 = VectorXd());

  void solveAsMaximumClique(const maxclique::Params& params = {});  ///< not built (PMC)
  void solveAsMSRCSDR(const sdp::Params& params = {});              ///< not built (SCS)

  const Solution& getSolution() const { return soln_; }
  Affinity getAffinityMatrix();      ///< dense symmetric + identity (clipper.cpp:131-136)
  Constraint getConstraintMatrix();  ///< dense symmetric + identity (clipper.cpp:140-145)

  void setMatrixData(const Affinity& M, const Constraint& C);              ///< clipper.cpp:149-158
  /// clipper.cpp:162-166. Read the way the reference's solver reads what it keeps — through the UPPER triangle
  /// (selfadjointView<Upper>, clipper.cpp:194-271): entries below the diagonal are ignored (reported as a warning
  /// through the C ABI's clipper_hip_last_error()); a lower-triangular matrix is an empty one. One divergence, by design: a stored NON-ZERO
  /// diagonal entry — which the reference would count once on top of the identity it adds, and which its own
  /// contract excludes (clipper.h:137-138) — is refused here (the diagonal is implicit in every storage of this
  /// build); explicit zeros on the diagonal are dropped.
  void setSparseMatrixData(const SpAffinity& M, const SpConstraint& C);

  Association getInitialAssociations();   ///< clipper.cpp:117-120
  Association getSelectedAssociations();  ///< clipper.cpp:124-127

  void setParallelize(bool parallelize) { parallelize_ = parallelize; }

  // ---- additions of this build -----------------------------------------------------------
  void setDevice(int device);        ///< HIP device ordinal (default 0); before the first call
  /// Multi-GPU through the class, no launcher: M is column-sharded over the listed devices of this process
  /// (one shard per entry; an ordinal may repeat — several logical shards on one GPU), each pass ends with the exchange
  /// of the shards' partial products. Over the C ABI's clipper_hip_create_group; one entry = setDevice. Before the first call.
  void setDevices(const std::vector<int>& devices);
  void setStorage(Storage storage);  ///< default F32_CSC (compressed; dense fp32 where it does not apply); before the first call
  /// Problems of up to 2048 associations are solved by ONE launch that keeps M on chip (the resident
  /// solver); false = always the streaming launches. Same result either way. Switching
  /// it off takes effect at the next solve(), switching it back on at the next build of M.
  void setResidentSolver(bool on);
  bool lastSolveWasResident() const;  ///< which of the two the last solve() ran on
  /// Row views: once most of u is zero a pass streams the slices of M[live rows, :]
  /// instead of M (same sums up to the order of the partial sums). false = every pass streams M. Any time.
  void setRowViews(bool on);
  long long lastSolvePassesOnAView() const;  ///< how many passes of the last solve() streamed a view
  /// The live sub-problem: once a row view exists and the penalty is large, the solve continues on the
  /// associations that can still be selected — provably the same result. false = the passes keep streaming the view. Any time.
  void setLiveSubproblem(bool on);
  long long lastSolvePassesOnTheSubproblem() const;
  struct PathStats {
    long long n_passes = 0, n_trials = 0;
    double affinity_kernel_ms = 0, d = 0;
  };
  PathStats getPathStats() const { return stats_; }

 private:
  Params params_;
  invariants::PairwiseInvariantPtr invariant_;
  bool parallelize_ = true;  ///< OpenMP for user-defined (host-evaluated) invariants only
  Association A_;
  Solution soln_;
  PathStats stats_;
  int device_ = 0;
  std::vector<int> devices_;  ///< more than one entry: column shards (setDevices)
  bool subproblem_ = true;
  Storage storage_ = Storage::F32_CSC;
  bool resident_ = true;
  bool row_views_ = true;
  clipper_hip_ctx* h_ = nullptr;

  clipper_hip_ctx* handle();
  void check(int rc, const char* what) const;
  void scoreCustomInvariantOnHost(const invariants::Data& D1, const invariants::Data& D2);
};

}  // namespace clipper
