// clipper.cpp — host side of the clipper:: facade (include/clipper/*.h) over the C ABI of
// include/clipper_hip.h. Mirrors the reference's src/clipper.cpp, src/utils.cpp and
// src/invariants/*.cpp member for member; every O(m^2) operation of the dense-cluster path
// (affinity build for the built-in invariants, the mat-vecs and vector algebra of solve())
// is a call into libclipper_hip.so. Nothing here is a CPU implementation of that path:
// if the GPU library reports an error, the facade throws.
#include "clipper/clipper.h"

#include <cmath>
#include <cstring>
#include <exception>
#include <functional>
#include <iostream>
#include <queue>
#include <random>
#include <stdexcept>
#include <utility>

#include "clipper/utils.h"
#include "clipper_hip.h"

namespace clipper {

// ------------------------------------------------------------------------------------------
// invariants — host functors (direct calls / API parity; CLIPPER itself evaluates the
// built-ins on the device)
// ------------------------------------------------------------------------------------------
namespace invariants {

namespace {
inline double dist(const Datum& p, const Datum& q, std::ptrdiff_t first, std::ptrdiff_t n) {
  double acc = 0.0;
  for (std::ptrdiff_t k = first; k < first + n; ++k) {
    const double t = p(k) - q(k);
    acc = std::fma(t, t, acc);
  }
  return std::sqrt(acc);
}
}  // namespace

// reference src/invariants/euclidean_distance.cpp:13-31
double EuclideanDistance::operator()(const Datum& ai, const Datum& aj, const Datum& bi,
                                     const Datum& bj) {
  const double l1 = dist(ai, aj, 0, ai.size());
  const double l2 = dist(bi, bj, 0, bi.size());
  if (params_.mindist > 0 && (l1 < params_.mindist || l2 < params_.mindist)) return 0.0;
  const double c = std::abs(l1 - l2);
  return (c < params_.epsilon) ? std::exp(-0.5 * c * c / (params_.sigma * params_.sigma)) : 0;
}

// reference src/invariants/pointnormal_distance.cpp:13-35
double PointNormalDistance::operator()(const Datum& ai, const Datum& aj, const Datum& bi,
                                       const Datum& bj) {
  const double l1 = dist(ai, aj, 0, 3);
  const double l2 = dist(bi, bj, 0, 3);
  const double alpha1 = std::acos(std::fma(ai(5), aj(5), std::fma(ai(4), aj(4), ai(3) * aj(3))));
  const double alpha2 = std::acos(std::fma(bi(5), bj(5), std::fma(bi(4), bj(4), bi(3) * bj(3))));
  const double dp = std::abs(l1 - l2);
  const double dn = std::abs(alpha1 - alpha2);
  if (dp < params_.epsp && dn < params_.epsn) {
    const double sp = std::exp(-0.5 * dp * dp / (params_.sigp * params_.sigp));
    const double sn = std::exp(-0.5 * dn * dn / (params_.sign * params_.sign));
    return sp * sn;
  }
  return 0.0;
}

}  // namespace invariants

// ------------------------------------------------------------------------------------------
// utils — reference src/utils.cpp
// ------------------------------------------------------------------------------------------
namespace utils {

VectorXd randvec(size_t n) {  // utils.cpp:22-29
  std::random_device rd;
  std::mt19937 gen(rd());
  std::uniform_real_distribution<double> dis(0, 1);
  VectorXd v(static_cast<std::ptrdiff_t>(n));
  for (size_t i = 0; i < n; ++i) v(static_cast<std::ptrdiff_t>(i)) = dis(gen);
  return v;
}

std::vector<int> findIndicesOfkLargest(const VectorXd& x, int k) {  // utils.cpp:33-55
  using T = std::pair<double, int>;
  if (k < 1) return {};
  if (k > x.size()) k = static_cast<int>(x.size());  // the reference pops an empty queue here
  std::priority_queue<T, std::vector<T>, std::greater<T>> q;
  for (std::ptrdiff_t i = 0; i < x.size(); ++i) {
    if (q.size() < static_cast<size_t>(k)) {
      q.push({x(i), static_cast<int>(i)});
    } else if (q.top().first < x(i)) {
      q.pop();
      q.push({x(i), static_cast<int>(i)});
    }
  }
  std::vector<int> indices(static_cast<size_t>(k));
  for (int i = 0; i < k; ++i) {
    indices[static_cast<size_t>(k - i - 1)] = q.top().second;
    q.pop();
  }
  return indices;
}

std::vector<int> findIndicesWhereAboveThreshold(const VectorXd& x, double thr) {  // :59-68
  std::vector<int> indices;
  indices.reserve(static_cast<size_t>(x.size()));
  for (std::ptrdiff_t i = 0; i < x.size(); ++i)
    if (x(i) > thr) indices.push_back(static_cast<int>(i));
  return indices;
}

VectorXd selectFromIndicator(const VectorXd& x, const VectorXi& ind) {  // utils.cpp:72-83
  std::ptrdiff_t cnt = 0;
  for (std::ptrdiff_t i = 0; i < ind.size(); ++i) cnt += ind(i);
  VectorXd y(cnt);
  std::ptrdiff_t idx = 0;
  for (std::ptrdiff_t i = 0; i < x.size(); ++i)
    if (ind(i)) y(idx++) = x(i);
  return y;
}

std::tuple<size_t, size_t> k2ij(size_t k, size_t n) {  // utils.cpp:87-97
  k += 1;
  const size_t l = n * (n - 1) / 2 - k;
  const size_t o = static_cast<size_t>(std::floor((std::sqrt(1 + 8 * l) - 1) / 2.));
  const size_t p = l - o * (o + 1) / 2;
  const size_t i = n - (o + 1);
  const size_t j = n - p;
  return {i - 1, j - 1};
}

Association selectInlierAssociations(const Solution& soln, const Association& A) {  // :101-108
  Association Ainliers(static_cast<std::ptrdiff_t>(soln.nodes.size()), 2);
  for (size_t i = 0; i < soln.nodes.size(); ++i) {
    Ainliers(static_cast<std::ptrdiff_t>(i), 0) = A(soln.nodes[i], 0);
    Ainliers(static_cast<std::ptrdiff_t>(i), 1) = A(soln.nodes[i], 1);
  }
  return Ainliers;
}

}  // namespace utils

// ------------------------------------------------------------------------------------------
// CLIPPER
// ------------------------------------------------------------------------------------------

CLIPPER::CLIPPER(const invariants::PairwiseInvariantPtr& invariant, const Params& params)
    : params_(params), invariant_(invariant) {}

CLIPPER::~CLIPPER() {
  if (h_) clipper_hip_destroy(h_);
}

void CLIPPER::setDevice(int device) {
  if (h_) throw std::logic_error("CLIPPER::setDevice must be called before the first GPU call");
  device_ = device;
  devices_.clear();
}

void CLIPPER::setDevices(const std::vector<int>& devices) {
  if (h_) throw std::logic_error("CLIPPER::setDevices must be called before the first GPU call");
  if (devices.empty()) throw std::invalid_argument("CLIPPER::setDevices: an empty device list");
  devices_ = devices;
  device_ = devices.front();
}

void CLIPPER::setStorage(Storage storage) {
  if (h_) throw std::logic_error("CLIPPER::setStorage must be called before the first GPU call");
  storage_ = storage;
}

void CLIPPER::setResidentSolver(bool on) {
  resident_ = on;
  if (h_) check(clipper_hip_set_resident(h_, on ? 0 : 1), "set_resident");
}

bool CLIPPER::lastSolveWasResident() const { return h_ != nullptr && clipper_hip_last_solver(h_) == 1; }

void CLIPPER::setRowViews(bool on) {
  row_views_ = on;
  if (h_) check(clipper_hip_set_row_view(h_, on ? 0 : 1), "set_row_view");
}

void CLIPPER::setLiveSubproblem(bool on) {
  subproblem_ = on;
  if (h_) check(clipper_hip_set_subproblem(h_, on ? 0 : 1), "set_subproblem");
}

long long CLIPPER::lastSolvePassesOnTheSubproblem() const {
  clipper_hip_view_stats_t st{};
  if (h_ == nullptr || clipper_hip_get_view_stats(h_, &st) < 0) return 0;
  return st.sub_passes;
}

long long CLIPPER::lastSolvePassesOnAView() const {
  clipper_hip_view_stats_t st{};
  if (h_ == nullptr || clipper_hip_get_view_stats(h_, &st) < 0) return 0;
  return st.view_passes;
}

clipper_hip_ctx* CLIPPER::handle() {
  if (!h_) {
    h_ = devices_.size() > 1
             ? clipper_hip_create_group(devices_.data(), static_cast<int>(devices_.size()), static_cast<int>(storage_))
             : clipper_hip_create(device_, static_cast<int>(storage_));
    if (!h_)
      throw std::runtime_error(std::string("clipper: cannot create the GPU context: ") +
                               clipper_hip_last_error());
    if (!resident_) check(clipper_hip_set_resident(h_, 1), "set_resident");
    if (!row_views_) check(clipper_hip_set_row_view(h_, 1), "set_row_view");
    if (!subproblem_) check(clipper_hip_set_subproblem(h_, 1), "set_subproblem");
  }
  return h_;
}

void CLIPPER::check(int rc, const char* what) const {
  if (rc < 0)
    throw std::runtime_error(std::string("clipper: ") + what + " failed (" + std::to_string(rc) +
                             "): " + clipper_hip_last_error());
}

// clipper.cpp:21-65
void CLIPPER::scorePairwiseConsistency(const invariants::Data& D1, const invariants::Data& D2,
                                       const Association& A) {
  if (D1.rows() != D2.rows())
    throw std::invalid_argument("clipper: D1 and D2 must have the same number of rows");
  if (A.size() == 0) A_ = utils::createAllToAll(D1.cols(), D2.cols());  // :24
  else A_ = A;                                                           // :25
  const int64_t m = A_.rows();

  auto euclid = std::dynamic_pointer_cast<invariants::EuclideanDistance>(invariant_);
  auto pointn = std::dynamic_pointer_cast<invariants::PointNormalDistance>(invariant_);
  // a subclass that overrides operator() must not be short-circuited to the built-in kernel
  const bool exact_euclid = euclid && typeid(*invariant_) == typeid(invariants::EuclideanDistance);
  const bool exact_pointn =
      pointn && typeid(*invariant_) == typeid(invariants::PointNormalDistance);

  if (exact_euclid) {
    const auto& p = euclid->params();
    check(clipper_hip_affinity_euclidean(handle(), D1.data(), static_cast<int>(D1.rows()),
                                         D1.cols(), D2.data(), D2.cols(), A_.data(), m, p.sigma,
                                         p.epsilon, p.mindist, params_.affinityeps),
          "scorePairwiseConsistency[EuclideanDistance]");
  } else if (exact_pointn) {
    const auto& p = pointn->params();
    check(clipper_hip_affinity_pointnormal(handle(), D1.data(), static_cast<int>(D1.rows()),
                                           D1.cols(), D2.data(), D2.cols(), A_.data(), m, p.sigp,
                                           p.epsp, p.sign, p.epsn, params_.affinityeps),
          "scorePairwiseConsistency[PointNormalDistance]");
  } else {
    scoreCustomInvariantOnHost(D1, D2);
  }
  clipper_hip_timings_t tm;
  if (clipper_hip_get_timings(h_, &tm) == 0) stats_.affinity_kernel_ms = tm.affinity_kernel_ms;
}

// User-defined invariants (C++ subclasses, Python trampolines) are opaque virtual functions
// and can only run where they were written: on the host, one call per pair, exactly the
// reference's loop (clipper.cpp:31-56, OpenMP when parallelize_). The resulting matrix is
// then handed to the GPU through setMatrixData's entry point with C = pattern(M)
// (clipper.cpp:61-64); solve() runs on the device as for the built-ins.
void CLIPPER::scoreCustomInvariantOnHost(const invariants::Data& D1, const invariants::Data& D2) {
  const std::ptrdiff_t m = A_.rows();
  const std::ptrdiff_t d = D1.rows();
  Affinity M = Affinity::Zero(m, m);
  auto column = [d](const invariants::Data& D, std::ptrdiff_t c) {
    invariants::Datum v(d);
    for (std::ptrdiff_t k = 0; k < d; ++k) v(k) = D(k, c);
    return v;
  };
  const long long npairs = static_cast<long long>(m) * (m - 1) / 2;
  std::exception_ptr failure;  // an exception must not leave an OpenMP region
#pragma omp parallel for schedule(static) if (parallelize_)
  for (long long k = 0; k < npairs; ++k) {
    try {
      size_t i, j;
      std::tie(i, j) = utils::k2ij(static_cast<size_t>(k), static_cast<size_t>(m));
      if (A_(i, 0) == A_(j, 0) || A_(i, 1) == A_(j, 1)) continue;  // :35-38
      const invariants::Datum d1i = column(D1, A_(i, 0)), d1j = column(D1, A_(j, 0));
      const invariants::Datum d2i = column(D2, A_(i, 1)), d2j = column(D2, A_(j, 1));
      const double scr = (*invariant_)(d1i, d1j, d2i, d2j);  // :52
      if (scr > params_.affinityeps) M(i, j) = scr;           // :53-55
    } catch (...) {
#pragma omp critical(clipper_custom_invariant_failure)
      if (!failure) failure = std::current_exception();
    }
  }
  if (failure) std::rethrow_exception(failure);
  Constraint C = Constraint::Zero(m, m);
  for (std::ptrdiff_t j = 0; j < m; ++j)
    for (std::ptrdiff_t i = 0; i < j; ++i)
      if (M(i, j) != 0.0) C(i, j) = 1.0;  // :63-64
  check(clipper_hip_set_matrix(handle(), M.data(), C.data(), m), "upload of the custom-invariant M");
}

// clipper.cpp:69-78
void CLIPPER::solve(const VectorXd& _u0) {
  const int64_t n = clipper_hip_num_associations(handle());
  VectorXd u0;
  if (_u0.size() == 0) u0 = utils::randvec(static_cast<size_t>(n));
  else u0 = _u0;
  if (u0.size() != n) throw std::invalid_argument("clipper: u0 has the wrong length");

  clipper_params_t p;
  clipper_params_default(&p);
  p.tol_u = params_.tol_u;
  p.tol_F = params_.tol_F;
  p.tol_Fop = params_.tol_Fop;
  p.maxiniters = params_.maxiniters;
  p.maxoliters = params_.maxoliters;
  p.beta = params_.beta;
  p.maxlsiters = params_.maxlsiters;
  p.eps = params_.eps;
  p.affinityeps = params_.affinityeps;
  p.rescale_u0 = params_.rescale_u0 ? 1 : 0;
  p.rounding = static_cast<int>(params_.rounding);
  VectorXd u(n);
  clipper_solve_info_t info;
  check(clipper_hip_solve(h_, u0.data(), &p, u.data(), &info), "solve");
  std::vector<int> nodes(static_cast<size_t>(info.num_nodes));
  if (info.num_nodes > 0)
    check(clipper_hip_get_nodes(h_, nodes.data(), info.num_nodes), "solve (nodes)");

  soln_.t = info.seconds;  // clipper.cpp:312-322
  soln_.ifinal = info.ifinal;
  std::swap(soln_.nodes, nodes);
  soln_.u0 = u0;
  soln_.u = u;
  soln_.score = info.score;
  stats_.n_passes = info.n_passes;
  stats_.n_trials = info.n_trials;
  stats_.d = info.d;
}

// Without PMC the reference prints a warning and returns an empty clique
// (maxclique.cpp:141-145); this build never has PMC.
void CLIPPER::solveAsMaximumClique(const maxclique::Params&) {
  std::cout << "PMC is not built. Maximum clique solver is unavailable." << std::endl;
  soln_.t = 0;
  soln_.ifinal = 0;
  soln_.nodes.clear();
  soln_.u = VectorXd::Zero(clipper_hip_num_associations(handle()));
  soln_.score = -1;
}

// Without SCS the reference prints a warning and returns an empty solution (sdp.cpp:298-302).
void CLIPPER::solveAsMSRCSDR(const sdp::Params&) {
  std::cout << "SCS is not built. SDP solver is unavailable." << std::endl;
  soln_.t = 0;
  soln_.ifinal = 0;
  soln_.nodes.clear();
  soln_.u = VectorXd::Zero(clipper_hip_num_associations(handle()));
  soln_.score = -1;
}

Association CLIPPER::getInitialAssociations() { return A_; }  // clipper.cpp:117-120

Association CLIPPER::getSelectedAssociations() {  // clipper.cpp:124-127
  return utils::selectInlierAssociations(soln_, A_);
}

Affinity CLIPPER::getAffinityMatrix() {  // clipper.cpp:131-136
  const int64_t m = clipper_hip_num_associations(handle());
  Affinity M(m, m);
  check(clipper_hip_get_matrix(h_, M.data(), nullptr), "getAffinityMatrix");
  return M;
}

Constraint CLIPPER::getConstraintMatrix() {  // clipper.cpp:140-145
  const int64_t m = clipper_hip_num_associations(handle());
  Constraint C(m, m);
  check(clipper_hip_get_matrix(h_, nullptr, C.data()), "getConstraintMatrix");
  return C;
}

void CLIPPER::setMatrixData(const Affinity& M, const Constraint& C) {  // clipper.cpp:149-158
  if (M.rows() != M.cols() || C.rows() != M.rows() || C.cols() != M.cols())
    throw std::invalid_argument("clipper: M and C must be square and of equal size");
  check(clipper_hip_set_matrix(handle(), M.data(), C.data(), M.rows()), "setMatrixData");
}

void CLIPPER::setSparseMatrixData(const SpAffinity& M, const SpConstraint& C) {  // :162-166
#ifdef CLIPPER_HAVE_EIGEN
  auto csc = [](const SpMat& S_, std::vector<int64_t>& cp, std::vector<int32_t>& ri,
                std::vector<double>& va) {
    SpMat S = S_;
    S.makeCompressed();
    cp.assign(S.outerIndexPtr(), S.outerIndexPtr() + S.cols() + 1);
    ri.assign(S.innerIndexPtr(), S.innerIndexPtr() + S.nonZeros());
    va.assign(S.valuePtr(), S.valuePtr() + S.nonZeros());
  };
  std::vector<int64_t> mcp, ccp;
  std::vector<int32_t> mri, cri;
  std::vector<double> mva, cva;
  csc(M, mcp, mri, mva);
  csc(C, ccp, cri, cva);
  check(clipper_hip_set_sparse(handle(), M.rows(), mcp.data(), mri.data(), mva.data(), ccp.data(),
                               cri.data(), cva.data()),
        "setSparseMatrixData");
#else
  if (M.rows() != M.cols() || C.rows() != M.rows())
    throw std::invalid_argument("clipper: sparse M and C must be square and of equal size");
  check(clipper_hip_set_sparse(handle(), M.rows(), M.colptr.data(), M.rowidx.data(),
                               M.values.data(), C.colptr.data(), C.rowidx.data(), C.values.data()),
        "setSparseMatrixData");
#endif
}

}  // namespace clipper
