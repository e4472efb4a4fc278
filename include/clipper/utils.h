/**
 * @file utils.h
 * @brief Utilities of the clipper:: facade (mirror of the reference include/clipper/utils.h
 *        and src/utils.cpp; same names, argument meaning and results).
 */
#pragma once

#include <chrono>
#include <ostream>
#include <string>
#include <tuple>
#include <vector>

#include "clipper/invariants/abstract.h"
#include "clipper/types.h"

namespace clipper {
struct Solution;
namespace utils {

/// n x 1 vector with entries drawn from U[0,1) (utils.cpp:22-29; std::random_device seeded).
VectorXd randvec(size_t n);

/// Indices of the k largest elements (utils.cpp:33-55): min-heap of (value,index), strict '<'
/// replacement, output sorted descending by (value,index); k < 1 returns {}.
std::vector<int> findIndicesOfkLargest(const VectorXd& x, int k);

/// Indices i with x[i] > thr, ascending (utils.cpp:59-68).
std::vector<int> findIndicesWhereAboveThreshold(const VectorXd& x, double thr);

/// All-to-all association hypothesis, row j + i*n2 = (i, j) (utils.h:61-71).
inline Association createAllToAll(size_t n1, size_t n2) {
  Association A(static_cast<std::ptrdiff_t>(n1 * n2), 2);
  for (size_t i = 0; i < n1; ++i) {
    for (size_t j = 0; j < n2; ++j) {
      A(static_cast<std::ptrdiff_t>(j + i * n2), 0) = static_cast<int>(i);
      A(static_cast<std::ptrdiff_t>(j + i * n2), 1) = static_cast<int>(j);
    }
  }
  return A;
}

/// Elements of x selected by a 0/1 indicator (utils.cpp:72-83).
VectorXd selectFromIndicator(const VectorXd& x, const VectorXi& ind);

/// Rows of A at soln.nodes, in that order (utils.cpp:101-108).
Association selectInlierAssociations(const Solution& soln, const Association& A);

/// Flat index k -> (i, j), i < j, row-major order of the strict upper triangle (utils.cpp:87-97).
std::tuple<size_t, size_t> k2ij(size_t k, size_t n);

/// Simple named profiling timer (utils.h:107-163).
class Timer {
 public:
  Timer() = default;
  Timer(const std::string& name) : name_(name) {}
  void start() {
    t1_ = std::chrono::high_resolution_clock::now();
    running_ = true;
  }
  void stop() {
    t2_ = std::chrono::high_resolution_clock::now();
    if (running_) {
      total_ += std::chrono::duration<double>(t2_ - t1_).count();
      running_ = false;
      count_++;
    }
  }
  void reset() { total_ = 0; }
  double getElapsedSeconds() const { return total_; }

  friend std::ostream& operator<<(std::ostream& os, const Timer& t) {
    if (!t.name_.empty()) os << t.name_ << ": ";
    os << t.total_ << " s (" << t.count_ << "x)";
    return os;
  }
  friend Timer operator+(const Timer& lhs, const Timer& rhs) {
    Timer t;
    t.total_ = lhs.total_ + rhs.total_;
    return t;
  }

 private:
  double total_ = 0;
  std::string name_;
  int count_ = 0;
  bool running_ = false;
  std::chrono::time_point<std::chrono::high_resolution_clock> t1_, t2_;
};

}  // namespace utils
}  // namespace clipper
