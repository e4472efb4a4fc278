// The exact densest-subgraph rounding of the host (clipper_amd/csrc/dsd_host.h): the certified replay
// (two maximum flows when the peeling finds the optimum) against the reference's procedure as it stands
// (one flow per bisection step, dsd.cpp:200-241) on random graph families — the node sets must be equal.
// g++ only; driven by tests/test_dsd_host.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../clipper_amd/csrc/dsd_host.h"

using clipper_hip::dsd::densest_subgraph;
using clipper_hip::dsd::densest_subgraph_by_bisection;

static int failures = 0, cases = 0, fallbacks = 0;
static long total_flows = 0;

static void check(std::vector<double> W, int k, int64_t n, const char* what, int id) {
  std::vector<double> W2 = W;
  int flows = 0;
  const auto a = densest_subgraph(W, k, n, &flows);
  const auto b = densest_subgraph_by_bisection(W2, k, n);
  ++cases;
  if (flows <= 0) { ++fallbacks; if (fallbacks <= 5) std::printf("fallback: %s #%d k=%d n=%ld\n", what, id, k, static_cast<long>(n)); } else total_flows += flows;
  if (a != b) {
    ++failures;
    std::printf("MISMATCH %s #%d k=%d n=%ld: replay %zu nodes (flows %d), bisection %zu nodes\n", what, id, k,
                static_cast<long>(n), a.size(), flows, b.size());
  }
}

static void sym(std::vector<double>& W, int k, int a, int b, double w) {
  W[static_cast<size_t>(a) * k + b] = w;
  W[static_cast<size_t>(b) * k + a] = w;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 60;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const int64_t totals[4] = {0, 10000, 300000, 1000000};  // 0: n = k (dsd::solve on the whole graph); 10^6: the
                                                            // last steps fall below the capacities' resolution
  for (int r = 0; r < reps; ++r) {
    const int k = 2 + static_cast<int>(rng() % 70);
    const int64_t n = totals[r % 4] ? totals[r % 4] : k;
    // 1. random weights, random sparsity
    {
      std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
      const double p = U(rng);
      for (int a = 0; a < k; ++a)
        for (int b = a + 1; b < k; ++b)
          if (U(rng) < p) sym(W, k, a, b, U(rng));
      check(W, k, n, "random weights", r);
    }
    // 2. unweighted: exact ties, densities that ARE bisection midpoints
    {
      std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
      const double p = U(rng);
      for (int a = 0; a < k; ++a)
        for (int b = a + 1; b < k; ++b)
          if (U(rng) < p) sym(W, k, a, b, 1.0);
      check(W, k, n, "unweighted", r);
    }
    // 3. a planted cluster (what the solver's support looks like) in a sparse fringe
    {
      std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
      const int c = 2 + static_cast<int>(rng() % static_cast<unsigned>(k - 1));
      for (int a = 0; a < k; ++a)
        for (int b = a + 1; b < k; ++b) {
          if (a < c && b < c) sym(W, k, a, b, 0.5 + 0.5 * U(rng));
          else if (U(rng) < 0.1) sym(W, k, a, b, U(rng));
        }
      check(W, k, n, "planted cluster", r);
    }
    // 4. several disjoint cliques of equal density (the union is the answer) + one heavier edge
    {
      std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
      const int q = 2 + static_cast<int>(rng() % 5);
      for (int a = 0; a < k; ++a)
        for (int b = a + 1; b < k; ++b)
          if (a / q == b / q && b / q < k / q) sym(W, k, a, b, 0.25);
      if (r & 1) sym(W, k, 0, k - 1, 0.25 * q);
      check(W, k, n, "equal cliques", r);
    }
    // 5. a peeling trap: a large light clique beside a small heavy one joined by a path
    {
      std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
      const int c = k / 2;
      for (int a = 0; a < k; ++a)
        for (int b = a + 1; b < k; ++b) {
          if (b < c) sym(W, k, a, b, 0.2 + 0.01 * U(rng));
          else if (a >= c && b < c + 4 && b < k) sym(W, k, a, b, 1.0);
          else if (b == a + 1) sym(W, k, a, b, 0.6 * U(rng));
        }
      check(W, k, n, "two clusters", r);
    }
  }
  // 6. weights far above 1 (a caller's own matrix): sink capacities m/2 + 2g - degree go negative
  for (int r = 0; r < reps; ++r) {
    const int k = 2 + static_cast<int>(rng() % 24);
    std::vector<double> W(static_cast<size_t>(k) * k, 0.0);
    const double wmax = 1.0 + 40.0 * U(rng);
    for (int a = 0; a < k; ++a)
      for (int b = a + 1; b < k; ++b)
        if (U(rng) < 0.6) sym(W, k, a, b, wmax * U(rng));
    check(W, k, totals[r % 4] ? totals[r % 4] : k, "heavy weights", r);
  }
  // degenerate inputs
  for (int k = 2; k <= 5; ++k) {
    std::vector<double> Z(static_cast<size_t>(k) * k, 0.0);
    check(Z, k, k, "all zero", k);
    sym(Z, k, 0, 1, 0.3);
    check(Z, k, 10000, "one edge", k);
  }
  std::printf("%d cases, %d mismatches, %d fell back to the plain procedure, %.2f flows per case otherwise\n", cases,
              failures, fallbacks, cases > fallbacks ? static_cast<double>(total_flows) / (cases - fallbacks) : 0.0);
  return failures ? 1 : 0;
}
