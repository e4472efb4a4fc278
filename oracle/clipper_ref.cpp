/*
 * clipper_ref.cpp — CPU ORACLE for the CLIPPER dense-cluster hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load this library; the product path
 * (clipper_amd/, libclipper_hip.so) never links, imports or calls it.
 *
 * What it is: an Eigen-free fp64 restatement of the reference's algorithm, function
 * by function, each citing the reference file:line (relative to /root/reference) it
 * follows. The reference itself cannot be compiled in this image (Eigen3 is absent:
 * CMakeLists.txt:40 `find_package(Eigen3 REQUIRED)`; every TU includes <Eigen/Dense>),
 * so this restatement is pinned against the reference's own golden vectors instead:
 *   - test/affinity_test.cpp:33-107  exact 12x12 affinity matrix "from MATLAB",
 *                                    all-to-all order, unit diagonal, symmetry, M==C
 *   - test/clipper_test.cpp:34-66    3 selected associations with A(i,0)==A(i,1)
 *   - test/clipper_test.cpp:115-133  get*Matrix -> setMatrixData round trip
 *   - examples/matlab/ex3_planecloud.m:18-33,79-98  PointNormalDistance ground truth
 * (tests/test_oracle_golden.py). PointNormalDistance *values* and everything at
 * m >= 1e3 have no reference-side golden data: there parity is GPU-vs-oracle only.
 *
 * Arithmetic conventions (shared with the HIP kernels so decisions agree bit for bit):
 *   - squared norms / dot products are sequential fma chains over the components;
 *   - compiled with -ffp-contract=off: every other expression rounds exactly as written;
 *   - sqrt is correctly rounded; exp/acos come from libm (the reference calls std::exp /
 *     std::acos, euclidean_distance.cpp:30, pointnormal_distance.cpp:21-22,29-30).
 * Eigen's own summation order inside .norm()/.sum()/.dot() and the sparse self-adjoint
 * product is not reproducible without Eigen; the effect is ~1e-14 relative.
 */
#include "clipper_ref.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <queue>
#include <string>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;

/* Strictly-upper CSC, the storage the reference keeps in M_ / C_
 * (clipper.cpp:61-64 `M.sparseView()`, Eigen::SparseMatrix<double> is column-major). */
struct Csc {
  int64_t n = 0;
  std::vector<int64_t> colptr; /* n+1 */
  std::vector<int32_t> row;
  std::vector<double> val;
  void clear(int64_t n_) {
    n = n_;
    colptr.assign(static_cast<size_t>(n_) + 1, 0);
    row.clear();
    val.clear();
  }
};

}  // namespace

struct clipper_ref_ctx {
  int64_t m = 0;
  std::vector<int32_t> A; /* column-major m x 2 */
  Csc M, C;
  std::vector<int32_t> nodes;
  /* the order of the additions inside M x, C x (clipper_ref_set_sum_mode): 0 = the reference's, 1 = the same additions
   * with the columns swept from the last to the first, 2 = every output accumulated in extended precision */
  int sum_mode = 0;
};

namespace {

/* ---- invariants ------------------------------------------------------------------ */

inline double dist_fma(const double* p, const double* q, int d) {
  double acc = 0.0;
  for (int k = 0; k < d; ++k) {
    const double t = p[k] - q[k];
    acc = std::fma(t, t, acc);
  }
  return std::sqrt(acc);
}

/* EuclideanDistance::operator() — euclidean_distance.cpp:13-31 */
inline double score_euclidean(const double* ai, const double* aj, const double* bi,
                              const double* bj, int d, double sigma, double epsilon,
                              double mindist) {
  const double l1 = dist_fma(ai, aj, d); /* :18 */
  const double l2 = dist_fma(bi, bj, d); /* :19 */
  if (mindist > 0 && (l1 < mindist || l2 < mindist)) return 0.0; /* :23-25 */
  const double c = std::fabs(l1 - l2); /* :28 */
  return (c < epsilon) ? std::exp(-0.5 * c * c / (sigma * sigma)) : 0.0; /* :30 */
}

inline double dot3_fma(const double* p, const double* q) {
  return std::fma(p[2], q[2], std::fma(p[1], q[1], p[0] * q[0]));
}

/* PointNormalDistance::operator() — pointnormal_distance.cpp:13-35.
 * Datum = [x y z nx ny nz]. acos of a dot product > 1 is NaN, which makes the
 * comparison at :28 false and the score 0, exactly as in the reference. */
inline double score_pointnormal(const double* ai, const double* aj, const double* bi,
                                const double* bj, double sigp, double epsp, double sign,
                                double epsn) {
  const double l1 = dist_fma(ai, aj, 3); /* :17 */
  const double l2 = dist_fma(bi, bj, 3); /* :18 */
  const double alpha1 = std::acos(dot3_fma(ai + 3, aj + 3)); /* :21 */
  const double alpha2 = std::acos(dot3_fma(bi + 3, bj + 3)); /* :22 */
  const double dp = std::fabs(l1 - l2);         /* :25 */
  const double dn = std::fabs(alpha1 - alpha2); /* :26 */
  if (dp < epsp && dn < epsn) {                 /* :28 */
    const double sp = std::exp(-0.5 * dp * dp / (sigp * sigp)); /* :29 */
    const double sn = std::exp(-0.5 * dn * dn / (sign * sign)); /* :30 */
    return sp * sn;                                             /* :31 */
  }
  return 0.0;
}

/* ---- utils ----------------------------------------------------------------------- */

/* utils::k2ij — utils.cpp:87-97 */
inline void k2ij(uint64_t k, uint64_t n, uint64_t* i, uint64_t* j) {
  k += 1;
  const uint64_t l = n * (n - 1) / 2 - k;
  const uint64_t o =
      static_cast<uint64_t>(std::floor((std::sqrt(static_cast<double>(1 + 8 * l)) - 1) / 2.));
  const uint64_t p = l - o * (o + 1) / 2;
  const uint64_t ii = n - (o + 1);
  const uint64_t jj = n - p;
  *i = ii - 1;
  *j = jj - 1;
}

/* utils::createAllToAll — utils.h:61-71 */
void create_all_to_all(int64_t n1, int64_t n2, int32_t* A) {
  const int64_t m = n1 * n2;
  for (int64_t i = 0; i < n1; ++i) {
    for (int64_t j = 0; j < n2; ++j) {
      A[j + i * n2] = static_cast<int32_t>(i);
      A[m + j + i * n2] = static_cast<int32_t>(j);
    }
  }
}

/* utils::findIndicesOfkLargest — utils.cpp:33-55. Min-heap of (value,index), strict
 * `<` replacement, output descending. The reference pops an empty queue when k > n
 * (undefined behaviour); here k is clamped to n. */
std::vector<int32_t> k_largest(const double* x, int64_t n, int k) {
  using T = std::pair<double, int>;
  if (k < 1) return {};
  if (k > n) k = static_cast<int>(n);
  std::priority_queue<T, std::vector<T>, std::greater<T>> q;
  for (int64_t i = 0; i < n; ++i) {
    if (q.size() < static_cast<size_t>(k)) {
      q.push({x[i], static_cast<int>(i)});
    } else if (q.top().first < x[i]) {
      q.pop();
      q.push({x[i], static_cast<int>(i)});
    }
  }
  std::vector<int32_t> indices(static_cast<size_t>(k));
  for (int i = 0; i < k; ++i) {
    indices[static_cast<size_t>(k - i - 1)] = q.top().second;
    q.pop();
  }
  return indices;
}

/* utils::findIndicesWhereAboveThreshold — utils.cpp:59-68 */
std::vector<int32_t> above_threshold(const double* x, int64_t n, double thr) {
  std::vector<int32_t> idx;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] > thr) idx.push_back(static_cast<int32_t>(i));
  return idx;
}

/* ---- sparse self-adjoint product ---------------------------------------------------
 * y = S.selfadjointView<Upper>() * x for strictly-upper CSC S (clipper.cpp:194,202,
 * 205,219,240-241,268,271). Column sweep as Eigen's sparse self-adjoint kernel does:
 * entry (i,j), i<j, contributes val*x[j] to y[i] and val*x[i] to y[j]. Single thread
 * (Eigen's product is not parallel). */
void symv_upper_reordered(const Csc& S, const double* x, double* y, int mode);

void symv_upper(const Csc& S, const double* x, double* y, int mode = 0) {
  if (mode != 0) return symv_upper_reordered(S, x, y, mode);
  const int64_t n = S.n;
  for (int64_t i = 0; i < n; ++i) y[i] = 0.0;
  for (int64_t j = 0; j < n; ++j) {
    const double xj = x[j];
    double yj = 0.0;
    for (int64_t p = S.colptr[static_cast<size_t>(j)]; p < S.colptr[static_cast<size_t>(j) + 1];
         ++p) {
      const int32_t i = S.row[static_cast<size_t>(p)];
      const double v = S.val[static_cast<size_t>(p)];
      /* selfadjointView<Eigen::Upper>: only entries on or above the diagonal are read. The
       * members the reference builds itself hold nothing else (clipper.cpp:61-64, 151-157);
       * a matrix handed to setSparseMatrixData (clipper.cpp:162-166) may: an entry below the
       * diagonal is ignored, a stored diagonal counts ONCE. */
      if (i > j) continue;
      if (i == j) {
        yj += v * xj;
        continue;
      }
      y[i] += v * xj;
      yj += v * x[i];
    }
    y[j] += yj;
  }
}

/* The SAME product with its additions in another order — not the reference's: what the reference's answer is worth
 * where its decisions sit on the last bits of these sums (tests/test_oracle_golden.py).
 *   mode 1: columns swept from the last to the first, a column's entries from the last to the first;
 *   mode 2: every output accumulated in extended precision (x87 long double, 64-bit mantissa) and rounded once —
 *           the product nearly as if the sums were exact. */
void symv_upper_reordered(const Csc& S, const double* x, double* y, int mode) {
  const int64_t n = S.n;
  if (mode == 2) {
    std::vector<long double> acc(static_cast<size_t>(n), 0.0L);
    for (int64_t j = 0; j < n; ++j) {
      const long double xj = x[j];
      for (int64_t p = S.colptr[static_cast<size_t>(j)]; p < S.colptr[static_cast<size_t>(j) + 1]; ++p) {
        const int32_t i = S.row[static_cast<size_t>(p)];
        const long double v = S.val[static_cast<size_t>(p)];
        if (i > j) continue;
        if (i == j) {
          acc[static_cast<size_t>(j)] += v * xj;
          continue;
        }
        acc[static_cast<size_t>(i)] += v * xj;
        acc[static_cast<size_t>(j)] += v * static_cast<long double>(x[i]);
      }
    }
    for (int64_t i = 0; i < n; ++i) y[i] = static_cast<double>(acc[static_cast<size_t>(i)]);
    return;
  }
  for (int64_t i = 0; i < n; ++i) y[i] = 0.0;
  for (int64_t j = n - 1; j >= 0; --j) {
    const double xj = x[j];
    double yj = 0.0;
    for (int64_t p = S.colptr[static_cast<size_t>(j) + 1] - 1; p >= S.colptr[static_cast<size_t>(j)]; --p) {
      const int32_t i = S.row[static_cast<size_t>(p)];
      const double v = S.val[static_cast<size_t>(p)];
      if (i > j) continue;
      if (i == j) {
        yj += v * xj;
        continue;
      }
      y[i] += v * xj;
      yj += v * x[i];
    }
    y[j] += yj;
  }
}

/* dense column-major m x m (upper triangle significant) -> strictly-upper CSC keeping
 * exact non-zeros: Eigen's sparseView() with default reference/epsilon drops only
 * values that are exactly 0 (clipper.cpp:61,151-157). */
void dense_upper_to_csc(const double* D, int64_t m, Csc* S) {
  S->clear(m);
  for (int64_t j = 0; j < m; ++j) {
    const double* col = D + j * m;
    for (int64_t i = 0; i < j; ++i) {
      if (col[i] != 0.0) {
        S->row.push_back(static_cast<int32_t>(i));
        S->val.push_back(col[i]);
      }
    }
    S->colptr[static_cast<size_t>(j) + 1] = static_cast<int64_t>(S->row.size());
  }
}

/* ---- affinity build: CLIPPER::scorePairwiseConsistency — clipper.cpp:21-65 ---------- */

template <class ScoreFn>
int score_pairwise(clipper_ref_ctx* h, const double* D1, int d, int64_t n1, const double* D2,
                   int64_t n2, const int32_t* A, int64_t m_in, double affinityeps,
                   int parallelize, int dense_temp, ScoreFn score) {
  /* :24-25 — empty A means all-to-all */
  int64_t m = m_in;
  if (A == nullptr || m_in == 0) {
    m = n1 * n2;
    h->A.assign(static_cast<size_t>(2 * m), 0);
    create_all_to_all(n1, n2, h->A.data());
  } else {
    h->A.assign(A, A + 2 * m);
  }
  h->m = m;
  h->nodes.clear();
  const int32_t* A0 = h->A.data();
  const int32_t* A1 = h->A.data() + m;
  for (int64_t r = 0; r < m; ++r) {
    if (A0[r] < 0 || A0[r] >= n1 || A1[r] < 0 || A1[r] >= n2) {
      g_err = "association index out of range";
      return -1;
    }
  }

  if (dense_temp < 0) dense_temp = (m <= 24000) ? 1 : 0; /* 8*m^2 <= 4.6 GB */

  if (dense_temp) {
    /* The reference's route: zero dense m x m (:29), flat loop over the m(m-1)/2
     * unordered pairs with k2ij unranking (:31-56), then sparseView (:61). */
    std::vector<double> M(static_cast<size_t>(m) * static_cast<size_t>(m), 0.0);
    const int64_t npairs = m * (m - 1) / 2;
#pragma omp parallel for schedule(static) if (parallelize)
    for (int64_t k = 0; k < npairs; ++k) {
      uint64_t i, j;
      k2ij(static_cast<uint64_t>(k), static_cast<uint64_t>(m), &i, &j); /* :33 */
      if (A0[i] == A0[j] || A1[i] == A1[j]) continue;                    /* :35-38 */
      const double* d1i = D1 + static_cast<int64_t>(A0[i]) * d;          /* :45-46 */
      const double* d1j = D1 + static_cast<int64_t>(A0[j]) * d;
      const double* d2i = D2 + static_cast<int64_t>(A1[i]) * d; /* :49-50 */
      const double* d2j = D2 + static_cast<int64_t>(A1[j]) * d;
      const double scr = score(d1i, d1j, d2i, d2j); /* :52 */
      if (scr > affinityeps) M[static_cast<size_t>(j) * m + i] = scr; /* :53-55, M(i,j) */
    }
    dense_upper_to_csc(M.data(), m, &h->M); /* :61 */
  } else {
    /* Same pairs, same scores, CSC assembled column by column without the dense
     * temporary (which is 8*m^2 bytes: 80 GB at m = 100k). */
    std::vector<std::vector<std::pair<int32_t, double>>> cols(static_cast<size_t>(m));
#pragma omp parallel for schedule(dynamic, 16) if (parallelize)
    for (int64_t j = 0; j < m; ++j) {
      auto& col = cols[static_cast<size_t>(j)];
      const double* d1j = D1 + static_cast<int64_t>(A0[j]) * d;
      const double* d2j = D2 + static_cast<int64_t>(A1[j]) * d;
      for (int64_t i = 0; i < j; ++i) {
        if (A0[i] == A0[j] || A1[i] == A1[j]) continue;
        const double* d1i = D1 + static_cast<int64_t>(A0[i]) * d;
        const double* d2i = D2 + static_cast<int64_t>(A1[i]) * d;
        const double scr = score(d1i, d1j, d2i, d2j);
        if (scr > affinityeps) col.emplace_back(static_cast<int32_t>(i), scr);
      }
    }
    h->M.clear(m);
    for (int64_t j = 0; j < m; ++j) {
      for (const auto& e : cols[static_cast<size_t>(j)]) {
        h->M.row.push_back(e.first);
        h->M.val.push_back(e.second);
      }
      h->M.colptr[static_cast<size_t>(j) + 1] = static_cast<int64_t>(h->M.row.size());
    }
  }

  /* :63-64 — C_ = M_; C_.coeffs() = 1 */
  h->C = h->M;
  std::fill(h->C.val.begin(), h->C.val.end(), 1.0);
  return 0;
}

/* ---- small dense helpers for the solver (fp64, sequential order) ------------------- */

inline double vsum(const std::vector<double>& x) {
  double s = 0.0;
  for (double v : x) s += v;
  return s;
}
inline double vdot(const std::vector<double>& x, const std::vector<double>& y) {
  double s = 0.0;
  for (size_t i = 0; i < x.size(); ++i) s += x[i] * y[i];
  return s;
}
inline double vnorm(const std::vector<double>& x) { return std::sqrt(vdot(x, x)); }

}  // namespace

/* ==================================================================================== */

extern "C" {

clipper_ref_t* clipper_ref_create(void) { return new clipper_ref_ctx(); }
void clipper_ref_destroy(clipper_ref_t* h) { delete h; }
const char* clipper_ref_last_error(void) { return g_err.c_str(); }

int clipper_ref_affinity_euclidean(clipper_ref_t* h, const double* D1, int d, int64_t n1,
                                   const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                   double sigma, double epsilon, double mindist,
                                   double affinityeps, int parallelize, int dense_temp) {
  if (!h || !D1 || !D2 || d < 1) {
    g_err = "invalid argument";
    return -1;
  }
  return score_pairwise(h, D1, d, n1, D2, n2, A, m, affinityeps, parallelize, dense_temp,
                        [=](const double* ai, const double* aj, const double* bi,
                            const double* bj) {
                          return score_euclidean(ai, aj, bi, bj, d, sigma, epsilon, mindist);
                        });
}

int clipper_ref_affinity_pointnormal(clipper_ref_t* h, const double* D1, int d, int64_t n1,
                                     const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                     double sigp, double epsp, double sign, double epsn,
                                     double affinityeps, int parallelize, int dense_temp) {
  if (!h || !D1 || !D2 || d != 6) {
    g_err = "PointNormalDistance needs d == 6";
    return -1;
  }
  return score_pairwise(h, D1, d, n1, D2, n2, A, m, affinityeps, parallelize, dense_temp,
                        [=](const double* ai, const double* aj, const double* bi,
                            const double* bj) {
                          return score_pointnormal(ai, aj, bi, bj, sigp, epsp, sign, epsn);
                        });
}

int64_t clipper_ref_num_associations(const clipper_ref_t* h) { return h ? h->m : 0; }

int clipper_ref_get_associations(const clipper_ref_t* h, int32_t* A_out) {
  if (!h || !A_out) return -1;
  std::memcpy(A_out, h->A.data(), h->A.size() * sizeof(int32_t));
  return 0;
}

/* CLIPPER::setMatrixData — clipper.cpp:149-158 */
int clipper_ref_set_matrix(clipper_ref_t* h, const double* M, const double* C, int64_t m) {
  if (!h || !M || !C || m < 0) {
    g_err = "invalid argument";
    return -1;
  }
  h->m = m;
  h->nodes.clear();
  dense_upper_to_csc(M, m, &h->M);
  dense_upper_to_csc(C, m, &h->C);
  return 0;
}

/* CLIPPER::setSparseMatrixData — clipper.cpp:162-166 */
int clipper_ref_set_sparse(clipper_ref_t* h, int64_t m, const int64_t* Mcolptr,
                           const int32_t* Mrow, const double* Mval, const int64_t* Ccolptr,
                           const int32_t* Crow, const double* Cval) {
  if (!h || !Mcolptr || !Ccolptr) {
    g_err = "invalid argument";
    return -1;
  }
  h->m = m;
  h->nodes.clear();
  auto fill = [m](Csc* S, const int64_t* cp, const int32_t* r, const double* v) {
    S->n = m;
    S->colptr.assign(cp, cp + m + 1);
    const int64_t nnz = cp[m];
    S->row.assign(r, r + nnz);
    S->val.assign(v, v + nnz);
  };
  fill(&h->M, Mcolptr, Mrow, Mval);
  fill(&h->C, Ccolptr, Crow, Cval);
  return 0;
}

/* CLIPPER::getAffinityMatrix / getConstraintMatrix — clipper.cpp:131-145 */
int clipper_ref_get_matrix(const clipper_ref_t* h, double* M_out, double* C_out) {
  if (!h) return -1;
  const int64_t m = h->m;
  auto densify = [m](const Csc& S, double* D) {
    std::fill(D, D + m * m, 0.0);
    for (int64_t j = 0; j < m; ++j) {
      for (int64_t p = S.colptr[static_cast<size_t>(j)]; p < S.colptr[static_cast<size_t>(j) + 1];
           ++p) {
        const int64_t i = S.row[static_cast<size_t>(p)];
        const double v = S.val[static_cast<size_t>(p)];
        if (i > j) continue; /* selfadjointView<Upper> (clipper.cpp:133,142): the lower triangle is not read */
        if (i == j) {
          D[j * m + j] += v; /* a stored diagonal (sparse setter) appears once */
        } else {
          D[j * m + i] += v;
          D[i * m + j] += v;
        }
      }
    }
    for (int64_t i = 0; i < m; ++i) D[i * m + i] += 1.0;
  };
  if (M_out) densify(h->M, M_out);
  if (C_out) densify(h->C, C_out);
  return 0;
}

/* (test infrastructure of the test infrastructure) the order of the additions inside the products: see symv_upper_reordered */
int clipper_ref_set_sum_mode(clipper_ref_t* h, int mode) {
  if (!h || mode < 0 || mode > 2) {
    g_err = "sum mode must be 0 (the reference's order), 1 (reversed) or 2 (extended precision)";
    return -1;
  }
  h->sum_mode = mode;
  return 0;
}

int64_t clipper_ref_nnz(const clipper_ref_t* h) {
  return h ? static_cast<int64_t>(h->M.row.size()) : 0;
}

int clipper_ref_matvec(const clipper_ref_t* h, const double* x, double* yM, double* yC) {
  if (!h || !x) return -1;
  if (yM) symv_upper(h->M, x, yM, h->sum_mode);
  if (yC) symv_upper(h->C, x, yC, h->sum_mode);
  return 0;
}

/* CLIPPER::solve + findDenseClique — clipper.cpp:69-78, 172-323 */
int clipper_ref_solve(clipper_ref_t* h, const double* u0_in, const clipper_params_t* P,
                      double* u_out, clipper_solve_info_t* info) {
  if (!h || !u0_in || !P) {
    g_err = "invalid argument (u0 and params are required)";
    return -1;
  }
  const auto t1 = std::chrono::high_resolution_clock::now(); /* :174 */

  const int64_t n = h->M.n; /* :180 */
  const size_t N = static_cast<size_t>(n);
  int64_t n_passes = 0, n_trials = 0;

  std::vector<double> gradF(N), gradFnew(N), u(N), unew(N), Mu(N), Cu(N), Cbu(N); /* :184-190 */
  const std::vector<double> u0(u0_in, u0_in + n);

  /* one pass = the pair (M_off*x, C_off*x); the reference evaluates the two products
   * wherever it needs them, here they are counted as it evaluates them. */
  auto matvec_M = [&](const std::vector<double>& x, std::vector<double>& y) {
    symv_upper(h->M, x.data(), y.data(), h->sum_mode);
  };
  auto matvec_C = [&](const std::vector<double>& x, std::vector<double>& y) {
    symv_upper(h->C, x.data(), y.data(), h->sum_mode);
  };

  /* :193-198 — one power-method step, then normalise */
  if (P->rescale_u0) {
    matvec_M(u0, Mu);
    ++n_passes;
    for (size_t i = 0; i < N; ++i) u[i] = Mu[i] + u0[i];
  } else {
    u = u0;
  }
  {
    const double nrm = vnorm(u);
    for (size_t i = 0; i < N; ++i) u[i] /= nrm;
  }

  /* :200-209 — initial penalty d */
  double d = 0;
  {
    const double s = vsum(u);
    matvec_C(u, Cu);
    ++n_passes;
    for (size_t i = 0; i < N; ++i) Cbu[i] = 1.0 * s - Cu[i] - u[i]; /* :202 */
    int64_t cnt = 0;
    for (size_t i = 0; i < N; ++i) cnt += (Cbu[i] > P->eps && u[i] > P->eps); /* :203 */
    if (cnt > 0) { /* :204 */
      matvec_M(u, Mu);
      double acc = 0.0;
      for (size_t i = 0; i < N; ++i)
        if (Cbu[i] > P->eps && u[i] > P->eps) acc += (Mu[i] + u[i]) / Cbu[i]; /* :205-208 */
      d = acc / static_cast<double>(cnt);
    }
  }

  double F = 0; /* :215 */
  int i = 0, j = 0, k = 0;
  for (i = 0; i < P->maxoliters; ++i) { /* :218 */
    /* :219-220 */
    {
      const double s = vsum(u);
      matvec_M(u, Mu);
      matvec_C(u, Cu);
      ++n_passes;
      for (size_t q = 0; q < N; ++q)
        gradF[q] = (1 + d) * u[q] - (d * 1.0) * s + Mu[q] + Cu[q] * d;
      F = vdot(u, gradF);
    }

    for (j = 0; j < P->maxiniters; ++j) { /* :226 */
      double alpha = 1;                   /* :227 */
      double Fnew = 0, deltaF = 0;        /* :233 */
      for (k = 0; k < P->maxlsiters; ++k) { /* :234 */
        for (size_t q = 0; q < N; ++q) {
          const double t = u[q] + alpha * gradF[q]; /* :235 */
          unew[q] = (t > 0) ? t : 0.0;              /* :236 cwiseMax(0) */
        }
        {
          /* :237 — Eigen's normalize(): z = squaredNorm(); if (z > 0) x /= sqrt(z) */
          const double z = vdot(unew, unew);
          if (z > 0) {
            const double nrm = std::sqrt(z);
            for (size_t q = 0; q < N; ++q) unew[q] /= nrm;
          }
        }
        {
          const double s = vsum(unew); /* :238-241 */
          matvec_M(unew, Mu);
          matvec_C(unew, Cu);
          ++n_passes;
          ++n_trials;
          for (size_t q = 0; q < N; ++q)
            gradFnew[q] = (1 + d) * unew[q] - (d * 1.0) * s + Mu[q] + Cu[q] * d;
        }
        Fnew = vdot(unew, gradFnew); /* :242 */
        deltaF = Fnew - F;           /* :244 */
        if (deltaF < -P->eps) {      /* :246 */
          alpha = alpha * P->beta;   /* :248 */
        } else {
          break; /* :250 */
        }
      }
      double deltau = 0.0; /* :253 */
      for (size_t q = 0; q < N; ++q) {
        const double t = unew[q] - u[q];
        deltau += t * t;
      }
      deltau = std::sqrt(deltau);

      F = Fnew; /* :256-258 */
      u.swap(unew);
      gradF.swap(gradFnew);

      if (deltau < P->tol_u || std::fabs(deltaF) < P->tol_F) break; /* :261 */
    }

    /* :268-280 — increase d */
    {
      const double s = vsum(u);
      matvec_C(u, Cu);
      ++n_passes;
      for (size_t q = 0; q < N; ++q) Cbu[q] = 1.0 * s - Cu[q] - u[q]; /* :268 */
      int64_t cnt = 0;
      for (size_t q = 0; q < N; ++q) cnt += (Cbu[q] > P->eps && u[q] > P->eps); /* :269 */
      if (cnt > 0) {                                                             /* :270 */
        matvec_M(u, Mu);
        double acc = 0.0;
        for (size_t q = 0; q < N; ++q)
          if (Cbu[q] > P->eps && u[q] > P->eps) acc += std::fabs((Mu[q] + u[q]) / Cbu[q]);
        d += acc / static_cast<double>(cnt); /* :274-276 */
      } else {
        break; /* :278-280 */
      }
    }
  }

  /* :287-310 — rounding */
  std::vector<int32_t> nodes;
  if (P->rounding == CLIPPER_ROUNDING_NONZERO) {
    nodes = above_threshold(u.data(), n, 0.0);
  } else if (P->rounding == CLIPPER_ROUNDING_DSD_HEU) {
    const int omega = static_cast<int>(std::round(F)); /* :305 */
    nodes = k_largest(u.data(), n, omega);             /* :308 */
  } else {
    g_err = "Rounding::DSD (exact densest sub-graph, dsd.cpp) is outside the hot-path scope";
    return -2;
  }

  const auto t2 = std::chrono::high_resolution_clock::now();
  const double elapsed =
      static_cast<double>(std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count()) /
      1e9;

  h->nodes = nodes; /* :312-322 */
  if (u_out) std::memcpy(u_out, u.data(), N * sizeof(double));
  if (info) {
    info->score = F;
    info->seconds = elapsed;
    info->d = d;
    info->ifinal = i;
    info->num_nodes = static_cast<int32_t>(nodes.size());
    info->n_passes = n_passes;
    info->n_trials = n_trials;
  }
  return 0;
}

int clipper_ref_get_nodes(const clipper_ref_t* h, int32_t* out, int32_t capacity) {
  if (!h || !out) return -1;
  const int32_t k = static_cast<int32_t>(h->nodes.size());
  if (capacity < k) return -1;
  std::memcpy(out, h->nodes.data(), static_cast<size_t>(k) * sizeof(int32_t));
  return k;
}

/* utils::selectInlierAssociations — utils.cpp:101-108 */
int clipper_ref_get_selected_associations(const clipper_ref_t* h, int32_t* A_out,
                                          int32_t capacity) {
  if (!h || !A_out) return -1;
  const int32_t k = static_cast<int32_t>(h->nodes.size());
  if (capacity < k) return -1;
  if (h->A.empty()) return (k == 0) ? 0 : -1;
  for (int32_t r = 0; r < k; ++r) {
    A_out[r] = h->A[static_cast<size_t>(h->nodes[static_cast<size_t>(r)])];
    A_out[k + r] = h->A[static_cast<size_t>(h->m + h->nodes[static_cast<size_t>(r)])];
  }
  return k;
}

void clipper_ref_k2ij(uint64_t k, uint64_t n, uint64_t* i, uint64_t* j) { k2ij(k, n, i, j); }

void clipper_ref_create_all_to_all(int64_t n1, int64_t n2, int32_t* A_out) {
  create_all_to_all(n1, n2, A_out);
}

int clipper_ref_k_largest(const double* x, int64_t n, int32_t k, int32_t* idx_out) {
  const std::vector<int32_t> r = k_largest(x, n, k);
  std::memcpy(idx_out, r.data(), r.size() * sizeof(int32_t));
  return static_cast<int>(r.size());
}

double clipper_ref_score_euclidean(const double* ai, const double* aj, const double* bi,
                                   const double* bj, int d, double sigma, double epsilon,
                                   double mindist) {
  return score_euclidean(ai, aj, bi, bj, d, sigma, epsilon, mindist);
}

double clipper_ref_score_pointnormal(const double* ai, const double* aj, const double* bi,
                                     const double* bj, double sigp, double epsp, double sign,
                                     double epsn) {
  return score_pointnormal(ai, aj, bi, bj, sigp, epsp, sign, epsn);
}

int clipper_ref_omp_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

} /* extern "C" */
