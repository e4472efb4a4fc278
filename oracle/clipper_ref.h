/*
 * clipper_ref.h — C ABI of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is an Eigen-free fp64 restatement of the reference's dense-cluster
 * hot path (/root/reference/src/clipper.cpp, src/utils.cpp, src/invariants/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Entry points mirror include/clipper_hip.h one-to-one so tests diff the two paths
 * on identical argument lists.
 */
#ifndef CLIPPER_REF_H
#define CLIPPER_REF_H

#include "../include/clipper_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clipper_ref_ctx clipper_ref_t;

clipper_ref_t* clipper_ref_create(void);
void clipper_ref_destroy(clipper_ref_t* h);
const char* clipper_ref_last_error(void);

/* CLIPPER::scorePairwiseConsistency with EuclideanDistance (clipper.cpp:21-65,
 * euclidean_distance.cpp:13-31). D1 is d x n1, D2 is d x n2 (column-major fp64);
 * A is column-major m x 2 int32, or NULL / m==0 for the all-to-all hypothesis.
 * `parallelize` mirrors CLIPPER::setParallelize (OpenMP on/off).
 * `dense_temp`: 1 = build through the reference's dense m x m temporary and
 * sparseView scan (clipper.cpp:29,61); 0 = build the CSC directly (same result,
 * for m where 8*m^2 bytes does not fit); -1 = choose by size. */
int clipper_ref_affinity_euclidean(clipper_ref_t* h, const double* D1, int d, int64_t n1,
                                   const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                   double sigma, double epsilon, double mindist,
                                   double affinityeps, int parallelize, int dense_temp);

/* Same with PointNormalDistance (pointnormal_distance.cpp:13-35); d must be 6. */
int clipper_ref_affinity_pointnormal(clipper_ref_t* h, const double* D1, int d, int64_t n1,
                                     const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                     double sigp, double epsp, double sign, double epsn,
                                     double affinityeps, int parallelize, int dense_temp);

/* Number of associations currently held (rows of A_, = dimension of M_). */
int64_t clipper_ref_num_associations(const clipper_ref_t* h);
/* CLIPPER::getInitialAssociations (clipper.cpp:117-120): column-major m x 2. */
int clipper_ref_get_associations(const clipper_ref_t* h, int32_t* A_out);

/* CLIPPER::setMatrixData (clipper.cpp:149-158): dense column-major m x m inputs,
 * strict upper triangle kept, exact zeros dropped. */
int clipper_ref_set_matrix(clipper_ref_t* h, const double* M, const double* C, int64_t m);
/* CLIPPER::setSparseMatrixData (clipper.cpp:162-166): CSC stored as given. */
int clipper_ref_set_sparse(clipper_ref_t* h, int64_t m,
                           const int64_t* Mcolptr, const int32_t* Mrow, const double* Mval,
                           const int64_t* Ccolptr, const int32_t* Crow, const double* Cval);
/* CLIPPER::getAffinityMatrix / getConstraintMatrix (clipper.cpp:131-145): dense
 * symmetric with the identity added. Either output may be NULL. */
int clipper_ref_get_matrix(const clipper_ref_t* h, double* M_out, double* C_out);
/* Stored non-zeros of the strict upper triangle of M_. */
int64_t clipper_ref_nnz(const clipper_ref_t* h);

/* CLIPPER::solve -> findDenseClique (clipper.cpp:69-78,172-323). u0 must be given
 * (the reference's random default is non-deterministic; parity needs one u0 on
 * both paths). u_out may be NULL. Nodes are fetched with clipper_ref_get_nodes. */
int clipper_ref_solve(clipper_ref_t* h, const double* u0, const clipper_params_t* params,
                      double* u_out, clipper_solve_info_t* info);
int clipper_ref_get_nodes(const clipper_ref_t* h, int32_t* nodes_out, int32_t capacity);
/* CLIPPER::getSelectedAssociations (clipper.cpp:124-127): column-major k x 2. */
int clipper_ref_get_selected_associations(const clipper_ref_t* h, int32_t* A_out, int32_t capacity);

/* The order of the additions inside every product M_off*x, C_off*x of the solver and of clipper_ref_matvec:
 * 0 = the reference's (Eigen's column sweep; the default and the only PARITY mode), 1 = the same additions with columns
 * and entries swept backwards, 2 = every output accumulated in extended precision. Modes 1 and 2 exist to measure how much
 * of the reference's answer is decided by rounding (tests/test_oracle_golden.py::test_summation_order_*). */
int clipper_ref_set_sum_mode(clipper_ref_t* h, int mode);

/* One symmetric product pair y_M = M_off*x, y_C = C_off*x (clipper.cpp:194,202,...);
 * exposed so tests can check the GPU mat-vec in isolation. */
int clipper_ref_matvec(const clipper_ref_t* h, const double* x, double* yM, double* yC);

/* utils (utils.cpp:33-55, 87-97; utils.h:61-71) */
void clipper_ref_k2ij(uint64_t k, uint64_t n, uint64_t* i, uint64_t* j);
void clipper_ref_create_all_to_all(int64_t n1, int64_t n2, int32_t* A_out);
int clipper_ref_k_largest(const double* x, int64_t n, int32_t k, int32_t* idx_out);

/* Single-pair invariant scores (for value-level tests). */
double clipper_ref_score_euclidean(const double* ai, const double* aj, const double* bi,
                                   const double* bj, int d, double sigma, double epsilon,
                                   double mindist);
double clipper_ref_score_pointnormal(const double* ai, const double* aj, const double* bi,
                                     const double* bj, double sigp, double epsp, double sign,
                                     double epsn);

int clipper_ref_omp_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* CLIPPER_REF_H */
