/*
 * clipper_hip.h — C ABI of the MI355X (gfx950) implementation of CLIPPER's dense-cluster
 * hot path. This is the drop-in boundary: plain pointers, sizes and PODs only, no C++
 * or framework types. The reference has no FFI of its own for this path (it is one C++
 * shared library, CMakeLists.txt:93); each entry point below names the reference member
 * function (file:line relative to /root/reference) whose work it takes over, and
 * clipper_amd/csrc/host/ holds the `clipper::CLIPPER` facade + `clipperpy` module that a
 * maintainer binds to it.
 *
 * Conventions
 *   - Matrices are column-major fp64 (Eigen default). A is column-major m x 2 int32.
 *   - Every function returns 0 on success, <0 on failure (CLIPPER_HIP_E_*); the message
 *     is available from clipper_hip_last_error() (thread-local). Nothing throws.
 *   - Calls are synchronous from the caller's view; one context = one problem instance,
 *     not thread-safe per context (same as clipper::CLIPPER), independent across contexts.
 *   - There is NO CPU fallback: without a usable HIP device every call fails loudly.
 *
 * Storage of the affinity matrix on device
 *   CLIPPER_HIP_STORE_F32_CSC (the default of every front end) and CLIPPER_HIP_STORE_F64_CSC
 *   hold ONLY the stored (nonzero) entries of M_off — both triangles, as the reference's
 *   Eigen::SparseMatrix does (include/clipper/types.h:15) — as per-column lists of (value,
 *   row byte) cut into slices of 64 columns x 128 rows, padded to groups of 4 entries and to
 *   nothing else (5.7 bytes per stored entry with fp32 values, 9.6 with fp64).
 *   The solver's passes stream the slices; a dense store is materialised only while a getter,
 *   the exact-DSD gather or an explicit C needs one and is dropped again. Column shards hold
 *   the slices of their own columns. The values, the fp64 products and every decision are
 *   those of the dense storage of the same value type; only the zeros are skipped. These
 *   storages apply whenever C == pattern(M) (always on the scorePairwiseConsistency path,
 *   clipper.cpp:63-64); setMatrixData with any other C falls back to the dense store of the
 *   same value type. gemv_bytes reports the bytes the slices hold = what one pass streams.
 *   CLIPPER_HIP_STORE_F32 / _F64 keep M_off dense in HBM as column slices: a context that
 *   owns global columns [c0, c0+W) stores S[j][c] = M(j, c0+c) for all m rows j, row pitch W
 *   (multiple of 64 elements, zero padded): 4*m^2 (8*m^2) bytes. On the
 *   scorePairwiseConsistency path C is not stored: the mat-vec kernel derives C_off*x from
 *   the same pass over M. setMatrixData with any other C stores a second dense matrix.
 *   All vectors, accumulators, scalars and branch operands of the solver are fp64 in every
 *   mode; fp32 is a STORAGE type of M's entries only.
 */
#ifndef CLIPPER_HIP_H
#define CLIPPER_HIP_H

#include <stddef.h>

#include "clipper_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clipper_hip_ctx clipper_hip_t;

enum {
  CLIPPER_HIP_STORE_F32 = 0,
  CLIPPER_HIP_STORE_F64 = 1,
  CLIPPER_HIP_STORE_F32_CSC = 2,
  CLIPPER_HIP_STORE_F64_CSC = 3
};

enum {
  CLIPPER_HIP_OK = 0,
  CLIPPER_HIP_E_INVALID = -1,  /* bad argument                       */
  CLIPPER_HIP_E_NOMEM = -2,    /* device or host allocation failed   */
  CLIPPER_HIP_E_HIP = -3,      /* HIP runtime error                  */
  CLIPPER_HIP_E_NODEVICE = -4, /* no gfx950 device visible           */
  CLIPPER_HIP_E_STATE = -5,    /* call out of order (no matrix yet)  */
  CLIPPER_HIP_E_COMM = -6,     /* RCCL error / communicator missing  */
  CLIPPER_HIP_E_SCOPE = -7,    /* not available in this configuration */
  CLIPPER_HIP_E_INTERNAL = -8  /* an unexpected C++ exception was caught at the boundary */
};

/* Timings of the most recent calls, from HIP events on the context's own stream. */
typedef struct clipper_hip_timings_t {
  double affinity_kernel_ms; /* affinity fill kernel(s) only                           */
  double affinity_total_ms;  /* H2D of D1,D2,A + gather + fill, host wall clock         */
  double solve_total_ms;     /* host wall clock of clipper_hip_solve                    */
  double gemv_avg_us;        /* mean duration of the mat-vec kernel (k_gemv: one pass over
                                M for a whole line-search window) over the last solve
                                (only when profiling is on; else 0)                     */
  double gemv_min_us;
  int64_t gemv_launches;     /* number of mat-vec launches that were timed              */
  double gemv_bytes;         /* algorithmic bytes of one launch: s*m*(owned columns) for a
                                dense store; for the compressed storage the bytes the slices
                                hold (headers, lengths, value and row quads) + their directory */
  double gemv_useful_bytes;  /* compressed storage: stored entries (both triangles, no quad
                                padding) x (value + row byte); dense store: = gemv_bytes     */
  double affinity_bytes;     /* bytes of M the last affinity build wrote (dense store or slices) */
  double exchange_avg_us;    /* column shards: mean duration of the sampled per-pass exchanges (reduce
                                launch + all-gather) of the last solve, HIP events on the solver
                                stream (profiling on; 0: none sampled / one shard)            */
  int64_t exchange_samples;
  double exchange_bytes;     /* bytes this rank contributes to one exchange                   */
} clipper_hip_timings_t;

/* The row view of the last solve (clipper_hip_set_row_view below). */
typedef struct clipper_hip_view_stats_t {
  int64_t builds;        /* row views built during the last solve                           */
  int64_t rows;          /* rows of the last one built (0: none)                            */
  int64_t bytes;         /* bytes its slices hold                                           */
  int64_t view_passes;   /* passes of the last solve that streamed a view instead of M      */
  int64_t passes;        /* all passes of the last solve                                    */
  double build_ms;       /* host wall clock spent building views (drain + compaction + fill + plan) */
  double view_pass_avg_us; /* mean duration of the sampled pass launches that streamed a view
                              (profiling on; 0 = none sampled)                               */
  int64_t view_pass_samples;
  int64_t resident_launches; /* launches of the resident solver on a view (k_rv_resident.hip.h) that RAN: the passes
                                on a view that ran inside one are counted in view_passes, not sampled   */
  int64_t resident_giveups;  /* launches of it that gave up and changed nothing (a unit that did not become resident
                                within the exchange's time-out — e.g. another tenant on the device —, a refused LDS
                                plan): the streaming launches did their work; the context then streams the views of
                                its next solves (1, 2, 4, ... 64 of them) before it tries again              */
  int64_t resident_iterations; /* solver iterations (one exchange each) that ran inside the launches that ran  */
  double resident_us;          /* their duration on the device's own wall clock: unit 0's first instruction to
                                  the last unit's commit (100 MHz ticks; free, always on)                      */
  double resident_event_us;    /* the same by HIP events around the launches (clipper_hip_set_profiling(h, 2):
                                  an event pair costs stream time, so not in a timed region); 0 = not timed   */
  int64_t resident_entries;    /* stored entries (quads x 4, padding included) of the view the last launch ran on */
  int64_t resident_units;      /* workgroups (one per CU: each holds its columns of the view in LDS) of that launch */
  /* the live sub-problem (csrc/k_subproblem.hip.h, clipper_hip_set_subproblem below) */
  int64_t sub_entries;         /* hand-overs of the last solve to the sub-problem of the associations that can still be selected */
  int64_t sub_leaves;          /* ... and hand-overs back (a column outside it could have come back to life)   */
  int64_t sub_passes;          /* passes of the last solve that ran on the sub-problem (not counted in view_passes) */
  int64_t sub_rows;            /* associations of the sub-problem (0: none was prepared)                       */
  int64_t sub_bytes;           /* bytes its slices hold                                                        */
  double sub_build_ms;         /* host wall clock spent preparing it (selection, gather, fill, plan)           */
  double sub_pass_avg_us;      /* mean duration of the sampled window passes on it (profiling on; 0 = none)    */
  int64_t sub_pass_samples;
  int64_t sub_dense;           /* 1: the sub-problem was mostly non-zero and was kept as a dense fp32 store (its passes: k_gemv) */
} clipper_hip_view_stats_t;

/* ---- life cycle --------------------------------------------------------------------- */

int clipper_hip_device_count(void);

/* One device holds the whole matrix. Replaces the construction of CLIPPER's members
 * M_, C_ (clipper.h:155-156). `storage` is CLIPPER_HIP_STORE_F32 / _F64. */
clipper_hip_t* clipper_hip_create(int device, int storage);

/* In-process column sharding: `nshards` slices driven by one host thread; devices[p] may
 * repeat (several logical shards on one GPU — used to test the sharded protocol on a
 * 1-GPU box). Slices exchange their (M_off*x, C_off*x) pieces by device-to-device copies. */
clipper_hip_t* clipper_hip_create_group(const int* devices, int nshards, int storage);

/* Multi-process column sharding, one process per GPU: this context is shard `rank` of
 * `world`. The per-pass exchange is an RCCL all-gather over xGMI; call
 * clipper_hip_comm_init before the first solve. */
clipper_hip_t* clipper_hip_create_rank(int device, int storage, int rank, int world);
/* ncclGetUniqueId into a 128-byte buffer (rank 0), to be broadcast by the launcher. */
int clipper_hip_comm_unique_id(void* id128);
/* ncclCommInitRank on this context's device with the broadcast id. */
int clipper_hip_comm_init(clipper_hip_t* h, const void* id128);
/* The same exchange through the CALLER instead of RCCL: once per pass the library hands `fn` this
 * rank's block (`bytes` bytes of host memory) and a buffer of world * bytes for the blocks of all
 * ranks, rank order (e.g. a torch.distributed / MPI all-gather). `fn` returns 0, or non-zero to
 * abort the solve with CLIPPER_HIP_E_COMM. Called from the thread that runs clipper_hip_solve.
 * One host round trip per pass: a portability and test back-end (it lets several processes share
 * ONE device), not a fast path. */
typedef int (*clipper_hip_allgather_fn)(void* user, const void* sendbuf, void* recvbuf, size_t bytes);
int clipper_hip_comm_init_callback(clipper_hip_t* h, clipper_hip_allgather_fn fn, void* user);

void clipper_hip_destroy(clipper_hip_t* h);
const char* clipper_hip_last_error(void);

/* ---- affinity build ----------------------------------------------------------------- */

/* CLIPPER::scorePairwiseConsistency (clipper.cpp:21-65) with the built-in
 * EuclideanDistance invariant (euclidean_distance.cpp:13-31, Params .h:22-27).
 * D1: d x n1, D2: d x n2 column-major fp64 host buffers; A: column-major m x 2 int32
 * host buffer, or NULL / m == 0 for the all-to-all hypothesis (utils.h:61-71).
 * Host buffers are only read during the call. */
int clipper_hip_affinity_euclidean(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                                   const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                   double sigma, double epsilon, double mindist,
                                   double affinityeps);

/* Same with PointNormalDistance (pointnormal_distance.cpp:13-35, Params .h:25-31); d == 6. */
int clipper_hip_affinity_pointnormal(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                                     const double* D2, int64_t n2, const int32_t* A, int64_t m,
                                     double sigp, double epsp, double sign, double epsn,
                                     double affinityeps);

/* The same work split at the PCIe boundary, so a caller (and bench.py) can keep the inputs
 * resident in HBM and time the device work alone:
 *   clipper_hip_stage_inputs   = clipper.cpp:24-25 (A or all-to-all) + H2D of D1, D2, A +
 *                                gather of the per-association point tables
 *   clipper_hip_affinity_*_staged = the pair loop clipper.cpp:31-56 + :61-64 on device.
 * clipper_hip_affinity_euclidean(...) == stage_inputs(...) then ..._euclidean_staged(...). */
int clipper_hip_stage_inputs(clipper_hip_t* h, const double* D1, int d, int64_t n1,
                             const double* D2, int64_t n2, const int32_t* A, int64_t m);
int clipper_hip_affinity_euclidean_staged(clipper_hip_t* h, double sigma, double epsilon,
                                          double mindist, double affinityeps);
int clipper_hip_affinity_pointnormal_staged(clipper_hip_t* h, double sigp, double epsp,
                                            double sign, double epsn, double affinityeps);

/* rows of A_ (= dimension of M_); CLIPPER::getInitialAssociations (clipper.cpp:117-120) */
int64_t clipper_hip_num_associations(const clipper_hip_t* h);
int clipper_hip_get_associations(const clipper_hip_t* h, int32_t* A_out /* col-major m x 2 */);

/* ---- matrix get / set ---------------------------------------------------------------- */

/* CLIPPER::setMatrixData (clipper.cpp:149-158): dense column-major m x m host matrices;
 * only the strict upper triangle is used. Keeps the current association list if its
 * length is m (so getSelectedAssociations keeps working), else clears it. */
int clipper_hip_set_matrix(clipper_hip_t* h, const double* M, const double* C, int64_t m);

/* CLIPPER::setSparseMatrixData (clipper.cpp:162-166): CSC, int64 column pointers, int32 row
 * indices. The reference reads the upper triangle of what it is given (selfadjointView<Upper>,
 * clipper.cpp:194-271) and so does this: an entry (i, j) with i < j stands for the symmetric pair,
 * entries below the diagonal are NOT read (a full symmetric matrix counts through its upper half; a
 * lower-triangular one is an empty matrix, as in the reference). The same (row, column) stored twice
 * is an error with the compressed storage. The diagonal is implicit in this library: a stored
 * non-zero diagonal entry (outside the reference's contract, clipper.h:137-138; it would count once
 * on top of the identity there) is refused with CLIPPER_HIP_E_INVALID; explicit zeros are dropped.
 * When entries below the diagonal were ignored the call still returns 0 and clipper_hip_last_error() holds a
 * warning that says how many (a caller that stored both triangles, or only the lower one, can tell). */
int clipper_hip_set_sparse(clipper_hip_t* h, int64_t m, const int64_t* Mcolptr,
                           const int32_t* Mrow, const double* Mval, const int64_t* Ccolptr,
                           const int32_t* Crow, const double* Cval);

/* CLIPPER::getAffinityMatrix / getConstraintMatrix (clipper.cpp:131-145): dense symmetric
 * fp64 with the identity added. Either pointer may be NULL. Moves 8*m^2 bytes per matrix
 * over PCIe: for tests and interoperability, not for the hot loop. */
int clipper_hip_get_matrix(clipper_hip_t* h, double* M_out, double* C_out);

/* ---- solver -------------------------------------------------------------------------- */

/* CLIPPER::solve -> findDenseClique (clipper.cpp:69-78, 172-323). u0: m doubles (host),
 * required (the facade supplies utils::randvec when the caller gives none). u_out (m
 * doubles) may be NULL. Rounding is done on the host: NONZERO and DSD_HEU with the reference's
 * exact tie-breaking (utils.cpp:33-68); DSD (exact densest subgraph of the graph induced by
 * nnz(u), dsd.cpp:171-320, Goldberg's flow algorithm) gathers that sub-matrix from the device
 * first — on a multi-process shard it returns CLIPPER_HIP_E_SCOPE. */
int clipper_hip_solve(clipper_hip_t* h, const double* u0, const clipper_params_t* params,
                      double* u_out, clipper_solve_info_t* info);

/* Split form: clipper_hip_stage_u0 copies u0 to HBM; clipper_hip_solve_staged runs
 * findDenseClique on it (device loop + D2H of u + host rounding).
 * clipper_hip_solve(u0, ...) == stage_u0(u0) then solve_staged(...). */
int clipper_hip_stage_u0(clipper_hip_t* h, const double* u0);
int clipper_hip_solve_staged(clipper_hip_t* h, const clipper_params_t* params, double* u_out,
                             clipper_solve_info_t* info);

/* Solution::nodes (clipper.h:69); returns the count or <0. */
int clipper_hip_get_nodes(const clipper_hip_t* h, int32_t* nodes_out, int32_t capacity);
/* CLIPPER::getSelectedAssociations (clipper.cpp:124-127): column-major k x 2. */
int clipper_hip_get_selected_associations(const clipper_hip_t* h, int32_t* A_out,
                                          int32_t capacity);

/* dsd::solve(M_, S) (dsd.cpp:274-320, the reference's `clipper::dsd::solve`): exact densest
 * subgraph (Goldberg) of the current affinity matrix, restricted to the k nodes S (NULL / k <= 0:
 * all nodes). The induced sub-matrix is gathered from the device, the flow algorithm runs on the
 * host. Returns the number of nodes written to nodes_out (ascending) or <0. */
int clipper_hip_densest_subgraph(clipper_hip_t* h, const int32_t* S, int32_t k,
                                 int32_t* nodes_out, int32_t capacity);

/* ---- before the path: putative associations ------------------------------------------------ */

/* k nearest neighbours in P1 of every point of P0 (both d x n column-major as `clipper::Data`:
 * each point contiguous; d = 2 | 3; knn <= 16), squared L2 distances ascending, brute force on
 * the device — what the nanoflann kd-tree queries of benchmarks/bm_utils.cpp:147-176 return.
 * idx_out / sqd_out (may be NULL): n0 x knn row-major; -1 / 1e300 where P1 has fewer points.
 * Among exactly equal distances the lower index comes first. Stand-alone: needs no context. */
int clipper_hip_knn(int device, const double* P0, int64_t n0, const double* P1, int64_t n1, int d,
                    int knn, int32_t* idx_out, double* sqd_out);

/* utils::distance_based_correspondences (benchmarks/bm_utils.cpp:147-232): associations
 * (i, nn_k(i)) within `radius`, i ascending / neighbours by distance; with enforce_1to1 one row
 * per point of P1 (ascending) — its closest claimant. A_out: column-major n x 2 with n = the
 * return value (<0: error); capacity in rows (n0*knn always suffices). */
int64_t clipper_hip_distance_based_correspondences(int device, const double* P0, int64_t n0,
                                                   const double* P1, int64_t n1, int d, int knn,
                                                   double radius, int enforce_1to1,
                                                   int32_t* A_out, int64_t capacity);

/* Line-search window: how many consecutive step sizes alpha, alpha*beta, ... of the
 * backtracking line search (clipper.cpp:234-251) one pass over M evaluates at once. The
 * trial sequence, the accepted trial and the result are those of the reference for every
 * window; only the number of passes over M changes. 0 = automatic (6 for m >= 8500 on slices,
 * m >= 6000 on a dense store; 4 for m >= 2000; else 1);
 * 1, 4, 6 or 8 forces a size (also: environment CLIPPER_HIP_WINDOW). Takes effect at the next
 * affinity build / set_matrix. clipper_hip_window returns the size in use. */
int clipper_hip_set_window(clipper_hip_t* h, int window);
int clipper_hip_window(const clipper_hip_t* h);

/* The resident solver: when the slices of the current matrix fit the LDS of the workgroups that
 * share them (m up to a few thousand, one device, C == pattern(M), automatic window), the whole of
 * findDenseClique (clipper.cpp:172-323) runs as ONE launch that keeps M on chip; otherwise — and
 * always with mode 1, or CLIPPER_HIP_RESIDENT=0 in the environment — as the streaming launches
 * (decision + pass, tail) per iteration. Same trial sequence and result either way. Mode 1 takes
 * effect at the next solve; back to 0 at the next affinity build / set_matrix (the plan is made there).
 * clipper_hip_last_solver: what the last solve ran on, 0 = streaming launches, 1 = resident. */
int clipper_hip_set_resident(clipper_hip_t* h, int mode);
int clipper_hip_last_solver(const clipper_hip_t* h);

/* The row view. Row r of every line-search candidate max(u + alpha g, 0) (clipper.cpp:235-236) is
 * exactly zero unless u[r] > 0 or g[r] > 0, and such a row adds exact zeros to M x: once the
 * projected gradient ascent has driven most of u to zero (a few iterations on registration data)
 * a pass needs only the rows that are still live. With M in slices (C == pattern(M)) the solver then
 * builds the slices of M[live rows, :] — scored again from the staged points on the
 * scorePairwiseConsistency path with a built-in invariant, filtered out of M's own slices for a
 * matrix that was handed over (setMatrixData / setSparseMatrixData, custom invariants) — and
 * streams THOSE while the device-side check "no live row outside the view" holds; any pass for
 * which it does not hold streams M itself. Same trial sequence, same sums up to the
 * order of the partial sums. mode 0 = automatic, 1 = never (also: CLIPPER_HIP_ROW_VIEW=0).
 * A view small enough for the LDS of the chip (at most 1024 rows, a few MB: the headline problem's 524
 * rows x 10 000 columns) is not streamed at all: the iterations on it run as ONE launch of workgroups that
 * each keep complete columns of the view on chip, until the solve ends or a row outside the view becomes
 * live (csrc/k_rv_resident.hip.h); mode 2 (also: CLIPPER_HIP_VIEW_RESIDENT=0) keeps the views but streams them. */
int clipper_hip_set_row_view(clipper_hip_t* h, int mode);
int clipper_hip_get_view_stats(const clipper_hip_t* h, clipper_hip_view_stats_t* out);

/* The live sub-problem: once a row view exists and the penalty d is large, no association outside a small set S (the
 * view's rows and the few columns with many entries among them) can get a positive gradient again — by a bound on
 * clipper.cpp:238-241 that the decision checks for every candidate it plans — and the solve continues on the
 * associations of S as a problem of its own (csrc/k_subproblem.hip.h): the same launches on M[S,S]. Exact: what is left
 * out is provably zero. mode 0 = automatic (one shard, built-in invariants, m >= 12 000: smaller problems' views are taken
 * by the resident solver), 1 = never (also: CLIPPER_HIP_SUBPROBLEM=0), 2 = automatic, but M[S,S] always as slices (a
 * sub-problem that is mostly non-zero — the inlier block — is otherwise kept as a dense fp32 store; also
 * CLIPPER_HIP_SUB_DENSE=0). */
int clipper_hip_set_subproblem(clipper_hip_t* h, int mode);

/* The products of clipper_hip_matvec through a row view built for the given rows (ascending association
 * indices): yM = M_off[:, rows] x[rows], yC likewise — what a pass of the solver computes when it
 * streams the view. For tests of the view's storage and of the rectangular fill kernel that writes it. */
int clipper_hip_view_matvec(clipper_hip_t* h, const int32_t* rows, int64_t nrows, const double* x,
                            double* yM, double* yC);

/* How the current matrix is stored: CLIPPER_HIP_STORE_F32_CSC only while the compressed copy is
 * in use (one shard, C == pattern(M)); a context created with it otherwise reports _F32. */
int clipper_hip_storage_in_use(const clipper_hip_t* h);

/* One pass of the mat-vec kernel: yM = M_off*x, yC = C_off*x (the products at
 * clipper.cpp:194,202,205,219,240-241,268,271). x, yM, yC: m doubles on the host. */
int clipper_hip_matvec(clipper_hip_t* h, const double* x, double* yM, double* yC);

/* ---- measurement ---------------------------------------------------------------------- */

/* When on (1), sampled mat-vec launches of a solve are bracketed by HIP events on the stream they
 * run on; clipper_hip_get_timings then reports their mean / min duration. 2: also an event pair around
 * every launch of the resident solver on a row view (clipper_hip_view_stats_t::resident_event_us). */
int clipper_hip_set_profiling(clipper_hip_t* h, int on);
int clipper_hip_get_timings(const clipper_hip_t* h, clipper_hip_timings_t* out);
/* Launches the mat-vec kernel `reps` times back to back on resident data and returns the
 * mean kernel time in microseconds (events on the launch stream). */
int clipper_hip_bench_matvec(clipper_hip_t* h, int reps, double* avg_us);
/* Device name, CU count, HBM bytes (for bench.py's report). */
/* ---- host-side neighbours of the path (no device work) -----------------
 * utils::read_ply (benchmarks/bm_utils.cpp:24-79): x, y, z of the `vertex` element of an ascii or
 * binary PLY file -> pts_out, column-major 3 x n (one datum per column, as invariants::Data).
 * pts_out == NULL: returns the vertex count only. Returns n or a negative status. */
int64_t clipper_hip_read_ply_xyz(const char* path, double* pts_out, int64_t capacity);
/* utils::generate_synthetic_correspondences (bm_utils.cpp:277-341): m putative associations with
 * outlier ratio rho — ni = round(m (1 - rho)) inliers drawn without replacement from the p good
 * associations Agood (column-major p x 2), placed LAST; m - ni outliers sampled uniformly without
 * repetition from all n0 * n1 pairs that are not in Agood, placed FIRST. A_out: m x 2, Agt_out:
 * ni x 2 (capacity m x 2), both column-major; *ni_out = ni. The reference seeds a mt19937 from
 * random_device; here the seed is an argument (mt19937_64), so a call is reproducible.
 * CLIPPER_HIP_E_STATE when Agood holds fewer than ni associations (the reference returns {}). */
int clipper_hip_generate_synthetic_correspondences(int64_t n0, int64_t n1, const int32_t* Agood,
                                                   int64_t p, int64_t m, double rho, uint64_t seed,
                                                   int32_t* A_out, int32_t* Agt_out,
                                                   int64_t* ni_out);
/* utils::get_precision_recall (bm_utils.cpp:345-371): TP counted row by row of A (na x 2) against
 * the set of rows of Agt (ngt x 2); (0, 0) when either is empty. */
int clipper_hip_precision_recall(const int32_t* A, int64_t na, const int32_t* Agt, int64_t ngt,
                                 double* precision, double* recall);
/* Least-squares rigid transform T (column-major 4 x 4) with D2[:, A(i,1)] ~ R D1[:, A(i,0)] + t over
 * the k >= 3 associations A (column-major k x 2): centroids, 3 x 3 cross-covariance, SVD, reflection
 * fix (Arun / Horn) — the consumer of the selected associations in examples/python/ex4_bunny.ipynb.
 * D1: 3 x n1, D2: 3 x n2, column-major. */
int clipper_hip_estimate_rigid_transform(const double* D1, int64_t n1, const double* D2, int64_t n2,
                                         const int32_t* A, int64_t k, double* T_out);

/* Measurement only (context created with CLIPPER_HIP_STAMPS=1 in the environment): per workgroup of
 * the last pass launch {start, decision done, end, info} on the 100 MHz device wall clock. */
int clipper_hip_debug_stamps(clipper_hip_t* h, int64_t* out, int capacity);

/* Test infrastructure: occupies `workgroups` wave slots with `lds_bytes` of LDS each on `device` for `milliseconds`
 * (a kernel that sleeps) and returns when it is over — "another tenant on the device" for the tests of the resident
 * solvers' time-outs (tests/test_gpu_rv_resident.py runs it in a second process). */
int clipper_hip_debug_occupy(int device, int workgroups, int lds_bytes, double milliseconds);

int clipper_hip_device_info(const clipper_hip_t* h, char* name64, int* cus, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CLIPPER_HIP_H */
