// kernels.hip.h — hand-written CDNA4 (gfx950, wave64) kernels of the CLIPPER hot path.
//
// Compiled with -ffp-contract=off: every fp64 expression rounds exactly as written and
// fused multiply-adds appear only where fma() is spelled out — the same convention as the
// CPU oracle, so that threshold decisions (c < epsilon, scr > affinityeps, deltaF < -eps)
// are taken on bit-identical operands wherever the operation order can be shared.
//
// Data layout (see include/clipper_hip.h): a shard owns global columns [c0, c0+W) of the
// symmetric matrix M_off and stores S[j][c] = M(j, c0+c), j = 0..m-1, row pitch ld = W
// (multiple of 64 elements, zero padded). Because M is symmetric, column c of S is row
// (c0+c) of M, so   (M_off x)[c0+c] = sum_j S[j][c] * x[j]:
// every lane owns output columns, walks down the rows with a wave-uniform x[j] (scalar
// load), and never needs a cross-lane reduction; a wave's 64 lanes read 64 x 16 B =
// 1 KiB of one row per load instruction — fully coalesced HBM streaming.
//
// Reference sites (relative to /root/reference):
//   k_affinity_*  : src/clipper.cpp:31-56 + src/invariants/euclidean_distance.cpp:13-31,
//                   src/invariants/pointnormal_distance.cpp:13-35
//   k_gemv[_csc]   : every `M_.selfadjointView<Upper>() * v` / `C_...* v` in
//                   src/clipper.cpp:194,202,205,219,240-241,268,271 (one pass over M serves a
//                   whole window of line-search candidates), preceded by the decisions of
//                   findDenseClique's control flow :244-262, :268-280 (decide)
//   k_tail        : the O(m) algebra of findDenseClique, src/clipper.cpp:219-220 (gradient),
//                   :235-242, :253 (step, projection, trial objective), :268-274 (penalty terms)
//   k_affinity_sym: the same scores as k_affinity_*, upper block triangle + mirrored stores
//
// Files (all in namespace clipper_hip):
//   k_solver.hip.h    solver state, decide (the head of every pass launch), k_init, k_tail, k_scal_fold
//   k_gemv.hip.h      the dense pass: k_gemv, k_gemv_plain, k_reduce_pass (column shards), k_reduce, k_spread
//   k_slices.hip.h    the compressed storage: layout, the pass on it (k_gemv_slices), packers, k_slice_expand
//   k_csc.hip.h       producers of the slices: emission from the fill kernel's LDS image, groups
//   k_resident.hip.h  the resident solver: findDenseClique as one launch for problems that fit on chip
//   k_rv_resident.hip.h  the resident solver on a row view: the iterations that stream a view, as one launch
//   k_affinity.hip.h  k_gather_points, k_affinity_* (plain, compacting strips, symmetric tiles + emission)
//   k_matrix.hip.h    k_from_dense_upper, k_from_csc, k_gather_sub
//   k_rowview.hip.h   the row list of a row view of M (the live rows of the solver's current points)
//   k_subproblem.hip.h  the live sub-problem: column counts of a view, the selection, the hand-over and the way back
//   k_knn.hip.h       brute-force k-nearest neighbours (putative associations)
#pragma once

#include "k_solver.hip.h"
#include "k_gemv.hip.h"
#include "k_csc.hip.h"
#include "k_resident.hip.h"
#include "k_rv_resident.hip.h"
#include "k_affinity.hip.h"
#include "k_matrix.hip.h"
#include "k_rowview.hip.h"
#include "k_subproblem.hip.h"
#include "k_knn.hip.h"
