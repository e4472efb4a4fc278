"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference benchmark's association
utilities, /root/reference/benchmarks/bm_utils.cpp. Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of the measurement tools may import this module.

PARITY UNPINNED: the reference holds no golden vectors or tests for these functions (they live
in benchmarks/, outside its test suite); what is followed, line by line:
  bm_utils.cpp:147-176  distance_based_correspondences, search part: for every row of pcd0 the
                        knn nearest rows of pcd1 by squared L2 distance, ascending (nanoflann
                        KNNResultSet over a kd-tree; restated as an exact brute-force search —
                        the result set of a kd-tree query is the exact k nearest; among exactly
                        equal distances the lower index is put first, the tree's own order there
                        being an artefact of its traversal)
  bm_utils.cpp:187-229  conversion: rows (c0, c1) for c0 ascending / neighbours in result order,
                        kept if sd <= radius^2; enforce_1to1: one row per c1 (std::map: ascending),
                        the claimant with the smallest sd (std::min_element: the first minimum)
  bm_utils.cpp:277-341  generate_synthetic_correspondences: ni = round(m (1 - rho)) inliers drawn
                        without replacement from Agood, placed LAST; no = m - ni outliers sampled
                        uniformly from all pairs that are not in Agood, without repetition, placed
                        first. (The reference seeds from std::random_device; here a Generator.)
  bm_utils.cpp:345-371  get_precision_recall: TP = rows of A contained in Agt
  bm_utils.cpp:117-143  generate_bounded_normal_noise (rejection: ||v|| <= beta), :110 scale_to_cube
"""
from __future__ import annotations

import numpy as np


def knn_bruteforce(pcd0: np.ndarray, pcd1: np.ndarray, knn: int):
    """pcd0: n0 x d, pcd1: n1 x d (rows = points). Returns (idx n0 x knn, sqd n0 x knn); -1 / inf
    where pcd1 has fewer than knn points."""
    pcd0, pcd1 = np.asarray(pcd0, np.float64), np.asarray(pcd1, np.float64)
    n0, n1 = len(pcd0), len(pcd1)
    idx = np.full((n0, knn), -1, np.int64)
    sqd = np.full((n0, knn), np.inf)
    for i in range(n0):
        df = pcd0[i][None, :] - pcd1
        d2 = np.zeros(n1)
        for k in range(pcd1.shape[1]):          # coordinate order, no fused operations
            d2 = d2 + df[:, k] * df[:, k]
        order = np.lexsort((np.arange(n1), d2))[:knn]   # by distance, then by index
        idx[i, :len(order)] = order
        sqd[i, :len(order)] = d2[order]
    return idx, sqd


def distance_based_correspondences(pcd0, pcd1, knn: int, radius: float, enforce_1to1: bool) -> np.ndarray:
    idx, sqd = knn_bruteforce(pcd0, pcd1, knn)
    r2 = radius * radius
    rows, claim = [], {}
    for i in range(idx.shape[0]):
        for j in range(idx.shape[1]):
            c1, sd = int(idx[i, j]), float(sqd[i, j])
            if c1 < 0:
                continue
            if sd <= r2:
                rows.append((i, c1))
                if enforce_1to1:
                    claim.setdefault(c1, []).append((i, sd))
    if enforce_1to1:
        rows = []
        for c1 in sorted(claim):
            best = 0
            for q in range(1, len(claim[c1])):
                if claim[c1][q][1] < claim[c1][best][1]:
                    best = q
            rows.append((claim[c1][best][0], c1))
    return np.array(rows, dtype=np.int32).reshape(-1, 2)


def generate_synthetic_correspondences(n0: int, n1: int, Agood: np.ndarray, m: int, rho: float,
                                       rng: np.random.Generator):
    """Returns (A m x 2, Agt ni x 2) or None when Agood holds too few associations."""
    assert 0.0 <= rho <= 1.0
    ni = int(round(m * (1.0 - rho)))
    no = m - ni
    p = len(Agood)
    if ni > p:
        return None
    sel = rng.permutation(p)[:ni]
    good = {(int(a), int(b)) for a, b in Agood}
    A = np.zeros((m, 2), np.int32)
    Agt = Agood[sel].astype(np.int32)
    A[no:] = Agt
    seen = set()
    k = 0
    while k < no:
        flat = int(rng.integers(0, n0 * n1))
        if flat in seen:
            continue
        seen.add(flat)
        row = (flat // n1, flat % n1)           # k2ij_full, :262-273
        if row in good:
            continue
        A[k] = row
        k += 1
    return A, Agt


def get_precision_recall(A: np.ndarray, Agt: np.ndarray):
    if len(A) == 0 or len(Agt) == 0:
        return 0.0, 0.0
    gt = {(int(a), int(b)) for a, b in Agt}
    tp = sum((int(a), int(b)) in gt for a, b in A)
    return tp / len(A), tp / len(Agt)


def generate_bounded_normal_noise(n: int, sigma: float, beta: float, rng: np.random.Generator) -> np.ndarray:
    eta = np.zeros((n, 3))
    for i in range(n):
        while True:
            v = rng.normal(0.0, sigma, 3)
            if np.linalg.norm(v) <= beta:
                break
        eta[i] = v
    return eta


def scale_to_cube(pts: np.ndarray, s: float = 1.0) -> np.ndarray:
    d = pts.max(axis=0) - pts.min(axis=0)
    return pts * (s / d.max())
