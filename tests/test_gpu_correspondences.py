"""GPU tests of the putative-association generator: the brute-force nearest-
neighbour kernels and utils::distance_based_correspondences of the reference benchmark, through
the C ABI, against the numpy restatement oracle/bm_utils_ref.py — bit-exact indices, squared
distances equal to the last bit (same fp64 operations in the same order)."""
import json
import os

import numpy as np
import pytest

from clipper_amd import _abi as abi
from oracle import bm_utils_ref as ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clouds(n0, n1, d, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n0, d)), rng.random((n1, d))


@pytest.mark.parametrize("n0,n1,d,knn", [(1, 1, 3, 1), (5, 3, 3, 4), (257, 1023, 3, 1), (300, 1025, 3, 3),
                                          (1000, 2500, 3, 5), (700, 3000, 2, 8), (64, 5000, 3, 16)])
def test_knn_against_brute_force(n0, n1, d, knn):
    P0, P1 = _clouds(n0, n1, d, seed=n0 + n1)
    idx, sqd = abi.knn(P0.T, P1.T, knn)
    ridx, rsqd = ref.knn_bruteforce(P0, P1, knn)
    assert np.array_equal(idx, ridx.astype(np.int32))
    have = ridx >= 0
    assert np.array_equal(sqd[have], rsqd[have])            # identical fp64 operations
    assert np.all(sqd[~have] >= 1e299)


def test_knn_ties_keep_the_lower_index():
    # duplicated points: exactly equal distances
    rng = np.random.default_rng(1)
    base = rng.random((40, 3))
    P1 = np.concatenate([base, base, base])     # every point three times: indices j, j+40, j+80
    P0 = base[:10] + 1e-3
    idx, _ = abi.knn(P0.T, P1.T, 3)
    for i in range(10):
        assert idx[i].tolist() == [i, i + 40, i + 80]


@pytest.mark.parametrize("knn,radius,one", [(1, 0.05, True), (1, 0.05, False), (3, 0.08, False),
                                             (4, 0.08, True), (2, 1e-6, True), (5, 10.0, False)])
def test_distance_based_correspondences(knn, radius, one):
    pts = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_points.json")))["points"])
    pts = ref.scale_to_cube(pts, 1.0)
    rng = np.random.default_rng(7)
    noisy = pts + ref.generate_bounded_normal_noise(len(pts), 0.01, 0.0554, rng)
    A = abi.distance_based_correspondences(pts.T, noisy.T, knn, radius, one)
    Ar = ref.distance_based_correspondences(pts, noisy, knn, radius, one)
    assert A.shape == Ar.shape and np.array_equal(A, Ar)
    if one:                                       # one row per point of the second cloud
        assert len(np.unique(A[:, 1])) == len(A)


def test_reference_benchmark_recipe_end_to_end():
    """benchmarks/main.cpp:156-195 on the bunny sample: noisy copy, ground-truth associations by
    1-NN within the noise bound, synthetic putative set, affinity + solve, precision / recall."""
    from clipper_amd import synth
    pts = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_points.json")))["points"])
    pts = ref.scale_to_cube(pts, 1.0)
    rng = np.random.default_rng(3)
    noisy = pts + ref.generate_bounded_normal_noise(len(pts), 0.01, 0.0554, rng)
    Agt0 = abi.distance_based_correspondences(pts.T, noisy.T, 1, 0.0554, True)
    assert len(Agt0) > 0.5 * len(pts)
    out = ref.generate_synthetic_correspondences(len(pts), len(noisy), Agt0, 256, 0.8, rng)
    assert out is not None
    A, Agt = out
    g = abi.HipClipper(storage=abi.STORE_F32_CSC)
    g.score_pairwise_consistency_euclidean(pts.T, noisy.T, A, sigma=0.015, epsilon=0.05)
    s = g.solve(np.random.default_rng(4).random(len(A)))
    p, r = ref.get_precision_recall(A[s.nodes], Agt)
    assert p >= 0.9 and r >= 0.5
