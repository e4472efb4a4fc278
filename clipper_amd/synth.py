"""Deterministic synthetic registration problems used by bench.py and tests.

The recipes restate the reference's benchmark driver so that the measured workload is the
one the reference itself times (citations relative to /root/reference):
  * bounded-normal point noise sigma = 0.01, ||eta|| <= 5.54*sigma
        benchmarks/main.cpp:31-32, benchmarks/bm_utils.cpp:131-143
  * putative associations = outliers first, then inliers; ni = round(m*(1-rho))
        benchmarks/bm_utils.cpp:277-349 (rows [0,no) outliers, [no,m) inliers, :311-315)
  * EuclideanDistance{sigma=0.015, epsilon=0.05}       benchmarks/main.cpp:221
Everything is seeded with numpy's default_rng, so tests, bench and the CPU baseline see
bit-identical inputs. Data generation is not part of the hot path and runs on the host.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

EUCLID_BENCH_PARAMS = dict(sigma=0.015, epsilon=0.05, mindist=0.0)


@dataclass
class Problem:
    D1: np.ndarray      # d x n1
    D2: np.ndarray      # d x n2
    A: np.ndarray       # m x 2 int32 (outliers first, then inliers)
    Agt: np.ndarray     # ni x 2 ground-truth inlier associations
    u0: np.ndarray      # m, explicit initial vector (same on every path)
    meta: dict


def rotation_axis_angle(axis=(1.0, 2.0, 3.0), angle=0.7):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def bounded_normal_noise(rng, n, sigma=0.01, beta=5.54 * 0.01):
    """benchmarks/bm_utils.cpp:131-143: N(0, sigma^2) per coordinate, redrawn until ||v|| <= beta."""
    eta = rng.normal(0.0, sigma, size=(n, 3))
    bad = np.linalg.norm(eta, axis=1) > beta
    while bad.any():
        eta[bad] = rng.normal(0.0, sigma, size=(int(bad.sum()), 3))
        bad = np.linalg.norm(eta, axis=1) > beta
    return eta


def _associations(rng, n_points, m, rho):
    ni = int(round(m * (1.0 - rho)))
    no = m - ni
    if ni > n_points:
        raise ValueError("not enough points for the requested number of inliers")
    inl = rng.permutation(n_points)[:ni]
    Agt = np.stack([inl, inl], axis=1).astype(np.int32)
    # unique outliers (a, b), a != b, drawn uniformly from all pairs
    seen = set()
    out = np.zeros((no, 2), dtype=np.int32)
    k = 0
    while k < no:
        need = no - k
        cand = rng.integers(0, n_points, size=(int(need * 1.2) + 16, 2))
        for a, b in cand:
            if a == b:
                continue
            key = int(a) * n_points + int(b)
            if key in seen:
                continue
            seen.add(key)
            out[k] = (a, b)
            k += 1
            if k == no:
                break
    A = np.concatenate([out, Agt], axis=0).astype(np.int32)
    return A, Agt


def make_euclidean_problem(m: int, rho: float, seed: int = 12345, n_points: int | None = None):
    """cfg2/cfg3 and the 100k/300k sweeps of BASELINE.json: uniform-cube 3-D points."""
    rng = np.random.default_rng(seed)
    n = int(n_points or m)
    P = rng.random((n, 3))
    R = rotation_axis_angle()
    t = np.array([0.5, -0.3, 0.8])
    Q = P @ R.T + t + bounded_normal_noise(rng, n)
    A, Agt = _associations(rng, n, m, rho)
    u0 = np.random.default_rng(seed + 1).random(m)
    meta = dict(kind="euclidean", m=m, rho=rho, seed=seed, n_points=n,
                invariant=dict(EUCLID_BENCH_PARAMS))
    return Problem(D1=np.ascontiguousarray(P.T), D2=np.ascontiguousarray(Q.T), A=A, Agt=Agt,
                   u0=u0, meta=meta)


def _random_unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make_pointnormal_problem(m: int, rho: float = 0.9, seed: int = 12345,
                             n_points: int | None = None):
    """cfg4: 6-DoF surfel registration, datum = [x y z nx ny nz], default invariant params."""
    rng = np.random.default_rng(seed)
    n = int(n_points or m)
    P = rng.random((n, 3)) * 10.0
    N = _random_unit(rng, n)
    R = rotation_axis_angle()
    t = np.array([0.5, -0.3, 0.8])
    Q = P @ R.T + t + bounded_normal_noise(rng, n)
    # normal noise: rotate by a small random rotation vector (sigma = 1 degree), renormalise
    w = rng.normal(0.0, np.deg2rad(1.0), size=(n, 3))
    Nq = N @ R.T
    Nq = Nq + np.cross(w, Nq)
    Nq = Nq / np.linalg.norm(Nq, axis=1, keepdims=True)
    A, Agt = _associations(rng, n, m, rho)
    u0 = np.random.default_rng(seed + 1).random(m)
    D1 = np.ascontiguousarray(np.concatenate([P, N], axis=1).T)
    D2 = np.ascontiguousarray(np.concatenate([Q, Nq], axis=1).T)
    meta = dict(kind="pointnormal", m=m, rho=rho, seed=seed, n_points=n,
                invariant=dict(sigp=0.5, epsp=0.5, sign=0.10, epsn=0.35))
    return Problem(D1=D1, D2=D2, A=A, Agt=Agt, u0=u0, meta=meta)


def precision_recall(Ain: np.ndarray, Agt: np.ndarray):
    """benchmarks/bm_utils.cpp:353-371."""
    if len(Ain) == 0 or len(Agt) == 0:
        return 0.0, 0.0
    gt = {(int(a), int(b)) for a, b in Agt}
    tp = sum((int(a), int(b)) in gt for a, b in Ain)
    return tp / len(Ain), tp / len(Agt)
