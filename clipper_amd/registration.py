"""Host-side pieces either side of the hot path for the 3-D registration use case:
a PLY point reader, the putative-association generator of the reference's bunny example, and the
rigid-transform estimate that consumes the selected associations. Plain numpy; nothing here
touches the device — the path between them is clipper_hip_affinity_* / clipper_hip_solve.

Reference sites (relative to /root/reference): the data-set recipe restates
examples/python/ex4_bunny.ipynb cell 2 (sample the model, ground-truth transform, uniform noise in
a sigma-cube, outlier points uniform in a ball, nia correct + noa wrong associations); the
transform estimate is the closed form of Arun et al. that cell 6 obtains from open3d's
TransformationEstimationPointToPoint and benchmarks/bm_utils.cpp uses through Eigen::umeyama
(without scaling); the error measures are cell 7's."""
from __future__ import annotations

import struct

import numpy as np

_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h",
              "ushort": "H", "uint16": "H", "int": "i", "int32": "i", "uint": "I", "uint32": "I",
              "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_ply_xyz(path: str) -> np.ndarray:
    """Vertex positions (n, 3) float64 of an ascii or binary PLY file (x, y, z properties of the
    `vertex` element; other properties and elements are skipped; list properties in the vertex
    element are not supported)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, nvert, props, in_vertex = None, 0, [], False
        seen_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    if seen_vertex:
                        raise ValueError("two vertex elements")
                    nvert, seen_vertex = int(tok[2]), True
                elif not seen_vertex:
                    raise ValueError("elements before `vertex` are not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(k in names for k in "xyz"):
            raise ValueError("vertex element has no x, y, z")
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(nvert)]
            arr = np.array([[float(r[names.index(k)]) for k in "xyz"] for r in rows], dtype=np.float64)
            for c, k in enumerate("xyz"):      # through the declared type, as a binary file would carry it
                arr[:, c] = arr[:, c].astype(dict(props)[k]).astype(np.float64)
        elif fmt in ("binary_little_endian", "binary_big_endian"):
            end = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, end + t) for n, t in props])
            raw = np.frombuffer(f.read(dt.itemsize * nvert), dtype=dt, count=nvert)
            arr = np.stack([raw[k].astype(np.float64) for k in "xyz"], axis=1)
        else:
            raise ValueError(f"unknown PLY format {fmt!r}")
    return arr


def write_ply_xyz(path: str, pts: np.ndarray) -> None:
    """Binary little-endian PLY with float x, y, z (round-trip partner of read_ply_xyz)."""
    pts = np.asarray(pts, dtype="<f4")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(pts))
        f.write(b"property float x\nproperty float y\nproperty float z\nend_header\n")
        f.write(pts.tobytes())


def random_rotation(rng: np.random.Generator) -> np.ndarray:
    """Uniformly distributed rotation matrix (QR of a Gaussian matrix, sign-fixed)."""
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def uniform_in_ball(rng: np.random.Generator, n: int, radius: float) -> np.ndarray:
    """n points uniformly distributed in a 3-D ball."""
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v * (radius * rng.random(n) ** (1.0 / 3.0))[:, None]


def make_registration_dataset(model_points: np.ndarray, m: int, n1: int, n2o: int, outrat: float,
                              sigma: float, T_21: np.ndarray, seed: int = 0):
    """Two views of a model and m putative associations between them.

    model_points (N, 3): the full model; n1 points are sampled for view 1, transformed by T_21
    (4x4) into view 2 and perturbed by noise uniform in [-sigma/2, sigma/2]^3; n2o outlier points
    (uniform in a unit ball around view 2's centroid) are appended to view 2. The association list
    holds nia = m - round(m*outrat) correct pairs (i, i) followed by noa wrong pairs (i, j != i),
    all distinct. Returns D1 (3, n1), D2 (3, n1 + n2o) float64, A (m, 2) int32, Agt (nia, 2)."""
    rng = np.random.default_rng(seed)
    n2 = n1 + n2o
    noa = int(round(m * outrat))
    nia = m - noa
    if nia > n1:
        raise ValueError("more inlier associations than model points")
    if noa > n1 * n2 - n1:
        raise ValueError("more outlier associations than wrong pairs exist")
    pick = rng.choice(len(model_points), n1, replace=False)
    D1 = np.asarray(model_points, dtype=np.float64)[pick].T
    D2 = T_21[:3, :3] @ D1 + T_21[:3, 3:4]
    D2 = D2 + rng.uniform(-sigma / 2.0, sigma / 2.0, size=D2.shape)
    O2 = uniform_in_ball(rng, n2o, 1.0).T + D2.mean(axis=1, keepdims=True)
    D2 = np.hstack([D2, O2])
    good = rng.choice(n1, nia, replace=False)
    Agt = np.stack([good, good], axis=1).astype(np.int32)
    bad, seen = [], set()
    while len(bad) < noa:
        i, j = int(rng.integers(n1)), int(rng.integers(n2))
        if i == j or (i, j) in seen:
            continue
        seen.add((i, j))
        bad.append((i, j))
    A = np.concatenate([Agt, np.array(bad, dtype=np.int32).reshape(-1, 2)]).astype(np.int32)
    return np.ascontiguousarray(D1), np.ascontiguousarray(D2), A, Agt


def estimate_rigid_transform(D1: np.ndarray, D2: np.ndarray, A: np.ndarray) -> np.ndarray:
    """Least-squares rigid transform T (4x4) with D2[:, A[k,1]] ~ R D1[:, A[k,0]] + t over the
    associations A (k x 2): centroids, 3x3 cross-covariance, SVD, reflection fix."""
    A = np.asarray(A)
    if A.shape[0] < 3:
        raise ValueError("at least 3 associations are needed")
    P, Q = D1[:, A[:, 0]], D2[:, A[:, 1]]
    cp, cq = P.mean(axis=1, keepdims=True), Q.mean(axis=1, keepdims=True)
    H = (Q - cq) @ (P - cp).T
    U, _, Vt = np.linalg.svd(H)
    S = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ S @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = (cq - R @ cp).ravel()
    return T


def transform_error(T: np.ndarray, That: np.ndarray):
    """(rotation error [rad], translation error) of an estimate against the truth."""
    E = np.linalg.inv(T) @ That
    c = min(max((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0), 1.0)
    return abs(float(np.arccos(c))), float(np.linalg.norm(E[:3, 3]))


def precision_recall(Ain: np.ndarray, Agt: np.ndarray):
    """Fraction of selected associations that are correct / of correct ones that were selected."""
    sel = {tuple(r) for r in np.asarray(Ain).tolist()}
    gt = {tuple(r) for r in np.asarray(Agt).tolist()}
    hit = len(sel & gt)
    return (hit / len(sel) if sel else 0.0), (hit / len(gt) if gt else 0.0)


# ---- the reference benchmark's putative-association recipe (benchmarks/main.cpp:156-166) --------

def scale_to_cube(pts: np.ndarray, s: float = 1.0) -> np.ndarray:
    """bm_utils.cpp:110-115: uniform scale so that the longest bounding-box edge is `s`."""
    d = pts.max(axis=0) - pts.min(axis=0)
    return pts * (s / d.max())


def bounded_normal_noise(rng: np.random.Generator, n: int, sigma: float, beta: float) -> np.ndarray:
    """bm_utils.cpp:117-143: N(0, sigma^2 I) conditioned on ||v|| <= beta (batched rejection)."""
    eta = rng.normal(0.0, sigma, (n, 3))
    bad = np.linalg.norm(eta, axis=1) > beta
    while bad.any():
        eta[bad] = rng.normal(0.0, sigma, (int(bad.sum()), 3))
        bad = np.linalg.norm(eta, axis=1) > beta
    return eta


def ground_truth_associations(pcd0: np.ndarray, pcd1: np.ndarray, radius: float, device: int = 0):
    """main.cpp:85-91: one-to-one nearest-neighbour associations within `radius`, on the GPU
    (clipper_hip_distance_based_correspondences). pcd*: n x 3, rows = points."""
    from . import _abi as abi
    return abi.distance_based_correspondences(pcd0.T, pcd1.T, 1, radius, True, device=device)


def generate_synthetic_correspondences(n0: int, n1: int, Agood: np.ndarray, m: int, rho: float,
                                       rng: np.random.Generator):
    """bm_utils.cpp:277-341: m putative associations with outlier ratio rho — round(m (1 - rho))
    inliers drawn without replacement from Agood (placed last), the rest sampled uniformly without
    repetition from all n0*n1 pairs that are not in Agood (placed first). Returns (A, Agt) or None
    when Agood holds too few associations."""
    if not 0.0 <= rho <= 1.0:
        raise ValueError("outlier ratio must be in [0, 1]")
    ni = int(round(m * (1.0 - rho)))
    no = m - ni
    if ni > len(Agood):
        return None
    Agt = np.asarray(Agood)[rng.permutation(len(Agood))[:ni]].astype(np.int32)
    good = np.asarray(Agood, dtype=np.int64)
    good_flat = set((good[:, 0] * n1 + good[:, 1]).tolist())
    out = np.zeros((no, 2), dtype=np.int32)
    seen: set = set()
    k = 0
    while k < no:
        cand = rng.integers(0, n0 * n1, size=2 * (no - k) + 8)
        for flat in cand.tolist():
            if flat in seen:
                continue
            seen.add(flat)
            if flat in good_flat:
                continue
            out[k] = (flat // n1, flat % n1)
            k += 1
            if k == no:
                break
    return np.concatenate([out, Agt], axis=0), Agt
