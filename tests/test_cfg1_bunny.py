"""BASELINE.json cfg1 — "Stanford Bunny 3D registration, EuclideanDistance, ~100 putative
associations, CPU path (plumbing)": PLY points -> putative associations -> affinity -> solve ->
selected set -> rigid transform, on a 600-point sample of the reference's bunny model
(tests/golden/bunny_points.json, made by tests/golden/make_bunny_fixture.py). The CPU tests run
the chain through the oracle; the GPU test runs the same data through the C ABI and requires the
oracle's selection."""
import json
import os

import numpy as np
import pytest

from clipper_amd import registration as reg
from oracle import clipper_ref as ref

HERE = os.path.dirname(os.path.abspath(__file__))
INV = dict(sigma=0.01, epsilon=0.02, mindist=0.0)      # ex4_bunny.ipynb cell 4
CFG1 = dict(m=100, n1=100, n2o=25, outrat=0.9, sigma=0.01)   # cell 3 scaled by 0.1


@pytest.fixture(scope="module")
def bunny():
    d = json.load(open(os.path.join(HERE, "golden", "bunny_points.json")))
    pts = np.array(d["points"], dtype=np.float64)
    assert pts.shape == (600, 3) and d["n_source_vertices"] == 9992
    assert np.all(pts.min(axis=0) >= np.array(d["bbox_min"]) - 1e-6)
    assert np.all(pts.max(axis=0) <= np.array(d["bbox_max"]) + 1e-6)
    return pts


def _truth(seed):
    rng = np.random.default_rng(1000 + seed)
    T = np.eye(4)
    T[:3, :3] = reg.random_rotation(rng)
    T[:3, 3] = rng.uniform(-5, 5, 3)
    return T


def test_ply_reader_round_trip_binary_and_ascii(tmp_path, bunny):
    p = str(tmp_path / "b.ply")
    reg.write_ply_xyz(p, bunny)
    back = reg.read_ply_xyz(p)
    assert np.array_equal(back, bunny.astype(np.float32).astype(np.float64))
    a = tmp_path / "a.ply"
    a.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\n"
                 "property float y\nproperty float z\nproperty uchar red\nelement face 1\n"
                 "property list uchar int vertex_indices\nend_header\n"
                 "0 0 0 255\n1 2 3 0\n-1.5 0.25 8 7\n3 0 1 2\n")
    assert np.array_equal(reg.read_ply_xyz(str(a)), [[0, 0, 0], [1, 2, 3], [-1.5, 0.25, 8]])
    bad = tmp_path / "bad.ply"
    bad.write_text("plx\n")
    with pytest.raises(ValueError):
        reg.read_ply_xyz(str(bad))


def test_dataset_recipe_and_transform_estimate(bunny):
    T = _truth(0)
    D1, D2, A, Agt = reg.make_registration_dataset(bunny, seed=0, T_21=T, **CFG1)
    assert D1.shape == (3, 100) and D2.shape == (3, 125) and A.shape == (100, 2) and A.dtype == np.int32
    assert Agt.shape == (10, 2) and np.all(Agt[:, 0] == Agt[:, 1]) and np.array_equal(A[:10], Agt)
    assert len({tuple(r) for r in A.tolist()}) == 100 and np.all(A[10:, 0] != A[10:, 1])
    assert A[:, 0].max() < 100 and A[:, 1].max() < 125
    # inlier pairs obey the ground truth up to the sigma-cube noise
    res = D2[:, Agt[:, 1]] - (T[:3, :3] @ D1[:, Agt[:, 0]] + T[:3, 3:4])
    assert np.max(np.abs(res)) <= 0.005 + 1e-12
    # noiseless: the closed form recovers the transform exactly
    D2x = T[:3, :3] @ D1 + T[:3, 3:4]
    That = reg.estimate_rigid_transform(D1, D2x, np.stack([np.arange(100)] * 2, axis=1))
    rerr, terr = reg.transform_error(T, That)
    assert rerr < 1e-7 and terr < 1e-9
    assert abs(np.linalg.det(That[:3, :3]) - 1.0) < 1e-12
    with pytest.raises(ValueError):
        reg.estimate_rigid_transform(D1, D2, A[:2])


@pytest.mark.parametrize("seed", range(6))
def test_cfg1_plumbing_through_the_oracle(bunny, seed):
    T = _truth(seed)
    D1, D2, A, Agt = reg.make_registration_dataset(bunny, seed=seed, T_21=T, **CFG1)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(D1, D2, A, **INV)
    s = r.solve(np.random.default_rng(seed + 1).random(len(A)))
    Ain = r.get_selected_associations()
    prec, rec = reg.precision_recall(Ain, Agt)
    assert prec >= 0.8 and rec >= 0.9 and abs(len(Ain) - round(s.score)) <= 1   # 10 true inliers of 100
    rerr, terr = reg.transform_error(T, reg.estimate_rigid_transform(D1, D2, Ain))
    assert rerr < 0.15 and terr < 0.01     # sigma = 1 cm noise on a 15 cm model


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_cfg1_same_selection_on_the_gpu(bunny, seed):
    from clipper_amd import _abi as abi
    T = _truth(seed)
    D1, D2, A, Agt = reg.make_registration_dataset(bunny, seed=seed, T_21=T, **CFG1)
    u0 = np.random.default_rng(seed + 1).random(len(A))
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(D1, D2, A, **INV)
    sr = r.solve(u0)
    for storage in (abi.STORE_F32, abi.STORE_F64):
        g = abi.HipClipper(storage=storage)
        g.score_pairwise_consistency_euclidean(D1, D2, A, **INV)
        sg = g.solve(u0)
        assert sorted(sg.nodes.tolist()) == sorted(sr.nodes.tolist())
        assert abs(sg.score - sr.score) <= 1e-6 * abs(sr.score)
        Ain = g.get_selected_associations()
        rerr, terr = reg.transform_error(T, reg.estimate_rigid_transform(D1, D2, Ain))
        assert rerr < 0.15 and terr < 0.01
        g.close()
