"""GPU parity tests: the HIP path (through the C ABI, clipper_amd/_abi.py) against the CPU
oracle on identical inputs. Run on a real MI355X:  pytest tests -m gpu

Bars (BASELINE.json north_star / SURVEY.md 8c):
  * selected association set bit-identical (compared as produced, i.e. same order);
  * objective within 1e-6 relative of the fp64 oracle (we hold 1e-9 for fp64 storage);
  * affinity: identical non-zero pattern; values exact to fp32 rounding (fp32 storage) or
    within 4 ulp (fp64 storage, EuclideanDistance: only device exp vs libm exp differs).
    PointNormalDistance: 1e-12 relative — a 1-ulp difference between device acos and libm
    acos is amplified by the cancellation in |alpha1 - alpha2| and by 1/sign^2 in the exponent
    (0.35 * 1e-16 / 0.1^2 ~ 4e-15 absolute in the exponent).
"""
import numpy as np
import pytest

from clipper_amd import _abi as abi
from clipper_amd import synth
from oracle import clipper_ref as ref

pytestmark = pytest.mark.gpu

STORAGES = [abi.STORE_F32, abi.STORE_F64, abi.STORE_F32_CSC, abi.STORE_F64_CSC]
F64S = (abi.STORE_F64, abi.STORE_F64_CSC)   # fp64 values: the parity-exact modes
REL_SCORE = 1e-6


def _pair(storage=abi.STORE_F32, **pkw):
    return abi.HipClipper(abi.Params(**pkw), storage=storage), ref.RefClipper(ref.Params(**pkw))


def _check_affinity(g, r, storage, f64_rel=4 * 2.3e-16):
    Mg, Mr = g.get_affinity_matrix(), r.get_affinity_matrix()
    assert Mg.shape == Mr.shape
    assert np.array_equal(Mg != 0, Mr != 0), "non-zero pattern differs"
    assert np.array_equal(Mg, Mg.T)
    if storage not in F64S:
        assert np.array_equal(Mg.astype(np.float32), Mr.astype(np.float32)) or \
            np.max(np.abs(Mg - Mr.astype(np.float32).astype(np.float64))) <= 1.2e-7
    else:
        nz = Mr != 0
        assert np.max(np.abs(Mg[nz] - Mr[nz]) / Mr[nz], initial=0.0) <= f64_rel
    Cg, Cr = g.get_constraint_matrix(), r.get_constraint_matrix()
    assert np.array_equal(Cg, Cr)


def _check_solution(sg, sr, exact_counts=False, ordered=True, rel=REL_SCORE, same_ifinal=True):
    """ordered=False: compare the selected nodes as a set — for inputs whose solution has
    structurally tied entries of u (identical rows), where the heap order of utils.cpp:33-55
    depends on last-bit rounding. rel / same_ifinal are relaxed only for deliberately
    ill-conditioned cases whose homotopy runs for thousands of passes with d ~ 1e9 (there
    F = u'Mu - d u'Cb u amplifies last-bit differences of u by d)."""
    if ordered:
        assert sg.nodes.tolist() == sr.nodes.tolist(), "selected node list differs"
    assert sorted(sg.nodes.tolist()) == sorted(sr.nodes.tolist()), "selected node set differs"
    assert abs(sg.score - sr.score) <= rel * max(1.0, abs(sr.score))
    if same_ifinal:
        assert sg.ifinal == sr.ifinal
    if exact_counts:
        assert sg.n_trials == sr.n_trials


# ------------------------------------------------------------------------------------------
# the reference's own golden vectors, on the GPU path
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("storage", STORAGES)
def test_golden_affinity_matches_reference_Mtrue(golden, storage):
    g = golden["affinity_test"]
    model, data, Mtrue = np.array(g["model"]), np.array(g["data"]), np.array(g["Mtrue"])
    c = abi.HipClipper(storage=storage)
    c.score_pairwise_consistency_euclidean(model, data)       # all-to-all (clipper.cpp:24)
    assert np.array_equal(c.get_initial_associations(), ref.create_all_to_all(4, 3))
    M, Cm = c.get_affinity_matrix(), c.get_constraint_matrix()
    assert np.array_equal(np.diag(M), np.ones(12))             # affinity_test.cpp:83
    assert np.array_equal(M, M.T) and np.array_equal(Cm, Cm.T)  # :86-87
    assert np.array_equal(M, Cm)                                # :91
    assert np.array_equal(M, Mtrue)                             # :93-107 exact


@pytest.mark.parametrize("storage", STORAGES)
def test_golden_solve_known_answer(golden, storage):
    g = golden["affinity_test"]
    model, data = np.array(g["model"]), np.array(g["data"])
    c, r = _pair(storage)
    c.score_pairwise_consistency_euclidean(model, data)
    r.score_pairwise_consistency_euclidean(model, data)
    rng = np.random.default_rng(2024)
    u0s = [np.ones(12) / np.sqrt(12)] + [rng.random(12) for _ in range(12)]
    for k, u0 in enumerate(u0s):
        sg, sr = c.solve(u0), r.solve(u0)
        # the selected clique {0,4,8} has exactly equal affinities, so the three entries of u
        # converge to the same value: their heap order (utils.cpp:33-55) rests on last-bit
        # rounding — compare as sets
        _check_solution(sg, sr, exact_counts=True, ordered=False)
        assert np.allclose(sg.u, sr.u, rtol=0, atol=1e-9)
        Ain = c.get_selected_associations()
        Aref = r.get_selected_associations()
        assert sorted(map(tuple, Ain)) == sorted(map(tuple, Aref))
        if k == 0:   # clipper_test.cpp:62-66
            assert Ain.shape[0] == 3 and np.all(Ain[:, 0] == Ain[:, 1])
            assert sorted(sg.nodes.tolist()) == g["expected_inlier_nodes"]


@pytest.mark.parametrize("storage", STORAGES)
def test_golden_get_set_round_trip(golden, storage):
    # clipper_test.cpp:115-133
    g = golden["affinity_test"]
    model, data, Mtrue = np.array(g["model"]), np.array(g["data"]), np.array(g["Mtrue"])
    c = abi.HipClipper(storage=storage)
    c.score_pairwise_consistency_euclidean(model, data)
    M, Cm = c.get_affinity_matrix(), c.get_constraint_matrix()
    c2 = abi.HipClipper(storage=storage)
    c2.set_matrix_data(M, Cm)
    assert np.array_equal(c2.get_affinity_matrix(), Mtrue)
    assert np.array_equal(c2.get_constraint_matrix(), Mtrue)
    u0 = np.ones(12) / np.sqrt(12)
    s1, s2 = c.solve(u0), c2.solve(u0)
    assert s1.nodes.tolist() == s2.nodes.tolist() and s1.score == s2.score


def test_golden_planecloud_pointnormal(golden):
    g = golden["planecloud"]
    D1, D2, inv = np.array(g["D1"]), np.array(g["D2"]), g["invariant"]
    for storage in STORAGES:
        c, r = _pair(storage)
        c.score_pairwise_consistency_pointnormal(D1, D2, (), **inv)
        r.score_pairwise_consistency_pointnormal(D1, D2, (), **inv)
        _check_affinity(c, r, storage, f64_rel=1e-12)
        want = sorted(map(tuple, g["Agt_zero_based"]))
        found = 0
        rng = np.random.default_rng(2024)
        for u0 in [np.ones(16) / 4] + [rng.random(16) for _ in range(10)]:
            sg, sr = c.solve(u0), r.solve(u0)
            _check_solution(sg, sr)
            found += sorted(map(tuple, c.get_selected_associations().tolist())) == want
        assert found >= 1   # ex3_planecloud.m:90-98


@pytest.mark.parametrize("storage", STORAGES)
def test_dsd_20x20_weighted_matrix(golden, storage):
    # sdp_test.cpp:38-56: setMatrixData(M, C=(M>0)); solve()
    M = np.array(golden["dsd_test_20x20"]["M"])
    Cm = (M > 0).astype(float)
    c, r = _pair(storage)
    c.set_matrix_data(M, Cm)
    r.set_matrix_data(M, Cm)
    for u0 in [np.ones(20) / np.sqrt(20), np.random.default_rng(3).random(20)]:
        _check_solution(c.solve(u0), r.solve(u0))


# ------------------------------------------------------------------------------------------
# seeded synthetic problems: affinity, mat-vec and solve against the oracle
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("storage", STORAGES)
@pytest.mark.parametrize("m,rho,seed", [(1000, 0.90, 12345), (1037, 0.80, 7), (2000, 0.90, 99)])
def test_euclidean_parity(storage, m, rho, seed):
    p = synth.make_euclidean_problem(m, rho, seed)
    c, r = _pair(storage)
    c.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    _check_affinity(c, r, storage)
    # one mat-vec pass in isolation
    x = np.random.default_rng(seed + 5).random(m)
    (aM, aC), (rM, rC) = c.matvec(x), r.matvec(x)
    tol = 1e-6 if storage not in F64S else 1e-12
    assert np.allclose(aM, rM, rtol=tol, atol=tol)
    assert np.allclose(aC, rC, rtol=1e-12, atol=1e-12)
    sg, sr = c.solve(p.u0), r.solve(p.u0)
    _check_solution(sg, sr, exact_counts=(storage in F64S))
    assert np.array_equal(c.get_selected_associations(), r.get_selected_associations())
    if storage in F64S:
        assert abs(sg.score - sr.score) <= 1e-9 * abs(sr.score)
        assert np.allclose(sg.u, sr.u, rtol=0, atol=1e-9)
    # a pass evaluates a window of line-search trials: never more passes than trials (+ the 2
    # initial passes + one pair-mode pass per penalty update), no fewer than trials / window
    assert (sg.n_trials + 7) // 8 + 2 <= sg.n_passes <= sg.n_trials + 3 + sg.ifinal
    prec, rec = synth.precision_recall(c.get_selected_associations(), p.Agt)
    assert prec >= 0.95


@pytest.mark.parametrize("storage", STORAGES)
def test_pointnormal_parity(storage):
    p = synth.make_pointnormal_problem(600, 0.8, seed=11)
    c, r = _pair(storage)
    c.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **p.meta["invariant"])
    r.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **p.meta["invariant"])
    _check_affinity(c, r, storage, f64_rel=1e-12)
    _check_solution(c.solve(p.u0), r.solve(p.u0))


@pytest.mark.parametrize("d", [2, 3, 5])
def test_euclidean_any_dimension(d):
    rng = np.random.default_rng(d)
    n = 60
    D1 = rng.random((d, n))
    D2 = D1 + rng.normal(0, 0.005, size=(d, n))
    A = np.stack([rng.integers(0, n, 300), rng.integers(0, n, 300)], axis=1).astype(np.int32)
    c, r = _pair(abi.STORE_F64)
    kw = dict(sigma=0.02, epsilon=0.08, mindist=0.05)   # exercises the mindist branch too
    c.score_pairwise_consistency_euclidean(D1, D2, A, **kw)
    r.score_pairwise_consistency_euclidean(D1, D2, A, **kw)
    _check_affinity(c, r, abi.STORE_F64)
    u0 = rng.random(300)
    # random repeated endpoints + mindist make this a long (~3000 passes, d ~ 1e9) homotopy
    sg, sr = c.solve(u0), r.solve(u0)
    print(f"d={d}: rel dscore {abs(sg.score - sr.score) / max(1.0, abs(sr.score)):.2e}, ifinal {sg.ifinal}/{sr.ifinal}, "
          f"trials {sg.n_trials}/{sr.n_trials}")
    _check_solution(sg, sr, ordered=False, rel=REL_SCORE, same_ifinal=False)


def test_duplicate_and_repeated_associations():
    # distinctness constraint (clipper.cpp:35-38): shared endpoints give M = C = 0
    rng = np.random.default_rng(5)
    D1 = rng.random((3, 8))
    D2 = D1.copy()
    A = np.array([[0, 0], [0, 1], [1, 1], [2, 2], [2, 2], [3, 3], [4, 3]], dtype=np.int32)
    c, r = _pair(abi.STORE_F64)
    c.score_pairwise_consistency_euclidean(D1, D2, A)
    r.score_pairwise_consistency_euclidean(D1, D2, A)
    _check_affinity(c, r, abi.STORE_F64)
    M = c.get_affinity_matrix()
    assert M[0, 1] == 0 and M[3, 4] == 0 and M[5, 6] == 0
    u0 = np.full(7, 1 / np.sqrt(7))
    # u has exact structural ties, and the repeated endpoints make the homotopy long (d ~ 1e9)
    sg, sr = c.solve(u0), r.solve(u0)
    print(f"rel dscore {abs(sg.score - sr.score) / max(1.0, abs(sr.score)):.2e}, ifinal {sg.ifinal}/{sr.ifinal}, "
          f"trials {sg.n_trials}/{sr.n_trials}")
    _check_solution(sg, sr, ordered=False, rel=REL_SCORE, same_ifinal=False)


@pytest.mark.parametrize("storage", STORAGES)
def test_explicit_constraint_matrix(storage):
    # setMatrixData with C != pattern(M): second dense matrix on device
    rng = np.random.default_rng(17)
    n = 150
    M = np.triu(rng.random((n, n)) * (rng.random((n, n)) < 0.3), 1)
    Cm = np.triu((rng.random((n, n)) < 0.6).astype(float), 1)   # unrelated to pattern(M)
    M, Cm = M + M.T + np.eye(n), Cm + Cm.T + np.eye(n)
    c, r = _pair(storage)
    c.set_matrix_data(M, Cm)
    r.set_matrix_data(M, Cm)
    Mg = c.get_affinity_matrix()
    assert np.allclose(Mg, r.get_affinity_matrix(), rtol=1e-7, atol=0)
    assert np.array_equal(c.get_constraint_matrix(), r.get_constraint_matrix())
    x = rng.random(n)
    (aM, aC), (rM, rC) = c.matvec(x), r.matvec(x)
    assert np.allclose(aM, rM, rtol=1e-6) and np.allclose(aC, rC, rtol=1e-12)
    u0 = rng.random(n)
    _check_solution(c.solve(u0), r.solve(u0))


def test_sparse_setter_matches_dense_setter():
    import scipy.sparse as sp
    rng = np.random.default_rng(23)
    n = 200
    Mu = np.triu(rng.random((n, n)) * (rng.random((n, n)) < 0.2), 1)
    Cu = (Mu != 0).astype(float)
    Ms, Cs = sp.csc_matrix(Mu), sp.csc_matrix(Cu)
    c1, c2 = abi.HipClipper(storage=abi.STORE_F64), abi.HipClipper(storage=abi.STORE_F64)
    c1.set_matrix_data(Mu + Mu.T + np.eye(n), Cu + Cu.T + np.eye(n))
    c2.set_sparse_matrix_data(n, Ms.indptr, Ms.indices, Ms.data, Cs.indptr, Cs.indices, Cs.data)
    assert np.array_equal(c1.get_affinity_matrix(), c2.get_affinity_matrix())
    assert np.array_equal(c1.get_constraint_matrix(), c2.get_constraint_matrix())
    u0 = rng.random(n)
    s1, s2 = c1.solve(u0), c2.solve(u0)
    assert s1.nodes.tolist() == s2.nodes.tolist() and s1.score == s2.score


@pytest.mark.parametrize("kw", [
    dict(rescale_u0=0),
    dict(rounding=abi.ROUNDING_NONZERO),
    dict(maxiniters=3),
    dict(maxlsiters=1),
    dict(maxoliters=1),
    dict(maxoliters=0),
    dict(beta=0.5, tol_u=1e-6, tol_F=1e-7),
])
def test_solver_parameter_variants(kw):
    p = synth.make_euclidean_problem(500, 0.85, seed=31)
    c, r = _pair(abi.STORE_F64, **kw)
    c.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sg, sr = c.solve(p.u0), r.solve(p.u0)
    _check_solution(sg, sr, exact_counts=True)


def test_error_paths_fail_loudly():
    c = abi.HipClipper()
    with pytest.raises(RuntimeError):          # no matrix yet
        c.solve(np.ones(4))
    D = np.random.default_rng(0).random((3, 5))
    with pytest.raises(RuntimeError):          # association index out of range
        c.score_pairwise_consistency_euclidean(D, D, np.array([[0, 7]], dtype=np.int32))
    c.score_pairwise_consistency_euclidean(D, D)
    c.params.rounding = 7
    with pytest.raises(RuntimeError):          # unknown rounding mode
        c.solve(np.ones(25))


# ------------------------------------------------------------------------------------------
# column-sharded protocol with several logical shards on the one GPU of the test box
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_in_process_shards_on_one_device(nshards):
    p = synth.make_euclidean_problem(1500, 0.9, seed=77)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sr = r.solve(p.u0)
    for storage in STORAGES:
        g = abi.HipClipper(storage=storage, group=[0] * nshards)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        _check_affinity(g, r, storage)
        x = np.random.default_rng(1).random(1500)
        (aM, aC), (rM, rC) = g.matvec(x), r.matvec(x)
        assert np.allclose(aM, rM, rtol=1e-6) and np.allclose(aC, rC, rtol=1e-12)
        _check_solution(g.solve(p.u0), sr)


def test_rccl_exchange_with_a_one_rank_world(monkeypatch):
    # CLIPPER_HIP_FORCE_RCCL=1 routes the per-pass exchange of a 1-rank world through
    # ncclAllGather: exercises dlopen(librccl), ncclGetUniqueId, ncclCommInitRank and the
    # in-place all-gather on the one GPU of the test box.
    monkeypatch.setenv("CLIPPER_HIP_FORCE_RCCL", "1")
    p = synth.make_euclidean_problem(1200, 0.9, seed=3)
    g = abi.HipClipper(storage=abi.STORE_F32, rank=0, world=1)
    g.comm_init(g.unique_id())
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sg = g.solve(p.u0)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    _check_solution(sg, r.solve(p.u0))
    monkeypatch.delenv("CLIPPER_HIP_FORCE_RCCL")
    s1 = abi.HipClipper(storage=abi.STORE_F32)
    s1.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    ss = s1.solve(p.u0)
    assert ss.nodes.tolist() == sg.nodes.tolist() and ss.score == sg.score   # bit-identical


# ------------------------------------------------------------------------------------------
# BASELINE.json full size (m = 10k): oracle comparison + size-independent properties
# ------------------------------------------------------------------------------------------

def test_full_size_10k_properties_and_parity():
    m = 10000
    p = synth.make_euclidean_problem(m, 0.95, seed=12345)
    c = abi.HipClipper(storage=abi.STORE_F32)
    c.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    rng = np.random.default_rng(0)
    x, y = rng.random(m), rng.random(m)
    (ax, bx), (ay, by) = c.matvec(x), c.matvec(y)
    (axy, bxy) = c.matvec(2.0 * x - 0.5 * y)
    # linearity of the fused pass
    assert np.allclose(axy, 2.0 * ax - 0.5 * ay, rtol=1e-11, atol=1e-9)
    assert np.allclose(bxy, 2.0 * bx - 0.5 * by, rtol=1e-11, atol=1e-9)
    # symmetry: x' (M y) == y' (M x)
    assert abs(x @ ay - y @ ax) <= 1e-10 * abs(x @ ay)
    assert abs(x @ by - y @ bx) <= 1e-10 * abs(x @ by)
    # unit vectors read out columns: zero diagonal, entries in [0,1], C = pattern(M)
    for k in (0, 4999, m - 1):
        e = np.zeros(m)
        e[k] = 1.0
        col, pat = c.matvec(e)
        assert col[k] == 0 and np.all((col >= 0) & (col <= 1))
        assert np.array_equal(pat, (col != 0).astype(float))
    # determinism: two solves are bit-identical
    s1 = c.solve(p.u0)
    s2 = c.solve(p.u0)
    assert s1.nodes.tolist() == s2.nodes.tolist() and s1.score == s2.score
    assert np.array_equal(s1.u, s2.u)
    # parity with the oracle at full size
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sr = r.solve(p.u0)
    _check_solution(s1, sr)
    prec, rec = synth.precision_recall(c.get_selected_associations(), p.Agt)
    assert prec >= 0.95
    # the objective recomputed from the returned u on the device matrix: u'(M+I)u
    a, _ = c.matvec(s1.u)
    assert abs((s1.u @ a + s1.u @ s1.u) - sr.u @ (r.matvec(sr.u)[0] + sr.u)) <= 1e-6 * sr.score


# ------------------------------------------------------------------------------------------
# line-search window: V candidates per pass over M (CLIPPER_HIP_WINDOW = 1 | 4 | 6). The window
# only changes HOW MANY passes evaluate the reference's trial sequence, never which trials are
# accepted: every window size must reproduce the oracle's trial count and selection.
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("m,rho,seed", [(600, 0.9, 1), (1037, 0.8, 7), (3000, 0.9, 5), (5200, 0.95, 11)])
def test_window_sizes_agree(monkeypatch, m, rho, seed):
    p = synth.make_euclidean_problem(m, rho, seed=seed)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sr = r.solve(p.u0)
    sols = {}
    for V in (1, 4, 6, 8):
        monkeypatch.setenv("CLIPPER_HIP_WINDOW", str(V))
        for storage in STORAGES:
            g = abi.HipClipper(storage=storage)
            g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
            for rep in range(2):   # repeated solves on one context: counters re-arm themselves
                s = g.solve(p.u0)
                sols[(V, storage, rep)] = s
                _check_solution(s, sr, exact_counts=(storage in F64S))
            g.close()
    monkeypatch.delenv("CLIPPER_HIP_WINDOW")
    for storage in STORAGES:
        base = sols[(1, storage, 0)]
        # window 1: one trial per pass, + the 2 initial passes + one pair pass per penalty update
        assert base.n_trials + 2 <= base.n_passes <= base.n_trials + 3 + base.ifinal
        for V in (1, 4, 6, 8):
            for rep in range(2):
                s = sols[(V, storage, rep)]
                assert s.nodes.tolist() == base.nodes.tolist(), (V, storage, rep)
                assert s.ifinal == base.ifinal and s.n_trials == base.n_trials, (V, storage, rep)
                # windows > 1 form g_v = (M + d*C) x_v in one product, window 1 adds a + d*b:
                # rounding-level differences only
                assert abs(s.score - base.score) <= 1e-11 * abs(base.score)
                assert np.allclose(s.u, base.u, rtol=0, atol=1e-11)
                assert s.n_passes <= base.n_passes
                assert np.array_equal(s.u, sols[(V, storage, 0)].u)   # run-to-run bit-identical


@pytest.mark.parametrize("V", [1, 4, 6, 8])
def test_window_with_line_search_limits(monkeypatch, V):
    # maxlsiters cuts the backtracking inside a window (the last allowed trial is accepted
    # unconditionally, clipper.cpp:234), beta != 1/4 changes the step sizes of the window
    monkeypatch.setenv("CLIPPER_HIP_WINDOW", str(V))
    p = synth.make_euclidean_problem(1500, 0.93, seed=8)
    for kw in (dict(maxlsiters=1), dict(maxlsiters=2), dict(maxlsiters=3), dict(maxlsiters=5),
               dict(beta=0.5), dict(beta=0.1, maxlsiters=7), dict(maxiniters=3), dict(maxoliters=2),
               dict(maxiniters=0), dict(maxoliters=0), dict(maxiniters=1, maxoliters=1),
               dict(rescale_u0=False), dict(rounding=abi.ROUNDING_NONZERO)):
        g, r = _pair(abi.STORE_F64, **kw)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        _check_solution(g.solve(p.u0), r.solve(p.u0), exact_counts=True)
        g.close()


@pytest.mark.parametrize("V", [1, 4, 6, 8])
def test_window_sharded(monkeypatch, V):
    # column-sharded M: k_gemv + k_reduce_pass, exchange of the [V+1][W] blocks, k_tail
    monkeypatch.setenv("CLIPPER_HIP_WINDOW", str(V))
    p = synth.make_euclidean_problem(1500, 0.9, seed=77)
    g1 = abi.HipClipper(storage=abi.STORE_F32)
    g1.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s1 = g1.solve(p.u0)
    g = abi.HipClipper(storage=abi.STORE_F32, group=[0] * 3)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    for rep in range(2):
        s = g.solve(p.u0)
        assert s.nodes.tolist() == s1.nodes.tolist() and s.n_passes == s1.n_passes
        assert s.n_trials == s1.n_trials
        assert abs(s.score - s1.score) <= 1e-12 * abs(s1.score)


# ------------------------------------------------------------------------------------------
# affinity fill variants: the symmetric tile kernel (one shard, fp32: upper block triangle +
# transposed copies), the compacting strip kernels and the plain kernels must write the same bits
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("m", [1, 2, 63, 127, 128, 129, 255, 257, 1037, 2049])
def test_fill_kernels_agree_bitwise(monkeypatch, m):
    rho = 0.8 if m > 20 else 0.0
    p = synth.make_euclidean_problem(m, rho, seed=100 + m)
    mats = {}
    for mode in ("sym", "strip", "plain"):
        if mode == "sym":
            monkeypatch.delenv("CLIPPER_HIP_AFFINITY", raising=False)
        else:
            monkeypatch.setenv("CLIPPER_HIP_AFFINITY", mode)
        g = abi.HipClipper(storage=abi.STORE_F32)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        mats[mode] = g.get_affinity_matrix()
        g.close()
    monkeypatch.delenv("CLIPPER_HIP_AFFINITY", raising=False)
    assert np.array_equal(mats["sym"], mats["strip"])
    assert np.array_equal(mats["sym"], mats["plain"])
    assert np.array_equal(mats["sym"], mats["sym"].T)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    Mr = r.get_affinity_matrix()
    assert np.array_equal(mats["sym"] != 0, Mr != 0)
    assert np.max(np.abs(mats["sym"] - Mr.astype(np.float32).astype(np.float64)), initial=0.0) <= 1.2e-7


def test_fill_kernels_agree_bitwise_pointnormal(monkeypatch):
    p = synth.make_pointnormal_problem(1500, 0.85, seed=5)
    inv = p.meta["invariant"]
    mats = {}
    for mode in ("sym", "strip", "plain"):
        if mode == "sym":
            monkeypatch.delenv("CLIPPER_HIP_AFFINITY", raising=False)
        else:
            monkeypatch.setenv("CLIPPER_HIP_AFFINITY", mode)
        g = abi.HipClipper(storage=abi.STORE_F32)
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **inv)
        mats[mode] = g.get_affinity_matrix()
        g.close()
    monkeypatch.delenv("CLIPPER_HIP_AFFINITY", raising=False)
    assert np.array_equal(mats["sym"], mats["strip"])
    assert np.array_equal(mats["sym"], mats["plain"])


# ------------------------------------------------------------------------------------------
# Rounding::DSD — exact densest subgraph (Goldberg, src/dsd.cpp) of the graph induced by nnz(u):
# the sub-matrix is gathered from the device, the flow algorithm runs on the host
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("storage", STORAGES)
def test_dsd_golden_through_the_abi(golden, storage):
    g = golden["dsd_test_20x20"]
    M = np.array(g["M"])
    C = (M != 0).astype(float)
    c = abi.HipClipper(storage=storage)
    c.set_matrix_data(M, C)
    assert c.densest_subgraph().tolist() == g["dsd_nodes"]                          # dsd_test.cpp:14-43
    assert c.densest_subgraph([0, 1, 3, 5, 7, 12, 14, 15, 19]).tolist() == g["dsd_nodes"]  # :47-80
    for nshards in (2, 3):
        cs = abi.HipClipper(storage=storage, group=[0] * nshards)
        cs.set_matrix_data(M, C)
        assert cs.densest_subgraph().tolist() == g["dsd_nodes"]


@pytest.mark.parametrize("m,rho,seed", [(120, 0.8, 2), (200, 0.85, 4), (260, 0.9, 6)])
def test_dsd_rounding_matches_the_oracle(m, rho, seed):
    p = synth.make_euclidean_problem(m, rho, seed=seed)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    Mr = r.get_affinity_matrix()
    Mup = np.triu(Mr, 1)
    sr = ref.numpy_solve(Mup, (Mup != 0).astype(float), p.u0, ref.Params(rounding=ref.ROUNDING_DSD))
    g = abi.HipClipper(abi.Params(rounding=abi.ROUNDING_DSD), storage=abi.STORE_F64)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sg = g.solve(p.u0)
    assert sg.nodes.tolist() == sr.nodes.tolist() and len(sg.nodes) >= 3
    assert abs(sg.score - sr.score) <= 1e-9 * abs(sr.score)
    # the exact rounding never returns a sparser subgraph than the heuristic one
    gh = abi.HipClipper(abi.Params(rounding=abi.ROUNDING_DSD_HEU), storage=abi.STORE_F64)
    gh.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sh = gh.solve(p.u0)

    def density(nodes):
        sub = Mup[np.ix_(nodes, nodes)]
        return (sub + sub.T).sum() / 2.0 / max(1, len(nodes))
    assert density(sg.nodes.tolist()) >= density(sh.nodes.tolist()) - 1e-12

