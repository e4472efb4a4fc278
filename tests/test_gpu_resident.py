"""GPU tests of the resident solver (k_resident.hip.h): findDenseClique as ONE launch for problems
whose slices fit on chip. It must return what the streaming launches and the oracle return — the
same selected set, the same number of line-search trials, the objective to rounding — for every
size around the slice and unit edges, both value types, the solver's parameter variants, and it
must step aside (streaming launches) wherever it does not apply."""
import numpy as np
import pytest

from clipper_amd import _abi as abi
from clipper_amd import synth
from oracle import clipper_ref as ref

pytestmark = pytest.mark.gpu

INV = synth.EUCLID_BENCH_PARAMS
CSC = {abi.STORE_F32_CSC: "f32", abi.STORE_F64_CSC: "f64"}


def _solve(p, storage, mode, params=None, inv=None):
    g = abi.HipClipper(params or abi.Params(), storage=storage)
    g.set_resident(mode)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **(inv or INV))
    s = g.solve(p.u0)
    return g, s


def _assert_same(a, b, rel=1e-9, trials=0):
    """Same selected set in the same order, same outer iterations, the objective to rounding. Trial
    counts: equal, except where a decision of the line search sits on a rounding error and the two
    solvers add in different orders (`trials` = the slack; m = 2047 below: the streaming launches
    with their window of 4 take 67 trials where the oracle and the resident solver take 68)."""
    assert a.nodes.tolist() == b.nodes.tolist()
    assert abs(a.n_trials - b.n_trials) <= trials and a.ifinal == b.ifinal
    assert abs(a.score - b.score) <= rel * max(1.0, abs(b.score))


# one workgroup (m <= 512), several units (above), ragged last slices, a single association
@pytest.mark.parametrize("storage", list(CSC), ids=list(CSC.values()))
@pytest.mark.parametrize("m", [1, 2, 63, 64, 65, 100, 129, 300, 511, 513, 777, 1000, 1500, 2047, 2048])
def test_resident_equals_streaming_and_oracle(m, storage):
    p = synth.make_euclidean_problem(m, 0.8 if m > 10 else 0.0, seed=100 + m)
    gr, sr = _solve(p, storage, 0)
    gs, ss = _solve(p, storage, 1)
    # (m ~ 2000 with fp64 values is more than 64 units of LDS: the planner leaves it to the streaming launches)
    assert gr.last_solver == (0 if (storage == abi.STORE_F64_CSC and m > 1500) else 1) and gs.last_solver == 0
    _assert_same(sr, ss, trials=1 if m == 2047 else 0)
    assert np.allclose(sr.u, ss.u, rtol=0, atol=1e-7)   # both stop within tol_u = 1e-8 of the fixed point
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV)
    so = r.solve(p.u0)
    assert sorted(sr.nodes.tolist()) == sorted(so.nodes.tolist())
    assert abs(sr.score - so.score) <= 1e-6 * max(1.0, abs(so.score))
    if storage == abi.STORE_F64_CSC:
        assert abs(sr.n_trials - so.n_trials) <= (1 if m == 2047 else 0) and sr.ifinal == so.ifinal
    gr.close()
    gs.close()


VARIANTS = [
    dict(rescale_u0=False),
    dict(maxiniters=1),
    dict(maxiniters=3, maxoliters=4),
    dict(maxiniters=0, maxoliters=5),
    dict(maxoliters=1),
    dict(maxoliters=0),
    dict(maxlsiters=1, maxoliters=6),
    dict(maxlsiters=2, maxoliters=6),
    dict(beta=0.1),
    dict(beta=0.7, tol_u=1e-4, tol_F=1e-5),
    dict(eps=1e-3),
    dict(rounding=abi.ROUNDING_NONZERO),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=[",".join(f"{k}={v}" for k, v in kw.items()) for kw in VARIANTS])
@pytest.mark.parametrize("m", [200, 1300])
def test_resident_parameter_variants(m, kw):
    p = synth.make_euclidean_problem(m, 0.85, seed=7 * m)
    gr, sr = _solve(p, abi.STORE_F64_CSC, 0, abi.Params(**kw))
    gs, ss = _solve(p, abi.STORE_F64_CSC, 1, abi.Params(**kw))
    assert gr.last_solver == 1 and gs.last_solver == 0
    _assert_same(sr, ss)
    r = ref.RefClipper(ref.Params(**kw))
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV)
    so = r.solve(p.u0)
    assert sorted(sr.nodes.tolist()) == sorted(so.nodes.tolist())
    assert sr.n_trials == so.n_trials and sr.ifinal == so.ifinal
    gr.close()
    gs.close()


def test_resident_line_search_with_rejections():
    """A problem whose line search rejects step sizes (large inlier block, small beta steps): the
    walk of the window must count the reference's trials."""
    p = synth.make_euclidean_problem(900, 0.5, seed=11)
    gr, sr = _solve(p, abi.STORE_F64_CSC, 0, abi.Params(beta=0.5))
    gs, ss = _solve(p, abi.STORE_F64_CSC, 1, abi.Params(beta=0.5))
    assert gr.last_solver == 1
    _assert_same(sr, ss)
    assert sr.n_trials > sr.n_passes - 4 or sr.n_trials >= 1
    gr.close()
    gs.close()


def test_resident_steps_aside():
    p = synth.make_euclidean_problem(400, 0.8, seed=5)
    # a forced line-search window is a knob of the streaming launches
    g = abi.HipClipper(storage=abi.STORE_F32_CSC)
    g.set_window(4)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV)
    s4 = g.solve(p.u0)
    assert g.last_solver == 0
    g.set_window(0)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV)
    s0 = g.solve(p.u0)
    assert g.last_solver == 1
    _assert_same(s0, s4)
    # dense storages and an explicit C never take it
    gd = abi.HipClipper(storage=abi.STORE_F32)
    gd.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV)
    gd.solve(p.u0)
    assert gd.last_solver == 0
    M, C = g.get_affinity_matrix(), g.get_constraint_matrix()
    C2 = C.copy()
    C2[0, 1] = C2[1, 0] = 1.0 - C2[0, 1]
    g.set_matrix_data(M, C2)
    g.solve(p.u0)
    assert g.last_solver == 0
    g.set_matrix_data(M, C)   # C == pattern(M) again: slices, resident
    sm = g.solve(p.u0)
    assert g.last_solver == 1
    _assert_same(sm, s0)
    # too large for the chip; too dense to win (more than 64 units: the streaming launches)
    for m, rho in ((2500, 0.9), (2000, 0.1)):
        pl = synth.make_euclidean_problem(m, rho, seed=6)
        gl = abi.HipClipper(storage=abi.STORE_F32_CSC)
        gl.score_pairwise_consistency_euclidean(pl.D1, pl.D2, pl.A, **INV)
        gl.solve(pl.u0)
        assert gl.last_solver == 0
        gl.close()
    for x in (g, gd):
        x.close()


def test_resident_context_reuse():
    """One context, problems of changing size and repeated solves: the epochs of the exchange flags
    keep counting, buffers grow, every result equals a fresh streaming context's."""
    g = abi.HipClipper(storage=abi.STORE_F32_CSC)
    for m, seed in [(1200, 1), (150, 2), (1200, 1), (2000, 3), (700, 4), (700, 4)]:
        p = synth.make_euclidean_problem(m, 0.85, seed=seed)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV)
        a = g.solve(p.u0)
        b = g.solve(p.u0)
        assert g.last_solver == 1
        assert a.nodes.tolist() == b.nodes.tolist() and a.score == b.score and np.array_equal(a.u, b.u)
        gs, ss = _solve(p, abi.STORE_F32_CSC, 1)
        _assert_same(a, ss)
        gs.close()
    g.close()


def test_resident_pointnormal_and_sparse_setter():
    p = synth.make_pointnormal_problem(900, 0.9, seed=21)
    inv = p.meta["invariant"]
    res = []
    for mode in (0, 1):
        g = abi.HipClipper(storage=abi.STORE_F32_CSC)
        g.set_resident(mode)
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **inv)
        res.append((g, g.solve(p.u0)))
    assert res[0][0].last_solver == 1 and res[1][0].last_solver == 0
    _assert_same(res[0][1], res[1][1])
    # the same matrix through setSparseMatrixData (lists -> slices): resident again, same answer
    M = res[0][0].get_affinity_matrix()
    iu = np.triu_indices(M.shape[0], 1)
    keep = M[iu] != 0
    rows, cols, vals = iu[0][keep], iu[1][keep], M[iu][keep]
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    colptr = np.zeros(M.shape[0] + 1, dtype=np.int64)
    np.add.at(colptr, cols + 1, 1)
    colptr = np.cumsum(colptr)
    g2 = abi.HipClipper(storage=abi.STORE_F32_CSC)
    g2.set_sparse_matrix_data(M.shape[0], colptr, rows.astype(np.int32), vals,
                              colptr, rows.astype(np.int32), np.ones_like(vals))
    s2 = g2.solve(p.u0)
    assert g2.last_solver == 1
    _assert_same(s2, res[0][1])
    for g, _ in res:
        g.close()
    g2.close()


def test_resident_fallbacks(monkeypatch):
    """The two ways out of the resident solver: a one-XCD launch whose units are not all claimed is
    repeated in the placement-free mode (still one launch), a time-out hands the solve to the
    streaming launches; the answer is the same every way."""
    p = synth.make_euclidean_problem(1000, 0.85, seed=31)
    g0, s0 = _solve(p, abi.STORE_F32_CSC, 1)
    monkeypatch.setenv("CLIPPER_HIP_RESIDENT_HOME", "11")          # no such XCD
    g1, s1 = _solve(p, abi.STORE_F32_CSC, 0)
    assert g1.last_solver == 1
    s1b = g1.solve(p.u0)                                            # the mode is not tried again
    monkeypatch.delenv("CLIPPER_HIP_RESIDENT_HOME")
    monkeypatch.setenv("CLIPPER_HIP_RESIDENT_XCD", "0")             # placement-free from the start
    g2, s2 = _solve(p, abi.STORE_F32_CSC, 0)
    assert g2.last_solver == 1
    monkeypatch.delenv("CLIPPER_HIP_RESIDENT_XCD")
    monkeypatch.setenv("CLIPPER_HIP_RESIDENT_TIMEOUT_TICKS", "-1")  # every wait is "late"
    g3, s3 = _solve(p, abi.STORE_F32_CSC, 0)
    assert g3.last_solver == 0
    monkeypatch.delenv("CLIPPER_HIP_RESIDENT_TIMEOUT_TICKS")
    s3b = g3.solve(p.u0)                                            # ... and stays with them for this matrix
    assert g3.last_solver == 0
    g3.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **INV) # a new build: resident again
    s3c = g3.solve(p.u0)
    assert g3.last_solver == 1
    for s in (s1, s1b, s2, s3, s3b, s3c):
        _assert_same(s, s0)
    for g in (g0, g1, g2, g3):
        g.close()
