"""The live sub-problem (include/clipper_hip.h: clipper_hip_set_subproblem; csrc/k_subproblem.hip.h): once a row view
exists and the penalty d of clipper.cpp:268-280 is large, no association outside a small set S can get a positive
gradient again (a bound on clipper.cpp:238-241 the decision checks for every candidate it plans), and the solve continues
on the associations of S as a problem of its own. What is left out is provably zero: every result must be the oracle's,
and the three routes — no views, views, views + sub-problem — must agree with each other."""
import os
import subprocess
import sys

import numpy as np
import pytest

from clipper_amd import _abi as abi
from clipper_amd import synth
from oracle import clipper_ref as ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same_list(s, sr):
    """the oracle's selected list (utils.cpp:33-55: descending by (value, index)); entries whose u agree to rounding
    may have swapped places — fp32 values move the last digits of u (tests/test_gpu_configs.py)"""
    na, nb = np.asarray(s.nodes), np.asarray(sr.nodes)
    assert na.size == nb.size and sorted(na.tolist()) == sorted(nb.tolist())
    ua, ub = np.asarray(s.u), np.asarray(sr.u)
    tol = max(1e-9, 4 * float(np.max(np.abs(ua - ub))))
    for k in np.nonzero(na != nb)[0]:
        assert abs(ua[na[k]] - ua[nb[k]]) < tol and abs(ub[na[k]] - ub[nb[k]]) < tol, (int(k), int(na[k]), int(nb[k]))


_ORACLE = {}


def _oracle(p, pointnormal=False, params=None, **prm):
    key = (p.meta["kind"], p.meta["m"], p.meta["rho"], p.meta["seed"], None if params is None else bytes(params))
    if key not in _ORACLE:  # (one CPU solve per problem: the storages share it)
        r = ref.RefClipper(params) if params is not None else ref.RefClipper()
        if pointnormal:
            r.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **prm)
        else:
            r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **prm)
        _ORACLE[key] = r.solve(p.u0)
    return _ORACLE[key]


def _gpu(p, storage, route, pointnormal=False, params=None, **prm):
    g = abi.HipClipper(storage=storage)
    g.set_row_view(1 if route == "noviews" else 0)
    g.set_subproblem({"sub": 0, "sub_slices": 2}.get(route, 1))   # (2: the sub-problem always as slices, never a dense store)
    if pointnormal:
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **prm)
    else:
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **prm)
    if params is not None:
        g.params = params
    s = g.solve(p.u0)
    return g, s, g.view_stats()


@pytest.mark.parametrize("storage", [abi.STORE_F32_CSC, abi.STORE_F64_CSC])
@pytest.mark.parametrize("m,rho", [(24000, 0.95), (16000, 0.9), (13000, 0.85)])
def test_the_sub_problem_does_not_change_the_result(storage, m, rho):
    p = synth.make_euclidean_problem(m, rho, seed=4100 + m)
    sr = _oracle(p, **synth.EUCLID_BENCH_PARAMS)
    g0, s0, st0 = _gpu(p, storage, "views", **synth.EUCLID_BENCH_PARAMS)
    g1, s1, st1 = _gpu(p, storage, "sub", **synth.EUCLID_BENCH_PARAMS)
    assert st0.sub_entries == 0 and st0.sub_passes == 0
    assert st1.sub_entries >= 1 and st1.sub_passes > 0, "the solve never ran on the sub-problem"
    assert st1.sub_rows >= st1.rows > 0 and st1.sub_rows <= 2 * st1.rows + 1024
    for s in (s0, s1):
        _same_list(s, sr)
        assert abs(s.score - sr.score) <= 1e-6 * abs(sr.score)
        assert s.ifinal == sr.ifinal
    # outside S the solution is zero on every route; inside it the same point to rounding
    # (u itself: the iteration stops at |du| < 1e-8, and at d ~ 1e4 two orders of the same sums end 1e-8 apart)
    assert abs(s1.score - s0.score) <= 1e-10 * abs(s0.score)
    assert np.allclose(s1.u, s0.u, rtol=0, atol=1e-7)
    assert np.allclose(s1.u, sr.u, rtol=0, atol=1e-7)
    # (trial counts: accept tests that sit on rounding errors — DESIGN.md section 5 —: the route with views alone is as
    # far from the oracle's count as this one; the two routes within a few of each other)
    assert abs(s1.n_trials - s0.n_trials) <= max(3, s0.n_trials // 20), (s1.n_trials, s0.n_trials, sr.n_trials)
    assert abs(s1.n_trials - sr.n_trials) <= max(4, sr.n_trials // 10), (s1.n_trials, s0.n_trials, sr.n_trials)
    # a sub-problem that is mostly non-zero (here: the inliers are consistent with each other) is kept as a dense fp32
    # store where the values are fp32; forced to slices it must give the same answer
    # (whether it is depends on the problem: here the live rows still hold many outliers when the view is built)
    assert st1.sub_dense in (0, 1) and (storage == abi.STORE_F32_CSC or st1.sub_dense == 0)
    g2, s2b, st2 = _gpu(p, storage, "sub_slices", **synth.EUCLID_BENCH_PARAMS)
    assert st2.sub_entries >= 1 and st2.sub_dense == 0
    _same_list(s2b, sr)
    assert s2b.ifinal == sr.ifinal and abs(s2b.score - s1.score) <= 1e-10 * abs(s1.score)
    assert np.allclose(s2b.u, s1.u, rtol=0, atol=1e-7)
    g2.close()
    # the same context solves again: bit-reproducible, the hand-over included
    s2 = g1.solve(p.u0)
    assert np.array_equal(s2.u, s1.u) and s2.n_trials == s1.n_trials and s2.nodes.tolist() == s1.nodes.tolist()
    print(f"m={m} rho={rho} storage={storage}: sub-problem of {st1.sub_rows} (view {st1.rows}), {st1.sub_passes} of {st1.passes} passes on it, "
          f"{st1.view_passes} on the view; trials {s1.n_trials} / views {s0.n_trials} / oracle {sr.n_trials}")
    g0.close()
    g1.close()


def test_sweep_size_m30000_against_the_routes_without_it():
    """the sweep size (the oracle's answer for it is asserted in test_gpu_configs.py on the default route, which is
    this one): here the three routes against each other, and what ran where"""
    p = synth.make_euclidean_problem(30000, 0.95, seed=12345)
    res = {}
    for route in ("noviews", "views", "sub"):
        g, s, st = _gpu(p, abi.STORE_F32_CSC, route, **synth.EUCLID_BENCH_PARAMS)
        res[route] = (s, st)
        g.close()
    s0 = res["noviews"][0]
    for route in ("views", "sub"):
        s = res[route][0]
        _same_list(s, s0)
        assert s.ifinal == s0.ifinal
        assert abs(s.score - s0.score) <= 1e-10 * abs(s0.score)
        assert np.allclose(s.u, s0.u, rtol=0, atol=1e-7)
    st = res["sub"][1]
    assert st.sub_dense == 1, "the sub-problem of the sweep size is the dense inlier block: a dense fp32 store"
    assert st.sub_entries == 1 and st.sub_leaves == 0
    assert st.sub_passes * 10 >= st.passes * 7, (st.sub_passes, st.passes)   # the long outer iterations run on it
    assert st.view_passes < res["views"][1].view_passes


@pytest.mark.parametrize("storage", [abi.STORE_F32_CSC, abi.STORE_F64_CSC])
def test_pointnormal_problem_takes_the_sub_problem_too(storage):
    p = synth.make_pointnormal_problem(14000, 0.9, seed=99)
    sr = _oracle(p, pointnormal=True)
    g0, s0, st0 = _gpu(p, storage, "views", pointnormal=True)
    g1, s1, st1 = _gpu(p, storage, "sub", pointnormal=True)
    for s in (s0, s1):
        _same_list(s, sr)
        assert abs(s.score - sr.score) <= 1e-6 * abs(sr.score) and s.ifinal == sr.ifinal
    assert st1.sub_entries >= 1 and st1.sub_passes > 0
    print(f"pointnormal storage={storage}: views {st0.builds} ({st0.rows} rows), sub-problem entries {st1.sub_entries} "
          f"({st1.sub_rows} associations, {st1.sub_passes} of {st1.passes} passes)")
    g0.close()
    g1.close()


_CHILD = r"""
import sys, json, hashlib
import numpy as np
sys.path.insert(0, {root!r})
from clipper_amd import _abi as abi
from clipper_amd import synth
p = synth.make_euclidean_problem({m}, 0.9, seed={seed})
g = abi.HipClipper(storage={storage})
g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
s = g.solve(p.u0)
st = g.view_stats()
print(json.dumps(dict(nodes=s.nodes.tolist(), score=s.score, ifinal=s.ifinal, trials=s.n_trials, passes=s.n_passes,
                      entries=st.sub_entries, leaves=st.sub_leaves, sub_passes=st.sub_passes, view_passes=st.view_passes,
                      u=hashlib.sha256(np.ascontiguousarray(s.u).tobytes()).hexdigest())))
"""


def _child(m, seed, storage, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, m=m, seed=seed, storage=storage)], env=e,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("k", [1, 2, 5, 11])
def test_a_solve_that_is_handed_back_and_over_again(k):
    """CLIPPER_HIP_SUB_TEST_LEAVE = k: the k-th launch after every hand-over is told that a column outside the
    sub-problem could come back to life. Its decision leaves the pass prepared, the point goes back to the full
    problem (u = 0, gradF < 0 outside S), the pass runs on the row view, the next decision hands the solve over
    again — until the solve's hand-overs are used up and it ends on the view. Every transition of the protocol, at
    several points of the iteration; the result must be the one of the undisturbed solve."""
    m, seed = 16000, 31
    base = _child(m, seed, abi.STORE_F32_CSC, {})
    off = _child(m, seed, abi.STORE_F32_CSC, {"CLIPPER_HIP_SUBPROBLEM": "0"})
    dist = _child(m, seed, abi.STORE_F32_CSC, {"CLIPPER_HIP_SUB_TEST_LEAVE": str(k)})
    assert base["entries"] == 1 and base["leaves"] == 0 and off["entries"] == 0
    assert dist["leaves"] >= 1 and dist["entries"] >= 2, dist
    for r in (off, dist):
        assert sorted(r["nodes"]) == sorted(base["nodes"]) and r["ifinal"] == base["ifinal"]
        assert abs(r["score"] - base["score"]) <= 1e-10 * abs(base["score"])
        assert abs(r["trials"] - base["trials"]) <= max(2, base["trials"] // 20)
    assert dist["sub_passes"] + dist["view_passes"] <= dist["passes"]
    print(f"k={k}: {dist['entries']} hand-overs, {dist['leaves']} back, {dist['sub_passes']} + {dist['view_passes']} of {dist['passes']} "
          f"passes on the sub-problem / the view; trials {dist['trials']} (undisturbed {base['trials']}, without {off['trials']})")


def test_custom_solver_parameters_move_the_hand_over_around():
    p = synth.make_euclidean_problem(13000, 0.9, seed=8)
    for kw in (dict(maxiniters=20), dict(beta=0.5), dict(tol_u=1e-6, tol_F=1e-7, maxoliters=3)):
        prm = ref.Params(**kw)
        sr = _oracle(p, params=prm, **synth.EUCLID_BENCH_PARAMS)
        gp = abi.Params(**kw)
        g, s, st = _gpu(p, abi.STORE_F64_CSC, "sub", params=gp, **synth.EUCLID_BENCH_PARAMS)
        _same_list(s, sr)
        assert abs(s.score - sr.score) <= 1e-6 * abs(sr.score) and s.ifinal == sr.ifinal, kw
        print(f"{kw}: sub-problem entries {st.sub_entries}, {st.sub_passes} of {st.passes} passes; trials {s.n_trials} (oracle {sr.n_trials})")
        g.close()
