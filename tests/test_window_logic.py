"""CPU check of the solver's device algorithm (windowed line search, pair / window passes,
transition iterations) through its executable numpy model (tests/window_model.py) against the
oracle's plain restatement of findDenseClique (oracle/clipper_ref.py: numpy_solve).

For every window size the model must evaluate exactly the oracle's trial sequence, take the same
accept / reject decisions and end in the same point; only the number of passes over M changes."""
import numpy as np
import pytest

from oracle import clipper_ref as ref
from tests import window_model as wm


def _random_problem(m, density, seed, clique=None):
    rng = np.random.default_rng(seed)
    M = np.triu(rng.random((m, m)) * (rng.random((m, m)) < density), 1)
    if clique:
        idx = rng.choice(m, clique, replace=False)
        for a in idx:
            for b in idx:
                if a < b:
                    M[a, b] = 0.8 + 0.2 * rng.random()
    C = (M != 0).astype(float)
    return M, C, rng.random(m)


@pytest.mark.parametrize("V", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("m,density,seed,clique", [(40, 0.3, 1, 8), (120, 0.15, 2, 15), (200, 0.1, 3, 25)])
def test_window_model_matches_plain_line_search(V, m, density, seed, clique):
    Mup, Cup, u0 = _random_problem(m, density, seed, clique)
    p = ref.Params()
    s_ref = ref.numpy_solve(Mup, Cup, u0, p)
    P = wm.Params()
    r = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, P, V)
    assert r.n_trials == s_ref.n_trials
    assert r.ifinal == s_ref.ifinal
    assert abs(r.F - s_ref.score) <= 1e-10 * max(1.0, abs(s_ref.score))
    assert np.allclose(r.u, s_ref.u, rtol=0, atol=1e-10)
    assert abs(r.d - s_ref.d) <= 1e-10 * max(1.0, abs(s_ref.d))
    # passes: 2 initial + one pair pass per penalty update + ceil(run / V) per line search
    assert r.n_passes <= r.n_trials + 3 + r.ifinal
    assert r.n_passes >= (r.n_trials + V - 1) // V
    if V == 1:
        assert all(j == 0 for j in r.accepted)


@pytest.mark.parametrize("kw", [dict(maxlsiters=1), dict(maxlsiters=2), dict(maxlsiters=4),
                                dict(beta=0.5), dict(beta=0.1, maxlsiters=7), dict(maxiniters=3),
                                dict(maxiniters=0), dict(maxoliters=0), dict(maxoliters=2),
                                dict(rescale_u0=False)])
def test_window_model_parameter_variants(kw):
    Mup, Cup, u0 = _random_problem(150, 0.12, 7, 18)
    p = ref.Params(**{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
    s_ref = ref.numpy_solve(Mup, Cup, u0, p)
    for V in (1, 3, 6):
        r = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, wm.Params(**kw), V)
        assert r.n_trials == s_ref.n_trials, (V, kw)
        assert r.ifinal == s_ref.ifinal
        assert abs(r.F - s_ref.score) <= 1e-10 * max(1.0, abs(s_ref.score))
        assert np.allclose(r.u, s_ref.u, rtol=0, atol=1e-10)


def test_window_saves_passes_when_the_line_search_backtracks():
    # a problem whose line search rejects several step sizes in a row: the window must fold
    # each run of rejections into one pass
    Mup, Cup, u0 = _random_problem(300, 0.08, 11, 40)
    s_ref = ref.numpy_solve(Mup, Cup, u0, ref.Params())
    r1 = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, wm.Params(), 1)
    r6 = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, wm.Params(), 6)
    assert r1.n_trials == r6.n_trials == s_ref.n_trials
    rejected = r1.n_trials - len(r1.accepted)
    if rejected > 0:
        assert r6.n_passes < r1.n_passes
    assert r6.accepted and sum(j + 1 for j in r6.accepted) <= r6.n_trials


@pytest.mark.parametrize("V", [4, 6])
@pytest.mark.parametrize("m,density,seed,clique", [(40, 0.3, 1, 8), (120, 0.15, 2, 15), (200, 0.1, 3, 25), (300, 0.08, 11, 40)])
def test_the_window_in_use_changes_passes_not_trials(V, m, density, seed, clique):
    """SolverState::weff (round 6): while line searches accept their first trial a pass multiplies candidate 0 alone and
    the decision walks that one candidate; after a rejection the whole window again. Whatever the policy predicts, the
    trials, the decisions and the result are the oracle's — a wrong guess costs a pass, nothing else."""
    Mup, Cup, u0 = _random_problem(m, density, seed, clique)
    s_ref = ref.numpy_solve(Mup, Cup, u0, ref.Params())
    r = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, wm.Params(), V, adaptive=True)
    rf = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, wm.Params(), V, adaptive=False)
    assert r.n_trials == rf.n_trials == s_ref.n_trials and r.ifinal == s_ref.ifinal
    assert abs(r.F - s_ref.score) <= 1e-10 * max(1.0, abs(s_ref.score))
    assert np.allclose(r.u, s_ref.u, rtol=0, atol=1e-10)
    assert r.n_passes <= r.n_trials + 3 + r.ifinal
    for kw in (dict(beta=0.5), dict(maxiniters=3, maxoliters=40), dict(rescale_u0=False)):   # (a line search capped at two trials runs the loops to their limits: minutes)
        p = ref.Params(**{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
        s2 = ref.numpy_solve(Mup, Cup, u0, p)
        r2 = wm.solve(Mup + Mup.T, Cup + Cup.T, u0, wm.Params(**kw), V, adaptive=True)
        assert r2.n_trials == s2.n_trials and r2.ifinal == s2.ifinal, kw
        assert np.allclose(r2.u, s2.u, rtol=0, atol=1e-10), kw
