"""The window in use (csrc/k_solver.hip.h: SolverState::weff). While line searches accept their first trial — the whole
first two outer iterations of every problem looked at — a pass on the slices multiplies candidate 0 of the pending window
alone (one LDS gather and two fmas per entry instead of three and seven); a candidate-0 pass whose candidate is REJECTED
is discarded and the same window multiplied again, whole. Every candidate that is ever walked therefore sits at the
window index, and is formed by the expressions, it has with full windows: the solve must be BIT FOR BIT the one with
`CLIPPER_HIP_ADAPTIVE_WINDOW=0` — the same u, the same trials — in a few more or (never) fewer launches."""
import numpy as np
import pytest

from clipper_amd import _abi as abi
from clipper_amd import synth

pytestmark = pytest.mark.gpu


def _solve(monkeypatch, p, storage, adaptive, row_view=0, subproblem=0, pointnormal=False, **kw):
    monkeypatch.setenv("CLIPPER_HIP_ADAPTIVE_WINDOW", "1" if adaptive else "0")
    g = abi.HipClipper(storage=storage)      # (the switch is read when the context is created)
    g.set_row_view(row_view)
    g.set_subproblem(subproblem)
    for k, v in kw.items():
        setattr(g.params, k, v)
    if pointnormal:
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
    else:
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = g.solve(p.u0)
    s2 = g.solve(p.u0)
    assert np.array_equal(s2.u, s.u) and s2.n_passes == s.n_passes   # the context solves again: the policy re-arms itself
    g.close()
    return s


@pytest.mark.parametrize("storage", [abi.STORE_F32_CSC, abi.STORE_F64_CSC])
@pytest.mark.parametrize("m,rho,seed,row_view", [
    (3000, 0.9, 3, 0),        # window of 4, no views at this size
    (6000, 0.5, 777 + 6000, 1),   # long line searches (130 - 150 trials), every pass on M
    (10000, 0.95, 12345, 0),  # the headline: passes on M, then the resident launch on the view
    (10000, 0.95, 12345, 2),  # ... the view's iterations streamed
    (16000, 0.9, 31, 0),      # passes on M, on a view, on the live sub-problem
])
def test_bit_identical_to_full_windows(monkeypatch, storage, m, rho, seed, row_view):
    p = synth.make_euclidean_problem(m, rho, seed=seed)
    on = _solve(monkeypatch, p, storage, True, row_view=row_view)
    off = _solve(monkeypatch, p, storage, False, row_view=row_view)
    assert np.array_equal(on.u, off.u), float(np.max(np.abs(on.u - off.u)))
    assert on.n_trials == off.n_trials and on.ifinal == off.ifinal and on.score == off.score and on.d == off.d
    assert on.nodes.tolist() == off.nodes.tolist()
    # a wrong guess costs one pass; there are one or two per solve (the first rejection after the quiet start, and after
    # a long quiet run late in an outer iteration)
    assert off.n_passes <= on.n_passes <= off.n_passes + 8, (on.n_passes, off.n_passes)
    print(f"m={m} rho={rho} storage={storage} views={row_view}: trials {on.n_trials}, passes {on.n_passes} (full windows: {off.n_passes})")


def test_solver_parameters_and_pointnormal(monkeypatch):
    p = synth.make_euclidean_problem(9000, 0.93, seed=5)
    for kw in (dict(beta=0.5), dict(maxlsiters=3), dict(maxlsiters=1), dict(maxiniters=3, maxoliters=40), dict(rescale_u0=0),
               dict(tol_u=1e-6, tol_F=1e-7)):
        on = _solve(monkeypatch, p, abi.STORE_F64_CSC, True, **kw)
        off = _solve(monkeypatch, p, abi.STORE_F64_CSC, False, **kw)
        assert np.array_equal(on.u, off.u) and on.n_trials == off.n_trials and on.ifinal == off.ifinal, kw
    q = synth.make_pointnormal_problem(5000, 0.9)
    on = _solve(monkeypatch, q, abi.STORE_F32_CSC, True, pointnormal=True)
    off = _solve(monkeypatch, q, abi.STORE_F32_CSC, False, pointnormal=True)
    assert np.array_equal(on.u, off.u) and on.n_trials == off.n_trials
