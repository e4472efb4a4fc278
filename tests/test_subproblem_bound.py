"""The inequality the live sub-problem rests on, checked on the CPU against the reference's own iteration.

For a candidate x >= 0 of the line search (clipper.cpp:234-251) and a column c with x[c] = 0,

    gradF[c] = -d sum(x) + sum_j (M[c,j] + d C[c,j]) x[j]  <=  -d sum(x) + (1 + d) sqrt(N_c) |x|

because no stored entry of M exceeds 1 and C is its pattern (clipper.cpp:58-64), N_c being the number of stored entries of
column c in the rows where x > 0 (Cauchy-Schwarz). The solver fixes a set S (the view's rows and every column with more
than 0.4 sum(x)^2 entries in them) and one N for all columns outside it — the largest count outside S plus |S \\ view| —
and keeps multiplying M[S,S] alone while d^2 sum(x)^2 >= kappa (1 + d)^2 N |x|^2 (k_solver.hip.h: sub_bound). This test
runs the oracle's numpy restatement of findDenseClique with a hook on every candidate and asserts what the kernels assume:
while the inequality holds, the gradient is negative on every column outside S (so the projection max(x + a gradF, 0)
keeps it at zero: clipper.cpp:238), and the support of every later candidate stays inside S. The GPU suite
(test_gpu_subproblem.py) checks the consequence — the same solve; this file checks the reason, without a GPU."""
import numpy as np
import pytest

from clipper_amd import synth
from oracle import clipper_ref as ref

THETA = 0.4       # k_subproblem.hip.h: SUB_THETA
KAPPA_ENTER = 1.10
KAPPA_STAY = 1.01


def _iterate(Ms, Cs, u0, p, hook):
    """findDenseClique as oracle/clipper_ref.py:numpy_solve states it (clipper.cpp:172-323), calling
    hook(d, unew, gradFnew, accepted_u, accepted_gradF) for every candidate of every line search."""
    u = Ms @ u0 + u0
    u = u / np.linalg.norm(u)
    d = 0.0
    Cbu = u.sum() - Cs @ u - u
    idx = (Cbu > p.eps) & (u > p.eps)
    if idx.sum() > 0:
        d = float(np.mean((Ms @ u + u)[idx] / Cbu[idx]))
    for _i in range(p.maxoliters):
        gradF = (1 + d) * u - d * u.sum() + Ms @ u + Cs @ u * d
        F = float(u @ gradF)
        for _j in range(p.maxiniters):
            alpha = 1.0
            for _k in range(p.maxlsiters):
                unew = np.maximum(u + alpha * gradF, 0)
                unew = unew / np.sqrt(float(unew @ unew))
                gradFnew = (1 + d) * unew - d * unew.sum() + Ms @ unew + Cs @ unew * d
                hook(d, unew, gradFnew, u, gradF)
                Fnew = float(unew @ gradFnew)
                deltaF = Fnew - F
                if deltaF < -p.eps:
                    alpha *= p.beta
                else:
                    break
            deltau = float(np.linalg.norm(unew - u))
            F, u, gradF = Fnew, unew, gradFnew
            if deltau < p.tol_u or abs(deltaF) < p.tol_F:
                break
        Cbu = u.sum() - Cs @ u - u
        idx = (Cbu > p.eps) & (u > p.eps)
        if idx.sum() == 0:
            break
        d += float(np.mean(np.abs((Ms @ u + u)[idx] / Cbu[idx])))
    return u


@pytest.mark.parametrize("m,rho,seed,pointnormal", [(1200, 0.9, 11, False), (1500, 0.95, 12, False), (900, 0.8, 13, False),
                                                    (1000, 0.9, 14, True)])
def test_gradient_is_negative_outside_the_subproblem_while_the_bound_holds(m, rho, seed, pointnormal):
    if pointnormal:
        prob = synth.make_pointnormal_problem(m, rho, seed=seed)
        Mup = ref.numpy_affinity_pointnormal(prob.D1, prob.D2, prob.A)
    else:
        prob = synth.make_euclidean_problem(m, rho, seed=seed)
        Mup, _ = ref.numpy_affinity_euclidean(prob.D1, prob.D2, prob.A, **synth.EUCLID_BENCH_PARAMS)
    Ms = Mup + Mup.T
    Cs = (Ms != 0).astype(float)
    assert Ms.max() <= 1.0   # the premise: no stored entry exceeds 1
    p = ref.Params()
    st = {"S": None, "N": 0, "checked": 0, "left": 0, "entered_at_d": None, "worst": -np.inf}

    def hook(d, x, g, u_acc, g_acc):
        s, z = float(x.sum()), float(x @ x)
        if st["S"] is None:
            # the view as the solver builds it: the rows that are live at the accepted point (u > 0 or gradF > 0)
            R = (u_acc > 0) | (g_acc > 0)
            if R.sum() * 3 > m:
                return
            cnt = (Ms[R] != 0).sum(axis=0)
            S = R | (cnt > THETA * s * s)
            N = int(cnt[~S].max(initial=0)) + int((S & ~R).sum())
            if (~S).sum() == 0 or not np.all(x[~S] == 0):
                return
            if d * d * s * s >= KAPPA_ENTER * (1 + d) ** 2 * N * z:
                st["S"], st["N"], st["entered_at_d"] = S, N, d
            else:
                return
        S, N = st["S"], st["N"]
        # by induction the candidate's support is inside S: every earlier candidate had a negative gradient outside
        assert np.all(x[~S] == 0)
        if d * d * s * s >= KAPPA_STAY * (1 + d) ** 2 * N * z:
            st["checked"] += 1
            worst = float(g[~S].max())
            st["worst"] = max(st["worst"], worst)
            assert worst < 0.0
            # and the bound itself, column by column, with the column's own count
            Nc = (Ms[np.ix_(x > 0, ~S)] != 0).sum(axis=0)
            assert np.all(Nc <= N)
            assert np.all(g[~S] <= -d * s + (1 + d) * np.sqrt(Nc * z) + 1e-12 * (1 + d))
        else:
            st["left"] += 1
            st["S"] = None   # handed back: a new selection at the next opportunity

    u = _iterate(Ms, Cs, prob.u0, p, hook)
    # the loop above is the oracle's: same result as numpy_solve
    sol = ref.numpy_solve(Mup, np.triu(Cs, 1), prob.u0, p)
    assert np.max(np.abs(sol.u - u)) < 1e-12
    # the property was exercised, not vacuous: the bound is met once the penalty has grown, and the candidates of the
    # rest of the solve are checked against it (7 of 14 / 20 of 36 / 50 of 67 on these three problems)
    assert st["entered_at_d"] is not None and st["entered_at_d"] > 1.0
    assert st["checked"] >= 5
    assert st["worst"] < 0.0


def test_bound_needs_the_rows_of_S_outside_the_view():
    """N counts the view's rows only; a row of S outside the view may come back to life inside the sub-problem and then
    adds one entry to a column outside (NOTEBOOK.md round 6, "The bound's N"): the constructed case where leaving
    |S \\ view| out would let the inequality pass although the column's true count is larger."""
    # column c has entries in two view rows and in one row r of S \\ view
    cnt_view = 2
    s_minus_view = 1
    x = np.array([0.5, 0.5, np.sqrt(0.5)])      # the two view rows and r, all live now
    z = float(x @ x)
    s = float(x.sum())
    d = 50.0
    true_Nc = 3
    g_upper_true = -d * s + (1 + d) * np.sqrt(true_Nc * z)
    g_upper_short = -d * s + (1 + d) * np.sqrt(cnt_view * z)
    g_upper_used = -d * s + (1 + d) * np.sqrt((cnt_view + s_minus_view) * z)
    assert g_upper_short < g_upper_true          # the short bound is not a bound
    assert g_upper_used >= g_upper_true - 1e-12  # the one in use is
