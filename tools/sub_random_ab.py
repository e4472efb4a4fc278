#!/usr/bin/env python3
"""Randomized A/B of the live sub-problem against the route with row views alone (clipper_hip_set_subproblem 0 vs 1):
N seeded random (m, outlier ratio, seed, value type[, solver parameters, invariant]) — node set, ifinal, objective must
agree; trial and pass counts are compared at 2 %. ONE context per problem (the matrix is built once, solved on both routes).
  python tools/sub_random_ab.py [N=60] [seed=1] [m_lo=12000] [m_hi=40000] [params|pn|params+pn|leave]
`leave`: with CLIPPER_HIP_SUB_TEST_LEAVE set by the caller the hand-backs are exercised as well. maxiniters < 5 is left
out of the random parameters: there the reference's own answer depends on the order of its sums (NOTEBOOK.md).
With random parameters the COUNTS may part at the end of a solve without any route being wrong: a line search at the
converged point is decided by the rounding of F (4 or 99 trials, same u), and where every inner loop is cut off by
maxiniters far from converged a trial accepted on one route and rejected on another moves u by 1e-7 — between ANY two
routes, which one being a matter of tol_F (tools/sub_case_dump.py shows both, NOTEBOOK.md round 6). Two node lists that differ only in associations whose u lies within 2 max|du| of the selection boundary in both
solves are reported as a TIE, not as a different result."""
import sys
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
opt = sys.argv[5] if len(sys.argv) > 5 else ""
vary_params, pointnormal = "params" in opt, "pn" in opt
bad = wrong = entered = left = ties = 0
for k in range(N):
    m = int(rng.integers(lo, hi))
    rho = float(rng.choice([0.8, 0.88, 0.92, 0.95, 0.97]))
    seed = int(rng.integers(1, 10**6))
    storage = abi.STORE_F64_CSC if rng.integers(0, 2) else abi.STORE_F32_CSC
    p = synth.make_pointnormal_problem(m, rho, seed=seed) if pointnormal else synth.make_euclidean_problem(m, rho, seed=seed)
    kw = {}
    if vary_params:
        kw = {"beta": float(rng.choice([0.25, 0.5, 0.1])), "maxlsiters": int(rng.choice([99, 20])),
              "maxiniters": int(rng.choice([200, 20, 5])), "maxoliters": int(rng.choice([1000, 40, 6])),
              "tol_u": float(rng.choice([1e-8, 1e-6, 1e-10])), "tol_F": float(rng.choice([1e-9, 1e-7, 1e-12])),
              "rescale_u0": int(rng.integers(0, 2)), "eps": float(rng.choice([1e-9, 1e-7]))}
    g = abi.HipClipper(storage=storage)
    g.set_row_view(2 if rng.integers(0, 4) == 0 else 0)   # (a quarter of the problems with views never resident: the sub-problem also behind small views)
    for key, val in kw.items():
        setattr(g.params, key, val)
    if pointnormal:
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
    else:
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    out = []
    for sub in (0, 1):
        g.set_subproblem(sub)
        s = g.solve(p.u0)
        st = g.view_stats()
        out.append((s, st.builds, st.rows, st.sub_entries, st.sub_leaves, st.sub_passes, st.sub_rows))
    g.close()
    (s1, b1, r1, e1, l1, sp1, sr1), (s2, b2, r2, e2, l2, sp2, sr2) = out
    entered += 1 if e1 else 0
    left += l1
    du = float(np.max(np.abs(s1.u - s2.u)))
    odd = set(s1.nodes.tolist()) ^ set(s2.nodes.tolist())
    tie = False
    if odd and len(s1.nodes) == len(s2.nodes) and len(s1.nodes) < m:
        tie = True
        for s in (s1, s2):   # every association the lists disagree on sits at the boundary of BOTH selections
            srt = np.sort(s.u)[::-1]
            edge = 0.5 * (srt[len(s.nodes) - 1] + srt[len(s.nodes)])
            tie = tie and all(abs(s.u[i] - edge) <= 2 * du for i in odd)
    ties += 1 if tie else 0
    same = ((not odd or tie) and s1.ifinal == s2.ifinal and
            abs(s1.score - s2.score) <= 1e-8 * abs(s2.score) and du <= 1e-6)
    ok = same and abs(s1.n_trials - s2.n_trials) <= max(2, s2.n_trials // 50) and abs(s1.n_passes - s2.n_passes) <= max(2, s2.n_passes // 50)
    wrong += 0 if same else 1
    bad += 0 if ok else 1
    print(f"{'DIFFERENT RESULT: ' if not same else ''}{f'TIE at the selection boundary {sorted(odd)}: ' if tie else ''}{'ok ' if ok else 'BAD'} m={m} rho={rho} seed={seed} storage={storage}: view rows {r1} "
          f"sub-problem {sr1} entries {e1} leaves {l1} passes on it {sp1} | passes {s1.n_passes}/{s2.n_passes} trials {s1.n_trials}/{s2.n_trials} "
          f"ifinal {s1.ifinal}/{s2.ifinal} dscore {abs(s1.score - s2.score) / abs(s2.score):.1e} max|du| {du:.1e}"
          + (f" {kw}" if kw else ""), flush=True)
print(f"{N} cases, {entered} handed over to the sub-problem ({left} hand-backs), {bad} BAD (counts), {ties} ties at the selection boundary, {wrong} with a DIFFERENT RESULT")
sys.exit(1 if wrong else 0)
