#!/usr/bin/env python3
"""Randomized A/B of the resident solver on a row view against the streamed views (clipper_hip_set_row_view 0 vs 2):
N seeded random (m, outlier ratio, seed, value type) — node list, ifinal, score, trial and pass counts must agree.
  python tools/rvr_random_ab.py [N=60] [seed=1] [m_lo=2500] [m_hi=24000] [modes=0,2] [params|pn|params+pn]
What found the norms of a window left with a live row outside the view (round 4). A "BAD" line with equal results and
trial counts about 98 apart is a line search that runs into maxlsiters on rounding noise in one order of summation
and not in the other (profiles/r04_rvr_random_ab.txt): not a defect."""
import sys
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 24000
modes = tuple(int(x) for x in sys.argv[5].split(',')) if len(sys.argv) > 5 else (0, 2)   # e.g. 2,1: streamed views against no views
vary_params = len(sys.argv) > 6 and "params" in sys.argv[6]
pointnormal = len(sys.argv) > 6 and "pn" in sys.argv[6]   # PointNormalDistance problems (6-DoF surfels) instead of EuclideanDistance   # also randomize the solver parameters (they move the exits around)
bad = 0
nres = 0
for k in range(N):
    m = int(rng.integers(lo, hi))
    rho = float(rng.choice([0.7, 0.8, 0.88, 0.92, 0.95, 0.97, 0.985]))
    seed = int(rng.integers(1, 10**6))
    storage = abi.STORE_F64_CSC if rng.integers(0, 2) else abi.STORE_F32_CSC
    p = synth.make_pointnormal_problem(m, rho, seed=seed) if pointnormal else synth.make_euclidean_problem(m, rho, seed=seed)
    kw = {}
    if vary_params:
        kw = {"beta": float(rng.choice([0.25, 0.5, 0.1])), "maxlsiters": int(rng.choice([99, 20, 12])),
              "maxiniters": int(rng.choice([200, 20, 5, 2])), "maxoliters": int(rng.choice([1000, 40, 6])),
              "tol_u": float(rng.choice([1e-8, 1e-6, 1e-10])), "tol_F": float(rng.choice([1e-9, 1e-7, 1e-12])),
              "rescale_u0": int(rng.integers(0, 2)), "eps": float(rng.choice([1e-9, 1e-7]))}
    out = []
    for mode in modes:
        g = abi.HipClipper(storage=storage)
        g.set_row_view(mode)
        for key, val in kw.items():
            setattr(g.params, key, val)
        if pointnormal:
            g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
        else:
            g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        s = g.solve(p.u0)
        st = g.view_stats()
        out.append((s, st.builds, st.rows, st.resident_launches))
        g.close()
    (s1, b1, r1, l1), (s2, b2, r2, l2) = out
    nres += 1 if l1 else 0
    ok = (sorted(s1.nodes.tolist()) == sorted(s2.nodes.tolist()) and s1.ifinal == s2.ifinal and abs(s1.score - s2.score) <= 1e-9 * abs(s2.score)
          and abs(s1.n_trials - s2.n_trials) <= max(2, s2.n_trials // 50) and abs(s1.n_passes - s2.n_passes) <= max(2, s2.n_passes // 50))
    # (the node SET: near-equal entries of u trade places in the ordered list when the partial sums associate differently)
    same = sorted(s1.nodes.tolist()) == sorted(s2.nodes.tolist()) and s1.ifinal == s2.ifinal and abs(s1.score - s2.score) <= 1e-8 * abs(s2.score)
    if not same:
        print("DIFFERENT RESULT:", end=" ")
    wrong = globals().get("wrong", 0) + (0 if same else 1)
    globals()["wrong"] = wrong
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} m={m} rho={rho} seed={seed} storage={storage}: views {b1}/{b2} rows {r1}/{r2} resident launches {l1} | "
          f"passes {s1.n_passes}/{s2.n_passes} trials {s1.n_trials}/{s2.n_trials} ifinal {s1.ifinal}/{s2.ifinal} "
          f"dscore {abs(s1.score - s2.score) / abs(s2.score):.1e}" + (f" {kw}" if kw else ""), flush=True)
print(f"{N} cases, {nres} with a resident launch, {bad} BAD (counts), {globals().get('wrong', 0)} with a DIFFERENT RESULT")
sys.exit(1 if globals().get('wrong', 0) else 0)
