#!/usr/bin/env python3
"""Randomized check that the window in use (SolverState::weff) changes nothing but the number of launches: N seeded random
problems (size, outlier ratio, value type, solver parameters, views on / streamed / off, PointNormal) solved with
CLIPPER_HIP_ADAPTIVE_WINDOW = 1 and = 0 — u must be BIT-IDENTICAL, the trial count and ifinal equal.
  python tools/window_random_ab.py [N=60] [seed=1] [m_lo=2500] [m_hi=30000]"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 30000
wrong = 0
extra = []
for k in range(N):
    m = int(rng.integers(lo, hi))
    rho = float(rng.choice([0.5, 0.8, 0.9, 0.95, 0.97]))
    seed = int(rng.integers(1, 10**6))
    storage = abi.STORE_F64_CSC if rng.integers(0, 2) else abi.STORE_F32_CSC
    pn = rng.integers(0, 5) == 0 and m <= 12000
    view_mode = int(rng.choice([0, 0, 2, 1]))
    kw = {}
    if rng.integers(0, 2):
        kw = {"beta": float(rng.choice([0.25, 0.5, 0.1])), "maxlsiters": int(rng.choice([99, 20, 3])),
              "maxiniters": int(rng.choice([200, 20, 5])), "maxoliters": int(rng.choice([1000, 40, 6])),
              "tol_u": float(rng.choice([1e-8, 1e-6])), "tol_F": float(rng.choice([1e-9, 1e-7])),
              "rescale_u0": int(rng.integers(0, 2)), "eps": float(rng.choice([1e-9, 1e-7]))}
    p = synth.make_pointnormal_problem(m, rho, seed=seed) if pn else synth.make_euclidean_problem(m, rho, seed=seed)
    out = []
    for adaptive in ("1", "0"):
        os.environ["CLIPPER_HIP_ADAPTIVE_WINDOW"] = adaptive
        g = abi.HipClipper(storage=storage)
        g.set_row_view(view_mode)
        for key, val in kw.items():
            setattr(g.params, key, val)
        if pn:
            g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
        else:
            g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        out.append(g.solve(p.u0))
        g.close()
    a, b = out
    same = np.array_equal(a.u, b.u) and a.n_trials == b.n_trials and a.ifinal == b.ifinal and a.score == b.score
    wrong += 0 if same else 1
    extra.append(a.n_passes - b.n_passes)
    print(f"{'ok ' if same else 'DIFFERENT'} m={m} rho={rho} seed={seed} storage={storage} views={view_mode} pn={int(pn)}: trials {a.n_trials}/{b.n_trials} "
          f"passes {a.n_passes}/{b.n_passes} ifinal {a.ifinal}/{b.ifinal} max|du| {float(np.max(np.abs(a.u - b.u))):.1e}" + (f" {kw}" if kw else ""), flush=True)
print(f"{N} cases, {wrong} NOT bit-identical; extra passes with the window in use: min {min(extra)}, median {int(np.median(extra))}, max {max(extra)}")
sys.exit(1 if wrong else 0)
