#!/usr/bin/env python3
"""Column shards against one shard: seeded random problems, the in-process group driver with 2 ... 8 logical shards
on the one device (the kernels, the per-shard fills and views, the per-pass exchange of the multi-GPU path) against the
single-shard solve — node set, ifinal, objective.
  python tools/shards_random_ab.py [N=60] [seed=3] [m_lo=2500] [m_hi=30000]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 30000
wrong = 0
for k in range(N):
    m = int(rng.integers(lo, hi))
    rho = float(rng.choice([0.7, 0.8, 0.88, 0.92, 0.95, 0.97]))
    seed = int(rng.integers(1, 10**6))
    storage = [abi.STORE_F32_CSC, abi.STORE_F64_CSC, abi.STORE_F32][int(rng.integers(0, 3))]
    if storage == abi.STORE_F32 and m > 16000:
        storage = abi.STORE_F32_CSC   # (a dense store of 4 m^2 bytes per shard set: keep it small)
    P = int(rng.choice([2, 3, 4, 8]))
    pn = bool(rng.integers(0, 4) == 0)
    p = synth.make_pointnormal_problem(m, rho, seed=seed) if pn else synth.make_euclidean_problem(m, rho, seed=seed)
    res = []
    for group in (None, [0] * P):
        g = abi.HipClipper(storage=storage, group=group)
        if pn:
            g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
        else:
            g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        s = g.solve(p.u0)
        st = g.view_stats()
        res.append((s, st.builds, st.rows, st.view_passes))
        g.close()
    (s1, b1, r1, v1), (s2, b2, r2, v2) = res
    same = sorted(s1.nodes.tolist()) == sorted(s2.nodes.tolist()) and s1.ifinal == s2.ifinal and abs(s1.score - s2.score) <= 1e-8 * abs(s2.score)
    wrong += 0 if same else 1
    print(f"{'ok ' if same else 'DIFFERENT RESULT'} m={m} rho={rho} seed={seed} storage={storage} {'pointnormal' if pn else 'euclidean'} shards {P}: "
          f"views {b1}/{b2} rows {r1}/{r2} view passes {v1}/{v2} | passes {s1.n_passes}/{s2.n_passes} trials {s1.n_trials}/{s2.n_trials} "
          f"ifinal {s1.ifinal}/{s2.ifinal} nodes {len(s1.nodes)}/{len(s2.nodes)} dscore {abs(s1.score - s2.score) / abs(s2.score):.1e}", flush=True)
print(f"{N} cases, {wrong} with a DIFFERENT RESULT")
sys.exit(1 if wrong else 0)
