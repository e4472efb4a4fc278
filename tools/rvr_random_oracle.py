#!/usr/bin/env python3
"""Seeded random problems in the range where a row view can go to the resident solver, default parameters: the
default path (resident on a view where it fits) against the ORACLE — node list, ifinal, objective (1e-6).
  python tools/rvr_random_oracle.py [N=40] [seed=5] [m_lo=2000] [m_hi=9000]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth
from oracle import clipper_ref as ref

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 9000
bad = nres = 0
for k in range(N):
    m = int(rng.integers(lo, hi))
    rho = float(rng.choice([0.8, 0.88, 0.92, 0.95, 0.97]))
    seed = int(rng.integers(1, 10**6))
    storage = abi.STORE_F64_CSC if rng.integers(0, 2) else abi.STORE_F32_CSC
    p = synth.make_euclidean_problem(m, rho, seed=seed)
    g = abi.HipClipper(storage=storage)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = g.solve(p.u0)
    st = g.view_stats()
    g.close()
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    t0 = time.time()
    so = r.solve(p.u0)
    ok = s.nodes.tolist() == so.nodes.tolist() and s.ifinal == so.ifinal and abs(s.score - so.score) <= 1e-6 * abs(so.score)
    bad += 0 if ok else 1
    nres += 1 if st.resident_launches else 0
    print(f"{'ok ' if ok else 'BAD'} m={m} rho={rho} seed={seed} storage={storage}: views {st.builds} rows {st.rows} resident launches {st.resident_launches} | "
          f"trials {s.n_trials} (oracle {so.n_trials}) ifinal {s.ifinal}/{so.ifinal} nodes {len(s.nodes)}/{len(so.nodes)} "
          f"dscore {abs(s.score - so.score) / abs(so.score):.1e} (oracle solve {time.time() - t0:.1f} s)", flush=True)
print(f"{N} cases, {nres} with a resident launch, {bad} BAD")
sys.exit(1 if bad else 0)
