#!/usr/bin/env python3
"""Is a build + solve bit-reproducible? Same problem, repeated in one context and in fresh ones."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clipper_amd import _abi as abi, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
storage = {"csc": abi.STORE_F32_CSC, "csc64": abi.STORE_F64_CSC, "f32": abi.STORE_F32}[sys.argv[2] if len(sys.argv) > 2 else "csc"]
p = synth.make_euclidean_problem(m, 0.95, seed=12345)
g = abi.HipClipper(storage=storage)
for rep in range(6):
    if rep == 3:
        g.close()
        g = abi.HipClipper(storage=storage)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    x = np.random.default_rng(5).random(m)
    yM, yC = g.matvec(x)
    s = g.solve(p.u0)
    print(rep, "matvec", hashlib.sha256(yM.tobytes()).hexdigest()[:12], hashlib.sha256(yC.tobytes()).hexdigest()[:12],
          "u", hashlib.sha256(s.u.tobytes()).hexdigest()[:12], "trials", s.n_trials, "passes", s.n_passes,
          "score %.13f" % s.score, "nodes", len(s.nodes), flush=True)
