#!/usr/bin/env python3
"""Fastest and mean pass on M of a profiled solve with row views off (every pass streams M), for A/B runs of
library variants (CLIPPER_HIP_LIB): the mean hides a few slow passes, the minimum shows the steady state.
  python tools/pass_min_probe.py 10000 100000"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipper_amd import _abi as abi  # noqa: E402
from clipper_amd import synth  # noqa: E402

for m in [int(x) for x in sys.argv[1:]] or [10000]:
    p = synth.make_euclidean_problem(m, 0.95, seed=12345)
    for views in (0, 1):
        g = abi.HipClipper(storage=abi.STORE_F32_CSC)
        g.set_row_view(0 if views else 1)
        g.set_profiling(True)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        g.solve(p.u0)
        s = g.solve(p.u0)
        t = g.timings()
        v = g.view_stats()
        print(json.dumps(dict(m=m, views=bool(views), solve_ms=round(s.t * 1e3, 4), passes=int(v.passes), trials=int(s.n_trials),
                              pass_on_M_avg_us=round(t.gemv_avg_us, 2), pass_on_M_min_us=round(t.gemv_min_us, 2),
                              pass_on_M_launches=int(t.gemv_launches), view_pass_avg_us=round(v.view_pass_avg_us, 2),
                              lib=os.path.basename(os.environ.get("CLIPPER_HIP_LIB", "product")))), flush=True)
        g.close()
