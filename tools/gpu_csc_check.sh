#!/bin/bash
# compressed-copy solver vs dense solver: same selected set, objective, counters; then bench A/B
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python - > $OUT/check.log 2>&1 <<PY
import numpy as np, torch, time
from clipper_amd import _abi as abi, synth
abi.load_library()
for m, rho in [(300, 0.9), (1000, 0.9), (2500, 0.95), (6000, 0.95), (10000, 0.95)]:
    p = synth.make_euclidean_problem(m, rho, seed=7)
    res = {}
    for name, st in [("f32", abi.STORE_F32), ("csc", abi.STORE_F32_CSC)]:
        g = abi.HipClipper(storage=st)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        t0 = time.perf_counter()
        sol = g.solve(p.u0)
        dt = time.perf_counter() - t0
        res[name] = (sol, dt, g.timings().gemv_bytes)
    a, b = res["f32"][0], res["csc"][0]
    same = np.array_equal(np.sort(a.nodes), np.sort(b.nodes))
    print(m, "same_set", same, "nodes", len(a.nodes), "score rel", abs(a.score - b.score) / abs(a.score),
          "passes", a.n_passes, b.n_passes, "trials", a.n_trials, b.n_trials,
          "ms", round(res["f32"][1] * 1e3, 3), round(res["csc"][1] * 1e3, 3),
          "bytes", res["f32"][2], res["csc"][2], flush=True)
PY
cat $OUT/check.log
for st in f32 csc f32 csc; do timeout 300 python bench.py --storage $st --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$st', 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], d['roofline'])" 2>&1 | tee -a $OUT/bench.log; done
