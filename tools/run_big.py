#!/usr/bin/env python3
"""One registration problem too large for a dense M on one GPU (BASELINE.json configs[4]:
m = 300 000, 360 GB dense fp32) on ONE MI355X with the compressed storage: affinity build +
solve, precision / recall against the ground truth, memory in use. No CPU oracle at this size.
  python tools/run_big.py --m 300000 [--reps 1]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from clipper_amd import _abi as abi  # noqa: E402
from clipper_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=300000)
    ap.add_argument("--rho", type=float, default=0.95)
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    abi.load_library()
    t0 = time.perf_counter()
    p = synth.make_euclidean_problem(a.m, a.rho)
    t_gen = time.perf_counter() - t0
    g = abi.HipClipper(storage=abi.STORE_F32_CSC)
    g.stage_inputs(p.D1, p.D2, p.A)
    g.set_profiling(True)
    rows = []
    for rep in range(a.reps + 1):   # the first build sizes the buffers (two fills)
        t0 = time.perf_counter()
        g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
        if rep == 0:
            g.stage_u0(p.u0)
        t1 = time.perf_counter()
        sol = g.solve_staged()
        t2 = time.perf_counter()
        tm = g.timings()
        free, total = torch.cuda.mem_get_info()
        prec, rec = synth.precision_recall(p.A[sol.nodes], p.Agt)
        rows.append(dict(m=a.m, rho=a.rho, rep=rep, storage_in_use=g.storage_in_use,
                         affinity_ms=round((t1 - t0) * 1e3, 2), affinity_kernel_ms=round(tm.affinity_kernel_ms, 2),
                         solve_ms=round((t2 - t1) * 1e3, 2), passes=int(sol.n_passes), trials=int(sol.n_trials),
                         pass_us=round(tm.gemv_avg_us, 1), pass_bytes=tm.gemv_bytes,
                         pass_GBps=round(tm.gemv_bytes / max(tm.gemv_avg_us, 1e-9) * 1e-3, 1),
                         dense_bytes=4.0 * a.m * a.m, hbm_used_GB=round((total - free) / 2**30, 1),
                         score=sol.score, nodes=int(len(sol.nodes)), ifinal=int(sol.ifinal),
                         precision=round(prec, 4), recall=round(rec, 4), gen_s=round(t_gen, 1)))
        print(json.dumps(rows[-1]), flush=True)
    g.close()


if __name__ == "__main__":
    main()
