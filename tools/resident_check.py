#!/usr/bin/env python3
"""Resident (one-launch) solver against the streaming launches and the CPU oracle on small problems:
selected set, trial count, objective, wall time of solve_staged.
  python tools/resident_check.py [--sizes 100,300,1000,2048,3000] [--reps 7] [--storage csc] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from clipper_amd import _abi as abi  # noqa: E402
from clipper_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100,300,1000,2048,3000")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--storage", default="csc")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    storage = {"csc": abi.STORE_F32_CSC, "csc64": abi.STORE_F64_CSC}[a.storage]
    for m in [int(x) for x in a.sizes.split(",")]:
        rho = 0.9 if m <= 5000 else 0.95
        p = synth.make_euclidean_problem(m, rho)
        inv = synth.EUCLID_BENCH_PARAMS
        row = dict(m=m, storage=a.storage)
        ref_nodes = None
        if not a.no_cpu:
            from oracle import clipper_ref as ref
            r = ref.RefClipper()
            r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **inv)
            t0 = time.perf_counter()
            sr = r.solve(p.u0)
            row.update(cpu_solve_ms=round((time.perf_counter() - t0) * 1e3, 3), cpu_trials=int(sr.n_trials),
                       cpu_score=sr.score)
            ref_nodes = sr.nodes.tolist()
        for mode, name in ((1, "streaming"), (0, "resident")):
            g = abi.HipClipper(storage=storage)
            g.set_resident(mode)
            g.stage_inputs(p.D1, p.D2, p.A)
            g.affinity_euclidean_staged(**inv)
            g.stage_u0(p.u0)
            sol = g.solve_staged()
            ts, ta = [], []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                g.affinity_euclidean_staged(**inv)
                t1 = time.perf_counter()
                sol = g.solve_staged()
                ts.append((time.perf_counter() - t1) * 1e3)
                ta.append((t1 - t0) * 1e3)
            row[name] = dict(solver=g.last_solver, solve_ms=round(float(np.median(ts)), 4),
                             solve_min_ms=round(float(np.min(ts)), 4),
                             affinity_ms=round(float(np.median(ta)), 4), passes=int(sol.n_passes),
                             trials=int(sol.n_trials), score=sol.score, nodes=len(sol.nodes),
                             ifinal=int(sol.ifinal),
                             same_as_cpu=(sol.nodes.tolist() == ref_nodes) if ref_nodes is not None else None)
            row[name + "_nodes_hash"] = hash(tuple(sol.nodes.tolist())) & 0xffffffff
            g.close()
        row["same_set"] = row["streaming_nodes_hash"] == row["resident_nodes_hash"]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
