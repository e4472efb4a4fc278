#!/usr/bin/env python3
"""Runs the BASELINE.json configurations on one MI355X next to the CPU oracle and prints one
JSON row per configuration (the table of BASELINE.md section 3).
  python tools/run_configs.py [--configs 1k,10k,pn5k,100k] [--reps 5] [--no-cpu]
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

CONFIGS = {
    # cfg1: 600-point sample of the reference's bunny model, ex4_bunny.ipynb recipe scaled by 0.1
    "bunny": dict(kind="bunny", m=100, rho=0.90),
    "1k": dict(kind="euclid", m=1000, rho=0.90),
    "10k": dict(kind="euclid", m=10000, rho=0.95),
    "pn5k": dict(kind="pointnormal", m=5000, rho=0.90),
    "30k": dict(kind="euclid", m=30000, rho=0.95),
    "100k": dict(kind="euclid", m=100000, rho=0.95),
    "300k": dict(kind="euclid", m=300000, rho=0.95),
}


def bunny_problem(m, rho, seed=0):
    from clipper_amd import registration as reg
    pts = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_points.json")))["points"])
    rng = np.random.default_rng(1000 + seed)
    T = np.eye(4)
    T[:3, :3] = reg.random_rotation(rng)
    T[:3, 3] = rng.uniform(-5, 5, 3)
    D1, D2, A, Agt = reg.make_registration_dataset(pts, m, m, m // 4, rho, 0.01, T, seed=seed)
    u0 = np.random.default_rng(seed + 1).random(m)
    p = synth.Problem(D1=D1, D2=D2, A=A, Agt=Agt, u0=u0, meta=dict(kind="bunny", T=T.tolist()))
    return p, dict(sigma=0.01, epsilon=0.02, mindist=0.0)


def run(name, cfg, reps, storage, with_cpu):
    m, rho = cfg["m"], cfg["rho"]
    if cfg["kind"] == "bunny":
        p, inv = bunny_problem(m, rho)
    elif cfg["kind"] == "euclid":
        p = synth.make_euclidean_problem(m, rho)
        inv = synth.EUCLID_BENCH_PARAMS
    else:
        p = synth.make_pointnormal_problem(m, rho)
        inv = p.meta["invariant"]
    g = abi.HipClipper(storage=storage)
    g.stage_inputs(p.D1, p.D2, p.A)

    def affinity():
        if cfg["kind"] in ("euclid", "bunny"):
            g.affinity_euclidean_staged(**inv)
        else:
            g.affinity_pointnormal_staged(**inv)

    affinity()
    g.stage_u0(p.u0)
    g.solve_staged()
    g.set_profiling(True)
    ta, ts, gemv, vgemv = [], [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        affinity()
        t1 = time.perf_counter()
        sol = g.solve_staged()
        t2 = time.perf_counter()
        ta.append((t1 - t0) * 1e3)
        ts.append((t2 - t1) * 1e3)
        gemv.append(g.timings().gemv_avg_us)
        vgemv.append(g.view_stats().view_pass_avg_us)
    tm = g.timings()
    vs = g.view_stats()
    row = dict(config=name, kind=cfg["kind"], m=m, rho=rho,
               storage={abi.STORE_F32: "f32", abi.STORE_F64: "f64", abi.STORE_F32_CSC: "csc",
                        abi.STORE_F64_CSC: "csc64"}[storage],
               pass_bytes=tm.gemv_bytes, useful_bytes=tm.gemv_useful_bytes, window=g.window,
               gpu_affinity_ms=round(float(np.median(ta)), 4), gpu_solve_ms=round(float(np.median(ts)), 4),
               passes=int(sol.n_passes), solver=["streaming", "resident"][g.last_solver],
               # the resident solver is one launch: no per-pass timings
               gemv_us=round(float(np.median(gemv)), 2) if np.median(gemv) > 0 else None,
               gemv_GBps=round(tm.gemv_bytes / float(np.median(gemv)) * 1e-3, 1) if np.median(gemv) > 0 else None,
               frac_of_8TBps=round(tm.gemv_bytes / float(np.median(gemv)) / 8e6, 4) if np.median(gemv) > 0 else None,
               # the row views of the solve (passes that streamed M[live rows, :] instead of M)
               passes_on_view=int(vs.view_passes), views_built=int(vs.builds), view_rows=int(vs.rows),
               view_bytes=int(vs.bytes), view_build_ms=round(vs.build_ms, 4),
               view_pass_us=round(float(np.median(vgemv)), 2) if np.median(vgemv) > 0 else None,
               # the live sub-problem (the passes behind the view that ran on M[S,S])
               passes_on_sub=int(vs.sub_passes), sub_rows=int(vs.sub_rows), sub_bytes=int(vs.sub_bytes),
               sub_entries=int(vs.sub_entries), sub_leaves=int(vs.sub_leaves), sub_build_ms=round(vs.sub_build_ms, 4),
               sub_pass_us=round(vs.sub_pass_avg_us, 2) if vs.sub_pass_samples > 0 else None,
               trials=int(sol.n_trials),
               score=sol.score, nodes=int(len(sol.nodes)), ifinal=int(sol.ifinal))
    prec, rec = synth.precision_recall(g.get_selected_associations(), p.Agt)
    row.update(precision=round(prec, 4), recall=round(rec, 4))
    if with_cpu:
        from oracle import clipper_ref as ref
        r = ref.RefClipper()
        t0 = time.perf_counter()
        if cfg["kind"] in ("euclid", "bunny"):
            r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **inv)
        else:
            r.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A, **inv)
        t1 = time.perf_counter()
        sr = r.solve(p.u0)
        t2 = time.perf_counter()
        row.update(cpu_affinity_ms=round((t1 - t0) * 1e3, 2), cpu_threads=ref.omp_threads(),
                   cpu_solve_ms=round((t2 - t1) * 1e3, 2), cpu_passes=int(sr.n_passes),
                   nnz_upper=int(r.nnz), density=round(r.nnz / (m * (m - 1) / 2), 4),
                   set_identical=bool(sr.nodes.tolist() == sol.nodes.tolist()),
                   rel_dscore=abs(sr.score - sol.score) / abs(sr.score))
    g.close()
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1k,10k,pn5k")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    storage = {"f32": abi.STORE_F32, "f64": abi.STORE_F64, "csc": abi.STORE_F32_CSC,
               "csc64": abi.STORE_F64_CSC}[a.storage]
    for name in a.configs.split(","):
        run(name, CONFIGS[name], a.reps, storage, not a.no_cpu)


if __name__ == "__main__":
    main()
