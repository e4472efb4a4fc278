#!/usr/bin/env python3
"""Soak test of the resident solver's lock-free exchange: many solves of random small problems on ONE
context (changing sizes, value types, repeated solves), every result compared with the streaming
launches; counts unexpected fall-backs (time-outs) and mismatches.
  python tools/resident_soak.py [--seconds 120] [--seed 1]"""
import argparse
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
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=1, help="concurrent host threads, each with its own contexts")
    a = ap.parse_args()
    if a.threads > 1:
        import threading
        ts = [threading.Thread(target=worker, args=(a.seconds, a.seed + 1000 * t, f"[thread {t}] ")) for t in range(a.threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    else:
        worker(a.seconds, a.seed, "")


def worker(seconds, seed, tag):
    class A:
        pass
    a = A()
    a.seconds, a.seed = seconds, seed
    rng = np.random.default_rng(a.seed)
    ctx = {st: (abi.HipClipper(storage=st), abi.HipClipper(storage=st)) for st in (abi.STORE_F32_CSC, abi.STORE_F64_CSC)}
    for gr, gs in ctx.values():
        gs.set_resident(1)
    t0 = time.time()
    n = solves = fallbacks = mismatches = streamed_by_plan = order_diff = trial_diff = max_trial_diff = 0
    max_rel = 0.0
    while time.time() - t0 < a.seconds:
        m = int(rng.integers(2, 2049))
        rho = float(rng.choice([0.5, 0.8, 0.9, 0.95]))
        if round(m * (1 - rho)) < 2:
            rho = 0.0
        st = abi.STORE_F32_CSC if rng.random() < 0.7 else abi.STORE_F64_CSC
        p = synth.make_euclidean_problem(m, rho, seed=int(rng.integers(1 << 30)))
        gr, gs = ctx[st]
        gr.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        gs.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        ss = gs.solve(p.u0)
        first = None
        for rep in range(int(rng.integers(1, 4))):
            sr = gr.solve(p.u0)
            solves += 1
            if first is None:
                first = gr.last_solver
                streamed_by_plan += 1 if first == 0 else 0
            elif gr.last_solver != first:
                fallbacks += 1
                print(f"{tag}fallback: m={m} rho={rho} storage={st} rep={rep}", flush=True)
            # the bar of the path: the same selected SET, the objective to 1e-6 relative
            ok = (sorted(sr.nodes.tolist()) == sorted(ss.nodes.tolist()) and sr.ifinal == ss.ifinal
                  and abs(sr.score - ss.score) <= 1e-6 * max(1.0, abs(ss.score)))
            if not ok:
                mismatches += 1
                print(f"{tag}MISMATCH: m={m} rho={rho} storage={st} resident {sr.score!r}/{sr.n_trials}/{len(sr.nodes)} "
                      f"streaming {ss.score!r}/{ss.n_trials}/{len(ss.nodes)}", flush=True)
            # softer observations: the order of the list (near-equal entries of u), the trial count
            order_diff += 0 if sr.nodes.tolist() == ss.nodes.tolist() else 1
            dt = abs(sr.n_trials - ss.n_trials)
            trial_diff += 1 if dt else 0
            max_trial_diff = max(max_trial_diff, dt)
            max_rel = max(max_rel, abs(sr.score - ss.score) / max(1.0, abs(ss.score)))
        n += 1
    print(f"{tag}{n} problems, {solves} solves in {time.time() - t0:.0f} s: {mismatches} mismatches, {fallbacks} fall-backs after a "
          f"resident solve, {streamed_by_plan} problems left to the streaming launches by the planner; "
          f"list order differs in {order_diff} solves, trial count in {trial_diff} (by at most {max_trial_diff}), "
          f"largest relative difference of the objective {max_rel:.1e}")
    for gr, gs in ctx.values():
        gr.close()
        gs.close()


if __name__ == "__main__":
    main()
