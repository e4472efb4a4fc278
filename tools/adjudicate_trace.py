#!/usr/bin/env python3
"""Where do the GPU routes and the oracles part ways on the two problems of profiles/r05_maxiniters_adjudication.md on
which BOTH oracles agreed and every GPU route did not (m = 8295, m = 9132; `maxiniters = 2`)?  

The solve is cut off after k = 1, 2, ... outer iterations (`maxoliters = k`: the state after k penalty updates) and
(d, F, trials) are compared per k between
  * the C++ oracle in its three summation modes — the reference's order, the same additions swept backwards, every
    output accumulated in extended precision (oracle/clipper_ref.h: clipper_ref_set_sum_mode),
  * `numpy_solve` on a scipy CSR product (rows summed left to right),
  * the GPU with fp64 values: no views, streamed views, the resident solver on a view.
The first k at which two of them differ in their trial count, and the relative size of the differences in d and F in
front of it, say whether a route computes something else (a defect) or the same thing to rounding (chaos).

  python tools/adjudicate_trace.py [--kmax 22] [--no-gpu] > profiles/r06_adjudication_trace.txt"""
import argparse
import sys

sys.path.insert(0, ".")
import numpy as np

CASES = [
    dict(m=8295, rho=0.97, seed=759923, kw=dict(beta=0.25, maxlsiters=20, maxiniters=2, maxoliters=1000, tol_u=1e-8, tol_F=1e-7,
                                                 rescale_u0=0, eps=1e-7)),
    dict(m=9132, rho=0.985, seed=944071, kw=dict(beta=0.1, maxlsiters=20, maxiniters=2, maxoliters=40, tol_u=1e-6, tol_F=1e-7,
                                                  rescale_u0=1, eps=1e-9)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kmax", type=int, default=22)
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    from clipper_amd import synth
    from oracle import clipper_ref as ref
    if not a.no_gpu:
        from clipper_amd import _abi as abi
    for c in CASES:
        p = synth.make_euclidean_problem(c["m"], c["rho"], seed=c["seed"])
        print(f"\n=== m = {c['m']}, rho = {c['rho']}, seed = {c['seed']}, {c['kw']}")
        routes = {}
        r = ref.RefClipper()
        r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)

        def oracle(mode):
            def run(k):
                kw = dict(c["kw"])
                kw["maxoliters"] = min(k, c["kw"]["maxoliters"])
                r.params = ref.Params(**kw)
                r.set_sum_mode(mode)
                s = r.solve(p.u0)
                r.set_sum_mode(0)
                return s
            return run
        routes["oracle"] = oracle(0)
        routes["oracle, reversed"] = oracle(1)
        routes["oracle, extended"] = oracle(2)
        if not a.no_gpu:
            def gpu(mode):
                g = abi.HipClipper(storage=abi.STORE_F64_CSC)
                g.set_row_view(mode)
                g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)

                def run(k):
                    kw = dict(c["kw"])
                    kw["maxoliters"] = min(k, c["kw"]["maxoliters"])
                    g.params = abi.Params(**kw)
                    return g.solve(p.u0)
                return run
            routes["gpu f64, no views"] = gpu(1)
            routes["gpu f64, views streamed"] = gpu(2)
            routes["gpu f64, resident on a view"] = gpu(0)
        names = list(routes)
        print("k | " + " | ".join(f"{n}: ifinal, trials, d, F" for n in names))
        first = None
        prev = None
        for k in range(1, a.kmax + 1):
            row = {n: routes[n](k) for n in names}
            cells = [f"{row[n].ifinal:2d} {row[n].n_trials:4d} {row[n].d:.12e} {row[n].score:.12e}" for n in names]
            print(f"{k:2d} | " + " | ".join(cells))
            trials = {n: row[n].n_trials for n in names}
            if first is None and len(set(trials.values())) > 1:
                first = k
                base = row["oracle"]
                print(f"   ^ first k with different trial counts: {trials}")
                if prev is not None:
                    pb = prev["oracle"]
                    for n in names[1:]:
                        print(f"     in front of it (k = {k - 1}) {n}: rel dd = {abs(prev[n].d - pb.d) / abs(pb.d):.3e}, "
                              f"rel dF = {abs(prev[n].score - pb.score) / abs(pb.score):.3e}, "
                              f"max |du| = {float(np.max(np.abs(np.asarray(prev[n].u) - np.asarray(pb.u)))):.3e}")
                for n in names[1:]:
                    print(f"     at it (k = {k}) {n}: rel dd = {abs(row[n].d - base.d) / abs(base.d):.3e}, "
                          f"rel dF = {abs(row[n].score - base.score) / abs(base.score):.3e}")
            prev = row
            if all(row[n].ifinal < k for n in names):   # every route has converged before k outer iterations
                break
        fin = {n: (prev[n].ifinal, prev[n].n_trials, sorted(prev[n].nodes.tolist())) for n in names}
        sets = {n: fin[n][2] == fin["oracle"][2] for n in names}
        print(f"final: " + "; ".join(f"{n}: ifinal {fin[n][0]}, {fin[n][1]} trials, the oracle's node set: {sets[n]}, "
                                      f"F {prev[n].score:.9f}" for n in names))


if __name__ == "__main__":
    main()
