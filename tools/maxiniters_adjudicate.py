#!/usr/bin/env python3
"""Adjudication of the `maxiniters = 2` divergences of profiles/r04_rvr_random_ab.txt:
the random problems on which two GPU paths ended on a different `ifinal` / objective are solved again by BOTH oracles —
the C++ restatement (`ref.RefClipper`, the reference's own product order: the upper triangle column by column) and
`ref.numpy_solve` on a scipy CSR matrix (the same algorithm, rows summed left to right) — and by the three GPU paths
with fp64 values (`F64_CSC`: the matrix is the oracle's up to 4 ulp of exp()).

  python tools/maxiniters_adjudicate.py cases  [log]            -> profiles/r05_adj_cases.json  (parses the log)
  python tools/maxiniters_adjudicate.py cpu    [first] [count]  -> profiles/r05_adj_cpu_<first>.json   (no GPU; minutes per case)
  python tools/maxiniters_adjudicate.py gpu                     -> gpurun_out/adj_gpu.json (copied to profiles/r05_adj_gpu.json) (needs the GPU; seconds per case)
  python tools/maxiniters_adjudicate.py table                   -> markdown on stdout (profiles/r05_maxiniters_adjudication.md)

If the two oracles part ways on a problem, the reference's answer there depends on the order of its own sums and no
implementation can be "bit-identical" to it; if they agree and a GPU path does not, the path has a defect."""
import glob
import json
import os
import re
import sys
import time

sys.path.insert(0, '.')
import numpy as np

OUT = "gpurun_out"
LINE = re.compile(r"^(?:DIFFERENT RESULT: )?(ok |BAD) m=(\d+) rho=([\d.]+) seed=(\d+) storage=(\d): .*?"
                  r"trials (\d+)/(\d+) ifinal (\d+)/(\d+) dscore ([\d.e+-]+) (\{.*\})\s*$")


def parse_cases(path):
    """Every line of the two parameter blocks whose two runs ended differently (ifinal or objective beyond 1e-8)."""
    cases, block = {}, None
    for line in open(path):
        if line.startswith("==========="):
            block = ("resident_vs_streamed" if "RANDOM SOLVER PARAMETERS" in line else
                     "streamed_vs_noviews" if "streamed views against no views" in line else None)
            continue
        mt = LINE.match(line)
        if not mt or block is None:
            continue
        _, m, rho, seed, storage, t1, t2, i1, i2, ds, kw = mt.groups()
        if int(i1) == int(i2) and float(ds) <= 1e-8:
            continue
        key = (int(m), float(rho), int(seed))
        c = cases.setdefault(key, {"m": int(m), "rho": float(rho), "seed": int(seed), "kw": eval(kw), "seen": []})
        c["seen"].append({"block": block, "storage": int(storage), "ifinal": [int(i1), int(i2)], "dscore": float(ds)})
    return sorted(cases.values(), key=lambda c: c["m"])


def load_cases():
    return json.load(open("profiles/r05_adj_cases.json"))


def problem(c):
    from clipper_amd import synth
    return synth, synth.make_euclidean_problem(c["m"], c["rho"], seed=c["seed"])


def sol_record(s, u=None):
    nodes = sorted(int(x) for x in s.nodes.tolist())
    return {"ifinal": int(s.ifinal), "score": float(s.score), "trials": int(s.n_trials), "n_nodes": len(nodes),
            "nodes_hash": hash_nodes(nodes)}


def hash_nodes(nodes):
    import hashlib
    return hashlib.sha256(np.asarray(nodes, dtype=np.int32).tobytes()).hexdigest()[:16]


def run_cpu(first, count):
    import scipy.sparse as sp
    from oracle import clipper_ref as ref
    cases = load_cases()[first:first + count]
    out = []
    for c in cases:
        synth, p = problem(c)
        kw = dict(c["kw"])
        t0 = time.time()
        r = ref.RefClipper(ref.Params(**kw))
        r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        t1 = time.time()
        sc = r.solve(p.u0)
        t2 = time.time()
        # the second oracle: the same loop, scipy's CSR product (each row summed left to right over BOTH triangles)
        # instead of the reference's walk over the upper triangle (clipper.cpp:195 selfadjointView<Upper>)
        M = r.get_affinity_matrix()
        np.fill_diagonal(M, 0.0)
        Ms = sp.csr_matrix(M)
        del M
        Cs = Ms.copy()
        Cs.data[:] = 1.0
        sn = ref.numpy_solve(None, None, p.u0, ref.Params(**kw), matvec=lambda x: (Ms @ x, Cs @ x))
        t3 = time.time()
        rec = {"m": c["m"], "rho": c["rho"], "seed": c["seed"], "kw": kw,
               "oracle_cpp": sol_record(sc), "oracle_numpy": sol_record(sn),
               "seconds": {"affinity": round(t1 - t0, 1), "cpp": round(t2 - t1, 1), "numpy": round(t3 - t2, 1)}}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        json.dump(out, open(f"profiles/r05_adj_cpu_{first:03d}.json", "w"), indent=1)


def run_gpu():
    from clipper_amd import _abi as abi
    out = []
    for c in load_cases():
        synth, p = problem(c)
        rec = {"m": c["m"], "rho": c["rho"], "seed": c["seed"]}
        for storage, sname in ((abi.STORE_F64_CSC, "f64"), (abi.STORE_F32_CSC, "f32")):
            for mode, name in ((0, "resident"), (2, "streamed"), (1, "noviews")):
                g = abi.HipClipper(storage=storage)
                g.set_row_view(mode)
                for key, val in c["kw"].items():
                    setattr(g.params, key, val)
                g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
                s = g.solve(p.u0)
                st = g.view_stats()
                r = sol_record(s)
                r.update(views=int(st.builds), rows=int(st.rows), resident_launches=int(st.resident_launches))
                rec[f"gpu_{sname}_{name}"] = r
                g.close()
        out.append(rec)
        print(json.dumps(rec), flush=True)
        json.dump(out, open(os.path.join(OUT, "adj_gpu.json"), "w"), indent=1)


def table():
    cpu = {}
    for f in sorted(glob.glob("profiles/r05_adj_cpu_*.json")):
        for r in json.load(open(f)):
            cpu[(r["m"], r["seed"])] = r
    gpu = {}
    gp = "profiles/r05_adj_gpu.json" if os.path.exists("profiles/r05_adj_gpu.json") else os.path.join(OUT, "adj_gpu.json")
    if os.path.exists(gp):
        for r in json.load(open(gp)):
            gpu[(r["m"], r["seed"])] = r
    cols = ["gpu_f64_resident", "gpu_f64_streamed", "gpu_f64_noviews", "gpu_f32_resident", "gpu_f32_streamed", "gpu_f32_noviews"]

    def same(a, b):
        return a["ifinal"] == b["ifinal"] and a["nodes_hash"] == b["nodes_hash"] and abs(a["score"] - b["score"]) <= 1e-6 * abs(b["score"])

    print("| m | rho | seed | parameters | C++ oracle: ifinal, score, trials | numpy oracle | oracles agree | "
          + " | ".join(c.replace("gpu_", "") for c in cols) + " |")
    print("|---|---|---|---|---|---|---|" + "---|" * len(cols))
    n_or = n_dis = 0
    agree_cnt = {c: [0, 0] for c in cols}
    both_cnt = {c: [0, 0] for c in cols}   # on the problems where the two oracles agree: equal to them / cases
    for c in load_cases():
        k = (c["m"], c["seed"])
        if k not in cpu:
            continue
        r = cpu[k]
        a, b = r["oracle_cpp"], r["oracle_numpy"]
        ok = same(a, b)
        n_or += 1
        n_dis += 0 if ok else 1
        kw = c["kw"]
        ptxt = f"beta {kw['beta']}, ls {kw['maxlsiters']}, in {kw['maxiniters']}, ol {kw['maxoliters']}, tol_u {kw['tol_u']:g}, tol_F {kw['tol_F']:g}, rescale {kw['rescale_u0']}, eps {kw['eps']:g}"
        cells = []
        for col in cols:
            g = gpu.get(k, {}).get(col)
            if g is None:
                cells.append("—")
                continue
            sa, sb = same(g, a), same(g, b)
            agree_cnt[col][0] += 1 if (sa or sb) else 0
            agree_cnt[col][1] += 1
            if ok:
                both_cnt[col][0] += 1 if (sa and sb) else 0
                both_cnt[col][1] += 1
            cells.append(f"{g['ifinal']}, {g['score']:.6f}, {g['trials']} " + ("= both" if sa and sb else "= C++" if sa else "= numpy" if sb else "**neither**"))
        print(f"| {c['m']} | {c['rho']} | {c['seed']} | {ptxt} | {a['ifinal']}, {a['score']:.6f}, {a['trials']} | "
              f"{b['ifinal']}, {b['score']:.6f}, {b['trials']} | {'yes' if ok else '**NO**'} | " + " | ".join(cells) + " |")
    print()
    print(f"{n_or} problems; the two oracles disagree with each other on {n_dis}.")
    # the north-star contract alone (node SET identical, objective to 1e-6), ifinal left out
    def contract(a, b):
        return a["nodes_hash"] == b["nodes_hash"] and abs(a["score"] - b["score"]) <= 1e-6 * abs(b["score"])
    oc = sum(1 for c in load_cases() if (c["m"], c["seed"]) in cpu and contract(cpu[(c["m"], c["seed"])]["oracle_cpp"], cpu[(c["m"], c["seed"])]["oracle_numpy"]))
    print(f"by the contract alone (same node set, objective within 1e-6; ifinal not compared): the two oracles agree on {oc} of {n_or};")
    for col in cols:
        k1 = sum(1 for c in load_cases() if (c["m"], c["seed"]) in cpu and (c["m"], c["seed"]) in gpu and col in gpu[(c["m"], c["seed"])]
                 and contract(gpu[(c["m"], c["seed"])][col], cpu[(c["m"], c["seed"])]["oracle_cpp"]))
        print(f"  {col} meets it against the C++ oracle on {k1}")
    for col in cols:
        if agree_cnt[col][1]:
            print(f"{col}: equal to at least one oracle on {agree_cnt[col][0]} of {agree_cnt[col][1]}; "
                  f"on the {both_cnt[col][1]} problems where the oracles agree with each other: equal to them on {both_cnt[col][0]}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    cmd = sys.argv[1]
    if cmd == "cases":
        cs = parse_cases(sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_rvr_random_ab.txt")
        json.dump(cs, open("profiles/r05_adj_cases.json", "w"), indent=1)
        print(len(cs), "cases;", sum(1 for c in cs if c["kw"]["maxiniters"] == 2), "with maxiniters = 2")
    elif cmd == "cpu":
        run_cpu(int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 10**6)
    elif cmd == "gpu":
        run_gpu()
    elif cmd == "table":
        table()
