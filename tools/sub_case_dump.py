#!/usr/bin/env python3
"""One problem of tools/sub_random_ab.py solved on the routes without and with the live sub-problem (and without row views
at all); u, the node lists and the counts of every route go to an .npz for a comparison with the oracle's answer off the box.
  python tools/sub_case_dump.py m rho seed storage out.npz [key=value ...]   (solver parameters as in clipper::Params; pn=1: PointNormal)
With cut=K the solve is also cut off after k = 1 .. K outer iterations (maxoliters = k) on every route: where do they part?
With sweep=a,b,c the solve is repeated with tol_F = a, b, c on every route: does the odd route out stay the same one?"""
import sys
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth

m, rho, seed, storage, out = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
kw = {}
cut = pn = 0
sweep = []
for a in sys.argv[6:]:
    k, v = a.split("=")
    if k == "cut":
        cut = int(v)
    elif k == "pn":
        pn = int(v)
    elif k == "sweep":
        sweep = [float(x) for x in v.split(",")]
    else:
        kw[k] = float(v) if k in ("beta", "tol_u", "tol_F", "eps") else int(v)
p = synth.make_pointnormal_problem(m, rho, seed=seed) if pn else synth.make_euclidean_problem(m, rho, seed=seed)


def score(g):
    if pn:
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
    else:
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)


res = {}
for name, view, sub in (("noview", 1, 1), ("views", 0, 1), ("sub", 0, 0)):   # (set_row_view / set_subproblem: 0 = on, 1 = never)
    g = abi.HipClipper(storage=storage)
    g.set_row_view(view)
    g.set_subproblem(sub)
    for key, val in kw.items():
        setattr(g.params, key, val)
    score(g)
    s = g.solve(p.u0)
    st = g.view_stats()
    print(name, "nodes", len(s.nodes), "score", repr(s.score), "ifinal", s.ifinal, "trials", s.n_trials, "passes", s.n_passes,
          "sub entries", st.sub_entries, "sub passes", st.sub_passes, "sub rows", st.sub_rows, flush=True)
    res[name + "_nodes"] = np.asarray(s.nodes)
    res[name + "_u"] = np.asarray(s.u)
    res[name + "_score"] = s.score
    g.close()
np.savez(out, **res)
a, b, c = (set(res[k + "_nodes"].tolist()) for k in ("noview", "views", "sub"))
print("noview ^ views", sorted(a ^ b), " noview ^ sub", sorted(a ^ c), " views ^ sub", sorted(b ^ c))
for k in ("views", "sub"):
    print(k, "max|du| against noview", float(np.max(np.abs(res[k + "_u"] - res["noview_u"]))))

if cut or sweep:
    ctx = {}
    for name, view, sub in (("noview", 1, 1), ("views", 0, 1), ("sub", 0, 0)):
        g = abi.HipClipper(storage=storage)
        g.set_row_view(view)
        g.set_subproblem(sub)
        for key, val in kw.items():
            setattr(g.params, key, val)
        score(g)
        ctx[name] = g
    for k in range(1, cut + 1):
        base = None
        for name, g in ctx.items():
            g.params.maxoliters = k
            s = g.solve(p.u0)
            st = g.view_stats()
            if base is None:
                base = s
            print(f"k={k} {name:7s} trials {s.n_trials:4d} passes {s.n_passes:4d} ifinal {s.ifinal} score {s.score!r} nodes {len(s.nodes)} "
                  f"sub entries {st.sub_entries} passes on it {st.sub_passes} | max|du| vs noview {float(np.max(np.abs(s.u - base.u))):.2e} "
                  f"node sets {'equal' if set(s.nodes.tolist()) == set(base.nodes.tolist()) else 'DIFFER'}", flush=True)
    for tol_F in sweep:
        base = None
        row = []
        for name, g in ctx.items():
            g.params.maxoliters = kw.get("maxoliters", 1000)
            g.params.tol_F = tol_F
            s = g.solve(p.u0)
            if base is None:
                base = s
            row.append(f"{name} trials {s.n_trials} passes {s.n_passes} nodes {len(s.nodes)} "
                       f"{'=' if set(s.nodes.tolist()) == set(base.nodes.tolist()) else 'DIFFER'} max|du| {float(np.max(np.abs(s.u - base.u))):.1e}")
        print(f"tol_F {tol_F:.2e}: " + " | ".join(row), flush=True)
    for g in ctx.values():
        g.close()
