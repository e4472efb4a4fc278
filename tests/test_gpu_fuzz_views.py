"""Differential fuzz of the solves that use ROW VIEWS (m >= 3000: tests/test_gpu_fuzz.py stays below that)
and of the exact DSD rounding, against the oracle. Random sizes on and off the slice edges, outlier
ratios, invariant and solver parameters, window sizes, both value types of the slices; a third of the
matrices is HANDED OVER (setMatrixData from the oracle's matrix: views then come from the filter,
k_slice_filter_rows, instead of the rectangular fill). Asserted per case, on the matrix the storage
holds: selected set, ifinal, objective to 1e-6 relative — with views and without.
FUZZ_VIEWS_CASES / FUZZ_DSD_CASES / FUZZ_VIEWS_SEED (environment) lengthen or vary the run; the log goes to
gpurun_out/fuzz_views.log."""
import os

import numpy as np
import pytest

from clipper_amd import _abi as abi
from clipper_amd import synth
from oracle import clipper_ref as ref

pytestmark = pytest.mark.gpu

N_VIEWS = int(os.environ.get("FUZZ_VIEWS_CASES", "10"))
N_DSD = int(os.environ.get("FUZZ_DSD_CASES", "8"))
SEED = int(os.environ.get("FUZZ_VIEWS_SEED", "20260927"))


def _same(a, b):
    return (sorted(a.nodes.tolist()) == sorted(b.nodes.tolist()) and a.ifinal == b.ifinal and
            abs(a.score - b.score) <= 1e-6 * max(1.0, abs(b.score)))


def test_fuzz_row_views_against_the_oracle():
    rng = np.random.default_rng(SEED)
    lines, failures, with_views = [], [], 0
    for case in range(N_VIEWS):
        m = int(rng.choice([3000, 3073, 3500, 4097, 5000, 6400, 7777, 9000]))
        rho = float(rng.choice([0.5, 0.8, 0.9, 0.95, 0.97]))
        if rho < 0.6 and m > 4100:   # (half of 9000 associations consistent: a minute of oracle time per case)
            rho = 0.8
        kw = dict(tol_u=float(rng.choice([1e-8, 1e-6])), tol_F=float(rng.choice([1e-9, 1e-7])),
                  maxiniters=int(rng.choice([200, 200, 50])), maxoliters=int(rng.choice([1000, 1000, 4])),
                  beta=float(rng.choice([0.25, 0.5])), rescale_u0=bool(rng.integers(0, 2)))
        inv = dict(sigma=float(rng.choice([0.01, 0.015, 0.03])), epsilon=float(rng.choice([0.03, 0.05, 0.1])),
                   mindist=float(rng.choice([0.0, 0.0, 0.05])))
        V = int(rng.choice([0, 0, 4, 6, 8]))
        storage = int(rng.choice([abi.STORE_F32_CSC, abi.STORE_F32_CSC, abi.STORE_F64_CSC]))
        handed_over = bool(rng.random() < 0.34)
        p = synth.make_euclidean_problem(m, rho, seed=int(rng.integers(1 << 30)))
        res = {}
        for views in (True, False):
            g = abi.HipClipper(abi.Params(**kw), storage=storage)
            g.set_row_view(0 if views else 1)
            g.set_window(V)
            if handed_over:
                src = abi.HipClipper(storage=abi.STORE_F64_CSC if storage == abi.STORE_F64_CSC else abi.STORE_F32_CSC)
                src.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **inv)
                Ms = src.get_affinity_matrix()
                src.close()
                g.set_matrix_data(Ms, (Ms != 0).astype(float))
            else:
                g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **inv)
            s = g.solve(p.u0)
            st = g.view_stats()
            res[views] = (s, int(st.builds), int(st.view_passes), int(st.passes))
            if views:
                rs = ref.RefClipper(ref.Params(**kw))      # the oracle on what this storage holds
                rs.set_matrix_data(g.get_affinity_matrix(), g.get_constraint_matrix())
                ss = rs.solve(p.u0)
            g.close()
        (s1, builds, vpasses, passes), (s0, b0, _, _) = res[True], res[False]
        with_views += 1 if vpasses > 0 else 0
        ok = _same(s1, ss) and _same(s0, ss) and b0 == 0
        lines.append(f"case {case:3d} m={m} rho={rho:.2f} V={V} storage={storage} handed_over={handed_over} builds={builds} "
                     f"view passes {vpasses}/{passes} nodes={len(s1.nodes)} score={s1.score:.9f} ifinal={s1.ifinal} "
                     f"trials {s1.n_trials}/{s0.n_trials}/{ss.n_trials} ok={ok} {kw} {inv}")
        if not ok:
            failures.append(lines[-1])
    lines.append(f"{N_VIEWS} cases, seed {SEED}: {len(failures)} failures, {with_views} solves ran passes on a view")
    _log("fuzz_views.log", lines)
    assert not failures, "\n".join(failures)
    assert with_views * 3 >= N_VIEWS, lines[-1]   # (coverage of the fuzz itself: a third of the solves at least used a view)


def test_fuzz_exact_dsd_rounding_against_the_oracle():
    """Rounding::DSD through the solve (clipper.cpp:294-300): nodes and score against the oracle's solve with
    the oracle's Goldberg procedure (oracle/dsd_ref.py) on the same matrix, and clipper_hip_densest_subgraph
    on random node lists of a sparse-ish graph."""
    from oracle import dsd_ref
    rng = np.random.default_rng(SEED + 1)
    lines, failures = [], []
    for case in range(N_DSD):
        m = int(rng.choice([60, 100, 129, 200, 260, 333]))
        rho = float(rng.choice([0.6, 0.8, 0.9]))
        storage = int(rng.choice([abi.STORE_F64, abi.STORE_F64_CSC]))
        p = synth.make_euclidean_problem(m, rho, seed=int(rng.integers(1 << 30)))
        g = abi.HipClipper(abi.Params(rounding=abi.ROUNDING_DSD), storage=storage)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        sg = g.solve(p.u0)
        M = g.get_affinity_matrix()
        Mup = np.triu(M, 1)
        sr = ref.numpy_solve(Mup, (Mup != 0).astype(float), p.u0, ref.Params(rounding=ref.ROUNDING_DSD))
        ok = sg.nodes.tolist() == sr.nodes.tolist() and abs(sg.score - sr.score) <= 1e-9 * max(1.0, abs(sr.score))
        S = np.sort(rng.choice(m, size=int(rng.integers(5, min(m, 70))), replace=False)).astype(np.int32)
        want = dsd_ref.densest_subgraph(M, S.tolist())
        got = g.densest_subgraph(S).tolist()
        ok = ok and got == want
        lines.append(f"case {case:3d} m={m} rho={rho:.2f} storage={storage} solve nodes {len(sg.nodes)} (oracle {len(sr.nodes)}) "
                     f"list of {S.size}: {len(got)} nodes (oracle {len(want)}) ok={ok}")
        if not ok:
            failures.append(lines[-1])
        g.close()
    lines.append(f"{N_DSD} cases, seed {SEED + 1}: {len(failures)} failures")
    _log("fuzz_dsd.log", lines)
    assert not failures, "\n".join(failures)


def _log(name, lines):
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", name), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
