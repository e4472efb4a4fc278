"""Differential fuzz of the GPU path against the oracle, in the suite: 60 random small problems
(fixed seed) — sizes on and off the slice edges, outlier ratios, invariant and solver parameters,
window sizes — each in ALL four storages, and with the automatic window the compressed storages
both ways: resident (one launch) and streaming launches.

What is asserted, without waivers:
  * every storage: the GPU result equals the oracle's on the matrix that storage holds (the
    GPU's own get_affinity_matrix(), i.e. fp32-rounded values for the fp32 storages): selected
    set, ifinal, objective to 1e-6 relative. Whatever differs there differs by summation order.
  * the fp64 storages (dense and slices): additionally equal to the plain fp64 oracle — the
    parity-exact modes (their matrix differs from the oracle's by <= 4 ulp of exp()).
The fp32 storages are also compared with the plain fp64 oracle and the outcome is LOGGED (not
asserted): that is the price of storing M in fp32, which CLIPPER_HIP_STORE_F64_CSC exists to avoid.
The log goes to gpurun_out/fuzz_parity.log (copied to profiles/ by hand)."""
import os

import numpy as np
import pytest

from clipper_amd import _abi as abi
from clipper_amd import synth
from oracle import clipper_ref as ref

pytestmark = pytest.mark.gpu

N_CASES = 60
SEED = 20260926
STORAGES = {abi.STORE_F32: "f32", abi.STORE_F64: "f64", abi.STORE_F32_CSC: "f32_csc", abi.STORE_F64_CSC: "f64_csc"}


def _same(a, b):
    return (sorted(a.nodes.tolist()) == sorted(b.nodes.tolist()) and a.ifinal == b.ifinal and
            abs(a.score - b.score) <= 1e-6 * max(1.0, abs(b.score)))


def _cases():
    rng = np.random.default_rng(SEED)
    for case in range(N_CASES):
        m = int(rng.choice([37, 64, 100, 129, 200, 333, 512, 700, 1000, 1500]))
        rho = float(rng.choice([0.0, 0.3, 0.6, 0.8, 0.9, 0.95]))
        if int(round(m * (1 - rho))) < 2:
            rho = 0.5
        kw = dict(tol_u=float(rng.choice([1e-8, 1e-6])), tol_F=float(rng.choice([1e-9, 1e-7])),
                  maxiniters=int(rng.choice([200, 200, 50, 20])), maxoliters=int(rng.choice([1000, 1000, 3])),
                  beta=float(rng.choice([0.25, 0.5, 0.1])), maxlsiters=int(rng.choice([99, 99, 3, 1])),
                  rescale_u0=bool(rng.integers(0, 2)),
                  rounding=int(rng.choice([abi.ROUNDING_NONZERO, abi.ROUNDING_DSD_HEU, abi.ROUNDING_DSD_HEU])))
        if kw["maxlsiters"] < 99:   # a crippled line search never converges: bound the homotopy
            kw["maxoliters"] = min(kw["maxoliters"], 10)
        inv = dict(sigma=float(rng.choice([0.01, 0.015, 0.05])), epsilon=float(rng.choice([0.02, 0.05, 0.2])),
                   mindist=float(rng.choice([0.0, 0.0, 0.05])))
        V = int(rng.choice([0, 0, 1, 4, 6, 8]))   # 0 = automatic (the compressed storages then solve resident)
        yield case, m, rho, kw, inv, V, int(rng.integers(1 << 30))


def test_fuzz_against_the_oracle_on_the_stored_matrix():
    lines, failures = [], []
    differs_from_f64_oracle = {}
    for case, m, rho, kw, inv, V, pseed in _cases():
        p = synth.make_euclidean_problem(m, rho, seed=pseed)
        r = ref.RefClipper(ref.Params(**kw))
        r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **inv)
        sr = r.solve(p.u0)
        runs = [(st, nm, 0) for st, nm in STORAGES.items()]
        if V == 0:  # the resident solver took the compressed storages: the streaming launches too
            runs += [(st, nm + "/streaming", 1) for st, nm in STORAGES.items() if nm.endswith("csc")]
        for storage, name, mode in runs:
            g = abi.HipClipper(abi.Params(**kw), storage=storage)
            g.set_window(V)
            g.set_resident(mode)
            g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **inv)
            sg = g.solve(p.u0)
            if mode == 1:
                assert g.last_solver == 0
            rs = ref.RefClipper(ref.Params(**kw))          # the oracle on what this storage holds
            rs.set_matrix_data(g.get_affinity_matrix(), g.get_constraint_matrix())
            ss = rs.solve(p.u0)
            on_stored, on_f64 = _same(sg, ss), _same(sg, sr)
            differs_from_f64_oracle[name] = differs_from_f64_oracle.get(name, 0) + (0 if on_f64 else 1)
            lines.append(f"case {case:2d} m={m:4d} rho={rho:.2f} V={V} {name:17s} solver={g.last_solver} nodes={len(sg.nodes):4d} "
                         f"score={sg.score:.9f} ifinal={sg.ifinal} trials={sg.n_trials} "
                         f"== oracle(stored M): {on_stored}  == oracle(fp64 M): {on_f64}  {kw} {inv}")
            if not on_stored or (name.startswith("f64") and not on_f64):
                failures.append(lines[-1])
            g.close()
    lines.append(f"{N_CASES} cases x {len(STORAGES)} storages, seed {SEED}: {len(failures)} failures; "
                 f"results that differ from the fp64 oracle on its own matrix: {differs_from_f64_oracle}")
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "fuzz_parity.log"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert not failures, "\n".join(failures)
