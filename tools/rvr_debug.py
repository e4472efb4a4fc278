import sys, os
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth
m, rho, seed = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
p = synth.make_euclidean_problem(m, rho, seed=seed)
res = {}
for mode in (0, 2):
    g = abi.HipClipper(storage=abi.STORE_F32_CSC)
    g.set_row_view(mode)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = g.solve(p.u0)
    st = g.view_stats()
    print(f"mode {mode}: passes {s.n_passes} trials {s.n_trials} ifinal {s.ifinal} score {s.score!r} nodes {len(s.nodes)} | views {st.builds} rows {st.rows} "
          f"view passes {st.view_passes} resident launches {st.resident_launches}", flush=True)
    res[mode] = s
    g.close()
print("same nodes:", res[0].nodes.tolist() == res[2].nodes.tolist(), "max |du|", float(np.abs(res[0].u - res[2].u).max()))
