import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from clipper_amd import _abi as abi, synth
import importlib.util
spec = importlib.util.spec_from_file_location('t', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/test_gpu_configs.py'))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
for m in [int(a) for a in sys.argv[1:]]:
    p = synth.make_euclidean_problem(m, 0.95, seed=12345)
    prm = synth.EUCLID_BENCH_PARAMS
    g = abi.HipClipper(storage=abi.STORE_F64_CSC)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **prm)
    rng = np.random.default_rng(300)
    rows = np.sort(rng.choice(m, size=64, replace=False)).astype(np.int32)
    S = t._reference_score_rows(p, rows, prm["sigma"], prm["epsilon"])
    x = rng.random(m); xm = np.zeros(m); xm[rows] = x[rows]
    oM = S.T @ x[rows]
    zM, zC = g.matvec(xm)
    yM, yC = g.view_matvec(rows, x)
    z2M, z2C = g.matvec(xm)
    bad = np.nonzero(np.abs(zM - oM) > 1e-9)[0]
    print(f"m={m}: |full-ref| {np.abs(zM-oM).max():.3e} |view-ref| {np.abs(yM-oM).max():.3e} |full2-ref| {np.abs(z2M-oM).max():.3e} bad columns {bad.size}: first {bad[:8]} last {bad[-4:] if bad.size else ''}", flush=True)
    if bad.size:
        print("   bad column groups (col//64) histogram head:", np.unique(bad // 64)[:10], "count", np.unique(bad//64).size, "of", (m+63)//64)
    g.close()
