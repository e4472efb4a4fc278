"""CPU tests of the compressed storage's SPECIFICATION (tests/slices_model.py): round trip, quad
padding, step offsets, the work list of a pass (coverage, splitting of heavy slices), and the pass
against the dense product. No GPU, no oracle."""
import numpy as np
import pytest

from tests import slices_model as sm


def _store(m, density, seed, dense_block=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    ld = -(-m // 64) * 64
    U = np.triu((rng.random((m, m)) < density) * rng.uniform(0.05, 1.0, (m, m)), 1)
    if dense_block:                        # the inlier block at the end of the matrix
        k = min(dense_block, m)
        U[m - k:, m - k:] = np.triu(rng.uniform(0.5, 1.0, (k, k)), 1)
    S = np.zeros((m, ld), dtype)
    S[:, :m] = (U + U.T).astype(dtype)
    return S


@pytest.mark.parametrize("m,density", [(1, 0.5), (63, 0.3), (64, 0.1), (65, 0.1), (129, 0.2), (200, 0.0),
                                        (200, 1.0), (333, 0.11)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_round_trip_and_padding(m, density, dtype):
    S = _store(m, density, seed=m, dtype=dtype)
    M = sm.build(S)
    assert np.array_equal(sm.expand(M), S)
    entries = 0
    for cg in range(M.ncg):
        for k in range(M.nchunks):
            s = cg * M.nchunks + k
            nq, steps = sm.read_slice(M, s)
            blk = S[k * 128:(k + 1) * 128, cg * 64:(cg + 1) * 64]
            cnt = np.count_nonzero(blk, axis=0)
            # a lane's list is its column's nonzeros rounded up to whole quads, nothing more
            assert np.array_equal(nq[:len(cnt)], (cnt + 3) // 4)
            assert (M.Lq[s] & 255) == nq.max() and (M.Lq[s] >> 8) == cnt.sum()
            entries += int(cnt.sum())
            for lanes, vals, rows in steps:
                assert np.all(rows[vals == 0] == 0)          # padding is (0, row 0)
                for j in range(len(lanes)):                  # rows ascend inside a quad
                    r = rows[j][vals[j] != 0].astype(int)
                    assert np.all(np.diff(r) > 0)
    assert entries == np.count_nonzero(S)
    # where the slices lie in the arena does not matter
    n = M.ncg * M.nchunks
    M2 = sm.build(S, order=np.random.default_rng(0).permutation(n))
    assert np.array_equal(sm.expand(M2), S)


@pytest.mark.parametrize("target", [1, 7, 40, 1000])
def test_work_list(target):
    S = _store(700, 0.1, seed=5, dense_block=200)
    M = sm.build(S)
    work, nslots = sm.plan(M, target)
    nstrips = -(-M.ncg // 4)
    seen = set()
    for w in work:
        assert (w.strip, w.slot) not in seen and w.slot < nslots
        seen.add((w.strip, w.slot))
    assert len(seen) == nstrips * nslots                     # every (strip, slot) exactly once
    maxq = (M.Lq & 255).reshape(M.ncg, M.nchunks)
    for st in range(nstrips):
        cover = {}
        for w in work:
            if w.strip != st:
                continue
            for k in range(w.t0, w.t1):
                mq = int(maxq[st * 4:(st + 1) * 4, k].max())
                lo, hi = w.q0, min(w.q1, mq)
                if w.t1 - w.t0 > 1:
                    assert w.q0 == 0 and w.q1 >= mq           # only single chunks are split
                assert lo % 16 == 0
                cover.setdefault(k, []).append((lo, hi))
        for k in range(M.nchunks):                           # every step of every chunk exactly once
            mq = int(maxq[st * 4:(st + 1) * 4, k].max())
            segs = sorted(cover.get(k, []))
            pos = 0
            for lo, hi in segs:
                assert lo == pos or (mq == 0 and lo == 0)
                pos = hi
            assert pos == mq
    costs = [sum(int(maxq[w.strip * 4:(w.strip + 1) * 4, k].max()) for k in range(w.t0, w.t1)) for w in work]
    assert work[0].t1 > work[0].t0 or all(c == 0 for c in costs)   # most expensive first


@pytest.mark.parametrize("m,V,dtype", [(130, 1, np.float32), (200, 4, np.float32), (333, 6, np.float32),
                                       (333, 6, np.float64), (420, 8, np.float32)])
def test_pass_equals_dense_product(m, V, dtype):
    S = _store(m, 0.15, seed=m + V, dense_block=300 if m >= 400 else 140, dtype=dtype)
    n = (S.shape[1] // 64) * (-(-m // 128))
    M = sm.build(S, order=np.random.default_rng(1).permutation(n))
    work, nslots = sm.plan(M, 9)
    if m >= 400:
        assert any(w.q0 > 0 for w in work)                   # the dense block's slices are split
    rng = np.random.default_rng(V)
    X = rng.random((m, V))
    d = 0.37
    a, g, b = sm.pass_window(M, work, nslots, X, d)
    Sd = S[:, :m].astype(np.float64)
    Cd = (Sd != 0).astype(np.float64)
    assert np.allclose(a, Sd.T @ X[:, 0], rtol=1e-13, atol=1e-13)
    assert np.allclose(b, Cd.T @ X[:, 0], rtol=1e-13, atol=1e-13)
    for v in range(1, V):
        assert np.allclose(g[v - 1], (Sd + d * Cd).T @ X[:, v], rtol=1e-13, atol=1e-13)
