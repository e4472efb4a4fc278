"""CPU tests of the compressed storage's SPECIFICATION (tests/csc_model.py): round trip, padding,
tile plan, and the pass against the dense product. No GPU, no oracle."""
import numpy as np
import pytest

from tests import csc_model as cm


def _store(m, density, seed, dense_block=0):
    rng = np.random.default_rng(seed)
    ld = -(-m // 64) * 64
    U = np.triu((rng.random((m, m)) < density) * rng.uniform(0.05, 1.0, (m, m)), 1)
    if dense_block:                        # the inlier block at the end of the matrix
        k = dense_block
        U[m - k:, m - k:] = np.triu(rng.uniform(0.5, 1.0, (k, k)), 1)
    S = np.zeros((m, ld), np.float32)
    S[:, :m] = (U + U.T).astype(np.float32)
    return S


@pytest.mark.parametrize("m,density", [(1, 0.5), (63, 0.3), (64, 0.1), (65, 0.1), (129, 0.2), (200, 0.0),
                                        (200, 1.0), (333, 0.11)])
def test_round_trip_and_padding(m, density):
    S = _store(m, density, seed=m)
    M = cm.build(S)
    assert np.array_equal(cm.expand(M), S)
    assert np.all(M.Lc % 4 == 0)
    # every group is exactly as long as its longest column, rounded up to 4; padding is (0, 0.0)
    nnz_stored = int(np.count_nonzero(M.vals))
    assert nnz_stored == int(np.count_nonzero(S))
    assert np.all(M.rows[M.vals == 0] == 0)
    for s in range(M.nstrips):
        for b in range(M.nblocks):
            blk = S[b * 64:(b + 1) * 64, s * 128:(s + 1) * 128]
            longest = int(np.count_nonzero(blk, axis=0).max(initial=0))
            assert M.Lc[s * M.nblocks + b] == (longest + 3) // 4 * 4
    # the order in which groups claim their space does not matter
    G = M.nstrips * M.nblocks
    M2 = cm.build(S, order=np.random.default_rng(0).permutation(G))
    assert np.array_equal(cm.expand(M2), S)


@pytest.mark.parametrize("target", [1, 7, 40, 1000])
def test_tile_plan(target):
    S = _store(700, 0.1, seed=5, dense_block=150)
    M = cm.build(S)
    tb = cm.plan_tiles(M, target)
    assert np.all(tb[:, 0] == 0) and np.all(tb[:, -1] == M.nblocks)
    assert np.all(np.diff(tb, axis=1) >= 0)
    cost = M.Lc.reshape(M.nstrips, M.nblocks).astype(float) + 2.0
    for s in range(M.nstrips):
        real = [(tb[s, t], tb[s, t + 1]) for t in range(tb.shape[1] - 1) if tb[s, t + 1] > tb[s, t]]
        sums = [cost[s, a:b].sum() for a, b in real]
        assert abs(sum(sums) - cost[s].sum()) < 1e-9
        # no tile exceeds its share by more than one block
        assert max(sums) <= cost[s].sum() / len(real) + cost[s].max() + 1e-9 or len(real) == 1


@pytest.mark.parametrize("m,V", [(130, 1), (200, 4), (333, 6)])
def test_pass_equals_dense_product(m, V):
    S = _store(m, 0.15, seed=m + V, dense_block=40)
    M = cm.build(S, order=np.random.default_rng(1).permutation((-(-S.shape[1] // 128)) * (-(-m // 64))))
    tb = cm.plan_tiles(M, 9)
    rng = np.random.default_rng(V)
    X = rng.random((m, V))
    d = 0.37
    a, g, b = cm.pass_window(M, tb, X, d)
    Sd = S[:, :m].astype(np.float64)
    Cd = (Sd != 0).astype(np.float64)
    assert np.allclose(a, Sd.T @ X[:, 0], rtol=1e-13, atol=1e-13)
    assert np.allclose(b, Cd.T @ X[:, 0], rtol=1e-13, atol=1e-13)
    for v in range(1, V):
        assert np.allclose(g[v - 1], (Sd + d * Cd).T @ X[:, v], rtol=1e-13, atol=1e-13)
