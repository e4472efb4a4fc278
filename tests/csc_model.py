"""Executable model (numpy) of the COMPRESSED STORAGE of M (CLIPPER_HIP_STORE_F32_CSC,
clipper_amd/csrc/kernels.hip.h "column-compressed copy"): the layout the fill kernels write
(csc_emit / csc_emit_lds), the expansion (k_csc_expand), the pass (csc_core) with its summation
order, and the cost-balanced row tiles (csc_plan in clipper_hip.hip). Test infrastructure:
tests/test_csc_model.py checks on the CPU that the format round-trips and that the pass equals
the dense product; the kernels themselves are checked on the GPU (tests/test_gpu_csc.py).

  group g = (strip s of 128 columns, block b of 64 rows), g = s * nblocks + b
  Lc[g]   padded list length (multiple of 4) = longest column list of the group, rounded up
  Pre[g]  start of the group in units of 128 entries
  entry k of column c = 2*lane + e of the group sits at
      base + ((e * LQ + k // 4) * 64 + lane) * 4 + k % 4,   base = Pre[g] * 128, LQ = Lc[g] / 4
  in BOTH flat arrays: vals (float32) and rows (uint8, row inside the block); padding = (0, 0.0)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

RB, CW = 64, 128


@dataclass
class Csc:
    m: int
    ld: int
    nstrips: int
    nblocks: int
    Lc: np.ndarray      # [nstrips * nblocks] uint32
    Pre: np.ndarray     # [nstrips * nblocks] uint64
    vals: np.ndarray    # float32
    rows: np.ndarray    # uint8


def build(S: np.ndarray, order: np.ndarray | None = None) -> Csc:
    """S: dense float32 m x ld store (zero diagonal, zero padding columns). `order`: the order in
    which the groups claim their space (the device's order varies from build to build)."""
    m, ld = S.shape
    nstrips, nblocks = -(-ld // CW), -(-m // RB)
    G = nstrips * nblocks
    Lc = np.zeros(G, np.uint32)
    lists = {}
    for s in range(nstrips):
        for b in range(nblocks):
            blk = S[b * RB:(b + 1) * RB, s * CW:(s + 1) * CW]
            cols = [np.flatnonzero(blk[:, c]) for c in range(blk.shape[1])]
            lists[(s, b)] = (blk, cols)
            longest = max((len(c) for c in cols), default=0)
            Lc[s * nblocks + b] = (longest + 3) & ~3
    Pre = np.zeros(G, np.uint64)
    cur = 0
    for g in (range(G) if order is None else order):
        Pre[g] = cur
        cur += int(Lc[g])
    vals = np.zeros(cur * CW, np.float32)
    rows = np.zeros(cur * CW, np.uint8)
    for (s, b), (blk, cols) in lists.items():
        g = s * nblocks + b
        LQ, base = int(Lc[g]) // 4, int(Pre[g]) * CW
        for c, rr in enumerate(cols):
            lane, e = c >> 1, c & 1
            for k, r in enumerate(rr):
                at = base + ((e * LQ + k // 4) * 64 + lane) * 4 + k % 4
                vals[at], rows[at] = blk[r, c], r
    return Csc(m, ld, nstrips, nblocks, Lc, Pre, vals, rows)


def expand(M: Csc) -> np.ndarray:
    """k_csc_expand"""
    S = np.zeros((M.m, M.ld), np.float32)
    for s in range(M.nstrips):
        for b in range(M.nblocks):
            g = s * M.nblocks + b
            LQ, base = int(M.Lc[g]) // 4, int(M.Pre[g]) * CW
            for cl in range(CW):
                c = s * CW + cl
                if c >= M.ld:
                    continue
                lane, e = cl >> 1, cl & 1
                for k in range(4 * LQ):
                    at = base + ((e * LQ + k // 4) * 64 + lane) * 4 + k % 4
                    if M.vals[at] != 0:
                        S[b * RB + int(M.rows[at]), c] = M.vals[at]
    return S


def plan_tiles(M: Csc, target_wgs: float) -> np.ndarray:
    """csc_plan: tb[s][0..ntmax] in blocks; tiles of (nearly) equal cost sum(Lc + 2) per strip,
    the number of tiles of a strip proportional to its cost."""
    L = M.Lc.reshape(M.nstrips, M.nblocks).astype(np.float64) + 2.0
    tot = L.sum(axis=1)
    Q = tot.sum() / target_wgs
    nts = np.minimum(np.maximum(1, np.floor(tot / Q + 0.5)).astype(int), M.nblocks)
    ntmax = int(nts.max())
    tb = np.full((M.nstrips, ntmax + 1), M.nblocks, np.int32)
    for s in range(M.nstrips):
        run, k = 0.0, 1
        tb[s, 0] = 0
        for b in range(M.nblocks):
            run += L[s, b]
            while k < nts[s] and run >= tot[s] * k / nts[s]:
                tb[s, k] = b + 1
                k += 1
    return tb


def pass_window(M: Csc, tb: np.ndarray, X: np.ndarray, d: float, NW: int = 8):
    """csc_core in window mode: X[m][V]. Returns (a, g[1..V-1], b) as the tail sums them: per tile
    the NH waves of a column phase in wave order, then the tiles in tile order."""
    V = X.shape[1]
    NH = NW // 2
    ntmax = tb.shape[1] - 1
    part = np.zeros((ntmax, V + 1, M.ld))
    for s in range(M.nstrips):
        for t in range(ntmax):
            b0, b1 = int(tb[s, t]), int(tb[s, t + 1])
            acc = np.zeros((NH, V + 1, CW))
            for h in range(NH):
                for b in range(b0 + h, b1, NH):
                    g = s * M.nblocks + b
                    LQ, base = int(M.Lc[g]) // 4, int(M.Pre[g]) * CW
                    for cl in range(CW):
                        lane, e = cl >> 1, cl & 1
                        for k in range(4 * LQ):
                            at = base + ((e * LQ + k // 4) * 64 + lane) * 4 + k % 4
                            mm = float(M.vals[at])
                            ii = 1.0 if mm != 0.0 else 0.0
                            r = b * RB + int(M.rows[at])
                            x = X[r] if r < M.m else np.zeros(V)
                            acc[h, 0, cl] += mm * x[0]
                            acc[h, V, cl] += ii * x[0]
                            w = mm + d * ii
                            for v in range(1, V):
                                acc[h, v, cl] += w * x[v]
            tile = acc[0].copy()
            for h in range(1, NH):
                tile += acc[h]
            hi = min(CW, M.ld - s * CW)
            part[t, :, s * CW:s * CW + hi] = tile[:, :hi]
    out = part[0].copy()
    for t in range(1, ntmax):
        out += part[t]
    return out[0, :M.m], out[1:V, :M.m], out[V, :M.m]
