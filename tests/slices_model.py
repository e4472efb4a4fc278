"""Executable model (numpy) of the COMPRESSED STORAGE of M (CLIPPER_HIP_STORE_F32_CSC / _F64_CSC,
clipper_amd/csrc/k_slices.hip.h): the byte layout of a slice (what slice_emit_lds and
k_slice_pack write), the expansion (k_slice_expand), the work list of a pass (slices_plan in
host_matrix.hpp) and the pass itself (slice_core) with its summation order. Test infrastructure:
tests/test_slices_model.py checks on the CPU that the format round-trips for any placement of
the slices and that the pass equals the dense product; the kernels themselves are checked on
the GPU (tests/test_gpu_csc.py).

  slice (cg, k) = columns [64 cg, 64 cg + 64) x rows [128 k, 128 k + 128), one column per lane
  slice := head {u32 nquads, u32 maxq, u32 bytes, u32 0} | nq[64] u8 | so[ceil(maxq/16)] u32
           (padded to 16 B) | step 0 | step 1 | ...
  step q := value quads of the lanes with q < nq[lane], lane order | their row quads (4 x u8) |
            padding to 16 B.   Entries of a lane ascend by row; a list is padded to whole quads
            with (value 0, row 0).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

W, SUB, SO = 64, 128, 16   # SL_W, SL_SUB, SL_SO


def so_bytes(maxq: int) -> int:
    return ((maxq + SO - 1) // SO * 4 + 15) & ~15


@dataclass
class Slices:
    m: int
    ld: int
    ncg: int
    nchunks: int
    dtype: np.dtype
    data: bytearray
    Pre: np.ndarray     # [ncg * nchunks] offset / 16
    Lq: np.ndarray      # [ncg * nchunks] maxq | entries << 8


def pack_slice(blk: np.ndarray, dtype) -> tuple[bytes, int, int]:
    """blk: SUB x 64 dense block of one slice (rows beyond m: zero). -> (bytes, maxq, entries)"""
    lists = [np.flatnonzero(blk[:, l]) for l in range(W)]
    nq = [(len(r) + 3) // 4 for r in lists]
    maxq = max(nq)
    esz = np.dtype(dtype).itemsize
    out = bytearray(16 + 64 + so_bytes(maxq))
    out[16:80] = bytes(nq)
    so = []
    for q in range(maxq):
        if q % SO == 0:
            so.append(len(out))
        act = [l for l in range(W) if q < nq[l]]
        vals = np.zeros((len(act), 4), dtype)
        rows = np.zeros((len(act), 4), np.uint8)
        for j, l in enumerate(act):
            r = lists[l][4 * q:4 * q + 4]
            vals[j, :len(r)] = blk[r, l]
            rows[j, :len(r)] = r
        out += vals.tobytes() + rows.tobytes()
        out += bytes((-len(out)) % 16)
    out[80:80 + 4 * len(so)] = np.asarray(so, np.uint32).tobytes()
    out[0:16] = np.asarray([sum(nq), maxq, len(out), 0], np.uint32).tobytes()
    assert len(out) == 16 + 64 + so_bytes(maxq) + sum(
        len([l for l in range(W) if q < nq[l]]) * 4 * esz + ((len([l for l in range(W) if q < nq[l]]) * 4 + 15) & ~15)
        for q in range(maxq))
    return bytes(out), maxq, int(sum(len(r) for r in lists))


def build(S: np.ndarray, order=None) -> Slices:
    """S: dense m x ld store (zero diagonal, zero padding columns), float32 or float64. `order`:
    the order in which the slices claim their space (the fill kernel's varies from build to
    build; the packers lay them out column group major)."""
    m, ld = S.shape
    ncg, nchunks = ld // W, -(-m // SUB)
    n = ncg * nchunks
    blobs = []
    for cg in range(ncg):
        for k in range(nchunks):
            blk = np.zeros((SUB, W), S.dtype)
            rows = S[k * SUB:(k + 1) * SUB, cg * W:(cg + 1) * W]
            blk[:rows.shape[0]] = rows
            blobs.append(pack_slice(blk, S.dtype))
    Pre = np.zeros(n, np.uint64)
    Lq = np.zeros(n, np.uint32)
    data = bytearray()
    for s in (range(n) if order is None else order):
        Pre[s] = len(data) // 16
        data += blobs[s][0]
    for s in range(n):
        Lq[s] = blobs[s][1] | (blobs[s][2] << 8)
    return Slices(m, ld, ncg, nchunks, S.dtype, data, Pre, Lq)


def read_slice(M: Slices, s: int):
    """-> (nq[64], steps) with steps[q] = (lanes, vals[n][4], rows[n][4])"""
    base = int(M.Pre[s]) * 16
    nquads, maxq, nbytes, zero = np.frombuffer(M.data, np.uint32, 4, base)
    assert zero == 0
    nq = np.frombuffer(M.data, np.uint8, 64, base + 16)
    so = np.frombuffer(M.data, np.uint32, (int(maxq) + SO - 1) // SO, base + 80)
    esz = np.dtype(M.dtype).itemsize
    off = 16 + 64 + so_bytes(int(maxq))
    steps = []
    for q in range(int(maxq)):
        if q % SO == 0:
            assert so[q // SO] == off
        lanes = [l for l in range(W) if q < nq[l]]
        n = len(lanes)
        vals = np.frombuffer(M.data, M.dtype, 4 * n, base + off).reshape(n, 4)
        rows = np.frombuffer(M.data, np.uint8, 4 * n, base + off + 4 * n * esz).reshape(n, 4)
        steps.append((lanes, vals, rows))
        off += 4 * n * esz + ((4 * n + 15) & ~15)
    assert off == nbytes and nquads == int(nq.sum())
    return nq, steps


def expand(M: Slices) -> np.ndarray:
    """k_slice_expand"""
    S = np.zeros((M.m, M.ld), M.dtype)
    for cg in range(M.ncg):
        for k in range(M.nchunks):
            _, steps = read_slice(M, cg * M.nchunks + k)
            for lanes, vals, rows in steps:
                for j, l in enumerate(lanes):
                    for e in range(4):
                        if vals[j, e] != 0:
                            S[k * SUB + int(rows[j, e]), cg * W + l] = vals[j, e]
    return S


@dataclass
class Work:
    strip: int
    slot: int
    t0: int
    t1: int
    q0: int
    q1: int


def plan(M: Slices, target: float, NW: int = 4):
    """slices_plan: the work list (most expensive first) and the number of partial-sum slots"""
    nstrips = -(-M.ncg // NW)
    maxq = (M.Lq & 255).reshape(M.ncg, M.nchunks)
    cost = np.zeros((nstrips, M.nchunks), int)
    for st in range(nstrips):
        cost[st] = maxq[st * NW:(st + 1) * NW].max(axis=0)
    T = max(8.0, float((cost + 2.0).sum()) / target)
    items, nslot_of = [], []
    for st in range(nstrips):
        slot, start, acc = 0, 0, 0.0

        def flush(end):
            nonlocal slot, start, acc
            if end > start:
                items.append((acc, Work(st, slot, start, end, 0, 1 << 30)))
                slot += 1
            start, acc = end, 0.0
        for k in range(M.nchunks):
            mq = int(cost[st, k])
            c = mq + 2.0
            if c > 1.5 * T and mq >= 2 * SO:
                flush(k)
                parts = min(math.ceil(c / T), -(-mq // SO))
                per = -(-(-(-mq // parts)) // SO) * SO
                for q0 in range(0, mq, per):
                    items.append((min(per, mq - q0) + 2.0, Work(st, slot, k, k + 1, q0, min(q0 + per, mq))))
                    slot += 1
                start = k + 1
            else:
                acc += c
                if acc >= T:
                    flush(k + 1)
        flush(M.nchunks)
        nslot_of.append(slot)
    nslots = max(1, max(nslot_of))
    for st in range(nstrips):
        for slot in range(nslot_of[st], nslots):
            items.append((0.0, Work(st, slot, 0, 0, 0, 0)))
    items.sort(key=lambda it: -it[0])   # stable, like std::stable_sort
    return [w for _, w in items], nslots


def pass_window(M: Slices, work, nslots: int, X: np.ndarray, d: float, NW: int = 4):
    """slice_core in window mode: X[m][V]. Returns (a, g[1..V-1], b) as the tail sums them: every
    workgroup's lanes accumulate their column over the chunks and steps of their work item (in
    that order), the slots are added in slot order."""
    V = X.shape[1]
    part = np.full((nslots, V + 1, M.ld), np.nan)
    for w in work:
        for wave in range(NW):
            cg = w.strip * NW + wave
            if cg >= M.ncg:
                continue
            acc = np.zeros((V + 1, W))
            for k in range(w.t0, w.t1):
                nq, steps = read_slice(M, cg * M.nchunks + k)
                for q, (lanes, vals, rows) in enumerate(steps):
                    if q < w.q0 or q >= w.q1:
                        continue
                    for j, l in enumerate(lanes):
                        for e in range(4):
                            mm = float(vals[j, e])
                            ii = 1.0 if mm != 0.0 else 0.0
                            r = k * SUB + int(rows[j, e])
                            x = X[r] if r < M.m else np.zeros(V)
                            acc[0, l] += mm * x[0]
                            acc[V, l] += ii * x[0]
                            ww = mm + d * ii
                            for v in range(1, V):
                                acc[v, l] += ww * x[v]
            part[w.slot, :, cg * W:(cg + 1) * W] = acc
    assert not np.isnan(part).any(), "every (strip, slot) must be written by exactly one workgroup"
    out = part[0].copy()
    for t in range(1, nslots):
        out += part[t]
    return out[0, :M.m], out[1:V, :M.m], out[V, :M.m]
