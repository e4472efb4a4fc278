"""Host-side plumbing for the column-sharded (N > 1 GPU) path: one process per GPU, launched
with torch.distributed.run. torch.distributed is used only for bootstrap and bookkeeping
(rendezvous, broadcasting the RCCL unique id, barriers, max-over-ranks of timings); the
per-pass exchange of the solver is an RCCL all-gather issued by libclipper_hip.so itself on
its own stream (clipper_hip_comm_init / ncclAllGather in csrc/clipper_hip.hip).

Layout contract shared with the kernels (kernels.hip.h, k_reduce_pass / k_tail): shard p of P owns
global columns [p*W, p*W + W), W = round_up(ceil(m / P), 64); the gathered sums are
ab[P][NSLOT][W] — block p holds, slot after slot, this shard's columns of every product of the
pass (slot 0 = M_off x_0, slots 1..V-1 = (M_off + d C_off) x_v, slot V = C_off x_0). The helpers
below model the two-slot case (a, b) the protocol test uses.
"""
from __future__ import annotations

import os

import numpy as np


def shard_pitch(m: int, world: int) -> int:
    """Columns per shard (the C side computes the same W in ensure_problem)."""
    per = -(-m // world)
    return -(-per // 64) * 64


def shard_columns(m: int, world: int, rank: int) -> tuple[int, int]:
    """[c0, c1) global columns actually owned by `rank` (c1 - c0 may be 0 for trailing ranks)."""
    W = shard_pitch(m, world)
    c0 = min(rank * W, m)
    return c0, min(c0 + W, m)


def pack_block(a_slice: np.ndarray, b_slice: np.ndarray, W: int) -> np.ndarray:
    """One rank's [a | b] block of length 2*W (zero padded), as k_reduce writes it."""
    blk = np.zeros(2 * W)
    blk[: len(a_slice)] = a_slice
    blk[W: W + len(b_slice)] = b_slice
    return blk


def unpack_gathered(ab: np.ndarray, m: int, world: int) -> tuple[np.ndarray, np.ndarray]:
    """ab[P][2][W] -> full-length (a, b), the indexing `ab_at` performs on the device."""
    W = ab.size // (2 * world)
    blocks = ab.reshape(world, 2, W)
    a = blocks[:, 0, :].reshape(-1)[:m]
    b = blocks[:, 1, :].reshape(-1)[:m]
    return a.copy(), b.copy()


def env_rank_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str | None = None):
    """Rendezvous on 127.0.0.1 by default (the container hostname may not resolve)."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    rank, local_rank, world = env_rank_world()
    if backend is None:
        backend = "cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def broadcast_bytes(payload: bytes | None, nbytes: int, src: int = 0) -> bytes:
    """Broadcast a fixed-size byte string (the 128-byte ncclUniqueId) through CPU tensors."""
    import torch
    import torch.distributed as dist

    t = torch.zeros(nbytes, dtype=torch.uint8)
    if dist.get_rank() == src:
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    dist.broadcast(t, src=src)
    return bytes(t.numpy().tobytes())


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_equal_over_ranks(arr: np.ndarray) -> bool:
    """True when every rank holds bit-identical data (replicated solver state check)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))
