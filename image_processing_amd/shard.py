"""Frame sharding across the GPUs of one node: frames are independent, so each rank takes a contiguous
block of the stream, runs its own ImgProc context, and only the per-frame moments (24-48 B per frame and range) are
gathered on the host, ordered by frame id.  No data-path collective, no RCCL: the gather is host-side control-plane
traffic over whatever torch.distributed backend the launcher initialised (gloo in the tests and in bench.py)."""
from __future__ import annotations

import numpy as np


def shard_bounds(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """[start, stop) of rank's block; blocks differ by at most one frame and tile [0, n_frames) in rank order."""
    if world < 1 or not 0 <= rank < world or n_frames < 0:
        raise ValueError(f"bad shard request n={n_frames} rank={rank} world={world}")
    return n_frames * rank // world, n_frames * (rank + 1) // world


def gather_by_frame(local: np.ndarray, n_frames: int, rank: int, world: int, dist=None, dst: int = 0):
    """local: this rank's int64 results [n_local, ...] for frames shard_bounds(n_frames, rank, world).
    Returns the full [n_frames, ...] array on rank `dst` (None elsewhere), frames in stream order."""
    local = np.ascontiguousarray(local, dtype=np.int64)
    start, stop = shard_bounds(n_frames, rank, world)
    if local.shape[0] != stop - start:
        raise ValueError(f"rank {rank} holds {local.shape[0]} results for frames [{start},{stop})")
    if world == 1:
        return local
    import torch

    tail = local.shape[1:]
    cap = (n_frames + world - 1) // world  # every rank pads to the largest block so gather() sees equal shapes
    buf = torch.zeros((cap,) + tail, dtype=torch.int64)
    buf[: local.shape[0]] = torch.from_numpy(local)
    sink = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, sink, dst=dst)
    if rank != dst:
        return None
    out = np.empty((n_frames,) + tail, np.int64)
    for r in range(world):
        a, b = shard_bounds(n_frames, r, world)
        out[a:b] = sink[r][: b - a].numpy()
    return out


def centroids(moments: np.ndarray) -> np.ndarray:
    """[..., >=3] integer moments -> [..., 2] (cx, cy) in float64, -1 where m00 == 0 (the library's convention)."""
    m = np.asarray(moments)
    m00 = m[..., 0].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        cx = np.where(m00 > 0, m[..., 1] / m00, -1.0)
        cy = np.where(m00 > 0, m[..., 2] / m00, -1.0)
    return np.stack([cx, cy], -1)
