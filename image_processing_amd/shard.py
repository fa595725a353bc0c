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


# ---------------------------------------------------------------------------------------------------------------------
# NUMA placement of a rank ("Streamed: bounded by per-GPU PCIe and host memcpy bandwidth / NUMA -- pin feeder
# threads to the GPU's NUMA node").  Pure sysfs reads, no HIP call: it runs before the rank touches the GPU, so the
# library's copy threads and every page-locked allocation made afterwards inherit the placement.
# ---------------------------------------------------------------------------------------------------------------------
def _read(path: str):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _parse_cpulist(text: str) -> set:
    cpus = set()
    for part in text.split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_pci_addresses(sysfs: str = "/sys") -> list:
    """PCI addresses of the node's GPUs in the order the HIP runtime enumerates them: the KFD topology's GPU nodes
    (simd_count > 0) in node order; if there is no KFD topology, the AMD (vendor 0x1002) DRM cards in card order."""
    import glob
    import os
    import re

    out = []
    nodes = glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*"))
    for node in sorted(nodes, key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30):
        props = _read(os.path.join(node, "properties"))
        if not props:
            continue
        kv = dict(ln.split(None, 1) for ln in props.splitlines() if len(ln.split(None, 1)) == 2)
        if int(kv.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        loc, dom = int(kv.get("location_id", "0")), int(kv.get("domain", "0"))
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    if out:
        return out
    cards = [c for c in glob.glob(os.path.join(sysfs, "class/drm/card*")) if re.fullmatch(r"card\d+", os.path.basename(c))]
    for card in sorted(cards, key=lambda p: int(os.path.basename(p)[4:])):
        if (_read(os.path.join(card, "device/vendor")) or "").lower() != "0x1002":
            continue
        out.append(os.path.basename(os.path.realpath(os.path.join(card, "device"))))
    return out


def pin_to_gpu_numa(local_rank: int, sysfs: str = "/sys", setaffinity=None) -> dict:
    """Restricts the calling process to the CPUs of the NUMA node its GPU hangs off.  Returns what it found and did
    ({"numa_node", "cpus", "pci", "pinned"}); any missing piece (no sysfs entry, numa_node -1, a visible-devices list it
    cannot read) leaves the affinity alone, silently -- placement is a speed matter, never a correctness one."""
    import os

    info = {"numa_node": None, "cpus": None, "pci": None, "pinned": False}
    try:
        idx = local_rank
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis:  # the rank's device is the local_rank-th entry of the list
                idx = int(vis.split(",")[local_rank])
                break
        gpus = gpu_pci_addresses(sysfs)
        if not 0 <= idx < len(gpus):
            return info
        info["pci"] = gpus[idx]
        node = _read(os.path.join(sysfs, "bus/pci/devices", gpus[idx], "numa_node"))
        if node is None:  # the fallback list holds directory names of class/drm/cardN/device, not always bus addresses
            return info
        info["numa_node"] = int(node)
        if info["numa_node"] < 0:
            return info
        cpus = _parse_cpulist(_read(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % info["numa_node"])) or "")
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else cpus
        cpus &= set(allowed)  # never widen what a container / launcher already restricted
        if not cpus:
            return info
        (setaffinity or os.sched_setaffinity)(0, cpus)
        info["cpus"], info["pinned"] = len(cpus), True
    except (OSError, ValueError, IndexError, AttributeError):
        pass
    return info
