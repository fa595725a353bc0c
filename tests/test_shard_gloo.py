"""CPU-only, world_size 2 over gloo: the multi-GPU path is "shard frames, process independently, gather moments by
frame id" (no collective on the data path).  The per-rank compute is stood in for by the oracle here (no GPU in this
suite); what is under test is the sharding / ordering / gather logic that bench.py and a multi-GPU host use."""
import os
import tempfile

import numpy as np
import pytest

from image_processing_amd.shard import centroids, gather_by_frame, shard_bounds


def test_shard_bounds_tile_the_stream():
    for n in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_centroids():
    m = np.array([[6, 6, 3], [0, 0, 0]], np.int64)
    assert centroids(m).tolist() == [[1.0, 0.5], [-1.0, -1.0]]


def _worker(rank, world, initfile, n_frames, w, h, outdir):
    import torch.distributed as dist

    from image_processing_amd.synth import DEFAULT_RANGE, synth_frame
    from oracle import Oracle

    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        orc = Oracle()
        a, b = shard_bounds(n_frames, rank, world)
        local = np.zeros((b - a, 1, 3), np.int64)
        for i in range(a, b):
            r = orc.chain(synth_frame(w, h, i), post_blur_k=5, morph=[(0, 0, 3, 3), (1, 0, 3, 3)],
                          ranges=[DEFAULT_RANGE], stages=False)
            local[i - a, 0] = r["moments"][0][:3].astype(np.int64)
        full = gather_by_frame(local, n_frames, rank, world, dist)
        if rank == 0:
            np.save(os.path.join(outdir, "full.npy"), full)
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8])
def test_two_ranks_gather_in_frame_order(n_frames):
    import torch.multiprocessing as mp

    from image_processing_amd.synth import DEFAULT_RANGE, synth_frame
    from oracle import Oracle

    w, h = 96, 64
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), n_frames, w, h, d), nprocs=2, join=True)
        full = np.load(os.path.join(d, "full.npy"))
    orc = Oracle()
    exp = np.stack([orc.chain(synth_frame(w, h, i), post_blur_k=5, morph=[(0, 0, 3, 3), (1, 0, 3, 3)],
                              ranges=[DEFAULT_RANGE], stages=False)["moments"][:, :3].astype(np.int64)
                    for i in range(n_frames)])
    assert full.shape == (n_frames, 1, 3) and (full == exp).all()
    assert (exp[:, 0, 0] > 0).any()
