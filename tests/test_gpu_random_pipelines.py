"""Randomised differential test: seeded random pipelines (encoding, blur sizes, 1-4 ranges incl. never-matching ones,
0-6 erode/dilate ops of every shape incl. custom elements with holes, random anchors, odd sizes) through the C ABI
against the oracle composed stage by stage.  Exercises the planners (chain kernel / front kernel + bit planes / stage
kernels, RECT merging, span splitting) on inputs nobody hand-picked.  Bar: bit-exact masks and moments."""
import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L

pytestmark = pytest.mark.gpu

KEYS = ("m00", "m10", "m01", "m20", "m11", "m02")


@pytest.fixture(scope="module")
def gpu():
    with ip.ImgProc(1024, 256, n_slots=1, max_batch=1) as g:
        yield g


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    enc = int(rng.choice([L.ENC_BGR8, L.ENC_RGB8, L.ENC_BAYER_RGGB8], p=[0.5, 0.2, 0.3]))
    w = int(rng.choice([4, 8, 60, 64, 236, 240, 244, 248, 252, 256, 500, 1000, 37, 101, 250]))
    h = int(rng.choice([3, 5, 8, 16, 33, 64, 97, 130, 200]))
    pre = int(rng.choice([0, 0, 3, 5]))
    post = int(rng.choice([0, 3, 5, 5, 7, 9]))
    nr = int(rng.integers(1, 5))
    ranges = []
    for _ in range(nr):
        lo = [int(rng.integers(0, 200)), int(rng.integers(0, 200)), int(rng.integers(0, 200))]
        hi = [lo[0] + int(rng.integers(10, 200)), lo[1] + int(rng.integers(30, 257)), lo[2] + int(rng.integers(30, 257))]
        if rng.random() < 0.1:
            hi[1] = lo[1] - 1  # empty channel interval: the whole mask is empty
        ranges.append((tuple(lo), tuple(min(v, 256) for v in hi)))
    ops = []
    for _ in range(int(rng.integers(0, 7))):
        kind = int(rng.choice([L.ERODE, L.DILATE]))
        shape = int(rng.choice([L.RECT, L.RECT, L.CROSS, L.ELLIPSE, L.CUSTOM]))
        kw, kh = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        anchor = (-1, -1) if rng.random() < 0.6 else (int(rng.integers(0, kw)), int(rng.integers(0, kh)))
        custom = None
        if shape == L.CUSTOM:
            custom = (rng.random((kh, kw)) < 0.6).astype(np.uint8)
            custom[int(rng.integers(0, kh)), int(rng.integers(0, kw))] = 1  # never all-empty
        ops.append(ip.MorphOp(kind, shape, kw, kh, anchor, custom))
    if enc == L.ENC_BAYER_RGGB8:
        frame = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if rng.random() < 0.5:  # smooth blobs so that morphology has something to chew on
        yy, xx = np.mgrid[0:h, 0:w]
        blob = ((np.sin(xx / 9.0) + np.cos(yy / 7.0)) > 0.3)
        frame = np.where(blob if frame.ndim == 2 else blob[:, :, None], frame, frame // 4).astype(np.uint8)
    return dict(enc=enc, pre=pre, post=post, ranges=ranges, ops=ops, frame=frame, second=bool(rng.random() < 0.5))


def oracle_run(oracle, c):
    front = oracle.chain(c["frame"], encoding=c["enc"], pre_blur_k=c["pre"], post_blur_k=c["post"], morph=[],
                         ranges=c["ranges"], stages=True)
    masks, moms = [], []
    for r in range(len(c["ranges"])):
        m = front["thresh"][r]
        for op in c["ops"]:
            elem = op.custom if op.shape == L.CUSTOM else oracle.structuring_element(op.shape, op.kw, op.kh)
            m = oracle.morph(m, op.kind, elem, op.anchor)
        masks.append(m)
        moms.append(oracle.moments(m))
    return masks, moms


@pytest.mark.parametrize("seed", range(160))
def test_random_pipeline(gpu, oracle, seed):
    c = random_case(seed)
    exp_masks, exp_moms = oracle_run(oracle, c)
    for force in (False, True):
        gpu.set_pipeline(encoding=c["enc"], pre_blur_k=c["pre"], post_blur_k=c["post"], morph=c["ops"],
                         second_order=c["second"], force_unfused=force)
        gpu.set_ranges(c["ranges"])
        res = gpu.process(c["frame"])
        tag = (seed, gpu.plan, c["frame"].shape, c["pre"], c["post"], [(o.kind, o.shape, o.kw, o.kh, o.anchor) for o in c["ops"]])
        for r in range(len(c["ranges"])):
            assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp_masks[r]).all(), tag
            n = 6 if c["second"] else 3
            assert [res[r][k] for k in KEYS[:n]] == [int(v) for v in exp_moms[r][:n]], tag
