"""GPU: BASELINE.json's other configurations at their full frame sizes.  Oracle comparison where the CPU oracle finishes
in seconds; size-independent properties (mask alphabet, moments == sums over the GPU's own mask, two independent GPU
paths agreeing bit for bit) where it does not."""
import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_bayer_rggb, synth_frame

pytestmark = pytest.mark.gpu
KEYS = ("m00", "m10", "m01")


def morph_tuples(ops):
    return [(m.kind, m.shape, m.kw, m.kh, m.anchor[0], m.anchor[1]) for m in ops]


def mask_sums(mask):
    ys, xs = np.nonzero(mask)
    return [len(xs), int(xs.sum()), int(ys.sum())]


def test_config3_bayer_2448x2048(oracle):
    """configs[2]: Basler-style 2448x2048 BayerRG8, demosaic + full chain, streamed through pinned slots."""
    w, h = 2448, 2048
    raws = [synth_bayer_rggb(w, h, i) for i in range(3)]
    with ip.ImgProc(w, h, n_slots=2) as g:
        g.set_pipeline(encoding=L.ENC_BAYER_RGGB8, post_blur_k=5, morph=ip.MORPH_BASELINE)
        g.set_ranges([DEFAULT_RANGE])
        assert g.plan == "fused chain kernel (Bayer front end inside)"
        res = g.process_batch(np.stack(raws))
        r0 = g.process(raws[0])
        m0 = g.fetch_stage(0, L.STAGE_MASK)
    exp = oracle.chain(raws[0], encoding=2, post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE],
                       stages=False)
    assert (m0 == exp["mask"][0]).all()
    assert [r0[0][k] for k in KEYS] == [int(v) for v in exp["moments"][0][:3]] == [res[0][0][k] for k in KEYS]
    assert all(r[0]["m00"] > 0 for r in res)


def test_config4_3840x2160_batch(oracle):
    """configs[3] frame shape: 3840x2160 BGR8 device batch; one frame against the oracle, all frames self-consistent."""
    w, h, n = 3840, 2160, 4
    frames = np.stack([synth_frame(w, h, i) for i in range(n)])
    with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
        g.set_pipeline(post_blur_k=5, morph=ip.MORPH_BASELINE)
        g.set_ranges([DEFAULT_RANGE])
        d_in, d_out = g.dev_alloc(frames.nbytes), g.dev_alloc(n * w * h)
        try:
            g.dev_upload(d_in, frames)
            res = g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out)
            masks = g.dev_download(d_out, (n, h, w))
        finally:
            g.dev_free(d_in)
            g.dev_free(d_out)
    assert set(np.unique(masks)) <= {0, 255}
    for i in range(n):
        assert [res[i][0][k] for k in KEYS] == mask_sums(masks[i])
    exp = oracle.chain(frames[2], post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE], stages=False)
    assert (masks[2] == exp["mask"][0]).all()


def test_config5_stress_7680x4320():
    """configs[4]: 7680x4320, 15x15 Gaussian + 9x9 elliptical erode/dilate + 8 simultaneous ranges: the bit-plane path
    (fused HSV + Gaussian + inRange kernel, strip morphology) and the per-op stage kernels (two independent GPU
    implementations) agree bit for bit, the moments equal the sums over the masks, and all eight masks and moment sets
    equal the oracle's on the full frame."""
    w, h = 7680, 4320
    bands = [(20 * i, 20 * i + 19) for i in range(8)]
    f = synth_frame(w, h, 3, hue_bands=bands)
    ranges = [((lo, 100, 80), (hi, 255, 255)) for lo, hi in bands]
    morph = [ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)]
    out = []
    with ip.ImgProc(w, h, n_slots=1) as g:
        for force in (False, True):
            g.set_pipeline(post_blur_k=15, morph=morph, force_unfused=force)
            g.set_ranges(ranges)
            res = g.process(f)
            masks = [g.fetch_stage(0, L.STAGE_MASK, r) for r in range(8)]
            out.append((res, masks))
    (res_a, masks_a), (res_b, masks_b) = out
    hit = 0
    for r in range(8):
        assert (masks_a[r] == masks_b[r]).all()
        assert [res_a[r][k] for k in KEYS] == [res_b[r][k] for k in KEYS] == mask_sums(masks_a[r])
        hit += res_a[r]["m00"] > 0
    assert hit >= 4
    # and the oracle itself on the full frame (the -O3 build of the same C source: integer arithmetic, ~25 s on one core)
    from oracle import Oracle
    exp = Oracle(fast=True).chain(f, post_blur_k=15, morph=morph_tuples(morph), ranges=ranges, stages=False)
    for r in range(8):
        assert (masks_a[r] == exp["mask"][r]).all(), r
        assert [res_a[r][k] for k in KEYS] == [int(v) for v in exp["moments"][r][:3]], r


@pytest.mark.parametrize("case", [("bgr", 1920, 1200), ("bgr", 1918, 1203), ("bayer", 2448, 2048)])
def test_full_size_noise_vs_oracle(oracle, case):
    """Full-size frames of structured noise (every pixel carries information, so every strip / segment boundary of the
    chain kernel shows up if it is wrong) for the three chains, two ranges, against the oracle: masks bit for bit,
    moments exact."""
    enc, w, h = case
    rng = np.random.default_rng(w * 7 + h)
    shape = (h, w) if enc == "bayer" else (h, w, 3)
    frame = rng.integers(0, 256, shape, dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    blob = (np.sin(xx / 37.0) * np.cos(yy / 29.0)) > 0.2
    frame = np.where(blob if enc == "bayer" else blob[:, :, None], frame, frame // 3).astype(np.uint8)
    ranges = [((0, 30, 60), (110, 256, 256)), ((90, 0, 0), (180, 140, 90))]
    presets = {"pose_estimate": dict(pre_blur_k=0, post_blur_k=0, morph=[]),
               "baseline": dict(pre_blur_k=0, post_blur_k=5, morph=ip.MORPH_BASELINE),
               "polygon_path": dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH)}
    with ip.ImgProc(w, h, n_slots=1) as g:
        for name, kw in presets.items():
            g.set_pipeline(encoding=L.ENC_BAYER_RGGB8 if enc == "bayer" else L.ENC_BGR8, second_order=True, **kw)
            g.set_ranges(ranges)
            assert "stage kernels" not in g.plan
            res = g.process(frame)
            exp = oracle.chain(frame, encoding=2 if enc == "bayer" else 0, pre_blur_k=kw["pre_blur_k"],
                               post_blur_k=kw["post_blur_k"], morph=morph_tuples(kw["morph"]), ranges=ranges, stages=False)
            for r in range(2):
                assert (g.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all(), (case, name, r)
                got = [res[r][k] for k in ("m00", "m10", "m01", "m20", "m11", "m02")]
                assert got == [int(v) for v in exp["moments"][r]], (case, name, r)
            assert sum(int(m[0]) for m in exp["moments"]) > 0 or name == "polygon_path"


@pytest.mark.parametrize("padded", [False, True])
def test_pipelined_slots_stage_large_frames(oracle, padded):
    """Three slots in flight at 1920x1200: the frames are staged by the helper threads (api.hip CopyPool), tight rows and
    rows with a stride; every frame's mask and moments must equal the lone-frame path's (single-threaded staging), and
    one of them the oracle's."""
    w, h, n = 1920, 1200, 7
    store = np.zeros((n, h, w + 40, 3), np.uint8)
    frames = []
    for i in range(n):
        f = synth_frame(w, h, 100 + i)
        if padded:
            store[i, :, :w] = f
            store[i, :, w:] = 255 - i  # the padding must never reach the device
            frames.append(store[i, :, :w])
        else:
            frames.append(f)
    morph = morph_tuples(ip.MORPH_BASELINE)
    with ip.ImgProc(w, h, n_slots=3) as g:
        g.set_pipeline(post_blur_k=5, morph=ip.MORPH_BASELINE, second_order=True)
        g.set_ranges([DEFAULT_RANGE])
        lone = []
        for f in frames:  # nothing else in flight: single-threaded staging
            r = g.process(np.ascontiguousarray(f))
            lone.append((r, g.fetch_stage(0, L.STAGE_MASK)))
        got = [None] * n
        for i in range(n + 2):  # submit i while i-1 and i-2 are in flight
            if i >= 2:
                j = i - 2
                got[j] = (g.wait(j % 3), g.fetch_stage(j % 3, L.STAGE_MASK))
            if i < n:
                g.submit(i % 3, frames[i])
        for i in range(n):
            assert got[i][0] == lone[i][0], i
            assert (got[i][1] == lone[i][1]).all(), i
    exp = oracle.chain(np.ascontiguousarray(frames[3]), post_blur_k=5, morph=morph, ranges=[DEFAULT_RANGE], stages=False)
    assert (got[3][1] == exp["mask"][0]).all()
    assert [got[3][0][0][k] for k in ("m00", "m10", "m01", "m20", "m11", "m02")] == [int(v) for v in exp["moments"][0]]


@pytest.mark.parametrize("padded", [False, True])
@pytest.mark.parametrize("preset", ["baseline", "polygon_path"])
def test_submit_pinned_skips_the_staging_copy(oracle, padded, preset):
    """imgproc_submit_pinned: frames in page-locked memory (tight rows, and rows with a stride -> one 2-D DMA) go to the
    device without the staging memcpy; three slots in flight; results equal imgproc_submit's, one frame the oracle's.
    A pageable buffer is accepted too (the runtime stages it)."""
    w, h, n = 1920, 1200, 6
    kw = dict(post_blur_k=5, morph=ip.MORPH_BASELINE) if preset == "baseline" else dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH)
    with ip.ImgProc(w, h, n_slots=3) as g:
        g.set_pipeline(second_order=True, **kw)
        g.set_ranges([DEFAULT_RANGE])
        pinned = g.host_alloc((n, h, w + (40 if padded else 0), 3))
        try:
            pinned[:] = 201
            views = []
            for i in range(n):
                pinned[i, :, :w] = synth_frame(w, h, 300 + i)
                views.append(pinned[i, :, :w])
            ref = []
            for f in views:
                r = g.process(np.ascontiguousarray(f))
                ref.append((r, g.fetch_stage(0, L.STAGE_MASK)))
            got = [None] * n
            for i in range(n + 2):
                if i >= 2:
                    j = i - 2
                    got[j] = (g.wait(j % 3), g.fetch_stage(j % 3, L.STAGE_MASK))
                if i < n:
                    g.submit_pinned(i % 3, views[i])
            for i in range(n):
                assert got[i][0] == ref[i][0], i
                assert (got[i][1] == ref[i][1]).all(), i
            g.submit_pinned(0, np.ascontiguousarray(views[2]))  # pageable memory: still correct
            assert g.wait(0) == ref[2][0]
            assert (g.fetch_stage(0, L.STAGE_HSV) == oracle.bgr2hsv(oracle.gaussian_blur(np.ascontiguousarray(views[2]), 3)
                                                                     if preset == "polygon_path" else np.ascontiguousarray(views[2]))).all()
            f4 = views[4].copy()  # before the page-locked block goes away
            # host_free refuses while views of the block are alive (they would dangle: round 2's segfault), then frees
            with pytest.raises(ip.pipeline.ImgProcError) as ei:
                g.host_free(pinned)
            status, ei = ei.value.status, None
            assert status == L.ESTATE
            with pytest.raises(ip.pipeline.ImgProcError):
                g.host_free(views[0])  # a view is not the array host_alloc handed out
        finally:
            views = f = ei = None  # ei's traceback holds host_free's frame, hence the array
            g.host_free(pinned)
    exp = oracle.chain(f4, pre_blur_k=kw.get("pre_blur_k", 0), post_blur_k=5, morph=morph_tuples(kw["morph"]), ranges=[DEFAULT_RANGE],
                       stages=False)
    assert (got[4][1] == exp["mask"][0]).all()
    assert [got[4][0][0][k] for k in ("m00", "m10", "m01", "m20", "m11", "m02")] == [int(v) for v in exp["moments"][0]]


@pytest.mark.parametrize("case", [("baseline", 1920, 1200, 96), ("polygon_path", 1920, 1200, 96), ("baseline", 1918, 1203, 96),
                                  ("bayer", 2448, 2048, 56)])
def test_overlapped_batches_take_tall_segments(oracle, case):
    """With another ticket in flight a large batch is cut into ~300-row segments instead of ~100-row ones
    (fused.hip, launch_fused `overlapped`): a batch of structured noise on tickets 1 and 2 while ticket 0 runs, equal to the
    same batch processed alone (other segment borders) mask for mask, and to the oracle on one frame.  Chain kernel, front
    kernel, a width that is not a multiple of 4 with a height that does not divide, and the Bayer chain."""
    preset, w, h, n = case
    uniq = 3
    bayer = preset == "bayer"
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    blob = (np.sin(xx / 41.0) * np.cos(yy / 23.0)) > 0.1
    if bayer:
        base = [np.where(blob, rng.integers(0, 256, (h, w), dtype=np.uint8), 40).astype(np.uint8) for _ in range(uniq)]
    else:
        base = [np.where(blob[:, :, None], rng.integers(0, 256, (h, w, 3), dtype=np.uint8), 40).astype(np.uint8) for _ in range(uniq)]
    frames = np.stack([np.roll(base[i % uniq], 8 * (i // uniq), axis=0) for i in range(n)])  # even shifts keep the mosaic phase
    kw = dict(baseline=dict(post_blur_k=5, morph=ip.MORPH_BASELINE), bayer=dict(post_blur_k=5, morph=ip.MORPH_BASELINE),
              polygon_path=dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH))[preset]
    enc = L.ENC_BAYER_RGGB8 if bayer else L.ENC_BGR8
    cn = 1 if bayer else 3
    rg = [((0, 30, 60), (120, 256, 256))]
    with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
        g.set_pipeline(encoding=enc, **kw)
        g.set_ranges(rg)
        assert "stage kernels" not in g.plan
        d_in = g.dev_alloc(frames.nbytes)
        d_m = [g.dev_alloc(n * w * h) for _ in range(4)]
        try:
            g.dev_upload(d_in, frames)
            alone = g.process_batch_device(d_in, w * h * cn, w * cn, w, h, n, d_m[0])
            for t in range(3):
                g.batch_enqueue(t, d_in, w * h * cn, w * cn, w, h, n, d_m[1 + t])  # tickets 1 and 2 see another one pending: overlapped
            for t in (2, 0, 1):
                assert g.batch_wait(t, n=n) == alone, t
            m_alone = g.dev_download(d_m[0], (n, h, w))
            for t in (1, 2):
                assert (g.dev_download(d_m[1 + t], (n, h, w)) == m_alone).all(), t
        finally:
            g.dev_free(d_in)
            for p in d_m:
                g.dev_free(p)
    i = n // 2
    exp = oracle.chain(frames[i], encoding=enc, morph=morph_tuples(kw["morph"]), ranges=rg, stages=False,
                       **{k: v for k, v in kw.items() if k != "morph"})
    assert (m_alone[i] == exp["mask"][0]).all()
    assert [alone[i][0][k] for k in KEYS] == [int(v) for v in exp["moments"][0][:3]] and alone[i][0]["m00"] > 0
