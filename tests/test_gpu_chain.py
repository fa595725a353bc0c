"""GPU parity of the whole chain through the C ABI (submit/wait, fetch_stage, batch paths) against the
oracle's chain on the same seeded synthetic frames; fused == unfused == oracle, bit for bit."""
import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch, synth_bayer_rggb, synth_frame

pytestmark = pytest.mark.gpu

KEYS = ("m00", "m10", "m01", "m20", "m11", "m02")


def morph_tuples(ops):
    return [(m.kind, m.shape, m.kw, m.kh, m.anchor[0], m.anchor[1]) for m in ops]


PRESETS = {
    "pose_estimate": dict(pre_blur_k=0, post_blur_k=0, morph=[]),  # pose_estimate.cpp:203-205
    "pose_estimate_full": dict(pre_blur_k=0, post_blur_k=5, morph=ip.MORPH_POSE_ESTIMATE),  # with :204,:206 enabled
    "baseline": dict(pre_blur_k=0, post_blur_k=5, morph=ip.MORPH_BASELINE),
    "polygon_path": dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH),  # objectPolygonPath.cpp:182-188
    "c1": dict(pre_blur_k=0, post_blur_k=0, morph=ip.MORPH_BASELINE),  # BASELINE config 1
}


@pytest.fixture(scope="module")
def gpu():
    with ip.ImgProc(1920, 1200, n_slots=2, max_batch=8) as g:
        yield g


def check_moments(res, exp, second=False):
    n = 6 if second else 3
    assert [res[k] for k in KEYS[:n]] == [int(v) for v in exp[:n]]
    if not second:
        assert res["m20"] == 0 and res["m11"] == 0 and res["m02"] == 0 and res["theta"] == 0.0
    if exp[0]:
        assert abs(res["cx"] - float(exp[1]) / float(exp[0])) <= 1e-9 * max(1.0, abs(res["cx"]))
        assert abs(res["cy"] - float(exp[2]) / float(exp[0])) <= 1e-9 * max(1.0, abs(res["cy"]))
        assert res["found"] == 1
    else:
        assert res["found"] == 0 and res["cx"] == -1.0


@pytest.mark.parametrize("name", list(PRESETS))
@pytest.mark.parametrize("hw", [(64, 96), (123, 517), (480, 640)])
def test_stages_unfused(gpu, oracle, name, hw):
    p = PRESETS[name]
    f = synth_frame(hw[1], hw[0], 5)
    ranges = [DEFAULT_RANGE, ((0, 0, 0), (90, 256, 256))]
    gpu.set_pipeline(keep_stages=True, second_order=True, **p)
    gpu.set_ranges(ranges)
    assert not gpu.is_fused
    res = gpu.process(f, slot=1)
    exp = oracle.chain(f, pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"], morph=morph_tuples(p["morph"]),
                       ranges=ranges)
    assert (gpu.fetch_stage(1, L.STAGE_BGR) == exp["bgr"]).all()
    assert (gpu.fetch_stage(1, L.STAGE_HSV) == exp["hsv"]).all()
    assert (gpu.fetch_stage(1, L.STAGE_BLUR) == exp["blur"]).all()
    for r in range(2):
        assert (gpu.fetch_stage(1, L.STAGE_THRESH, r) == exp["thresh"][r]).all()
        assert (gpu.fetch_stage(1, L.STAGE_MASK, r) == exp["mask"][r]).all()
        check_moments(res[r], exp["moments"][r], second=True)
    assert (gpu.fetch_stage(1, L.STAGE_MASKED) == oracle.mask_and(exp["bgr"], exp["mask"][0])).all()
    assert exp["moments"][0][0] > 0  # the scene really contains the marker


@pytest.mark.parametrize("name", list(PRESETS))
@pytest.mark.parametrize("hw", [(1, 1), (5, 3), (17, 13), (64, 96), (100, 250), (123, 517), (480, 640), (1200, 1920)])
@pytest.mark.parametrize("rgb", [False, True])
def test_chain_default_path(gpu, oracle, name, hw, rgb):
    """Whatever plan the library picks (fused when available) must equal the oracle."""
    if hw == (1200, 1920) and (rgb or name not in ("baseline", "polygon_path")):
        pytest.skip("full size once per headline preset")
    p = PRESETS[name]
    f = synth_frame(hw[1], hw[0], 11, rgb=rgb)
    gpu.set_pipeline(encoding=L.ENC_RGB8 if rgb else L.ENC_BGR8, **p)
    gpu.set_ranges([DEFAULT_RANGE])
    res = gpu.process(f)
    exp = oracle.chain(f, encoding=1 if rgb else 0, pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"],
                       morph=morph_tuples(p["morph"]), ranges=[DEFAULT_RANGE], stages=False)
    assert (gpu.fetch_stage(0, L.STAGE_MASK) == exp["mask"][0]).all()
    check_moments(res[0], exp["moments"][0])


@pytest.mark.parametrize("name", ["baseline", "polygon_path", "pose_estimate"])
def test_fused_equals_unfused(gpu, name):
    p = PRESETS[name]
    frames = synth_batch(517, 123, 3)
    outs = []
    for force in (False, True):
        gpu.set_pipeline(force_unfused=force, **p)
        gpu.set_ranges([DEFAULT_RANGE])
        r = []
        for f in frames:
            res = gpu.process(f)
            r.append((gpu.fetch_stage(0, L.STAGE_MASK), [res[0][k] for k in KEYS]))
        outs.append(r)
    for (m0, k0), (m1, k1) in zip(*outs):
        assert (m0 == m1).all() and k0 == k1


def test_edge_masks(gpu, oracle):
    gpu.set_pipeline(**PRESETS["baseline"])
    for val, rng in ((0, DEFAULT_RANGE), (255, ((0, 0, 0), (256, 256, 256))), (255, ((256, 0, 0), (0, 256, 256)))):
        f = np.full((37, 61, 3), val, np.uint8)
        gpu.set_ranges([rng])
        res = gpu.process(f)
        exp = oracle.chain(f, post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[rng], stages=False)
        assert (gpu.fetch_stage(0, L.STAGE_MASK) == exp["mask"][0]).all()
        check_moments(res[0], exp["moments"][0])


def test_bayer_chain(gpu, oracle):
    raw = synth_bayer_rggb(250, 100, 2)
    gpu.set_pipeline(encoding=L.ENC_BAYER_RGGB8, post_blur_k=5, morph=ip.MORPH_BASELINE)
    gpu.set_ranges([DEFAULT_RANGE])
    res = gpu.process(raw)
    exp = oracle.chain(raw, encoding=2, post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE],
                       stages=False)
    assert (gpu.fetch_stage(0, L.STAGE_MASK) == exp["mask"][0]).all()
    check_moments(res[0], exp["moments"][0])


@pytest.mark.parametrize("preset", ["baseline", "polygon_path"])
@pytest.mark.parametrize("enc", ["bgr", "bayer"])
def test_fetch_stage_on_demand(gpu, oracle, preset, enc):
    """keep_stages = 0 (the fast paths): every stage image can still be fetched; it is rebuilt from the slot's raw frame
    and equals the oracle's stage, which pins the fast path's internal stages too (its mask / moments match the same
    oracle run).  MASKED is frame & mask (objectPolygonPath.cpp:189-190) for the range asked for."""
    h, w = 96, 256
    kw = PRESETS[preset]
    ranges = [DEFAULT_RANGE, ((0, 0, 0), (90, 256, 120))]
    frame = synth_bayer_rggb(w, h, 5) if enc == "bayer" else synth_frame(w, h, 5)
    encoding = L.ENC_BAYER_RGGB8 if enc == "bayer" else L.ENC_BGR8
    gpu.set_pipeline(encoding=encoding, **kw)
    gpu.set_ranges(ranges)
    assert "stage kernels" not in gpu.plan
    res = gpu.process(frame)
    exp = oracle.chain(frame, encoding=2 if enc == "bayer" else 0, pre_blur_k=kw["pre_blur_k"], post_blur_k=kw["post_blur_k"],
                       morph=morph_tuples(kw["morph"]), ranges=ranges, stages=True)
    for r in (1, 0):
        masked = gpu.fetch_stage(0, L.STAGE_MASKED, r)
        assert (masked == (exp["bgr"] & exp["mask"][r][:, :, None])).all()
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all()
        check_moments(res[r], exp["moments"][r])
    assert (gpu.fetch_stage(0, L.STAGE_THRESH, 1) == exp["thresh"][1]).all()
    assert (gpu.fetch_stage(0, L.STAGE_BLUR) == exp["blur"]).all()
    assert (gpu.fetch_stage(0, L.STAGE_HSV) == exp["hsv"]).all()
    assert (gpu.fetch_stage(0, L.STAGE_BGR) == exp["bgr"]).all()
    gpu.set_ranges([DEFAULT_RANGE])  # parameters changed: the old frame's stages can no longer be rebuilt
    frame2 = synth_bayer_rggb(w, h, 6) if enc == "bayer" else synth_frame(w, h, 6)
    gpu.process(frame2)
    assert gpu.fetch_stage(0, L.STAGE_HSV).shape == (h, w, 3)
    gpu.set_ranges(ranges)
    with pytest.raises(ip.ImgProcError):
        gpu.fetch_stage(0, L.STAGE_BLUR)
    assert gpu.fetch_stage(0, L.STAGE_HSV).shape == (h, w, 3)  # already materialised: still served


@pytest.mark.parametrize("hw", [(3, 4), (4, 8), (3, 8), (4, 12), (5, 16), (7, 20), (6, 34), (9, 38), (5, 236), (33, 240), (64, 244),
                                (33, 248), (17, 252), (12, 256), (21, 260), (19, 250), (57, 484), (40, 496), (40, 500), (31, 746),
                                (131, 1000)])
@pytest.mark.parametrize("preset", ["pose_estimate", "baseline", "polygon_path"])
def test_bayer_front_end_noise(gpu, oracle, hw, preset):
    """The Bayer front end inside the chain / front kernels on full-range noise mosaics (every demosaiced value matters:
    no blur in `pose_estimate`, two ranges, one of them narrow): sizes around the 248-column strip, rows of 8 to 20 bytes
    (every lane's 8-byte window touches a row end), widths with w % 4 == 2 (right-aligned last strip), 3 to 7 rows."""
    h, w = hw
    raw = np.random.default_rng(h * 7919 + w).integers(0, 256, (h, w), dtype=np.uint8)
    ranges = [((0, 40, 60), (90, 256, 256)), ((85, 0, 100), (140, 200, 220))]
    kw = PRESETS[preset]
    gpu.set_pipeline(encoding=L.ENC_BAYER_RGGB8, second_order=True, **kw)
    gpu.set_ranges(ranges)
    assert "Bayer front end inside" in gpu.plan
    res = gpu.process(raw)
    exp = oracle.chain(raw, encoding=2, pre_blur_k=kw["pre_blur_k"], post_blur_k=kw["post_blur_k"],
                       morph=morph_tuples(kw["morph"]), ranges=ranges, stages=False)
    for r in range(2):
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all(), (preset, hw, r)
        check_moments(res[r], exp["moments"][r], second=True)
    gpu.set_pipeline(encoding=L.ENC_BAYER_RGGB8, force_unfused=True, **kw)  # the demosaic stage kernel agrees
    gpu.set_ranges(ranges)
    gpu.process(raw)
    for r in range(2):
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all(), ("unfused", preset, hw, r)


@pytest.mark.parametrize("hw", [(100, 256), (480, 640)])
def test_bayer_chain_fused_backend(gpu, oracle, hw):
    """BayerRG8 input with a fusable tail: the chain kernel demosaics while loading (BASELINE config 3's operators)."""
    h, w = hw
    raws = np.stack([synth_bayer_rggb(w, h, i) for i in range(3)])
    gpu.set_pipeline(encoding=L.ENC_BAYER_RGGB8, post_blur_k=5, morph=ip.MORPH_BASELINE)
    gpu.set_ranges([DEFAULT_RANGE])
    assert gpu.is_fused
    exps = [oracle.chain(r, encoding=2, post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE],
                         stages=False) for r in raws]
    res = gpu.process(raws[0])
    assert (gpu.fetch_stage(0, L.STAGE_MASK) == exps[0]["mask"][0]).all()
    check_moments(res[0], exps[0]["moments"][0])
    d_in = gpu.dev_alloc(raws.nbytes)
    d_out = gpu.dev_alloc(3 * w * h)
    try:
        gpu.dev_upload(d_in, raws)
        res = gpu.process_batch_device(d_in, w * h, w, w, h, 3, d_out)
        masks = gpu.dev_download(d_out, (3, 1, h, w))
        for i in range(3):
            assert (masks[i, 0] == exps[i]["mask"][0]).all()
            check_moments(res[i][0], exps[i]["moments"][0])
    finally:
        gpu.dev_free(d_in)
        gpu.dev_free(d_out)


def test_stress_pipeline_small(gpu, oracle):
    """BASELINE config 5's operators (15x15 Gaussian, 9x9 ellipse, 8 ranges) at a size the oracle finishes fast."""
    bands = [(20 * i, 20 * i + 19) for i in range(8)]
    f = synth_frame(320, 180, 4, hue_bands=bands)
    ranges = [((lo, 100, 80), (hi, 255, 255)) for lo, hi in bands]
    morph = [ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)]
    gpu.set_pipeline(post_blur_k=15, morph=morph)
    gpu.set_ranges(ranges)
    res = gpu.process(f)
    exp = oracle.chain(f, post_blur_k=15, morph=morph_tuples(morph), ranges=ranges, stages=False)
    for r in range(8):
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all()
        check_moments(res[r], exp["moments"][r])
    assert sum(int(exp["moments"][r][0] > 0) for r in range(8)) >= 4


@pytest.mark.parametrize("hw", [(50, 37), (33, 259), (40, 1022), (64, 260), (21, 2052)])
def test_stress_operators_ragged_batches(oracle, hw):
    """The stage path's bit-plane tail (inRange planes as bits, morphology once per group of frames) at widths that are no
    multiple of 4 (byte loads in the inRange kernel, tile kernel behind it), a multiple of 4 but not of 16 or 32, and with a
    padded row pitch; batches of 3 frames, with and without a mask buffer; 3 and 8 ranges."""
    h, w = hw
    n = 3
    pitch = w * 3 + 20
    rng = np.random.default_rng(h * w)
    buf = rng.integers(0, 256, (n, h, pitch), dtype=np.uint8)
    buf[:, h // 4: h - h // 4, (w // 5) * 3: (w - w // 5) * 3] //= 2
    frames = np.ascontiguousarray(buf[:, :, : w * 3].reshape(n, h, w, 3))
    morph = [ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)]
    for ranges in ([((0, 0, 0), (256, 256, 140)), ((10, 0, 0), (90, 200, 256)), ((256, 0, 0), (0, 256, 256))],
                   [((20 * i, 0, 0), (20 * i + 30, 256, 200)) for i in range(8)]):
        nr = len(ranges)
        with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
            g.set_pipeline(post_blur_k=7, morph=morph)
            g.set_ranges(ranges)
            d_in, d_out = g.dev_alloc(buf.nbytes), g.dev_alloc(n * nr * w * h)
            try:
                g.dev_upload(d_in, buf)
                res = g.process_batch_device(d_in, h * pitch, pitch, w, h, n, d_out)
                masks = g.dev_download(d_out, (n, nr, h, w))
                res_nomask = g.process_batch_device(d_in, h * pitch, pitch, w, h, n, None)
            finally:
                g.dev_free(d_in)
                g.dev_free(d_out)
        assert res == res_nomask
        for i in range(n):
            exp = oracle.chain(frames[i], post_blur_k=7, morph=morph_tuples(morph), ranges=ranges, stages=False)
            for r in range(nr):
                assert (masks[i, r] == exp["mask"][r]).all(), (hw, i, r)
                check_moments(res[i][r], exp["moments"][r])
        assert sum(int(m[0]) for m in exp["moments"]) > 0


def test_process_batch_host(gpu, oracle):
    frames = synth_batch(160, 120, 7)
    p = PRESETS["baseline"]
    gpu.set_pipeline(**p)
    gpu.set_ranges([DEFAULT_RANGE])
    res = gpu.process_batch(frames)
    assert len(res) == 7
    for f, r in zip(frames, res):
        exp = oracle.chain(f, post_blur_k=5, morph=morph_tuples(p["morph"]), ranges=[DEFAULT_RANGE], stages=False)
        check_moments(r[0], exp["moments"][0])


@pytest.mark.parametrize("name", ["baseline", "polygon_path"])
@pytest.mark.parametrize("force_unfused", [False, True])
def test_process_batch_device(gpu, oracle, name, force_unfused):
    p = PRESETS[name]
    w, h, n = 640, 480, 5
    frames = synth_batch(w, h, n)
    ranges = [DEFAULT_RANGE, ((0, 0, 0), (60, 256, 256))]
    gpu.set_pipeline(force_unfused=force_unfused, **p)
    gpu.set_ranges(ranges)
    d_in = gpu.dev_alloc(frames.nbytes)
    d_out = gpu.dev_alloc(n * 2 * w * h)
    try:
        gpu.dev_upload(d_in, frames)
        res = gpu.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out)
        masks = gpu.dev_download(d_out, (n, 2, h, w))
        for i in range(n):
            exp = oracle.chain(frames[i], pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"],
                               morph=morph_tuples(p["morph"]), ranges=ranges, stages=False)
            for r in range(2):
                assert (masks[i, r] == exp["mask"][r]).all()
                check_moments(res[i][r], exp["moments"][r])
        # no mask output requested: moments only
        res2 = gpu.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, None)
        assert [[m[k] for k in KEYS] for fr in res2 for m in fr] == [[m[k] for k in KEYS] for fr in res for m in fr]
    finally:
        gpu.dev_free(d_in)
        gpu.dev_free(d_out)


def test_full_size_properties(gpu, oracle):
    """BASELINE config 2 size (1920x1200): size-independent properties on a batch + oracle on one frame."""
    w, h, n = 1920, 1200, 8
    frames = synth_batch(w, h, n, unique=2)
    gpu.set_pipeline(**PRESETS["baseline"])
    gpu.set_ranges([DEFAULT_RANGE])
    d_in = gpu.dev_alloc(frames.nbytes)
    d_out = gpu.dev_alloc(n * w * h)
    try:
        gpu.dev_upload(d_in, frames)
        res = gpu.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out)
        masks = gpu.dev_download(d_out, (n, 1, h, w))
        assert set(np.unique(masks)) <= {0, 255}
        for i in range(n):
            ys, xs = np.nonzero(masks[i, 0])
            assert res[i][0]["m00"] == len(xs) and res[i][0]["m10"] == int(xs.sum()) and res[i][0]["m01"] == int(ys.sum())
        # frame i+2 is frame i rolled down by one row (synth_batch): away from the wrap rows the masks shift with it
        assert (masks[2, 0][8:-8] == masks[0, 0][7:-9]).all()
        exp = oracle.chain(frames[1], post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE],
                           stages=False)
        assert (masks[1, 0] == exp["mask"][0]).all()
        check_moments(res[1][0], exp["moments"][0])
    finally:
        gpu.dev_free(d_in)
        gpu.dev_free(d_out)


@pytest.mark.parametrize("graph", ["0", "1"])
def test_streaming_graph_replay_and_invalidation(oracle, graph, monkeypatch):
    """submit/wait with direct launches and with a replayed per-slot hipGraph (IMGPROC_GRAPH=1); new ranges or a new
    frame size must re-capture."""
    monkeypatch.setenv("IMGPROC_GRAPH", graph)
    morph = morph_tuples(ip.MORPH_BASELINE)
    with ip.ImgProc(640, 480, n_slots=3) as g:
        g.set_pipeline(post_blur_k=5, morph=ip.MORPH_BASELINE)
        g.set_ranges([DEFAULT_RANGE])
        rng2 = ((0, 0, 0), (60, 256, 256))
        plan = [(640, 480, DEFAULT_RANGE)] * 4 + [(640, 480, rng2)] * 3 + [(320, 200, rng2)] * 3 + [(640, 480, DEFAULT_RANGE)] * 2
        pending = []
        for i, (w, h, rng) in enumerate(plan):
            slot = i % 3
            if len(pending) == 3:
                s0, f0, r0 = pending.pop(0)
                res = g.wait(s0)
                exp = oracle.chain(f0, post_blur_k=5, morph=morph, ranges=[r0], stages=False)
                check_moments(res[0], exp["moments"][0])
                assert (g.fetch_stage(s0, L.STAGE_MASK) == exp["mask"][0]).all()
            if i and plan[i][2] != plan[i - 1][2]:
                for s0, f0, r0 in pending:  # ranges are context state: drain before changing them
                    res = g.wait(s0)
                    exp = oracle.chain(f0, post_blur_k=5, morph=morph, ranges=[r0], stages=False)
                    check_moments(res[0], exp["moments"][0])
                pending = []
                g.set_ranges([rng])
            f = synth_frame(w, h, 100 + i)
            g.submit(slot, f)
            pending.append((slot, f, rng))
        for s0, f0, r0 in pending:
            res = g.wait(s0)
            exp = oracle.chain(f0, post_blur_k=5, morph=morph, ranges=[r0], stages=False)
            check_moments(res[0], exp["moments"][0])


@pytest.mark.parametrize("name", ["baseline", "pose_estimate", "c1"])
@pytest.mark.parametrize("hw", [(64, 96), (123, 516), (480, 640), (1200, 1920)])
def test_fused_second_order_moments(gpu, oracle, name, hw):
    """m20/m11/m02 from the same fused launch, exact integers; theta from the central moments."""
    if hw == (1200, 1920) and name != "baseline":
        pytest.skip("full size once")
    p = PRESETS[name]
    f = synth_frame(hw[1], hw[0], 21)
    ranges = [DEFAULT_RANGE, ((0, 0, 0), (90, 256, 256))]
    gpu.set_pipeline(second_order=True, **p)
    gpu.set_ranges(ranges)
    assert gpu.is_fused
    res = gpu.process(f)
    exp = oracle.chain(f, post_blur_k=p["post_blur_k"], morph=morph_tuples(p["morph"]), ranges=ranges, stages=False)
    for r in range(2):
        check_moments(res[r], exp["moments"][r], second=True)
        m = [float(v) for v in exp["moments"][r]]
        if m[0]:
            mu20, mu02, mu11 = m[3] - m[1] * m[1] / m[0], m[5] - m[2] * m[2] / m[0], m[4] - m[1] * m[2] / m[0]
            assert abs(res[r]["theta"] - 0.5 * np.arctan2(2 * mu11, mu20 - mu02)) < 1e-9


def test_plans(gpu):
    """Which kernels each reference chain runs on (imgproc_pipeline_plan)."""
    gpu.set_ranges([DEFAULT_RANGE])
    expect = {"pose_estimate": "fused chain kernel", "baseline": "fused chain kernel", "c1": "fused chain kernel",
              "pose_estimate_full": "front kernel + bit-plane morphology (3 groups)",
              "polygon_path": "front kernel + bit-plane morphology (5 groups)"}
    for name, want in expect.items():
        gpu.set_pipeline(**PRESETS[name])
        assert gpu.plan == want, (name, gpu.plan)
    gpu.set_pipeline(pre_blur_k=3, post_blur_k=5, morph=[ip.E(3)])
    assert gpu.plan == "fused chain kernel"  # reach 1 + 2 + 1 = 4 columns: fits the halo lane
    gpu.set_pipeline(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_BASELINE)
    assert gpu.plan.startswith("front kernel")
    gpu.set_pipeline(post_blur_k=15, morph=[ip.E(9, L.ELLIPSE)])
    assert gpu.plan == "stage kernels + bit-plane morphology (1 groups)"
    gpu.set_pipeline(keep_stages=True, **PRESETS["polygon_path"])
    assert gpu.plan.startswith("stage kernels")
    gpu.set_pipeline(force_unfused=True, **PRESETS["polygon_path"])
    assert gpu.plan.startswith("stage kernels")


@pytest.mark.parametrize("pre,post,morph", [(3, 5, [ip.E(3)]), (3, 0, ip.MORPH_BASELINE), (3, 0, []), (3, 5, []),
                                            (3, 5, ip.MORPH_POSE_ESTIMATE), (0, 0, ip.MORPH_POLYGON_PATH)])
@pytest.mark.parametrize("hw", [(37, 64), (123, 516), (480, 640)])
def test_preblur_and_front_back_variants(gpu, oracle, pre, post, morph, hw):
    f = synth_frame(hw[1], hw[0], 31)
    ranges = [DEFAULT_RANGE, ((0, 0, 0), (90, 256, 256))]
    gpu.set_pipeline(pre_blur_k=pre, post_blur_k=post, morph=morph, second_order=True)
    gpu.set_ranges(ranges)
    assert not gpu.plan.startswith("stage kernels")
    res = gpu.process(f)
    exp = oracle.chain(f, pre_blur_k=pre, post_blur_k=post, morph=morph_tuples(morph), ranges=ranges, stages=False)
    for r in range(2):
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all()
        check_moments(res[r], exp["moments"][r], second=True)


def test_two_batches_in_flight(gpu, oracle):
    """imgproc_batch_enqueue / imgproc_batch_wait: two tickets in flight deliver the same results as the one-call form."""
    h, w, n = 64, 256, 4
    fa, fb = synth_batch(w, h, n, seed=11), synth_batch(w, h, n, seed=12)
    gpu.set_pipeline(**PRESETS["baseline"])
    gpu.set_ranges([DEFAULT_RANGE])
    d_a, d_b = gpu.dev_alloc(fa.nbytes), gpu.dev_alloc(fb.nbytes)
    d_ma, d_mb = gpu.dev_alloc(n * w * h), gpu.dev_alloc(n * w * h)
    try:
        gpu.dev_upload(d_a, fa)
        gpu.dev_upload(d_b, fb)
        ref_a = gpu.process_batch_device(d_a, w * h * 3, w * 3, w, h, n, d_ma)
        ref_b = gpu.process_batch_device(d_b, w * h * 3, w * 3, w, h, n, d_mb)
        gpu.batch_enqueue(0, d_a, w * h * 3, w * 3, w, h, n, d_ma)
        gpu.batch_enqueue(1, d_b, w * h * 3, w * 3, w, h, n, d_mb)
        with pytest.raises(ip.ImgProcError):
            gpu.batch_enqueue(1, d_b, w * h * 3, w * 3, w, h, n, d_mb)  # ticket busy
        with pytest.raises(ip.ImgProcError):
            gpu.process_batch_device(d_a, w * h * 3, w * 3, w, h, n, d_ma)  # uses ticket 0
        assert gpu.batch_wait(1, n=n) == ref_b
        assert gpu.batch_wait(0, n=n) == ref_a
        with pytest.raises(ip.ImgProcError):
            gpu.batch_wait(0)
        exp = oracle.chain(fb[2], post_blur_k=5, morph=morph_tuples(ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE], stages=False)
        assert (gpu.dev_download(d_mb, (n, h, w))[2] == exp["mask"][0]).all()
        check_moments(ref_b[2][0], exp["moments"][0])
    finally:
        for p in (d_a, d_b, d_ma, d_mb):
            gpu.dev_free(p)


@pytest.mark.parametrize("preset", ["baseline", "polygon_path", "stress", "bayer_baseline"])
def test_two_tickets_overlap_on_two_streams(gpu, oracle, preset):
    """Ticket 1 runs on a stream of its own (include/imgproc.h, imgproc_batch_enqueue), with its own inRange planes on the
    two-launch paths: two DIFFERENT batches in flight, several rounds, each equal to its one-call result and to the oracle;
    the kernel spans of the two tickets are reported on one clock."""
    h, w, n = 150, 520, 3
    bayer = preset == "bayer_baseline"
    kw = dict(pre_blur_k=0, post_blur_k=15, morph=[ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)]) if preset == "stress" else PRESETS[preset.replace("bayer_", "")]
    ranges = [((20 * r, 40, 40), (20 * r + 50, 255, 255)) for r in range(8)] if preset == "stress" else [DEFAULT_RANGE]
    nr = len(ranges)
    if bayer:
        fa, fb = (np.stack([synth_bayer_rggb(w, h, i, seed=s) for i in range(n)]) for s in (21, 22))
    else:
        fa, fb = synth_batch(w, h, n, seed=21), synth_batch(w, h, n, seed=22)
    gpu.set_pipeline(encoding=L.ENC_BAYER_RGGB8 if bayer else L.ENC_BGR8, **kw)
    gpu.set_ranges(ranges)
    cn = 1 if bayer else 3
    d_a, d_b = gpu.dev_alloc(fa.nbytes), gpu.dev_alloc(fb.nbytes)
    d_ma, d_mb = gpu.dev_alloc(n * nr * w * h), gpu.dev_alloc(n * nr * w * h)
    try:
        gpu.dev_upload(d_a, fa)
        gpu.dev_upload(d_b, fb)
        ref_a = gpu.process_batch_device(d_a, w * h * cn, w * cn, w, h, n, d_ma)
        mask_a = gpu.dev_download(d_ma, (n, nr, h, w)).copy()
        ref_b = gpu.process_batch_device(d_b, w * h * cn, w * cn, w, h, n, d_ma)
        mask_b = gpu.dev_download(d_ma, (n, nr, h, w)).copy()
        assert ref_a != ref_b
        d_m = [d_ma, d_mb, gpu.dev_alloc(n * nr * w * h)]
        for depth in (2, 3):  # batch k is A on even k, B on odd k, on ticket k % depth; collected after batch k + depth - 1 is enqueued
            gpu.set_profiling(True)
            spans = []
            for k in range(9):
                gpu.batch_enqueue(k % depth, d_b if k & 1 else d_a, w * h * cn, w * cn, w, h, n, d_m[k % depth])
                if k >= depth - 1:
                    j = k - depth + 1
                    assert gpu.batch_wait(j % depth, n=n) == (ref_b if j & 1 else ref_a), (depth, j)
                    spans.append(gpu.last_kernel_span())
            for j in range(9 - depth + 1, 9):
                assert gpu.batch_wait(j % depth, n=n) == (ref_b if j & 1 else ref_a), (depth, j)
                spans.append(gpu.last_kernel_span())
            gpu.set_profiling(False)
            for t in range(depth):  # the last batch each ticket took: k = 8, 7 (, 6)
                k = max(j for j in range(9) if j % depth == t)
                assert (gpu.dev_download(d_m[t], (n, nr, h, w)) == (mask_b if k & 1 else mask_a)).all(), (depth, t)
            assert all(0 <= t0 < t1 for t0, t1 in spans)
            assert all(b[1] > a[1] for a, b in zip(spans, spans[depth:]))  # one ticket's batches run in order
        with pytest.raises(ip.ImgProcError):
            gpu.batch_enqueue(L.MAX_TICKETS, d_a, w * h * cn, w * cn, w, h, n, d_ma)
        exp = oracle.chain(fb[1], encoding=L.ENC_BAYER_RGGB8 if bayer else L.ENC_BGR8, morph=morph_tuples(kw["morph"]), ranges=ranges,
                           stages=False, **{k: v for k, v in kw.items() if k != "morph"})
        for r in range(nr):
            assert (mask_b[1, r] == exp["mask"][r]).all()
            check_moments(ref_b[1][r], exp["moments"][r])
    finally:
        for p in [d_a, d_b] + (d_m if "d_m" in locals() else [d_ma, d_mb]):
            gpu.dev_free(p)


@pytest.mark.parametrize("w", [32, 33, 37, 101, 249, 250, 251, 255, 257, 499, 501, 997])
@pytest.mark.parametrize("preset", ["pose_estimate", "baseline", "polygon_path"])
def test_odd_widths_stay_on_the_fast_path(gpu, oracle, w, preset):
    """Widths that are not a multiple of 4 (tracker ROI crops, pose_estimate.cpp:197-202): the last strip is
    right-aligned, dword accesses are unaligned, and the chain still runs as one launch (front + bit planes: two)."""
    h, n = 45, 2
    kw = PRESETS[preset]
    for enc in ("bgr", "bayer"):
        if enc == "bayer" and (w & 1):
            continue
        rng = np.random.default_rng(w * 3 + len(preset))
        frames = rng.integers(0, 256, (n, h, w) if enc == "bayer" else (n, h, w, 3), dtype=np.uint8)
        frames[:, 5:30, 3:w - 2] //= 2  # some structure for the morphology
        ranges = [((0, 30, 40), (100, 256, 256))]
        gpu.set_pipeline(encoding=L.ENC_BAYER_RGGB8 if enc == "bayer" else L.ENC_BGR8, second_order=True, **kw)
        gpu.set_ranges(ranges)
        gpu.set_profiling(True)
        d_in, d_out = gpu.dev_alloc(frames.nbytes), gpu.dev_alloc(n * w * h)
        try:
            gpu.dev_upload(d_in, frames)
            cn = 1 if enc == "bayer" else 3
            res = gpu.process_batch_device(d_in, w * h * cn, w * cn, w, h, n, d_out)
            assert gpu.last_kernel_ms()[1] == (1 if preset != "polygon_path" else 2), (w, preset, enc)
            masks = gpu.dev_download(d_out, (n, h, w))
        finally:
            gpu.dev_free(d_in)
            gpu.dev_free(d_out)
        for i in range(n):
            exp = oracle.chain(frames[i], encoding=2 if enc == "bayer" else 0, pre_blur_k=kw["pre_blur_k"],
                               post_blur_k=kw["post_blur_k"], morph=morph_tuples(kw["morph"]), ranges=ranges, stages=False)
            assert (masks[i] == exp["mask"][0]).all(), (w, preset, enc, i)
            check_moments(res[i][0], exp["moments"][0], second=True)
    gpu.set_profiling(False)


@pytest.mark.parametrize("nr", [3, 4, 5, 8])
@pytest.mark.parametrize("preset", ["baseline", "polygon_path"])
def test_up_to_eight_ranges_in_one_launch(gpu, oracle, nr, preset):
    """3..8 ranges (imgproc_set_ranges takes 1..8) run in the capacity-4 / capacity-8 variants of the chain kernel."""
    h, w, n = 70, 500, 2
    kw = PRESETS[preset]
    frames = synth_batch(w, h, n, seed=77, hue_bands=tuple((20 * i, 20 * i + 19) for i in range(nr)))
    ranges = [((20 * i, 60, 60), (20 * i + 19, 256, 256)) for i in range(nr)]
    ranges[-1] = ((200, 0, 0), (100, 256, 256))  # an empty H interval: never matches
    gpu.set_pipeline(second_order=True, **kw)
    gpu.set_ranges(ranges)
    gpu.set_profiling(True)
    d_in, d_out = gpu.dev_alloc(frames.nbytes), gpu.dev_alloc(n * nr * w * h)
    try:
        gpu.dev_upload(d_in, frames)
        res = gpu.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out)
        assert gpu.last_kernel_ms()[1] == (1 if preset == "baseline" else 2)
        masks = gpu.dev_download(d_out, (n, nr, h, w))
    finally:
        gpu.dev_free(d_in)
        gpu.dev_free(d_out)
        gpu.set_profiling(False)
    for i in range(n):
        exp = oracle.chain(frames[i], pre_blur_k=kw["pre_blur_k"], post_blur_k=kw["post_blur_k"],
                           morph=morph_tuples(kw["morph"]), ranges=ranges, stages=False)
        for r in range(nr):
            assert (masks[i, r] == exp["mask"][r]).all(), (nr, preset, i, r)
            check_moments(res[i][r], exp["moments"][r], second=True)
        assert exp["moments"][nr - 1][0] == 0 and (preset != "baseline" or sum(int(m[0]) for m in exp["moments"]) > 0)
    res1 = gpu.process(frames[0])  # the streaming path takes the same variants
    exp = oracle.chain(frames[0], pre_blur_k=kw["pre_blur_k"], post_blur_k=kw["post_blur_k"], morph=morph_tuples(kw["morph"]),
                       ranges=ranges, stages=False)
    for r in range(nr):
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all()
        check_moments(res1[r], exp["moments"][r], second=True)


STRIP_PRESETS = dict(PRESETS)
# BASELINE configs[4]'s operators: the strip kernel's instantiation for an element of row spans (9x9 ellipse open)
STRIP_PRESETS["stress"] = dict(pre_blur_k=0, post_blur_k=15, morph=[ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)])


@pytest.mark.parametrize("preset", ["polygon_path", "pose_estimate_full", "stress"])
@pytest.mark.parametrize("hw", [(4, 8), (37, 64), (100, 252), (123, 516), (300, 1920), (211, 2100), (96, 4096)])
@pytest.mark.parametrize("seg_rows", [0, 48])
def test_bit_plane_strip_kernel(gpu, oracle, monkeypatch, preset, hw, seg_rows):
    """The strip form of the bit-plane morphology (instantiated for the two morphOps lists of the reference and for the
    stress preset's elliptical open; normally taken only for large jobs) forced on small and wide frames: one strip and
    several, several segments per plane, words cut by the right edge; equal to the oracle and to the tile kernel,
    second-order sums included."""
    h, w = hw
    n = 2
    kw = STRIP_PRESETS[preset]
    frames = np.random.default_rng(h + w).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[:, h // 5: h - h // 6, w // 7: w - w // 5] //= 3  # a large dark region: something survives the erosions
    ranges = [((0, 0, 0), (256, 256, 110)), DEFAULT_RANGE]
    out = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("IMGPROC_BITSTRIP", mode)
        if seg_rows:
            monkeypatch.setenv("IMGPROC_BITSTRIP_SEG_ROWS", str(seg_rows))
        with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
            g.set_pipeline(second_order=True, **kw)
            g.set_ranges(ranges)
            d_in, d_out = g.dev_alloc(frames.nbytes), g.dev_alloc(n * 2 * w * h)
            try:
                g.dev_upload(d_in, frames)
                res = g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out)
                out[mode] = (res, g.dev_download(d_out, (n, 2, h, w)))
            finally:
                g.dev_free(d_in)
                g.dev_free(d_out)
    assert out["2"][0] == out["0"][0]
    assert (out["2"][1] == out["0"][1]).all()
    for i in range(n):
        exp = oracle.chain(frames[i], pre_blur_k=kw["pre_blur_k"], post_blur_k=kw["post_blur_k"], morph=morph_tuples(kw["morph"]),
                           ranges=ranges, stages=False)
        for r in range(2):
            assert (out["2"][1][i, r] == exp["mask"][r]).all(), (preset, hw, i, r)
            check_moments(out["2"][0][i][r], exp["moments"][r], second=True)
    assert sum(int(m[0]) for m in exp["moments"]) > 0 or h < 30
