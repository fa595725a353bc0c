"""ABI behaviour under the call orders a pipelined caller produces: a frame in flight on a slot while a device batch
or a single-stage call runs, parameters changed between submit and wait, frames shorter than the blurs' reach, and two
contexts driven from two threads (SURVEY §8b "Threading")."""
import threading

import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch, synth_frame

from test_gpu_chain import KEYS, PRESETS, check_moments, morph_tuples

pytestmark = pytest.mark.gpu

RANGE2 = ((0, 0, 0), (90, 256, 256))


@pytest.fixture(scope="module")
def gpu():
    with ip.ImgProc(640, 480, n_slots=2, max_batch=4) as g:
        yield g


@pytest.mark.parametrize("keep", [False, True])
def test_batch_call_leaves_a_submitted_slot_alone(gpu, oracle, keep):
    """submit(0) -> batch_enqueue(stage by stage) -> mask_and -> wait(0) -> fetch_stage: the batch and single-stage paths
    work in the context's own planes, so slot 0's frame, result and stage images survive them."""
    p = PRESETS["polygon_path"]
    w, h, n = 320, 200, 3
    f = synth_frame(w, h, 41)
    batch = synth_batch(w, h, n, seed=42)
    gpu.set_pipeline(keep_stages=keep, force_unfused=True, **p)
    gpu.set_ranges([DEFAULT_RANGE])
    d_in, d_out = gpu.dev_alloc(batch.nbytes), gpu.dev_alloc(n * w * h)
    try:
        gpu.dev_upload(d_in, batch)
        gpu.submit(0, f)
        gpu.batch_enqueue(0, d_in, w * h * 3, w * 3, w, h, n, d_out)
        bres = gpu.batch_wait(0, n=n)
        other = gpu.mask_and(batch[0], np.full((h, w), 255, np.uint8))
        assert (other == batch[0]).all()
        res = gpu.wait(0)  # was IMGPROC_ESTATE when the batch path borrowed slot 0
        masks = gpu.dev_download(d_out, (n, h, w))
    finally:
        gpu.dev_free(d_in)
        gpu.dev_free(d_out)
    kw = dict(pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"], morph=morph_tuples(p["morph"]), ranges=[DEFAULT_RANGE])
    exp = oracle.chain(f, stages=True, **kw)
    check_moments(res[0], exp["moments"][0])
    assert (gpu.fetch_stage(0, L.STAGE_MASK) == exp["mask"][0]).all()
    assert (gpu.fetch_stage(0, L.STAGE_BGR) == exp["bgr"]).all()
    assert (gpu.fetch_stage(0, L.STAGE_HSV) == exp["hsv"]).all()
    assert (gpu.fetch_stage(0, L.STAGE_BLUR) == exp["blur"]).all()
    assert (gpu.fetch_stage(0, L.STAGE_MASKED) == (exp["bgr"] & exp["mask"][0][:, :, None])).all()
    for i in range(n):
        e = oracle.chain(batch[i], stages=False, **kw)
        assert (masks[i] == e["mask"][0]).all()
        check_moments(bres[i][0], e["moments"][0])


def test_wait_reports_the_ranges_of_the_submit(gpu, oracle):
    """set_ranges / set_pipeline between submit and wait (dynamic_reconfigure arriving while a frame is in flight):
    the result is the one the frame was submitted for."""
    p = PRESETS["baseline"]
    f = synth_frame(320, 200, 43)
    gpu.set_pipeline(second_order=True, **p)
    gpu.set_ranges([DEFAULT_RANGE, RANGE2])
    gpu.submit(0, f)
    gpu.set_ranges([RANGE2])
    gpu.set_pipeline(**p)  # second order off
    res = gpu.wait(0)
    exp = oracle.chain(f, post_blur_k=5, morph=morph_tuples(p["morph"]), ranges=[DEFAULT_RANGE, RANGE2], stages=False)
    assert len(res) == 2
    for r in range(2):
        check_moments(res[r], exp["moments"][r], second=True)
    gpu.submit(1, f)  # and the other way round: fewer ranges in flight than configured at wait
    gpu.set_ranges([DEFAULT_RANGE, RANGE2, DEFAULT_RANGE])
    res = gpu.wait(1)
    assert len(res) == 1
    check_moments(res[0], exp["moments"][1])


@pytest.mark.parametrize("h", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("w", [8, 32])
@pytest.mark.parametrize("preset", ["baseline", "polygon_path", "pose_estimate"])
def test_frames_shorter_than_the_blur_reach(gpu, oracle, preset, h, w):
    """BORDER_REFLECT_101 reflects more than once when h is below the blurs' reach; such frames must not take the chain
    kernel's single reflection."""
    p = PRESETS[preset]
    f = np.random.default_rng(100 * h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ranges = [((0, 0, 0), (100, 256, 256)), ((20, 40, 40), (160, 230, 230))]
    gpu.set_pipeline(**p)
    gpu.set_ranges(ranges)
    res = gpu.process(f)
    exp = oracle.chain(f, pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"], morph=morph_tuples(p["morph"]),
                       ranges=ranges, stages=False)
    for r in range(2):
        assert (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp["mask"][r]).all(), (preset, h, w, r)
        check_moments(res[r], exp["moments"][r])


@pytest.mark.parametrize("w", [1, 2, 3])
def test_narrow_tall_frames_with_eight_ranges(oracle, w):
    """A context sized for frames narrower than 4 px: the inRange BIT planes (4 bytes per row and range) are larger than
    the byte planes (w bytes per row and range) there, and the slot's plane buffer must hold them."""
    h = 3000
    ranges = [((10 * r, 0, 0), (10 * r + 60, 256, 256)) for r in range(8)]
    f = np.random.default_rng(7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    p = PRESETS["polygon_path"]
    with ip.ImgProc(w, h, n_slots=1) as g:
        g.set_pipeline(**p)
        g.set_ranges(ranges)
        res = g.process(f)
        masks = [g.fetch_stage(0, L.STAGE_MASK, r) for r in range(8)]
    exp = oracle.chain(f, pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"], morph=morph_tuples(p["morph"]),
                       ranges=ranges, stages=False)
    for r in range(8):
        assert (masks[r] == exp["mask"][r]).all(), (w, r)
        check_moments(res[r], exp["moments"][r])


def test_two_contexts_two_threads(oracle):
    """Different contexts are independent and may be driven from different threads (here: both on device 0)."""
    jobs = [("baseline", 320, 200, 51), ("polygon_path", 256, 160, 52)]
    out, err = {}, []

    def run(name, w, h, seed):
        try:
            p = PRESETS[name]
            with ip.ImgProc(w, h, n_slots=2, max_batch=2) as g:
                g.set_pipeline(**p)
                g.set_ranges([DEFAULT_RANGE])
                got = []
                for i in range(12):
                    f = synth_frame(w, h, seed + i)
                    g.submit(i & 1, f)
                    res = g.wait(i & 1)
                    got.append((f, res[0], g.fetch_stage(i & 1, L.STAGE_MASK)))
                out[name] = got
        except Exception as e:  # noqa: BLE001
            err.append((name, e))

    ts = [threading.Thread(target=run, args=j) for j in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for name, w, h, seed in jobs:
        p = PRESETS[name]
        for f, res, mask in out[name]:
            exp = oracle.chain(f, pre_blur_k=p["pre_blur_k"], post_blur_k=p["post_blur_k"], morph=morph_tuples(p["morph"]),
                               ranges=[DEFAULT_RANGE], stages=False)
            assert (mask == exp["mask"][0]).all(), name
            check_moments(res, exp["moments"][0])
