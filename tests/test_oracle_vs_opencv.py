"""CPU-only, opportunistic (SURVEY §8c "golden vectors"): when a real OpenCV is importable on the host that runs
the tests, every stage of the C oracle is compared with it bit for bit -- if they disagree, OpenCV wins and
oracle/cpu_ref.c is what must change.  In this image (and on the GPU box, which runs the same image) `cv2` is
absent, so the whole module SKIPS and parity stays UNPINNED; nothing here installs or fetches anything.
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed in this image: parity stays unpinned")

from image_processing_amd.synth import synth_bayer_rggb, synth_frame  # noqa: E402

RECT, CROSS, ELLIPSE = 0, 1, 2
CV_SHAPE = {RECT: cv2.MORPH_RECT, CROSS: cv2.MORPH_CROSS, ELLIPSE: cv2.MORPH_ELLIPSE}


def _frames():
    rng = np.random.default_rng(7)
    yield rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    yield synth_frame(160, 120, 5)
    g = np.arange(0, 256, 3, dtype=np.uint8)
    yield np.ascontiguousarray(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(86, -1, 3))


def test_bgr2hsv():  # objectPolygonPath.cpp:183, pose_estimate.cpp:203
    from oracle import Oracle

    o = Oracle()
    for f in _frames():
        assert (o.bgr2hsv(f) == cv2.cvtColor(f, cv2.COLOR_BGR2HSV)).all()
        assert (o.bgr2hsv(f, rgb=True) == cv2.cvtColor(f, cv2.COLOR_RGB2HSV)).all()


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11, 13, 15])
def test_gaussian_blur(k):  # objectPolygonPath.cpp:182,184 (k = 3, 5); 15 is BASELINE.json's stress preset
    from oracle import Oracle

    o = Oracle()
    for f in _frames():
        assert (o.gaussian_blur(f, k) == cv2.GaussianBlur(f, (k, k), 0, 0)).all(), k
    one = np.random.default_rng(k).integers(0, 256, (1, 9, 3), dtype=np.uint8)
    assert (o.gaussian_blur(one, k) == cv2.GaussianBlur(one, (k, k), 0, 0)).all()


def test_in_range():  # objectPolygonPath.cpp:186, pose_estimate.cpp:205
    from oracle import Oracle

    o = Oracle()
    f = next(_frames())
    for lo, hi in [((0, 0, 0), (256, 256, 256)), ((100, 120, 80), (130, 255, 255)), ((256, 0, 0), (0, 256, 256)),
                   ((10, 20, 30), (10, 20, 30)), ((-5, 0, 0), (300, 255, 255)), ((0, 0, 0), (-1, 255, 255))]:
        exp = cv2.inRange(f, np.array(lo, np.float64), np.array(hi, np.float64))
        assert (o.in_range(f, lo, hi) == exp).all(), (lo, hi)


@pytest.mark.parametrize("shape,kw,kh", [(RECT, 3, 3), (RECT, 8, 8), (ELLIPSE, 9, 9), (ELLIPSE, 5, 7), (CROSS, 5, 5),
                                         (ELLIPSE, 15, 15), (RECT, 2, 5)])
def test_morphology(shape, kw, kh):  # objectPolygonPath.cpp:31-32,416-432; pose_estimate.cpp:40-41,407-414
    from oracle import Oracle

    o = Oracle()
    e_cv = cv2.getStructuringElement(CV_SHAPE[shape], (kw, kh))
    e = o.structuring_element(shape, kw, kh)
    assert (e == e_cv).all()
    m = np.where(np.random.default_rng(kw * 31 + kh).random((57, 75)) < 0.6, 255, 0).astype(np.uint8)
    assert (o.erode(m, e) == cv2.erode(m, e_cv)).all()
    assert (o.dilate(m, e) == cv2.dilate(m, e_cv)).all()


def test_bayer_rggb():  # cv_bridge bayer_rggb8 -> bgr8 = COLOR_BayerBG2BGR (pose_estimate.cpp:180)
    from oracle import Oracle

    o = Oracle()
    for raw in (synth_bayer_rggb(96, 64, 3), np.random.default_rng(2).integers(0, 256, (38, 54), dtype=np.uint8)):
        assert (o.bayer_rggb2bgr(raw) == cv2.cvtColor(raw, cv2.COLOR_BayerBG2BGR)).all()


def test_moments_and_chain():  # cv::moments(mask, true); sequence of objectPolygonPath.cpp:182-188,414-433
    from oracle import Oracle

    o = Oracle()
    f = synth_frame(320, 200, 9)
    lo, hi = (100, 120, 80), (130, 255, 255)
    x = cv2.GaussianBlur(f, (3, 3), 0, 0)
    x = cv2.cvtColor(x, cv2.COLOR_BGR2HSV)
    x = cv2.GaussianBlur(x, (5, 5), 0, 0)
    m = cv2.inRange(x, np.array(lo, np.float64), np.array(hi, np.float64))
    e3 = cv2.getStructuringElement(cv2.MORPH_RECT, (3, 3))
    e8 = cv2.getStructuringElement(cv2.MORPH_RECT, (8, 8))
    morph = []
    for op, e, k in [(0, e3, 3), (0, e3, 3), (1, e8, 8), (0, e3, 3), (0, e3, 3), (1, e8, 8)]:
        m = cv2.erode(m, e) if op == 0 else cv2.dilate(m, e)
        morph.append((op, RECT, k, k))
    r = o.chain(f, pre_blur_k=3, post_blur_k=5, morph=morph, ranges=[(lo, hi)])
    assert (r["mask"][0] == m).all()
    mm = cv2.moments(m, True)
    got = r["moments"][0].tolist()
    assert got[:3] == [int(mm["m00"]), int(mm["m10"]), int(mm["m01"])]
    assert got[3:] == [int(mm["m20"]), int(mm["m11"]), int(mm["m02"])]
