"""GPU parity, stage by stage: every cv:: call on the reference's path, through the C ABI, against the
oracle on the same seeded inputs.  Bar: bit-exact (u8 / integer work)."""
import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import synth_bayer_rggb, synth_frame

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (1, 13), (13, 1), (17, 13), (16, 64), (37, 250), (64, 256), (123, 517), (480, 640), (150, 1000)]


@pytest.fixture(scope="module")
def gpu():
    with ip.ImgProc(1920, 1200, n_slots=2, max_batch=4) as g:
        yield g


@pytest.mark.parametrize("hw", SHAPES)
def test_bgr2hsv(gpu, oracle, hw):
    img = np.random.default_rng(hw[0] * 1000 + hw[1]).integers(0, 256, hw + (3,), dtype=np.uint8)
    assert (gpu.cvtColor_BGR2HSV(img) == oracle.bgr2hsv(img)).all()
    assert (gpu.cvtColor_BGR2HSV(img, rgb=True) == oracle.bgr2hsv(img, rgb=True)).all()


def test_bgr2hsv_all_colours_lattice(gpu, oracle, kat):
    g = np.arange(0, 256, 3, dtype=np.uint8)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(86, -1, 3)
    assert (gpu.cvtColor_BGR2HSV(lat) == oracle.bgr2hsv(lat)).all()
    src = np.array([[p for p, _ in kat["bgr2hsv"]]], np.uint8)
    exp = np.array([[q for _, q in kat["bgr2hsv"]]], np.uint8)
    assert (gpu.cvtColor_BGR2HSV(src) == exp).all()


@pytest.mark.parametrize("hw", SHAPES)
@pytest.mark.parametrize("k", [3, 5, 7, 9, 11, 13, 15, 21, 31])
def test_gaussian(gpu, oracle, hw, k):
    rng = np.random.default_rng(hw[0] * 31 + hw[1] + k)
    img3 = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    assert (gpu.GaussianBlur(img3, k) == oracle.gaussian_blur(img3, k)).all()
    img1 = rng.integers(0, 256, hw, dtype=np.uint8)
    assert (gpu.GaussianBlur(img1, k) == oracle.gaussian_blur(img1, k)).all()


def test_gaussian_kats(gpu, kat):
    img = np.zeros((21, 21), np.uint8)
    img[10, 1] = 255
    out = gpu.GaussianBlur(img, 5)
    assert out[10, 0] == 48 and out[10, 1] == 42  # reflect-101 traps of SURVEY §A.8
    img[:] = 0
    img[10, 10] = 255
    out = gpu.GaussianBlur(img, 5)
    assert out[10, 10] == 36 and out[10, 11] == 24 and out[11, 11] == 16 and out[12, 12] == 1


@pytest.mark.parametrize("hw", SHAPES)
def test_in_range(gpu, oracle, hw):
    img = np.random.default_rng(hw[0] + 7 * hw[1]).integers(0, 256, hw + (3,), dtype=np.uint8)
    for lo, hi in [((0, 0, 0), (256, 256, 256)), ((100, 120, 80), (130, 255, 255)), ((256, 0, 0), (0, 256, 256)),
                   ((10, 20, 30), (200, 220, 230)), ((-5, 0, 0), (300, 255, 255)), ((0, 0, 0), (-1, 255, 255)),
                   ((128, 128, 128), (128, 128, 128))]:
        assert (gpu.inRange(img, lo, hi) == oracle.in_range(img, lo, hi)).all()


@pytest.mark.parametrize("hw", SHAPES)
def test_morphology(gpu, oracle, hw):
    rng = np.random.default_rng(hw[0] * 3 + hw[1])
    binm = np.where(rng.random(hw) < 0.6, 255, 0).astype(np.uint8)
    gray = rng.integers(0, 256, hw, dtype=np.uint8)
    ops = [(L.RECT, 3, 3), (L.RECT, 8, 8), (L.ELLIPSE, 9, 9), (L.CROSS, 5, 5), (L.RECT, 2, 5), (L.ELLIPSE, 15, 15),
           (L.RECT, 31, 31)]
    for img in (binm, gray):
        for shape, kw, kh in ops:
            e = oracle.structuring_element(shape, kw, kh)
            for kind in (L.ERODE, L.DILATE):
                got = gpu.morphology(img, ip.MorphOp(kind, shape, kw, kh))
                assert (got == oracle.morph(img, kind, e)).all(), (shape, kw, kh, kind)
    e = (rng.random((7, 5)) < 0.5).astype(np.uint8)
    e[3, 2] = 1
    for kind in (L.ERODE, L.DILATE):
        got = gpu.morphology(gray, ip.MorphOp(kind, L.CUSTOM, 5, 7, (1, 6), e))
        assert (got == oracle.morph(gray, kind, e, (1, 6))).all()


def test_morph_kats(gpu):
    img = np.zeros((40, 40), np.uint8)
    img[20, 17] = 255
    ys, xs = np.nonzero(gpu.dilate(img, 8))
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (14, 21, 17, 24)
    assert (gpu.erode(np.full((7, 9), 255, np.uint8), 3) == 255).all()


@pytest.mark.parametrize("hw", SHAPES + [(1200, 1920)])
def test_moments(gpu, oracle, hw):
    img = np.where(np.random.default_rng(hw[0] ^ hw[1]).random(hw) < 0.4, 255, 0).astype(np.uint8)
    img[0, 0] = 7  # non-binary nonzero counts as set
    m = gpu.moments(img)
    exp = oracle.moments(img)
    assert [m[k] for k in ("m00", "m10", "m01", "m20", "m11", "m02")] == exp.tolist()
    assert m["cx"] == float(exp[1]) / float(exp[0]) and m["cy"] == float(exp[2]) / float(exp[0])
    z = gpu.moments(np.zeros(hw, np.uint8))
    assert z["found"] == 0 and z["cx"] == -1.0 and z["m00"] == 0


@pytest.mark.parametrize("hw", [(8, 10), (9, 11), (3, 3), (2, 9), (16, 3), (1, 1), (61, 127), (480, 640)])
def test_demosaic(gpu, oracle, hw):
    raw = np.random.default_rng(hw[0] + hw[1]).integers(0, 256, hw, dtype=np.uint8)
    assert (gpu.demosaic_rggb(raw) == oracle.bayer_rggb2bgr(raw)).all()


def test_mask_and(gpu, oracle):
    f = synth_frame(250, 37, 1)
    m = np.where(np.random.default_rng(0).random((37, 250)) < 0.5, 255, 0).astype(np.uint8)
    assert (gpu.mask_and(f, m) == oracle.mask_and(f, m)).all()


def test_strided_input(gpu, oracle):
    big = np.random.default_rng(9).integers(0, 256, (50, 70, 3), dtype=np.uint8)
    view = big[5:42, 3:64]  # row stride 210, not contiguous, odd offsets
    lib = L.load()
    out = np.empty((37, 61, 3), np.uint8)
    rc = lib.imgproc_cvt_bgr2hsv(gpu._ctx, view.ctypes.data, view.strides[0], out.ctypes.data, out.strides[0], 61, 37, 0)
    assert rc == 0
    assert (out == oracle.bgr2hsv(np.ascontiguousarray(view))).all()


def test_errors(gpu):
    with pytest.raises(ip.ImgProcError) as e:
        gpu.GaussianBlur(np.zeros((4, 4), np.uint8), 4)
    assert e.value.status == L.EINVAL
    with pytest.raises(ip.ImgProcError) as e:
        gpu.cvtColor_BGR2HSV(np.zeros((1300, 2000, 3), np.uint8))
    assert e.value.status == L.ESIZE
    with pytest.raises(ip.ImgProcError) as e:
        gpu.wait(1)
    assert e.value.status == L.ESTATE
    with pytest.raises(ip.ImgProcError):
        gpu.morphology(np.zeros((4, 4), np.uint8), ip.MorphOp(L.ERODE, L.RECT, 40, 3))


@pytest.mark.parametrize("hw,cn,k", [((1000, 1234), 3, 15), ((700, 2051), 1, 15), ((333, 901), 3, 9), ((40, 4096), 3, 7), ((1200, 1920), 3, 15)])
def test_gaussian_long_kernels_on_matrix_cores(oracle, monkeypatch, hw, cn, k):
    """k = 7..15 run as banded i8 contractions on the MFMA units (gauss_mfma.hip): several 16-row segments and 64-byte
    strips per image, row bytes that are no multiple of 64 or of 8, reflect-101 on all four borders; bit-identical to the
    oracle and to the strip kernel (IMGPROC_GAUSS_MFMA=0)."""
    rng = np.random.default_rng(hw[0] + hw[1] + k)
    img = rng.integers(0, 256, hw + ((3,) if cn == 3 else ()), dtype=np.uint8)
    img[: hw[0] // 3] |= 0xF0  # bright rows: sums near the top of the 8.8 range
    exp = oracle.gaussian_blur(img, k)
    with ip.ImgProc(hw[1], hw[0], n_slots=1) as g:
        got = g.GaussianBlur(img, k)
        monkeypatch.setenv("IMGPROC_GAUSS_MFMA", "0")
        other = g.GaussianBlur(img, k)
    assert (got == exp).all()
    assert (other == exp).all()


# ---- the fused HSV + long Gaussian + inRange kernel (gauss_fused.hip): bit planes pinned to the oracle ----------------
FUSED_RANGES8 = [((20 * i, 30, 40), (20 * i + 25, 256, 230)) for i in range(7)] + [((0, 0, 0), (256, 90, 256))]


@pytest.mark.parametrize("hw,k,rgb,nr", [((700, 2052), 15, False, 8), ((33, 97), 15, True, 8), ((1, 1), 15, False, 1), ((5, 40), 7, False, 3),
                                         ((17, 31), 9, True, 2), ((70, 36), 11, False, 5), ((49, 64), 13, False, 8), ((16, 32), 15, False, 4)])
def test_fused_hsv_gaussian_inrange_bit_planes(oracle, monkeypatch, hw, k, rgb, nr):
    """inRange(GaussianBlur_k(BGR2HSV(frame))) for 1..8 ranges through the fused kernel.  The morphology list is one erode
    with a 1x1 element (the identity), so the final mask IS the kernel's bit plane: compared with the oracle's inRange of
    its own blur, and with the three stage kernels (IMGPROC_GAUSS_FUSED=0 is read once per process, so the stage path is
    taken by force_unfused).  Sizes: many 32-pixel strips and 16-row blocks (2052x700, 8 ranges), single pixel, frames
    narrower than the kernel's reach, widths that end inside a word."""
    h, w = hw
    f = np.random.default_rng(1000 * h + w + k).integers(0, 256, (h, w, 3), dtype=np.uint8)
    f[h // 4: h - h // 4, w // 5: w - w // 5] //= 2  # a darker block: V bounds cut through it after the blur
    ranges = FUSED_RANGES8[8 - nr:]
    ident = [ip.E(1)]
    with ip.ImgProc(w, h, n_slots=1) as g:
        g.set_pipeline(encoding=L.ENC_RGB8 if rgb else L.ENC_BGR8, post_blur_k=k, morph=ident)
        g.set_ranges(ranges)
        assert "bit-plane" in g.plan
        res = g.process(f)
        got = [g.fetch_stage(0, L.STAGE_MASK, r) for r in range(nr)]
        g.set_pipeline(encoding=L.ENC_RGB8 if rgb else L.ENC_BGR8, post_blur_k=k, morph=ident, force_unfused=True)
        g.process(f)
        stage = [g.fetch_stage(0, L.STAGE_MASK, r) for r in range(nr)]
    bgr = f[:, :, ::-1] if rgb else f
    blur = oracle.gaussian_blur(oracle.bgr2hsv(np.ascontiguousarray(bgr)), k)
    for r in range(nr):
        exp = oracle.in_range(blur, ranges[r][0], ranges[r][1])
        assert (got[r] == exp).all(), (hw, k, r, int((got[r] != exp).sum()))
        assert (stage[r] == exp).all()
        assert res[r]["m00"] == int((exp != 0).sum())


def test_fused_hsv_gaussian_inrange_batch(oracle):
    """The same kernel with several frames in one launch (device batch, padded row pitch, frame stride with a gap)."""
    h, w, n, k = 83, 130, 3, 15
    pitch, fstride = w * 3 + 8, h * (w * 3 + 8) + 64
    rng = np.random.default_rng(5)
    buf = rng.integers(0, 256, (n, fstride), dtype=np.uint8)
    frames = np.stack([buf[i, : h * pitch].reshape(h, pitch)[:, : w * 3].reshape(h, w, 3) for i in range(n)])
    ranges = FUSED_RANGES8
    with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
        g.set_pipeline(post_blur_k=k, morph=[ip.E(1)])
        g.set_ranges(ranges)
        d_in, d_out = g.dev_alloc(buf.nbytes), g.dev_alloc(n * 8 * w * h)
        try:
            g.dev_upload(d_in, buf)
            res = g.process_batch_device(d_in, fstride, pitch, w, h, n, d_out)
            masks = g.dev_download(d_out, (n, 8, h, w))
        finally:
            g.dev_free(d_in)
            g.dev_free(d_out)
    for i in range(n):
        blur = oracle.gaussian_blur(oracle.bgr2hsv(np.ascontiguousarray(frames[i])), k)
        for r in range(8):
            exp = oracle.in_range(blur, ranges[r][0], ranges[r][1])
            assert (masks[i, r] == exp).all(), (i, r)
            assert res[i][r]["m00"] == int((exp != 0).sum())
