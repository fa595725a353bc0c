"""The two hand-derivable vectors of tests/golden/bayer_ed_kat.json (generator beside it): a 6x6 RGGB mosaic demosaiced
the way OpenCV's Bayer2RGB_ loop walks it, and the 15x15 sigma=0 Gaussian taps with the error diffusion written out.
CPU part: oracle and numpy restatement against the fixture.  GPU part (-m gpu): the demosaic stage kernel, the Bayer
front end inside the chain kernel and the library's tap derivation against the same fixture.
Still not an OpenCV output: parity stays unpinned."""
import json
import os
from decimal import Decimal

import numpy as np
import pytest

import np_ref

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "bayer_ed_kat.json")) as f:
        return json.load(f)


def cases(fx):
    b = fx["bayer_rggb_6x6"]
    yield "6x6", np.array(b["mosaic"], np.uint8), np.array(b["bgr"], np.uint8)
    for k in ("h2_all_zero", "h3", "w4"):
        yield k, np.array(b[k]["mosaic"], np.uint8), np.array(b[k]["bgr"], np.uint8)


def test_fixture_is_what_its_generator_makes(fx):
    import importlib.util

    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_bayer_ed_kat.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert mk.bayer_bg2bgr(mk.M) == fx["bayer_rggb_6x6"]["bgr"]
    assert mk.ed_taps(15)[2] == fx["gauss15_error_diffusion"]["taps"]


def test_worked_pixels_and_border_copies(fx):
    b = fx["bayer_rggb_6x6"]
    bgr = np.array(b["bgr"])
    for wk in b["worked"]:
        assert bgr[wk["y"], wk["x"]].tolist() == wk["bgr"], wk["why"]
    assert (bgr[:, 0] == bgr[:, 1]).all() and (bgr[:, 5] == bgr[:, 4]).all()
    assert (bgr[0] == bgr[1]).all() and (bgr[5] == bgr[4]).all()
    assert not np.array(b["h2_all_zero"]["bgr"]).any()


def test_oracle_and_numpy_demosaic(oracle, fx):
    for name, raw, exp in cases(fx):
        assert (oracle.bayer_rggb2bgr(raw) == exp).all(), name
        assert (np_ref.bayer_rggb2bgr(raw) == exp).all(), name


def test_error_diffusion_steps(oracle, fx):
    ed = fx["gauss15_error_diffusion"]
    assert ed["sigma"] == "2.60"
    err, taps = Decimal(0), []
    for st in ed["steps"]:
        adj = Decimal(st["g"]) * 256 + err
        assert abs(adj - Decimal(st["adj"])) < Decimal("1e-9")
        v = int(adj.to_integral_value(rounding="ROUND_HALF_EVEN"))
        assert v == st["v"] and abs(adj - v) < Decimal("0.45")  # no tap sits near a rounding tie
        err = adj - v
        taps.append(v)
    full = taps + [256 - 2 * sum(taps)] + taps[::-1]
    assert full == ed["taps"] == [1, 3, 6, 12, 20, 30, 36, 40, 36, 30, 20, 12, 6, 3, 1]
    assert round(float(ed["centre_adj"])) == full[7]  # the centre's own diffusion step lands on the same integer
    assert oracle.gaussian_taps(15) == full and np_ref.gaussian_taps(15) == full


@pytest.mark.gpu
def test_gpu_demosaic_and_taps(fx):
    import image_processing_amd as ip
    from image_processing_amd import _lib as L

    assert ip.gaussian_taps(15) == fx["gauss15_error_diffusion"]["taps"]
    with ip.ImgProc(64, 64, n_slots=1) as g:
        for name, raw, exp in cases(fx):
            assert (g.demosaic_rggb(raw) == exp).all(), name
        # the Bayer front end inside the chain kernel: a 48x40 mosaic whose top-left 6x6 corner is the fixture's (the
        # interior of the corner block and its left / top border copies do not depend on what lies beyond column / row 5)
        raw = np.random.default_rng(3).integers(0, 256, (40, 48), dtype=np.uint8)
        raw[:6, :6] = np.array(fx["bayer_rggb_6x6"]["mosaic"], np.uint8)
        exp = np.array(fx["bayer_rggb_6x6"]["bgr"], np.uint8)
        g.set_pipeline(encoding=L.ENC_BAYER_RGGB8)
        g.set_ranges([((0, 0, 0), (256, 256, 256))])
        assert "Bayer front end inside" in g.plan
        g.process(raw)
        assert (g.fetch_stage(0, L.STAGE_BGR)[:5, :5] == exp[:5, :5]).all()  # rebuilt by the stage kernels
        # the chain kernel itself only hands out masks: one range per corner pixel, exactly that pixel's expected HSV
        # (cvtColor of the fixture's BGR), eight ranges per launch; the pixel's own mask bit is set only if the kernel's
        # demosaic + HSV of that pixel equals the expectation
        from oracle import Oracle

        hsv = Oracle().bgr2hsv(exp)
        pix = [(y, x) for y in range(5) for x in range(5)]
        for i in range(0, len(pix), 8):
            grp = pix[i:i + 8]
            g.set_ranges([(tuple(int(v) for v in hsv[y, x]), tuple(int(v) for v in hsv[y, x])) for y, x in grp])
            assert g.is_fused
            g.process(raw)
            for r, (y, x) in enumerate(grp):
                m = g.fetch_stage(0, L.STAGE_MASK, r)
                assert m[y, x] == 255, (y, x)
                assert (m[:5, :5] == np.where((hsv[:5, :5] == hsv[y, x]).all(-1), 255, 0)).all(), (y, x)
