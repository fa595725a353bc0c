"""imgproc_process_batch_device_noalloc (the entry point that allocates nothing: its scratch is a per-GPU static device
array) against imgproc_process_batch_device, through the C ABI, two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch

pytestmark = pytest.mark.gpu

KEYS = ("m00", "m10", "m01", "m20", "m11", "m02", "cx", "cy", "theta", "found")
RANGES = [DEFAULT_RANGE, ((0, 0, 0), (90, 256, 256))]
PRESETS = {
    "baseline": dict(post_blur_k=5, morph=ip.MORPH_BASELINE),                   # chain kernel: moment sums in the scratch
    "polygon_path": dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH),  # front + bit planes: inRange planes too
}


def noalloc(g, d_in, w, h, n, d_out):
    res = (L.Result * max(n, 1))()
    rc = g.lib.imgproc_process_batch_device_noalloc(g._ctx, d_in, w * h * 3, w * 3, w, h, n, d_out, res)
    return rc, res


def as_lists(res, n, nr):
    return [[[getattr(res[i].r[r], k) for k in KEYS] for r in range(nr)] for i in range(n)]


@pytest.mark.parametrize("preset", list(PRESETS))
def test_noalloc_matches_process_batch_device(preset):
    with ip.ImgProc(640, 480, n_slots=1, max_batch=8) as g:
        g.set_pipeline(second_order=True, **PRESETS[preset])
        g.set_ranges(RANGES)
        assert not g.plan.startswith("stage kernels")
        # two calls in a row on the same GPU's scratch, the second with a smaller input (fewer, smaller frames)
        for (w, h, n, seed) in ((640, 480, 6, 5), (250, 131, 3, 6), (252, 131, 2, 7)):
            frames = synth_batch(w, h, n, seed=seed)
            d_in = g.dev_alloc(frames.nbytes)
            d_ref, d_out = g.dev_alloc(n * 2 * w * h), g.dev_alloc(n * 2 * w * h)
            try:
                g.dev_upload(d_in, frames)
                ref = (L.Result * n)()
                g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_ref, raw=ref)
                rc, got = noalloc(g, d_in, w, h, n, d_out)
                assert rc == 0, g.lib.imgproc_last_error(g._ctx)
                assert as_lists(got, n, 2) == as_lists(ref, n, 2)
                m_ref, m_got = g.dev_download(d_ref, (n, 2, h, w)), g.dev_download(d_out, (n, 2, h, w))
                assert (m_ref == m_got).all()
                assert sum(int(ref[i].r[0].m00) for i in range(n)) > 0
            finally:
                for p in (d_in, d_ref, d_out):
                    g.dev_free(p)


def test_noalloc_limits():
    with ip.ImgProc(1920, 1200, n_slots=1, max_batch=16) as g:
        w, h, n = 1920, 1200, 16
        d_in = g.dev_alloc(n * w * h * 3)
        try:
            g.set_pipeline(**PRESETS["polygon_path"])
            g.set_ranges([DEFAULT_RANGE] * 8)  # 128 inRange bit planes of 1920x1200 (288 000 B each) > 32 MiB of scratch
            rc, _ = noalloc(g, d_in, w, h, n, None)
            assert rc == L.ESIZE and b"scratch" in g.lib.imgproc_last_error(g._ctx)
            g.set_ranges([DEFAULT_RANGE])
            rc, _ = noalloc(g, d_in, w - 2, h, n, None)  # w % 4 != 0: byte planes, 16 x 2.3 MB > 32 MiB
            assert rc == L.ESIZE and b"scratch" in g.lib.imgproc_last_error(g._ctx)
            rc, res = noalloc(g, d_in, w, h, n, None)  # 16 bit planes fit
            assert rc == 0 and res[15].n_ranges == 1
            rc, _ = noalloc(g, d_in, w, h, 17, None)
            assert rc == L.ESIZE
            g.set_pipeline(post_blur_k=15, morph=ip.MORPH_BASELINE)  # stage-by-stage pipeline
            rc, _ = noalloc(g, d_in, w, h, 1, None)
            assert rc == L.EUNSUPPORTED
            g.set_pipeline(**PRESETS["baseline"])
            rc, res = noalloc(g, d_in, w, h, n, None)  # fits: only the moment sums live in the scratch
            assert rc == 0 and res[0].n_ranges == 1
        finally:
            g.dev_free(d_in)
