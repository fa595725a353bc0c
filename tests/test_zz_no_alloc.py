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


def test_noalloc_against_the_oracle_after_a_larger_call(oracle):
    """Straight against the oracle (not only against the sibling entry point, which runs the same kernels): a call with
    many frames x ranges first, then a smaller one -- whatever the first left in the scratch (moment sums, planes) must
    not show in the second."""
    from test_gpu_chain import morph_tuples

    for name, kw in PRESETS.items():
        with ip.ImgProc(320, 200, n_slots=1, max_batch=8) as g:
            g.set_pipeline(second_order=True, **kw)
            g.set_ranges(RANGES)
            big, small = synth_batch(320, 200, 8, seed=11), synth_batch(316, 120, 2, seed=12)
            for frames in (big, small):
                n, h, w = frames.shape[:3]
                d_in, d_out = g.dev_alloc(frames.nbytes), g.dev_alloc(n * 2 * w * h)
                try:
                    g.dev_upload(d_in, frames)
                    rc, got = noalloc(g, d_in, w, h, n, d_out)
                    assert rc == 0, g.lib.imgproc_last_error(g._ctx)
                    masks = g.dev_download(d_out, (n, 2, h, w))
                finally:
                    g.dev_free(d_in)
                    g.dev_free(d_out)
            for i in range(2):  # the second, smaller call
                exp = oracle.chain(small[i], pre_blur_k=kw.get("pre_blur_k", 0), post_blur_k=5, morph=morph_tuples(kw["morph"]),
                                   ranges=RANGES, stages=False)
                for r in range(2):
                    assert (masks[i, r] == exp["mask"][r]).all(), (name, i, r)
                    assert [getattr(got[i].r[r], k) for k in KEYS[:6]] == [int(v) for v in exp["moments"][r]], (name, i, r)


def test_noalloc_leaves_device_memory_alone():
    """The property the entry point exists for: no hipMalloc / hipFree inside the call.  Free device memory
    (hipMemGetInfo of the HIP runtime the library itself is linked to) is the same before and after the FIRST call of a fresh context, for the chain kernel
    and for the two-launch path whose sibling entry point grows a plane buffer on first use.  (The kernels' code objects
    are loaded by an earlier context, so that the runtime's own one-time module upload is not counted.)"""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # already in the process (libimgproc_hip.so links it): the same runtime instance

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    w, h, n = 640, 480, 8
    frames = synth_batch(w, h, n, seed=21)
    for kw in PRESETS.values():
        with ip.ImgProc(w, h, n_slots=1, max_batch=n) as warm:  # loads the code objects, fills the occupancy cache
            warm.set_pipeline(**kw)
            warm.set_ranges(RANGES)
            d = warm.dev_alloc(frames.nbytes)
            warm.dev_upload(d, frames)
            warm.process_batch_device(d, w * h * 3, w * 3, w, h, n, None, want_results=False)
            warm.dev_free(d)
        with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
            g.set_pipeline(**kw)
            g.set_ranges(RANGES)
            d_in, d_out = g.dev_alloc(frames.nbytes), g.dev_alloc(n * 2 * w * h)
            try:
                g.dev_upload(d_in, frames)
                g.synchronize()
                free0 = free_bytes()
                rc, _ = noalloc(g, d_in, w, h, n, d_out)  # cold: this context's first batch call
                assert rc == 0, g.lib.imgproc_last_error(g._ctx)
                free1 = free_bytes()
                assert free1 == free0, f"device memory changed by {free0 - free1} bytes inside the no-allocation call"
            finally:
                g.dev_free(d_in)
                g.dev_free(d_out)


def test_noalloc_two_contexts_two_threads():
    """Two contexts on GPU 0 driven from two threads share the GPU's scratch array.  A call that finds the array in use
    returns IMGPROC_ESTATE and touches nothing; every call that returns IMGPROC_OK has the right sums and masks."""
    import threading

    w, h, n = 640, 480, 8
    batches = [synth_batch(w, h, n, seed=31), synth_batch(w, h, n, seed=32)]
    stats = [dict(ok=0, busy=0, bad=[]), dict(ok=0, busy=0, bad=[])]
    start = threading.Barrier(2)

    def worker(t):
        with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
            g.set_pipeline(**PRESETS["polygon_path"])
            g.set_ranges(RANGES)
            d_in, d_ref, d_out = g.dev_alloc(batches[t].nbytes), g.dev_alloc(n * 2 * w * h), g.dev_alloc(n * 2 * w * h)
            try:
                g.dev_upload(d_in, batches[t])
                ref = (L.Result * n)()
                g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_ref, raw=ref)
                want, m_ref = as_lists(ref, n, 2), g.dev_download(d_ref, (n, 2, h, w))
                start.wait()
                for it in range(150):
                    rc, got = noalloc(g, d_in, w, h, n, d_out)
                    if rc == L.ESTATE:
                        stats[t]["busy"] += 1
                        continue
                    if rc != 0 or as_lists(got, n, 2) != want:
                        stats[t]["bad"].append((it, rc))
                    stats[t]["ok"] += 1
                if not (g.dev_download(d_out, (n, 2, h, w)) == m_ref).all():
                    stats[t]["bad"].append(("masks", 0))
            finally:
                for p in (d_in, d_ref, d_out):
                    g.dev_free(p)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not stats[0]["bad"] and not stats[1]["bad"], stats
    assert stats[0]["ok"] > 0 and stats[1]["ok"] > 0, stats
