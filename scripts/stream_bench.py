"""Streamed mode (SURVEY §8d mode ii): host frames -> pinned staging -> H2D -> chain -> 48 B of moments back, through the
slots (imgproc_process_batch).  PCIe-bound; reported next to the device-resident number, never as bench.py's value.
usage: python scripts/stream_bench.py [n_frames] [n_slots] [w h] [bayer]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch, synth_bayer_rggb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1200
bayer = len(sys.argv) > 5 and sys.argv[5] == "bayer"
polygon = len(sys.argv) > 5 and sys.argv[5] == "polygon"
if bayer:
    base = np.stack([synth_bayer_rggb(w, h, i) for i in range(4)])
else:
    base = synth_batch(w, h, 4)
frames = np.ascontiguousarray(np.concatenate([base] * (n // 4)))
with ip.ImgProc(w, h, n_slots=slots) as g:
    if polygon:
        g.set_pipeline(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH)
    else:
        g.set_pipeline(encoding=L.ENC_BAYER_RGGB8 if bayer else L.ENC_BGR8, post_blur_k=5, morph=ip.MORPH_BASELINE)
    g.set_ranges([DEFAULT_RANGE])
    res = (L.Result * len(frames))()
    import ctypes as C

    def run():
        rc = g.lib.imgproc_process_batch(g._ctx, frames.ctypes.data, frames.strides[0], frames.strides[1], w, h,
                                         len(frames), res)
        assert rc == 0, g.lib.imgproc_last_error(g._ctx)

    run()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    fb = frames[0].nbytes
    if os.environ.get("STREAM_PINNED", "1") == "1":
        # the same stream from page-locked memory through imgproc_submit_pinned: no staging memcpy on the critical path
        pin = g.host_alloc(frames.shape)
        pin[:] = frames
        res2 = (L.Result * len(frames))()

        def run_pinned():
            ns = slots
            for i in range(len(frames) + ns):
                if i >= ns:
                    rc = g.lib.imgproc_wait(g._ctx, (i - ns) % ns, C.byref(res2[i - ns]))
                    assert rc == 0, g.lib.imgproc_last_error(g._ctx)
                if i < len(frames):
                    rc = g.lib.imgproc_submit_pinned(g._ctx, i % ns, pin[i].ctypes.data, pin[i].strides[0], w, h)
                    assert rc == 0, g.lib.imgproc_last_error(g._ctx)

        run_pinned()
        t0 = time.perf_counter()
        for _ in range(reps):
            run_pinned()
        dtp = (time.perf_counter() - t0) / reps
        assert all(res2[i].r[0].m00 == res[i].r[0].m00 for i in range(len(frames)))
        print(f"streamed {'bayer' if bayer else 'bgr8'} {w}x{h} from page-locked frames (imgproc_submit_pinned): {len(frames) / dtp:.0f} frames/s, "
              f"{len(frames) * fb / dtp / 1e9:.1f} GB/s over PCIe, slots={slots}")
        g.host_free(pin)
    print(f"streamed {'bayer' if bayer else 'bgr8'} {w}x{h}: {len(frames) / dt:.0f} frames/s, {len(frames) * fb / dt / 1e9:.1f} GB/s over PCIe "
          f"(incl. host memcpy into pinned staging), slots={slots}, graphs={os.environ.get('IMGPROC_GRAPH', 'auto')}, preset={'polygon_path' if polygon else 'baseline'}, "
          f"m00[0]={res[0].r[0].m00}")
