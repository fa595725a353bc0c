"""BASELINE config 5 operators at 7680x4320 on one GPU: 15x15 Gaussian + 9x9 elliptical erode/dilate + 8 ranges."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import synth_frame

w, h, n = 7680, 4320, 4
bands = [(20 * i, 20 * i + 19) for i in range(8)]
f = synth_frame(w, h, 3, hue_bands=bands)
frames = np.stack([f] * n)
ranges = [((lo, 100, 80), (hi, 255, 255)) for lo, hi in bands]
with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
    g.set_pipeline(post_blur_k=15, morph=[ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)])
    g.set_ranges(ranges)
    d_in, d_out = g.dev_alloc(frames.nbytes), g.dev_alloc(n * 8 * w * h)
    g.dev_upload(d_in, frames)
    raw = (L.Result * n)()
    g.set_profiling(True)
    g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out, raw=raw)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out, raw=raw)
    dt = (time.perf_counter() - t0) / reps
    ms, nl = g.last_kernel_ms()
    print(f"stress 7680x4320: {n / dt:.1f} frames/s, {ms / n:.2f} ms per frame in {nl // n} launches, algorithmic {w * h * 11 * n / dt / 1e9:.0f} GB/s [{g.plan}]")
