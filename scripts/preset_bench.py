"""Device-resident batch throughput of the reference's other chains (not the headline config of bench.py).
usage: python scripts/preset_bench.py [preset ...]   presets: baseline polygon_path pose_estimate pose_estimate_full c1 baseline_8r"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch

P = {"pose_estimate": dict(), "pose_estimate_full": dict(post_blur_k=5, morph=ip.MORPH_POSE_ESTIMATE),
     "baseline": dict(post_blur_k=5, morph=ip.MORPH_BASELINE), "c1": dict(morph=ip.MORPH_BASELINE),
     "polygon_path": dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH),
     "baseline_8r": dict(post_blur_k=5, morph=ip.MORPH_BASELINE)}  # eight hue bands in one launch (capacity-8 variant)
R8 = [((20 * i, 60, 60), (20 * i + 19, 256, 256)) for i in range(8)]
w, h, n = 1920, 1200, int(os.environ.get("PRESET_BATCH", "64"))
frames = synth_batch(w, h, n, unique=4)
with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
    d_in = g.dev_alloc(frames.nbytes)
    d_out = g.dev_alloc((8 if any(a.endswith("_8r") for a in (sys.argv[1:] or list(P))) else 1) * n * w * h)
    g.dev_upload(d_in, frames)
    raw = (L.Result * n)()
    for name in (sys.argv[1:] or list(P)):
        g.set_pipeline(**P[name])
        g.set_ranges(R8 if name.endswith("_8r") else [DEFAULT_RANGE])
        g.set_profiling(True)
        for _ in range(2):
            g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out, raw=raw)
        t0 = time.perf_counter()
        reps = 10
        ks = []
        for _ in range(reps):
            g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, d_out, raw=raw)
            ks.append(g.last_kernel_ms()[0])
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:20s} {n / dt:9.0f} frames/s  kernels {np.mean(ks):.3f} ms per {n} frames  [{g.plan}]  m00[0]={raw[0].r[0].m00}")
