"""Single-frame latency (the live node: one frame per callback): submit+wait wall time from a host frame, and the chain
kernel alone for a one-frame device batch.  usage: python scripts/latency_bench.py [preset]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_frame

P = {"pose_estimate": dict(), "baseline": dict(post_blur_k=5, morph=ip.MORPH_BASELINE),
     "polygon_path": dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH)}
w, h = 1920, 1200
frame = synth_frame(w, h, 3)
with ip.ImgProc(w, h, n_slots=2, max_batch=1) as g:
    d_in = g.dev_alloc(frame.nbytes)
    d_out = g.dev_alloc(w * h)
    g.dev_upload(d_in, frame)
    raw = (L.Result * 1)()
    for name in (sys.argv[1:] or list(P)):
        g.set_pipeline(**P[name])
        g.set_ranges([DEFAULT_RANGE])
        g.set_profiling(True)
        ts, ks = [], []
        for i in range(60):
            t0 = time.perf_counter()
            g.process(frame)
            ts.append(time.perf_counter() - t0)
            g.process_batch_device(d_in, w * h * 3, w * 3, w, h, 1, d_out, raw=raw)
            ks.append(g.last_kernel_ms()[0])
        print(f"{name:16s} submit+wait median {np.median(ts[10:]) * 1e3:.3f} ms   kernels of one frame {np.median(ks[10:]) * 1e3:.1f} us  [{g.plan}]")
