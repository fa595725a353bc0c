"""dev tool: polygon_path / pose_estimate_full device batches with and without mask output (how much of the bit-plane
kernel is its byte stores).  usage: python scripts/nomask_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch

w, h, n = 1920, 1200, 256
frames = synth_batch(w, h, n, unique=4)
with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
    d_in = g.dev_alloc(frames.nbytes); d_out = g.dev_alloc(n * w * h)
    g.dev_upload(d_in, frames)
    raw = (L.Result * n)()
    for name, kw in (("polygon_path", dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH)),
                     ("pose_estimate_full", dict(post_blur_k=5, morph=ip.MORPH_POSE_ESTIMATE))):
        g.set_pipeline(**kw); g.set_ranges([DEFAULT_RANGE]); g.set_profiling(True)
        for out in (d_out, None):
            for _ in range(3):
                g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, out, raw=raw)
            ks = []
            for _ in range(10):
                g.process_batch_device(d_in, w * h * 3, w * 3, w, h, n, out, raw=raw)
                ks.append(g.last_kernel_ms()[0])
            print(f"{name:20s} masks={'yes' if out else 'no '} kernels {np.mean(ks):.3f} ms per {n} frames m00={raw[0].r[0].m00}")
