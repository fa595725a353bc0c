"""Device-resident batch throughput of BASELINE config 3 (2448x2048 BayerRG8, baseline chain).
usage: python scripts/bayer_bench.py [n_frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_bayer_rggb

w, h = 2448, 2048
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
base = [synth_bayer_rggb(w, h, i) for i in range(2)]
frames = np.stack([np.roll(base[i % 2], 2 * (i // 2), axis=0) for i in range(n)])  # even shifts keep the mosaic phase
with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
    d_in = g.dev_alloc(frames.nbytes)
    d_out = g.dev_alloc(n * w * h)
    g.dev_upload(d_in, frames)
    raw = (L.Result * n)()
    g.set_pipeline(encoding=L.ENC_BAYER_RGGB8, post_blur_k=5, morph=ip.MORPH_BASELINE)
    g.set_ranges([DEFAULT_RANGE])
    for _ in range(2):
        g.process_batch_device(d_in, w * h, w, w, h, n, d_out, raw=raw)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        g.process_batch_device(d_in, w * h, w, w, h, n, d_out, raw=raw)
    dt = (time.perf_counter() - t0) / reps
    print(f"bayer {w}x{h}: {n / dt:.0f} frames/s, {dt / n * 1e3:.3f} ms per frame, algorithmic {n / dt * w * h * 2 / 1e9:.0f} GB/s  [{g.plan}]  m00[0]={raw[0].r[0].m00}")
