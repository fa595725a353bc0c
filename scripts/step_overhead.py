"""dev tool: step time of the bench loop (two batches in flight) with and without the HIP profiling events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_batch

W, H, B = 1920, 1200, 256
base = synth_batch(W, H, 8)
frames = np.stack([np.roll(base[i % 8], i // 8, axis=0) for i in range(B)])
with ip.ImgProc(W, H, n_slots=1, max_batch=B) as g:
    d_in, d_out = g.dev_alloc(frames.nbytes), g.dev_alloc(B * W * H)
    g.dev_upload(d_in, frames)
    g.set_pipeline(post_blur_k=5, morph=ip.MORPH_BASELINE)
    g.set_ranges([DEFAULT_RANGE])
    raws = [(L.Result * B)(), (L.Result * B)()]
    for prof in (True, False, True, False):
        g.set_profiling(prof)
        def run(n):
            for i in range(n):
                g.batch_enqueue(i & 1, d_in, W * H * 3, W * 3, W, H, B, d_out)
                if i:
                    g.batch_wait((i - 1) & 1, raw=raws[(i - 1) & 1])
            g.batch_wait((n - 1) & 1, raw=raws[(n - 1) & 1])
        run(200)
        g.synchronize()
        t0 = time.perf_counter()
        run(200)
        g.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print(f"profiling events {'on ' if prof else 'off'}: {dt * 1e3:.4f} ms per step, {B / dt:.0f} frames/s")
