"""dev tool: the headline chain kernel with its memory traffic taken away step by step (what is left is its arithmetic):
  a) as benchmarked: 256 distinct frames, masks written
  b) no mask buffer (moments only): no stores
  c) frame stride 0: every "frame" is the same 6.9 MB (cache hits), masks written
  d) both: neither HBM loads nor stores
usage: python scripts/valu_only_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import synth_batch
import bench

cfg = bench.config_table()["c2"]
w, h, n = cfg["w"], cfg["h"], cfg["batch"]
frames = synth_batch(w, h, n, unique=cfg["uniq"], hue_bands=cfg["hue_bands"])
with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
    d_in = g.dev_alloc(frames.nbytes); d_out = g.dev_alloc(n * w * h)
    g.dev_upload(d_in, frames)
    raw = (L.Result * n)()
    g.set_pipeline(**cfg["pipe"]); g.set_ranges(cfg["ranges"]); g.set_profiling(True)
    for name, fstride, out in (("a) distinct frames, masks", frames[0].nbytes, d_out), ("b) distinct frames, no masks", frames[0].nbytes, None),
                               ("c) one frame 256 times, masks", 0, d_out), ("d) one frame 256 times, no masks", 0, None)):
        for _ in range(60):
            g.process_batch_device(d_in, fstride, frames.strides[1], w, h, n, out, raw=raw)
        ks = []
        for _ in range(40):
            g.process_batch_device(d_in, fstride, frames.strides[1], w, h, n, out, raw=raw)
            ks.append(g.last_kernel_ms()[0])
        print(f"{name:34s} kernel {np.mean(ks):.4f} ms (min {np.min(ks):.4f}) per {n} frames  m00={raw[0].r[0].m00}")
