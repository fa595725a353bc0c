"""dev tool: device batches of a bench configuration with and without mask output (how much of the bit-plane kernel is its
byte stores).  usage: python scripts/nomask_bench.py [pp pe c5 ...]   (names of bench.py's config table)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import synth_batch
import bench

table = bench.config_table()
for name in (sys.argv[1:] or ["pp", "pe"]):
    cfg = table[name]
    w, h, n, nr = cfg["w"], cfg["h"], cfg["batch"], len(cfg["ranges"])
    frames = synth_batch(w, h, n, unique=cfg["uniq"], hue_bands=cfg["hue_bands"]) if "hue_bands" in cfg else synth_batch(w, h, n, unique=cfg["uniq"])
    with ip.ImgProc(w, h, n_slots=1, max_batch=n) as g:
        d_in = g.dev_alloc(frames.nbytes); d_out = g.dev_alloc(n * nr * w * h)
        g.dev_upload(d_in, frames)
        raw = (L.Result * n)()
        g.set_pipeline(**cfg["pipe"]); g.set_ranges(cfg["ranges"]); g.set_profiling(True)
        for out in (d_out, None):
            for _ in range(3):
                g.process_batch_device(d_in, frames[0].nbytes, frames.strides[1], w, h, n, out, raw=raw)
            ks = []
            for _ in range(10):
                g.process_batch_device(d_in, frames[0].nbytes, frames.strides[1], w, h, n, out, raw=raw)
                ks.append(g.last_kernel_ms()[0])
            print(f"{name:4s} masks={'yes' if out else 'no '} kernels {np.mean(ks):.3f} ms per {n} frames m00={raw[0].r[0].m00}")
