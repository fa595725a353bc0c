import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_processing_amd as ip
from image_processing_amd import _lib as L
from image_processing_amd.synth import DEFAULT_RANGE, synth_frame
from oracle import Oracle
o = Oracle()
P = {"pose": dict(post_blur_k=0, morph=[]), "c1": dict(post_blur_k=0, morph=ip.MORPH_BASELINE),
     "blur": dict(post_blur_k=5, morph=[]), "base": dict(post_blur_k=5, morph=ip.MORPH_BASELINE),
     "e": dict(post_blur_k=0, morph=[ip.E(3)]), "d": dict(post_blur_k=0, morph=[ip.D(3)])}
with ip.ImgProc(1920, 1200) as g:
    for hw in [(64, 96), (123, 516), (480, 640)]:
        f = synth_frame(hw[1], hw[0], 11)
        for name, p in P.items():
            g.set_pipeline(**p); g.set_ranges([DEFAULT_RANGE])
            res = g.process(f)
            m = g.fetch_stage(0, L.STAGE_MASK)
            mt = [(x.kind, x.shape, x.kw, x.kh, -1, -1) for x in p["morph"]]
            e = o.chain(f, post_blur_k=p["post_blur_k"], morph=mt, ranges=[DEFAULT_RANGE], stages=False)
            d = m != e["mask"][0]
            ys, xs = np.nonzero(d)
            print(hw, name, "fused", g.is_fused, "ndiff", d.sum(), "of set", (e["mask"][0] > 0).sum(),
                  "mom", [res[0][k] for k in ("m00", "m10", "m01")], e["moments"][0][:3].tolist())
            if d.sum():
                print("   rows", np.unique(ys)[:20], "cols", np.unique(xs)[:40])
