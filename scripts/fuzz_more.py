"""dev tool: the randomised differential test of tests/test_gpu_random_pipelines.py on seeds beyond the suite's 160.
usage: python scripts/fuzz_more.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_processing_amd as ip
from image_processing_amd import _lib as L
from oracle import Oracle
import test_gpu_random_pipelines as T

first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 160, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
oracle = Oracle()
bad = 0
with ip.ImgProc(1024, 256, n_slots=1, max_batch=1) as gpu:
    for seed in range(first, first + n):
        c = T.random_case(seed)
        exp_masks, exp_moms = T.oracle_run(oracle, c)
        for force in (False, True):
            gpu.set_pipeline(encoding=c["enc"], pre_blur_k=c["pre"], post_blur_k=c["post"], morph=c["ops"],
                             second_order=c["second"], force_unfused=force)
            gpu.set_ranges(c["ranges"])
            res = gpu.process(c["frame"])
            for r in range(len(c["ranges"])):
                k = 6 if c["second"] else 3
                ok = (gpu.fetch_stage(0, L.STAGE_MASK, r) == exp_masks[r]).all() and \
                    [res[r][q] for q in T.KEYS[:k]] == [int(v) for v in exp_moms[r][:k]]
                if not ok:
                    bad += 1
                    print("MISMATCH seed", seed, "force", force, "range", r, gpu.plan, c["frame"].shape, c["pre"], c["post"], flush=True)
        # the device-batch path (grouped bit-plane tail, front kernel + bit planes, chain kernel) on three copies of the
        # frame, the middle one flipped: frames 0 and 2 must give the single-frame result just checked
        import numpy as np
        f = c["frame"]
        frames = np.ascontiguousarray(np.stack([f, f[::-1], f]))
        h, w = f.shape[:2]
        nr = len(c["ranges"])
        with ip.ImgProc(w, h, n_slots=1, max_batch=3) as gb:
            gb.set_pipeline(encoding=c["enc"], pre_blur_k=c["pre"], post_blur_k=c["post"], morph=c["ops"], second_order=c["second"])
            gb.set_ranges(c["ranges"])
            d_in, d_out = gb.dev_alloc(frames.nbytes), gb.dev_alloc(3 * nr * w * h)
            try:
                gb.dev_upload(d_in, frames)
                rb = gb.process_batch_device(d_in, frames[0].nbytes, frames.strides[1], w, h, 3, d_out)
                mb = gb.dev_download(d_out, (3, nr, h, w))
                # the same batch on tickets 1 and 2 at once (streams of their own, planes of their own)
                d_t = [gb.dev_alloc(3 * nr * w * h) for _ in range(2)]
                try:
                    for t in (1, 2):
                        gb.batch_enqueue(t, d_in, frames[0].nbytes, frames.strides[1], w, h, 3, d_t[t - 1])
                    for t in (2, 1):
                        if gb.batch_wait(t, n=3) != rb or not (gb.dev_download(d_t[t - 1], (3, nr, h, w)) == mb).all():
                            bad += 1
                            print("TICKET MISMATCH seed", seed, "ticket", t, gb.plan, f.shape, c["pre"], c["post"], flush=True)
                finally:
                    for q in d_t:
                        gb.dev_free(q)
            finally:
                gb.dev_free(d_in); gb.dev_free(d_out)
        for r in range(nr):
            k = 6 if c["second"] else 3
            for i in (0, 2):
                ok = (mb[i, r] == exp_masks[r]).all() and [rb[i][r][q] for q in T.KEYS[:k]] == [int(v) for v in exp_moms[r][:k]]
                if not ok:
                    bad += 1
                    print("BATCH MISMATCH seed", seed, "frame", i, "range", r, gb.plan, f.shape, c["pre"], c["post"], flush=True)
        if seed % 100 == 0:
            print("seed", seed, "mismatches so far", bad, flush=True)
print("done:", n, "seeds,", bad, "mismatches")
sys.exit(1 if bad else 0)
