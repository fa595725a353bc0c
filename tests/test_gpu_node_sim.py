"""GPU: the ROS-free node simulator (host/node_sim.cpp + node_shim.hpp) drives the C ABI the way the two nodes'
imageCb would, and must publish what the reference's host logic would publish given the oracle's pixels."""
import json
import os
import subprocess

import numpy as np
import pytest

from image_processing_amd.synth import synth_frame

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "image_processing_amd", "lib", "node_sim")


def run(args):
    out = subprocess.run([SIM] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_pose_estimate_callback(oracle, tmp_path):
    w, h = 1920, 1200
    f = synth_frame(w, h, 7)
    (tmp_path / "f.raw").write_bytes(f.tobytes())
    # find the marker with the oracle to place a plausible tracker ROI around it
    hsv = oracle.bgr2hsv(f)
    full = oracle.in_range(hsv, (100, 120, 80), (130, 255, 255))
    ys, xs = np.nonzero(full)
    x0, x1, y0, y1 = max(int(xs.min()) - 20, 8), min(int(xs.max()) + 20, w - 9), max(int(ys.min()) - 20, 8), min(int(ys.max()) + 20, h - 9)
    roi = (x0, y0, x1 - x0, y1 - y0)
    cam = (77, 33)  # camera ROI offset: roiCb rounds x to a multiple of 32, y to a multiple of 2 (pose_estimate.cpp:301-312)
    # dynamic_reconfigure hands (H_MIN.. H_MAX..); the node swaps H (pose_estimate.cpp:348-349), so to threshold H in
    # [100,130] an operator has to enter them swapped
    cfgv = (130, 120, 80, 100, 255, 255)
    o = run(["pose", tmp_path / "f.raw", w, h, *roi, *cam, *cfgv, tmp_path / "m.raw"])
    assert o["published"] and (o["H_MIN"], o["H_MAX"]) == (100, 130)
    rx, ry, rw, rh = int(roi[0] - 7.5), int(roi[1] - 7.5), int(roi[2] + 15), int(roi[3] + 15)
    assert (o["mask_w"], o["mask_h"]) == (rw, rh)
    crop = np.ascontiguousarray(f[ry:ry + rh, rx:rx + rw])
    exp_mask = oracle.in_range(oracle.bgr2hsv(crop), (100, 120, 80), (130, 255, 255))
    got_mask = np.frombuffer((tmp_path / "m.raw").read_bytes(), np.uint8).reshape(rh, rw)
    assert (got_mask == exp_mask).all()
    m = oracle.moments(exp_mask)
    assert o["m00"] == int(m[0]) > 0
    cx, cy = int(int(m[1]) / int(m[0])) + rx, int(int(m[2]) / int(m[0])) + ry
    assert (o["center_x"], o["center_y"]) == (cx, cy)
    assert (o["offsetX"], o["offsetY"]) == (64, 32)  # 77 % 32 = 13 <= 16 -> 64; 33 % 2 = 1 <= 1 -> 32
    assert o["position_x"] == (-960.0 + (64 + cx)) / 1 * 0.07758621 * 1e-3
    assert o["position_y"] == (-600.0 + (32 + cy)) / 1 * 0.07758621 * 1e-3


def test_pose_estimate_default_cfg_gives_empty_mask(tmp_path):
    """cfg defaults + the node's H swap => lo_H = 256 > hi_H = 0 => never matches."""
    w, h = 640, 480
    (tmp_path / "f.raw").write_bytes(synth_frame(w, h, 1).tobytes())
    o = run(["pose", tmp_path / "f.raw", w, h, 100, 100, 200, 150, 0, 0, 0, 0, 0, 256, 256, 256, tmp_path / "m.raw"])
    assert o["m00"] == 0 and (o["H_MIN"], o["H_MAX"]) == (256, 0)
    assert not np.frombuffer((tmp_path / "m.raw").read_bytes(), np.uint8).any()


def test_polygon_path_callback(oracle, tmp_path):
    w, h = 640, 480
    f = synth_frame(w, h, 9)
    (tmp_path / "f.raw").write_bytes(f.tobytes())
    rng = ((100, 120, 80), (130, 255, 255))
    o = run(["polygon", tmp_path / "f.raw", w, h, *rng[0], *rng[1], tmp_path / "m.raw", tmp_path / "p.raw"])
    seq = [0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0]  # objectPolygonPath.cpp:416-432
    morph = [(k, 0, 8 if k else 3, 8 if k else 3) for k in seq]
    exp = oracle.chain(f, pre_blur_k=3, post_blur_k=5, morph=morph, ranges=[rng])
    got = np.frombuffer((tmp_path / "m.raw").read_bytes(), np.uint8).reshape(h, w)
    assert (got == exp["mask"][0]).all()
    proc = np.frombuffer((tmp_path / "p.raw").read_bytes(), np.uint8).reshape(h, w, 3)
    assert (proc == oracle.mask_and(exp["bgr"], exp["mask"][0])).all()  # frame was blurred in place at :182
    assert [o["m00"], o["m10"], o["m01"]] == [int(v) for v in exp["moments"][0][:3]]
    assert o["m00"] > 0
