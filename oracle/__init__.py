"""ctypes front-end of the CPU oracle (oracle/cpu_ref.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from image_processing_amd/ (the product).  PARITY UNPINNED: see the
header of cpu_ref.c (OpenCV is absent here and the reference holds no fixtures).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MORPH_ERODE, MORPH_DILATE = 0, 1
SHAPE_RECT, SHAPE_CROSS, SHAPE_ELLIPSE = 0, 1, 2
ENC_BGR8, ENC_RGB8, ENC_BAYER_RGGB8 = 0, 1, 2


def build(force: bool = False) -> None:
    """Compile liborc.so / liborc_fast.so with gcc (building the checker is not using it)."""
    if force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liborc.so", "liborc_fast.so")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


class _ChainCfg(C.Structure):
    _fields_ = [
        ("encoding", C.c_int),
        ("pre_blur_k", C.c_int),
        ("post_blur_k", C.c_int),
        ("n_morph", C.c_int),
        ("morph", (C.c_int * 6) * 32),
        ("n_ranges", C.c_int),
        ("lo", (C.c_int32 * 3) * 8),
        ("hi", (C.c_int32 * 3) * 8),
    ]


_u8p = C.POINTER(C.c_uint8)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


class Oracle:
    def __init__(self, fast: bool = False):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "liborc_fast.so" if fast else "liborc.so"))
        L = self.lib
        L.orc_bgr2hsv.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.orc_bgr2hsv.restype = None
        L.orc_gaussian_taps.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.orc_gaussian_blur.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_in_range.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_in_range.restype = None
        L.orc_structuring_element.argtypes = [C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_morph.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_moments.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_moments.restype = None
        L.orc_bayer_rggb2bgr.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int]
        L.orc_bayer_rggb2bgr.restype = None
        L.orc_mask_and.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int]
        L.orc_mask_and.restype = None
        L.orc_hsv_tables.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_hsv_tables.restype = None
        L.orc_chain.argtypes = [C.POINTER(_ChainCfg), _u8p, C.c_size_t, C.c_int, C.c_int,
                                _u8p, _u8p, _u8p, _u8p, _u8p, C.POINTER(C.c_uint64)]

    # ---- per-stage -------------------------------------------------------------------
    def hsv_tables(self):
        s = (C.c_int * 256)()
        h = (C.c_int * 256)()
        self.lib.orc_hsv_tables(s, h)
        return np.array(s[:], dtype=np.int64), np.array(h[:], dtype=np.int64)

    def bgr2hsv(self, img: np.ndarray, rgb: bool = False) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        out = np.empty_like(img)
        self.lib.orc_bgr2hsv(_p(img), w * 3, _p(out), w * 3, w, h, 2 if rgb else 0)
        return out

    def gaussian_taps(self, k: int):
        t = (C.c_int * 31)()
        if self.lib.orc_gaussian_taps(k, t):
            raise ValueError(f"bad ksize {k}")
        return list(t[:k])

    def gaussian_blur(self, img: np.ndarray, k: int) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty_like(img)
        rc = self.lib.orc_gaussian_blur(_p(img), w * cn, _p(out), w * cn, w, h, cn, k)
        if rc:
            raise ValueError(f"orc_gaussian_blur rc={rc}")
        return out

    def in_range(self, img: np.ndarray, lo, hi) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        out = np.empty((h, w), np.uint8)
        self.lib.orc_in_range(_p(img), w * 3, _p(out), w, w, h, (C.c_int32 * 3)(*lo), (C.c_int32 * 3)(*hi))
        return out

    def structuring_element(self, shape: int, kw: int, kh: int) -> np.ndarray:
        e = np.empty((kh, kw), np.uint8)
        if self.lib.orc_structuring_element(shape, kw, kh, _p(e)):
            raise ValueError("bad structuring element")
        return e

    def morph(self, mask: np.ndarray, op: int, elem: np.ndarray, anchor=(-1, -1)) -> np.ndarray:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        elem = np.ascontiguousarray(elem, dtype=np.uint8)
        h, w = mask.shape
        out = np.empty_like(mask)
        rc = self.lib.orc_morph(_p(mask), w, _p(out), w, w, h, op, _p(elem), elem.shape[1], elem.shape[0],
                                anchor[0], anchor[1])
        if rc:
            raise ValueError(f"orc_morph rc={rc}")
        return out

    def erode(self, mask, elem, anchor=(-1, -1)):
        return self.morph(mask, MORPH_ERODE, elem, anchor)

    def dilate(self, mask, elem, anchor=(-1, -1)):
        return self.morph(mask, MORPH_DILATE, elem, anchor)

    def moments(self, mask: np.ndarray) -> np.ndarray:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        m = (C.c_uint64 * 6)()
        self.lib.orc_moments(_p(mask), w, w, h, m)
        return np.array(m[:], dtype=np.uint64)

    def bayer_rggb2bgr(self, raw: np.ndarray) -> np.ndarray:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        h, w = raw.shape
        out = np.empty((h, w, 3), np.uint8)
        self.lib.orc_bayer_rggb2bgr(_p(raw), w, _p(out), w * 3, w, h)
        return out

    def mask_and(self, bgr: np.ndarray, mask: np.ndarray) -> np.ndarray:
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        out = np.empty_like(bgr)
        self.lib.orc_mask_and(_p(bgr), w * 3, _p(mask), w, _p(out), w * 3, w, h)
        return out

    # ---- whole chain -----------------------------------------------------------------
    def chain(self, frame: np.ndarray, *, encoding=ENC_BGR8, pre_blur_k=0, post_blur_k=0, morph=(),
              ranges=(((0, 0, 0), (256, 256, 256)),), stages: bool = True):
        """morph: sequence of (op, shape, kw, kh[, ax, ay]).  Returns dict of stage outputs."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        cfg = _ChainCfg()
        cfg.encoding, cfg.pre_blur_k, cfg.post_blur_k = encoding, pre_blur_k, post_blur_k
        cfg.n_morph = len(morph)
        for i, m in enumerate(morph):
            m = tuple(m) + (-1, -1) if len(m) == 4 else tuple(m)
            for j in range(6):
                cfg.morph[i][j] = m[j]
        cfg.n_ranges = len(ranges)
        for i, (lo, hi) in enumerate(ranges):
            for c in range(3):
                cfg.lo[i][c], cfg.hi[i][c] = lo[c], hi[c]
        nr = len(ranges)
        out = {}
        if stages:
            out["bgr"] = np.empty((h, w, 3), np.uint8)
            out["hsv"] = np.empty((h, w, 3), np.uint8)
            out["blur"] = np.empty((h, w, 3), np.uint8)
            out["thresh"] = np.empty((nr, h, w), np.uint8)
        out["mask"] = np.empty((nr, h, w), np.uint8)
        mom = (C.c_uint64 * (6 * nr))()
        nul = C.cast(None, _u8p)
        stride = frame.strides[0]
        rc = self.lib.orc_chain(C.byref(cfg), _p(frame), stride, w, h,
                                _p(out["bgr"]) if stages else nul, _p(out["hsv"]) if stages else nul,
                                _p(out["blur"]) if stages else nul, _p(out["thresh"]) if stages else nul,
                                _p(out["mask"]), mom)
        if rc:
            raise ValueError(f"orc_chain rc={rc}")
        out["moments"] = np.array(mom[:], dtype=np.uint64).reshape(nr, 6)
        return out
