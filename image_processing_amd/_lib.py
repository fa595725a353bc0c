"""ctypes binding of the C ABI in include/imgproc.h.  Fails loudly when the HIP library is
missing: there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libimgproc_hip.so")

MAX_RANGES, MAX_MORPH, MAX_KSIZE = 8, 32, 31
MAX_TICKETS = 4  # device batches in flight per context
OK, EINVAL, ENOMEM, EHIP, ESTATE, ESIZE, ENODEV, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7
ENC_BGR8, ENC_RGB8, ENC_BAYER_RGGB8 = 0, 1, 2
ERODE, DILATE = 0, 1
RECT, CROSS, ELLIPSE, CUSTOM = 0, 1, 2, 3
STAGE_BGR, STAGE_HSV, STAGE_BLUR, STAGE_THRESH, STAGE_MASK, STAGE_MASKED = range(6)

u8p = C.POINTER(C.c_uint8)


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("max_width", C.c_int), ("max_height", C.c_int), ("n_slots", C.c_int),
                ("max_batch", C.c_int)]


class Morph(C.Structure):
    _fields_ = [("kind", C.c_int), ("shape", C.c_int), ("kw", C.c_int), ("kh", C.c_int), ("anchor_x", C.c_int),
                ("anchor_y", C.c_int), ("custom", u8p)]


class Pipeline(C.Structure):
    _fields_ = [("encoding", C.c_int), ("pre_blur_k", C.c_int), ("post_blur_k", C.c_int), ("n_morph", C.c_int),
                ("morph", Morph * MAX_MORPH), ("keep_stages", C.c_int), ("force_unfused", C.c_int),
                ("second_order", C.c_int)]


class Moments(C.Structure):
    _fields_ = [("m00", C.c_uint64), ("m10", C.c_uint64), ("m01", C.c_uint64), ("m20", C.c_uint64),
                ("m11", C.c_uint64), ("m02", C.c_uint64), ("cx", C.c_double), ("cy", C.c_double),
                ("theta", C.c_double), ("found", C.c_int), ("reserved", C.c_int)]


class Result(C.Structure):
    _fields_ = [("n_ranges", C.c_int), ("reserved", C.c_int), ("r", Moments * MAX_RANGES)]


I3 = C.c_int32 * 3
_SIGS = {
    "imgproc_version": (C.c_int, []),
    "imgproc_strerror": (C.c_char_p, [C.c_int]),
    "imgproc_last_error": (C.c_char_p, [C.c_void_p]),
    "imgproc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "imgproc_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config)]),
    "imgproc_destroy": (None, [C.c_void_p]),
    "imgproc_set_pipeline": (C.c_int, [C.c_void_p, C.POINTER(Pipeline)]),
    "imgproc_set_ranges": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(I3), C.POINTER(I3)]),
    "imgproc_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "imgproc_submit_pinned": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "imgproc_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Result)]),
    "imgproc_fetch_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "imgproc_process_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(Result)]),
    "imgproc_process_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p, C.POINTER(Result)]),
    "imgproc_process_batch_device_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                                       C.c_int, C.c_void_p, C.POINTER(Result)]),
    "imgproc_batch_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int]),
    "imgproc_batch_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Result)]),
    "imgproc_cvt_bgr2hsv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                      C.c_int]),
    "imgproc_gaussian_blur": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                        C.c_int, C.c_int]),
    "imgproc_in_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "imgproc_morphology": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                     C.POINTER(Morph)]),
    "imgproc_mask_moments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Moments)]),
    "imgproc_demosaic_rggb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                        C.c_int]),
    "imgproc_mask_and": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_size_t, C.c_int, C.c_int]),
    "imgproc_structuring_element": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imgproc_gaussian_taps": (C.c_int, [C.c_int, C.POINTER(C.c_int32)]),
    "imgproc_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "imgproc_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "imgproc_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "imgproc_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "imgproc_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "imgproc_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "imgproc_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "imgproc_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "imgproc_last_kernel_span": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "imgproc_pipeline_is_fused": (C.c_int, [C.c_void_p]),
    "imgproc_pipeline_plan": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "imgproc_synchronize": (C.c_int, [C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m image_processing_amd.build` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
