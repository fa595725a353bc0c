"""CPU-only: the C-ABI library loads, exports every symbol include/imgproc.h declares, and its
host-only entry points (no device needed) agree with the oracle.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import image_processing_amd as ip
from image_processing_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "imgproc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(imgproc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    lib = L.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(L.EXPORTS) == names  # the ctypes binding covers the whole header


def test_version_and_strerror():
    lib = L.load()
    assert lib.imgproc_version() == 0x000100
    assert lib.imgproc_strerror(0) == b"ok"
    for s in range(-7, 0):
        assert lib.imgproc_strerror(s) not in (b"ok", b"unknown status")
    assert lib.imgproc_strerror(-99) == b"unknown status"
    assert lib.imgproc_last_error(None) == b"null context"


def test_struct_layout_matches_header():
    # sizes the C side computes for the same structs (LP64): guards the ctypes mirror
    assert C.sizeof(L.Config) == 20
    assert C.sizeof(L.Morph) == 32
    assert C.sizeof(L.Pipeline) == 16 + 32 * 32 + 16
    assert C.sizeof(L.Moments) == 6 * 8 + 3 * 8 + 8
    assert C.sizeof(L.Result) == 8 + 8 * C.sizeof(L.Moments)


def test_host_only_entry_points_match_oracle(oracle):
    for shape in (0, 1, 2):
        for kw, kh in ((3, 3), (8, 8), (9, 9), (5, 7), (1, 1), (15, 15), (2, 3), (31, 31)):
            assert (ip.getStructuringElement(shape, kw, kh) == oracle.structuring_element(shape, kw, kh)).all()
    for k in (1, 3, 5, 7, 9, 15, 31):
        assert ip.gaussian_taps(k) == oracle.gaussian_taps(k)
    with pytest.raises(ip.ImgProcError):
        ip.gaussian_taps(4)
    with pytest.raises(ip.ImgProcError):
        ip.getStructuringElement(7, 3, 3)


def test_bad_create_arguments():
    lib = L.load()
    ctx = C.c_void_p()
    assert lib.imgproc_create(None, None) == L.EINVAL
    bad = L.Config(0, 0, 10, 1, 1)
    assert lib.imgproc_create(C.byref(ctx), C.byref(bad)) == L.EINVAL
    assert not ctx.value


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(ip.ImgProcError) as e:
        ip.ImgProc(64, 64)
    assert e.value.status in (L.ENODEV, L.EHIP)


def test_pinned_array_view_accounting():
    """ImgProc.host_free refuses to free a page-locked block while views of it are alive: the bookkeeping behind that
    (views refer to the array's owner object, whatever they were sliced from), on plain host memory."""
    import ctypes as C

    from image_processing_amd.pipeline import _array_over, _owner_refs

    raw = (C.c_uint8 * 600)()

    def alloc():
        ent = [C.addressof(raw), *_array_over(C.addressof(raw), 600, (2, 10, 10, 3), np.uint8), 0]
        ent[3] = _owner_refs(ent)
        return ent

    ent = alloc()
    arr = ent[1]
    assert arr.shape == (2, 10, 10, 3) and arr.ctypes.data == C.addressof(raw)
    live = lambda: _owner_refs(ent) - ent[3]  # noqa: E731
    assert live() == 0
    v = arr[0, :, :5]  # starts at the array's own address: must still count as a view, not as the array
    assert v.ctypes.data == arr.ctypes.data and live() == 1
    more = [arr[1], arr.reshape(-1), v[2:]]
    assert live() == 4
    del v, more
    assert live() == 0
