"""Host-side mirror of the C ABI: one ImgProc per GPU / camera namespace (it stands where the
node's objectTracker stands, pose_estimate.cpp:153-167), plus cv-style single-stage calls."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


class ImgProcError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {L.load().imgproc_strerror(status).decode()} ({status}) {detail}".strip())


@dataclass
class MorphOp:
    """One erode()/dilate() call of morphOps (objectPolygonPath.cpp:416-432)."""
    kind: int  # L.ERODE / L.DILATE
    shape: int = L.RECT
    kw: int = 3
    kh: int = 3
    anchor: tuple = (-1, -1)
    custom: np.ndarray | None = None


def E(k, shape=L.RECT):
    return MorphOp(L.ERODE, shape, k, k)


def D(k, shape=L.RECT):
    return MorphOp(L.DILATE, shape, k, k)


# the reference's two morphOps sequences and the BASELINE preset
MORPH_POLYGON_PATH = [E(3)] * 4 + [D(8)] * 2 + [E(3)] * 2 + [D(8)] * 2 + [E(3)] * 4  # objectPolygonPath.cpp:416-432
MORPH_POSE_ESTIMATE = [E(3)] * 2 + [D(8)] * 2 + [E(3)] * 2  # pose_estimate.cpp:407-414
MORPH_BASELINE = [E(3), D(3)]


def _array_over(address: int, nbytes: int, shape, dtype):
    """(array, owner): a numpy array of `shape` over raw memory, and the object every view of it refers to as its base
    (numpy collapses base chains to that owner, so its reference count tells whether views are alive)."""
    buf = (C.c_uint8 * nbytes).from_address(address)
    owner = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape)))
    return owner.reshape(shape), owner


def _owner_refs(ent) -> int:
    """Reference count of a pinned array's owner object (ent = [address, array, owner, count with no view alive]): every
    live view of the array adds one.  Always read through this one function, so the references of the call itself cancel."""
    import sys

    return sys.getrefcount(ent[2])


def _moments_to_dict(m: L.Moments) -> dict:
    return {k: getattr(m, k) for k in ("m00", "m10", "m01", "m20", "m11", "m02", "cx", "cy", "theta", "found")}


class ImgProc:
    def __init__(self, max_width: int, max_height: int, *, device: int = 0, n_slots: int = 2, max_batch: int = 1):
        self.lib = L.load()
        self._ctx = C.c_void_p()
        cfg = L.Config(device, max_width, max_height, n_slots, max_batch)
        rc = self.lib.imgproc_create(C.byref(self._ctx), C.byref(cfg))
        if rc:
            self._ctx = C.c_void_p()
            raise ImgProcError(rc, "imgproc_create")
        self.n_slots = n_slots
        self.n_ranges = 1
        self._slot_dims = {}
        self._keep = []

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            # page-locked arrays still out (host_alloc without host_free) go with the context: after close() they dangle
            for ent in list(getattr(self, "_pinned", {}).values()):
                self.lib.imgproc_host_free(self._ctx, ent[0])
            self._pinned = {}
            self.lib.imgproc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc: int, where: str):
        if rc:
            raise ImgProcError(rc, where, self.lib.imgproc_last_error(self._ctx).decode())

    # -- configuration -----------------------------------------------------------------------
    def set_pipeline(self, *, encoding=L.ENC_BGR8, pre_blur_k=0, post_blur_k=0, morph=(), keep_stages=False,
                     force_unfused=False, second_order=False):
        p = L.Pipeline()
        p.encoding, p.pre_blur_k, p.post_blur_k = encoding, pre_blur_k, post_blur_k
        p.n_morph = len(morph)
        if len(morph) > L.MAX_MORPH:
            raise ImgProcError(L.EINVAL, "set_pipeline", "too many morph ops")
        self._keep = []
        for i, m in enumerate(morph):
            pm = p.morph[i]
            pm.kind, pm.shape, pm.kw, pm.kh = m.kind, m.shape, m.kw, m.kh
            pm.anchor_x, pm.anchor_y = m.anchor
            if m.custom is not None:
                arr = np.ascontiguousarray(m.custom, dtype=np.uint8)
                self._keep.append(arr)
                pm.custom = arr.ctypes.data_as(L.u8p)
        p.keep_stages, p.force_unfused, p.second_order = int(keep_stages), int(force_unfused), int(second_order)
        self._chk(self.lib.imgproc_set_pipeline(self._ctx, C.byref(p)), "imgproc_set_pipeline")
        self.encoding = encoding

    def set_ranges(self, ranges):
        """ranges: [((Hmin,Smin,Vmin),(Hmax,Smax,Vmax)), ...] raw reconfigure ints (0..256)."""
        n = len(ranges)
        lo = (L.I3 * max(n, 1))()
        hi = (L.I3 * max(n, 1))()
        for i, (a, b) in enumerate(ranges):
            for c in range(3):
                lo[i][c], hi[i][c] = int(a[c]), int(b[c])
        self._chk(self.lib.imgproc_set_ranges(self._ctx, n, lo, hi), "imgproc_set_ranges")
        self.n_ranges = n

    @property
    def is_fused(self) -> bool:
        return bool(self.lib.imgproc_pipeline_is_fused(self._ctx))

    @property
    def plan(self) -> str:
        buf = C.create_string_buffer(256)
        self._chk(self.lib.imgproc_pipeline_plan(self._ctx, buf, 256), "imgproc_pipeline_plan")
        return buf.value.decode()

    # -- streaming ---------------------------------------------------------------------------
    def submit(self, slot: int, frame: np.ndarray):
        if frame.dtype != np.uint8 or frame.strides[-1] != 1 or (frame.ndim == 3 and frame.strides[1] != 3):
            frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        self._chk(self.lib.imgproc_submit(self._ctx, slot, frame.ctypes.data, frame.strides[0], w, h), "imgproc_submit")
        self._slot_dims[slot] = (h, w)

    def submit_pinned(self, slot: int, frame: np.ndarray):
        """frame: a uint8 view of page-locked memory (host_alloc); it must stay untouched until wait(slot)."""
        if frame.dtype != np.uint8 or frame.strides[-1] != 1 or (frame.ndim == 3 and frame.strides[1] != 3):
            raise ImgProcError(L.EINVAL, "submit_pinned", "frame must be a uint8 array with packed pixels")
        h, w = frame.shape[:2]
        self._chk(self.lib.imgproc_submit_pinned(self._ctx, slot, frame.ctypes.data, frame.strides[0], w, h),
                  "imgproc_submit_pinned")
        self._slot_dims[slot] = (h, w)

    def host_alloc(self, shape, dtype=np.uint8) -> np.ndarray:
        """A numpy array over page-locked host memory; release with host_free(arr).  The array does NOT own the memory:
        after host_free(arr) (or close()) the array and every view / slice of it point at freed pages, and touching
        one is a use-after-free (a segmentation fault, as gpurun_out/pinned.log of round 2 shows).  Copy what is
        needed (`arr[i].copy()`) and drop the views before host_free; host_free refuses while views are alive."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self.lib.imgproc_host_alloc(self._ctx, max(n, 1), C.byref(p)), "imgproc_host_alloc")
        ent = [p.value, *_array_over(p.value, max(n, 1), shape, dtype), 0]
        ent[3] = _owner_refs(ent)  # with no view alive: host_free compares against it through the same call
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[id(ent[1])] = ent
        return ent[1]

    def host_free(self, arr: np.ndarray, force: bool = False):
        """Releases a host_alloc array.  Every view of it (`arr[i]`, slices, reshapes) keeps a reference to the array's
        owner object; while any is alive the call raises ImgProcError(ESTATE) instead of leaving it dangling
        (force=True frees anyway).  Only the array host_alloc returned is accepted, not a view of it (EINVAL)."""
        pinned = getattr(self, "_pinned", {})
        ent = pinned.get(id(arr))
        if ent is None or ent[1] is not arr:
            raise ImgProcError(L.EINVAL, "host_free", "not an array handed out by host_alloc (a view of one?)")
        n_views = _owner_refs(ent) - ent[3]
        if not force and n_views:
            raise ImgProcError(L.ESTATE, "host_free", f"{n_views} view(s) of the array are alive (slices, reshapes, containers "
                               "holding them): drop them first, or pass force=True")
        del pinned[id(arr)]
        self._chk(self.lib.imgproc_host_free(self._ctx, ent[0]), "imgproc_host_free")

    def wait(self, slot: int):
        r = L.Result()
        self._chk(self.lib.imgproc_wait(self._ctx, slot, C.byref(r)), "imgproc_wait")
        return [_moments_to_dict(r.r[i]) for i in range(r.n_ranges)]

    def fetch_stage(self, slot: int, stage: int, rng: int = 0) -> np.ndarray:
        h, w = self._slot_dims[slot]
        out = np.empty((h, w) if stage in (L.STAGE_THRESH, L.STAGE_MASK) else (h, w, 3), np.uint8)
        self._chk(self.lib.imgproc_fetch_stage(self._ctx, slot, stage, rng, out.ctypes.data, out.strides[0]),
                  "imgproc_fetch_stage")
        return out

    def process(self, frame: np.ndarray, slot: int = 0):
        self.submit(slot, frame)
        return self.wait(slot)

    # -- batches -----------------------------------------------------------------------------
    def process_batch(self, frames: np.ndarray):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, h, w = frames.shape[:3]
        res = (L.Result * n)()
        self._chk(self.lib.imgproc_process_batch(self._ctx, frames.ctypes.data, frames.strides[0], frames.strides[1],
                                                 w, h, n, res), "imgproc_process_batch")
        return [[_moments_to_dict(res[i].r[j]) for j in range(res[i].n_ranges)] for i in range(n)]

    def process_batch_device(self, d_frames: int, frame_stride: int, row_stride: int, w: int, h: int, n: int,
                             d_masks: int | None = None, want_results: bool = True, raw=None):
        """raw: a preallocated (L.Result * n)() array to fill and return as is (no per-frame Python objects)."""
        res = raw if raw is not None else ((L.Result * n)() if want_results else None)
        self._chk(self.lib.imgproc_process_batch_device(self._ctx, d_frames, frame_stride, row_stride, w, h, n,
                                                        d_masks, res), "imgproc_process_batch_device")
        if res is None or raw is not None:
            return res
        return [[_moments_to_dict(res[i].r[j]) for j in range(res[i].n_ranges)] for i in range(n)]

    def batch_enqueue(self, ticket: int, d_frames: int, frame_stride: int, row_stride: int, w: int, h: int, n: int,
                      d_masks: int | None = None):
        """Puts a device batch on its ticket's stream and returns; tickets 0 .. L.MAX_TICKETS - 1 may be in flight, their
        d_masks must not overlap."""
        self._chk(self.lib.imgproc_batch_enqueue(self._ctx, d_frames, frame_stride, row_stride, w, h, n, d_masks, ticket),
                  "imgproc_batch_enqueue")

    def batch_wait(self, ticket: int, raw=None, n: int | None = None):
        """Blocks until the ticket's batch is done.  raw: (L.Result * n)() to fill and return; else n is needed to
        build the list of per-frame, per-range dicts; neither: results are dropped."""
        res = raw if raw is not None else ((L.Result * n)() if n else None)
        self._chk(self.lib.imgproc_batch_wait(self._ctx, ticket, res), "imgproc_batch_wait")
        if res is None or raw is not None:
            return res
        return [[_moments_to_dict(res[i].r[j]) for j in range(res[i].n_ranges)] for i in range(n)]

    # -- device memory -------------------------------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self.lib.imgproc_dev_alloc(self._ctx, nbytes, C.byref(p)), "imgproc_dev_alloc")
        return p.value

    def dev_free(self, ptr: int):
        self._chk(self.lib.imgproc_dev_free(self._ctx, ptr), "imgproc_dev_free")

    def dev_upload(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.imgproc_dev_upload(self._ctx, ptr, arr.ctypes.data, arr.nbytes), "imgproc_dev_upload")

    def dev_download(self, ptr: int, shape, dtype=np.uint8) -> np.ndarray:
        out = np.empty(shape, dtype)
        self._chk(self.lib.imgproc_dev_download(self._ctx, out.ctypes.data, ptr, out.nbytes), "imgproc_dev_download")
        return out

    # -- measurement ---------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._chk(self.lib.imgproc_set_profiling(self._ctx, int(on)), "imgproc_set_profiling")

    def last_kernel_ms(self):
        ms, n = C.c_float(), C.c_int()
        self._chk(self.lib.imgproc_last_kernel_ms(self._ctx, C.byref(ms), C.byref(n)), "imgproc_last_kernel_ms")
        return ms.value, n.value

    def last_kernel_span(self):
        """(start, end) of the last waited batch's kernels in ms on the device clock since set_profiling(True)."""
        t0, t1 = C.c_float(), C.c_float()
        self._chk(self.lib.imgproc_last_kernel_span(self._ctx, C.byref(t0), C.byref(t1)), "imgproc_last_kernel_span")
        return t0.value, t1.value

    def synchronize(self):
        self._chk(self.lib.imgproc_synchronize(self._ctx), "imgproc_synchronize")

    # -- cv-style single stages (each is one cv:: call of the reference) -------------------------
    def cvtColor_BGR2HSV(self, img: np.ndarray, rgb: bool = False) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        out = np.empty_like(img)
        self._chk(self.lib.imgproc_cvt_bgr2hsv(self._ctx, img.ctypes.data, img.strides[0], out.ctypes.data,
                                               out.strides[0], w, h, L.ENC_RGB8 if rgb else L.ENC_BGR8), "imgproc_cvt_bgr2hsv")
        return out

    def GaussianBlur(self, img: np.ndarray, ksize: int) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty_like(img)
        self._chk(self.lib.imgproc_gaussian_blur(self._ctx, img.ctypes.data, img.strides[0], out.ctypes.data,
                                                 out.strides[0], w, h, cn, ksize), "imgproc_gaussian_blur")
        return out

    def inRange(self, img: np.ndarray, lo, hi) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        out = np.empty((h, w), np.uint8)
        self._chk(self.lib.imgproc_in_range(self._ctx, img.ctypes.data, img.strides[0], out.ctypes.data, out.strides[0],
                                            w, h, L.I3(*[int(v) for v in lo]), L.I3(*[int(v) for v in hi])), "imgproc_in_range")
        return out

    def morphology(self, mask: np.ndarray, op: MorphOp) -> np.ndarray:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        out = np.empty_like(mask)
        m = L.Morph(op.kind, op.shape, op.kw, op.kh, op.anchor[0], op.anchor[1], None)
        keep = None
        if op.custom is not None:
            keep = np.ascontiguousarray(op.custom, dtype=np.uint8)
            m.custom = keep.ctypes.data_as(L.u8p)
        self._chk(self.lib.imgproc_morphology(self._ctx, mask.ctypes.data, mask.strides[0], out.ctypes.data,
                                              out.strides[0], w, h, C.byref(m)), "imgproc_morphology")
        return out

    def erode(self, mask, k=3, shape=L.RECT):
        return self.morphology(mask, MorphOp(L.ERODE, shape, k, k))

    def dilate(self, mask, k=3, shape=L.RECT):
        return self.morphology(mask, MorphOp(L.DILATE, shape, k, k))

    def moments(self, mask: np.ndarray) -> dict:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        m = L.Moments()
        self._chk(self.lib.imgproc_mask_moments(self._ctx, mask.ctypes.data, mask.strides[0], w, h, C.byref(m)),
                  "imgproc_mask_moments")
        return _moments_to_dict(m)

    def demosaic_rggb(self, raw: np.ndarray) -> np.ndarray:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        h, w = raw.shape
        out = np.empty((h, w, 3), np.uint8)
        self._chk(self.lib.imgproc_demosaic_rggb(self._ctx, raw.ctypes.data, raw.strides[0], out.ctypes.data,
                                                 out.strides[0], w, h), "imgproc_demosaic_rggb")
        return out

    def mask_and(self, bgr: np.ndarray, mask: np.ndarray) -> np.ndarray:
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        out = np.empty_like(bgr)
        self._chk(self.lib.imgproc_mask_and(self._ctx, bgr.ctypes.data, bgr.strides[0], mask.ctypes.data, mask.strides[0],
                                            out.ctypes.data, out.strides[0], w, h), "imgproc_mask_and")
        return out


def getStructuringElement(shape: int, kw: int, kh: int) -> np.ndarray:
    out = np.empty((kh, kw), np.uint8)
    rc = L.load().imgproc_structuring_element(shape, kw, kh, out.ctypes.data)
    if rc:
        raise ImgProcError(rc, "imgproc_structuring_element")
    return out


def gaussian_taps(k: int):
    t = (C.c_int32 * 31)()
    rc = L.load().imgproc_gaussian_taps(k, t)
    if rc:
        raise ImgProcError(rc, "imgproc_gaussian_taps")
    return list(t[:k])
