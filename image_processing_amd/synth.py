"""Seeded synthetic camera frames: splitmix64 noise background (V < 96),
1-3 filled ellipses of saturated in-range colour covering a few percent of the frame and
0.1 % salt-and-pepper of in-range colour (so the morphology passes have work to do).
Host-side test/bench data only; the generator is the same for the tests, the oracle leg and
the GPU leg, so all three see byte-identical frames."""
from __future__ import annotations

import numpy as np

SEED = 0x1A6E5EED
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (counter form: state_i = seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _hsv180_to_bgr(h: float, s: float, v: float):
    """Float HSV (OpenCV ranges: H 0..180, S,V 0..255) -> integer BGR; only used to pick paint."""
    hh = (h * 2.0) / 60.0
    c = (v / 255.0) * (s / 255.0)
    x = c * (1 - abs(hh % 2 - 1))
    m = v / 255.0 - c
    r, g, b = [(c, x, 0), (x, c, 0), (0, c, x), (0, x, c), (x, 0, c), (c, 0, x)][int(hh) % 6]
    return [int(round((t + m) * 255)) for t in (b, g, r)]


def synth_frame(w: int, h: int, index: int = 0, *, seed: int = SEED, hue_bands=((100, 130),),
                rgb: bool = False) -> np.ndarray:
    """One BGR8 (or RGB8) frame [h, w, 3]."""
    s = (seed + index) & 0xFFFFFFFFFFFFFFFF
    n = w * h * 3
    raw = splitmix64(s, (n + 7) // 8).view(np.uint8)[:n]
    img = ((raw.astype(np.uint16) * 96) >> 8).astype(np.uint8).reshape(h, w, 3)  # 0..95
    par = splitmix64(s ^ 0xABCDEF, 64)
    yy, xx = np.mgrid[0:h, 0:w]
    pi = 0
    for band in hue_bands:
        n_obj = 1 + int(par[pi] % 3) if len(hue_bands) == 1 else 1
        pi += 1
        for _ in range(n_obj):
            cx = int(par[pi] % max(w, 1)); cy = int(par[pi + 1] % max(h, 1))
            frac = 0.06 + 0.06 * (int(par[pi + 2] % 1000) / 1000.0)
            rx = max(2, int(w * frac)); ry = max(2, int(h * frac * (0.6 + (int(par[pi + 3] % 100) / 125.0))))
            hue = band[0] + 2 + int(par[pi + 4] % max(band[1] - band[0] - 4, 1))
            bgr = _hsv180_to_bgr(hue, 200 + int(par[pi + 5] % 50), 190 + int(par[pi + 6] % 60))
            pi += 7
            inside = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
            jitter = (raw[: w * h].reshape(h, w) & 7).astype(np.int16) - 3
            for c in range(3):
                ch = img[..., c]
                ch[inside] = np.clip(bgr[c] + jitter[inside], 0, 255).astype(np.uint8)
    # salt & pepper of in-range colour: 0.1 % of pixels
    k = max(1, (w * h) // 1000)
    pos = splitmix64(s ^ 0x5A17, k) % np.uint64(w * h)
    band = hue_bands[0]
    paint = _hsv180_to_bgr((band[0] + band[1]) / 2.0, 230, 230)
    flat = img.reshape(-1, 3)
    flat[pos.astype(np.int64)] = np.array(paint, np.uint8)
    if rgb:
        img = img[..., ::-1].copy()
    return img


def synth_bayer_rggb(w: int, h: int, index: int = 0, **kw) -> np.ndarray:
    """RGGB mosaic [h, w] of the same scene (sampled from the BGR frame)."""
    bgr = synth_frame(w, h, index, **kw)
    raw = np.empty((h, w), np.uint8)
    raw[0::2, 0::2] = bgr[0::2, 0::2, 2]  # R
    raw[0::2, 1::2] = bgr[0::2, 1::2, 1]  # G
    raw[1::2, 0::2] = bgr[1::2, 0::2, 1]  # G
    raw[1::2, 1::2] = bgr[1::2, 1::2, 0]  # B
    return raw


def synth_batch(w: int, h: int, n: int, *, unique: int | None = None, **kw) -> np.ndarray:
    """[n, h, w, 3]; frames beyond `unique` cycle (rolled by one row per cycle so no two are equal)."""
    unique = n if unique is None else min(unique, n)
    base = [synth_frame(w, h, i, **kw) for i in range(unique)]
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        out[i] = np.roll(base[i % unique], i // unique, axis=0)
    return out


DEFAULT_RANGE = ((100, 120, 80), (130, 255, 255))  # "blue marker" 
