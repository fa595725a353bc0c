"""Second, independent restatement of SURVEY.md Appendix A in numpy (vectorised, different
loop structure from oracle/cpu_ref.c: exact-rational table rounding, full 2-D Gaussian
kernel instead of two separable passes, shift-and-reduce morphology).  Its only job is to
catch transcription slips in the C oracle; like the oracle it is NOT an OpenCV output
(parity unpinned, see oracle/cpu_ref.c header)."""
from __future__ import annotations

from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np


def _round_half_even(fr: Fraction) -> int:
    fl = fr.numerator // fr.denominator
    rem = fr - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1):
        return fl + 1
    return fl


def hsv_tables():
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    for i in range(1, 256):
        sdiv[i] = _round_half_even(Fraction(255 << 12, i))
        hdiv[i] = _round_half_even(Fraction(180 << 12, 6 * i))
    return sdiv, hdiv


def bgr2hsv(img: np.ndarray, rgb: bool = False) -> np.ndarray:
    sdiv, hdiv = hsv_tables()
    a = img.astype(np.int64)
    b, g, r = (a[..., 2], a[..., 1], a[..., 0]) if rgb else (a[..., 0], a[..., 1], a[..., 2])
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + 2048) >> 12  # numpy >> on negative int64 is arithmetic
    h = np.where(h < 0, h + 180, h)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


def gaussian_taps(k: int):
    fixed = {1: [256], 3: [64, 128, 64], 5: [16, 64, 96, 64, 16], 7: [8, 28, 56, 72, 56, 28, 8]}
    if k in fixed:
        return fixed[k]
    getcontext().prec = 60
    sigma = Decimal(3) / Decimal(10) * ((Decimal(k) - 1) / 2 - 1) + Decimal(8) / Decimal(10)
    s2 = Decimal(-1) / (2 * sigma * sigma)
    g = [(s2 * (Decimal(i) - (Decimal(k) - 1) / 2) ** 2).exp() for i in range(k)]
    tot = sum(g)
    g = [x / tot for x in g]
    taps = [0] * k
    err = Decimal(0)
    acc = 0
    for i in range(k // 2):
        adj = g[i] * 256 + err
        v0 = int(adj.to_integral_value(rounding="ROUND_HALF_EVEN"))
        err = adj - v0
        taps[i] = taps[k - 1 - i] = v0
        acc += v0
    taps[k // 2] = 256 - 2 * acc
    return taps


def _reflect101_index(n: int, r: int) -> np.ndarray:
    idx = np.arange(-r, n + r)
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def gaussian_blur(img: np.ndarray, k: int) -> np.ndarray:
    taps = np.array(gaussian_taps(k), np.int64)
    r = k // 2
    a = img.astype(np.int64)
    if a.ndim == 2:
        a = a[..., None]
    h, w, cn = a.shape
    p = a[_reflect101_index(h, r)][:, _reflect101_index(w, r)]
    acc = np.zeros((h, w, cn), np.int64)
    for j in range(k):
        for i in range(k):
            acc += taps[j] * taps[i] * p[j:j + h, i:i + w]
    out = ((acc + 32768) >> 16).astype(np.uint8)
    return out.reshape(img.shape)


def in_range(img: np.ndarray, lo, hi) -> np.ndarray:
    ok = np.ones(img.shape[:2], bool)
    for c in range(3):
        l, u = int(lo[c]), int(hi[c])
        if l > u or l > 255 or u < 0:
            return np.zeros(img.shape[:2], np.uint8)
        ok &= (img[..., c] >= max(l, 0)) & (img[..., c] <= min(u, 255))
    return np.where(ok, 255, 0).astype(np.uint8)


def structuring_element(shape: int, kw: int, kh: int) -> np.ndarray:
    e = np.zeros((kh, kw), np.uint8)
    if shape == 0:
        e[:] = 1
    elif shape == 1:
        e[kh // 2, :] = 1
        e[:, kw // 2] = 1
    else:
        r, c = kh // 2, kw // 2
        for i in range(kh):
            dy = i - r
            if abs(dy) <= r:
                val = c * float(np.sqrt((r * r - dy * dy) * (1.0 / (r * r)))) if r else 0.0
                dx = int(np.rint(val))
                e[i, max(c - dx, 0):min(c + dx + 1, kw)] = 1
    return e


def morph(mask: np.ndarray, op: int, elem: np.ndarray, anchor=(-1, -1)) -> np.ndarray:
    kh, kw = elem.shape
    ax = kw // 2 if anchor[0] < 0 else anchor[0]
    ay = kh // 2 if anchor[1] < 0 else anchor[1]
    h, w = mask.shape
    neutral = 0 if op else 255
    pad = np.full((h + kh, w + kw), neutral, np.uint8)
    pad[ay:ay + h, ax:ax + w] = mask
    out = np.full((h, w), neutral, np.uint8)
    red = np.maximum if op else np.minimum
    for j in range(kh):
        for i in range(kw):
            if elem[j, i]:
                out = red(out, pad[j:j + h, i:i + w])
    return out


def moments(mask: np.ndarray):
    ys, xs = np.nonzero(mask)
    xs = xs.astype(object)
    ys = ys.astype(object)
    return [len(xs), int(xs.sum()) if len(xs) else 0, int(ys.sum()) if len(ys) else 0,
            int((xs * xs).sum()) if len(xs) else 0, int((xs * ys).sum()) if len(xs) else 0,
            int((ys * ys).sum()) if len(xs) else 0]


def bayer_rggb2bgr(raw: np.ndarray) -> np.ndarray:
    h, w = raw.shape
    out = np.zeros((h, w, 3), np.uint8)
    if h <= 2 or w <= 2:
        return out
    p = raw.astype(np.int64)
    c = p[1:-1, 1:-1]
    cross = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] + 2) >> 2
    diag = (p[:-2, :-2] + p[:-2, 2:] + p[2:, :-2] + p[2:, 2:] + 2) >> 2
    hor = (p[1:-1, :-2] + p[1:-1, 2:] + 1) >> 1
    ver = (p[:-2, 1:-1] + p[2:, 1:-1] + 1) >> 1
    yy, xx = np.mgrid[1:h - 1, 1:w - 1]
    ey, ex = yy & 1, xx & 1
    r_site = (ey == 0) & (ex == 0)
    b_site = (ey == 1) & (ex == 1)
    g_rrow = (ey == 0) & (ex == 1)
    R = np.where(r_site, c, np.where(b_site, diag, np.where(g_rrow, hor, ver)))
    G = np.where(r_site | b_site, cross, c)
    B = np.where(b_site, c, np.where(r_site, diag, np.where(g_rrow, ver, hor)))
    out[1:-1, 1:-1] = np.stack([B, G, R], -1).astype(np.uint8)
    out[1:-1, 0] = out[1:-1, 1]
    out[1:-1, -1] = out[1:-1, -2]
    out[0] = out[1]
    out[-1] = out[-2]
    return out
