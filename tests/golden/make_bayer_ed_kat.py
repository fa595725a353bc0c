"""Generator of tests/golden/bayer_ed_kat.json (committed with its output).

Two hand-derivable vectors for the stages that had none:

1. A 6x6 RGGB mosaic through cv_bridge's bayer_rggb8 -> bgr8 (= cv::COLOR_BayerBG2BGR).  The expected image is
   computed here by walking the mosaic the way OpenCV's Bayer2RGB_ loop does (a pointer that advances
   two columns per step over rows 1..H-2, a `blue` sign and a start_with_green flag that flip per row, then the
   column copies 0 <- 1 and W-1 <- W-2, then the row copies 0 <- 1 and H-1 <- H-2; all-zero output when H <= 2) --
   a third formulation, structurally unlike oracle/cpu_ref.c (per-pixel site classification) and tests/np_ref.py
   (whole-array shifts).  Four pixels are also worked by hand in `worked`, with the sums written out.
2. The 15x15 sigma=0 Gaussian taps (sigma = 0.3*((15-1)*0.5 - 1) + 0.8 = 2.6) with the error diffusion written out per tap:
   g[i], adj = 256*g[i] + err, v = round-half-even(adj), err = adj - v; centre = 256 - 2*sum(v).  50-digit decimals.

NOT an OpenCV output (OpenCV is absent from this image): this pins the restatements to SURVEY Appendix A, it does
not make parity green."""
import json
import os
from decimal import Decimal, getcontext

getcontext().prec = 50

M = [[10, 20, 30, 41, 50, 60],
     [71, 80, 90, 101, 110, 120],
     [131, 140, 150, 160, 171, 180],
     [190, 201, 210, 220, 230, 241],
     [250, 3, 14, 25, 36, 47],
     [58, 69, 70, 81, 92, 103]]


def bayer_bg2bgr(m):
    """The loop structure of OpenCV's Bayer2RGB_ for COLOR_BayerBG2BGR on a flat byte buffer."""
    h, w = len(m), len(m[0])
    src = [v for row in m for v in row]
    step = w
    dst = [0] * (h * w * 3)
    dstep = w * 3
    if h <= 2:
        return [[[0, 0, 0] for _ in range(w)] for _ in range(h)]
    blue, start_with_green = -1, False
    width = w - 2
    for y in range(h - 2):
        b = y * step                      # bayer0
        d0 = (y + 1) * dstep + 3 + 1      # dst0: row y+1, pixel 1, channel 1 (G)
        d, end = d0, b + width
        if width <= 0:
            for i in (-4, -3, -2, width * 3 - 1, width * 3, width * 3 + 1):
                dst[d0 + i] = 0
            blue, start_with_green = -blue, not start_with_green
            continue
        if start_with_green:
            t0 = (src[b + 1] + src[b + 2 * step + 1] + 1) >> 1
            t1 = (src[b + step] + src[b + step + 2] + 1) >> 1
            dst[d - blue], dst[d], dst[d + blue] = t0, src[b + step + 1], t1
            b += 1
            d += 3
        while b <= end - 2:
            t0 = (src[b] + src[b + 2] + src[b + 2 * step] + src[b + 2 * step + 2] + 2) >> 2
            t1 = (src[b + 1] + src[b + step] + src[b + step + 2] + src[b + 2 * step + 1] + 2) >> 2
            dst[d - blue], dst[d], dst[d + blue] = t0, t1, src[b + step + 1]
            t0 = (src[b + 2] + src[b + 2 * step + 2] + 1) >> 1
            t1 = (src[b + step + 1] + src[b + step + 3] + 1) >> 1
            dst[d + 3 - blue], dst[d + 3], dst[d + 3 + blue] = t0, src[b + step + 2], t1
            b += 2
            d += 6
        if b < end:
            t0 = (src[b] + src[b + 2] + src[b + 2 * step] + src[b + 2 * step + 2] + 2) >> 2
            t1 = (src[b + 1] + src[b + step] + src[b + step + 2] + src[b + 2 * step + 1] + 2) >> 2
            dst[d - blue], dst[d], dst[d + blue] = t0, t1, src[b + step + 1]
        for c in (-4, -3, -2):            # column 0 <- column 1
            dst[d0 + c] = dst[d0 + c + 3]
        for c in (-1, 0, 1):              # column W-1 <- column W-2
            dst[d0 + width * 3 + c] = dst[d0 + width * 3 + c - 3]
        blue, start_with_green = -blue, not start_with_green
    for i in range(w * 3):                # row 0 <- row 1, row H-1 <- row H-2
        dst[i] = dst[i + dstep]
        dst[i + (h - 1) * dstep] = dst[i + (h - 2) * dstep]
    return [[dst[(y * w + x) * 3:(y * w + x) * 3 + 3] for x in range(w)] for y in range(h)]


def ed_taps(k):
    sigma = Decimal("0.3") * ((k - 1) * Decimal("0.5") - 1) + Decimal("0.8")
    g = [(-(Decimal(i) - (k - 1) / Decimal(2)) ** 2 / (2 * sigma * sigma)).exp() for i in range(k)]
    tot = sum(g)
    g = [x / tot for x in g]
    err, steps, taps = Decimal(0), [], [0] * k
    for i in range(k // 2):
        adj = g[i] * 256 + err
        v = int(adj.to_integral_value(rounding="ROUND_HALF_EVEN"))
        err = adj - v
        taps[i] = taps[k - 1 - i] = v
        steps.append({"i": i, "g": f"{g[i]:.15f}", "adj": f"{adj:.12f}", "v": v, "err": f"{err:.12f}"})
    taps[k // 2] = 256 - 2 * sum(taps[:k // 2])
    return str(sigma), steps, taps, f"{g[k // 2] * 256 + err:.12f}"


if __name__ == "__main__":
    bgr = bayer_bg2bgr(M)
    worked = [
        {"y": 1, "x": 1, "site": "B", "bgr": [80, (20 + 71 + 90 + 140 + 2) >> 2, (10 + 30 + 131 + 150 + 2) >> 2],
         "why": "B = m[1][1]; G = (20+71+90+140+2)>>2 = 80; R = (10+30+131+150+2)>>2 = 80"},
        {"y": 1, "x": 2, "site": "G on a B row", "bgr": [(80 + 101 + 1) >> 1, 90, (30 + 150 + 1) >> 1],
         "why": "B = (80+101+1)>>1 = 91 (left/right); G = m[1][2]; R = (30+150+1)>>1 = 90 (up/down)"},
        {"y": 2, "x": 1, "site": "G on an R row", "bgr": [(80 + 201 + 1) >> 1, 140, (131 + 150 + 1) >> 1],
         "why": "B = (80+201+1)>>1 = 141 (up/down); G = m[2][1]; R = (131+150+1)>>1 = 141 (left/right)"},
        {"y": 2, "x": 2, "site": "R", "bgr": [(80 + 101 + 201 + 220 + 2) >> 2, (90 + 140 + 160 + 210 + 2) >> 2, 150],
         "why": "B = (80+101+201+220+2)>>2 = 151 (diagonals); G = (90+140+160+210+2)>>2 = 150; R = m[2][2]"},
    ]
    for wk in worked:
        assert bgr[wk["y"]][wk["x"]] == wk["bgr"], (wk, bgr[wk["y"]][wk["x"]])
    for y in range(6):  # the copies: columns first, then rows (so the corners are pixel (1,1) / (1,4) / (4,1) / (4,4))
        assert bgr[y][0] == bgr[y][1] and bgr[y][5] == bgr[y][4]
    assert bgr[0] == bgr[1] and bgr[5] == bgr[4]
    sigma, steps, taps, centre_adj = ed_taps(15)
    out = {
        "_provenance": "tests/golden/make_bayer_ed_kat.py: hand-derivable vectors from SURVEY.md Appendix A.7 / A.3; NOT OpenCV outputs "
                       "(OpenCV is absent from the image)",
        "bayer_rggb_6x6": {"mosaic": M, "bgr": bgr, "worked": worked,
                           "h2_all_zero": {"mosaic": M[:2], "bgr": bayer_bg2bgr(M[:2])},
                           "h3": {"mosaic": M[:3], "bgr": bayer_bg2bgr(M[:3])},
                           "w4": {"mosaic": [r[:4] for r in M], "bgr": bayer_bg2bgr([r[:4] for r in M])}},
        "gauss15_error_diffusion": {"sigma": sigma, "steps": steps, "centre_adj": centre_adj, "taps": taps},
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bayer_ed_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, taps)
