"""Definitional cross-checks of the CPU ORACLE's feature stages on textured images (CPU only, no GPU).

The reference ships no fixture for cv::FAST / IC_Angle / computeOrbDescriptor (PARITY UNPINNED), and the
hand-made images of test_oracle_kat.py exercise a handful of pixels.  Here every pixel of textured 160x120 synthetic frames
goes through restatements written from the DEFINITIONS, sharing no code and no evaluation order with oracle/*.cpp:

  * FAST-9/16 (OpenCV features2d fast.cpp / fast_score.cpp): a pixel is a corner at threshold t iff
    some contiguous arc of >= 9 of its 16 ring pixels is entirely brighter than v + t or entirely darker than v - t; its
    score is the LARGEST t at which it is still a corner = max over the sixteen 9-arcs of the arc's minimum contrast, minus
    one; 3x3 non-maximum suppression keeps strictly greater scores, non-corners and the 3-pixel border count as 0; row-major
    emission.  (cornerScore's pairwise min/max ladder in the oracle is an optimisation of exactly this.)
  * IC_Angle (reference src/ORBextractor.cc:76-103): integer moments over the radius-15 disc given by umax, then
    cv::fastAtan2 (appendix A.4) in float32 without FMA, restated with numpy float32 scalars.
  * computeOrbDescriptor (:107-146): 256 comparisons of the blurred level at cvRound-ed rotated pattern points, float32
    products and sums without FMA, (float)cos / sin of the float angle.
"""
import math
import os
import re

import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]          # SURVEY.md 8 table (src/ORBextractor.cc:452-468)


def fast_definitional(img, t, want_map=False):
    """(x, y, score) of FAST-9/16 corners at threshold t with 3x3 NMS, row-major, straight from the definition"""
    h, w = img.shape
    v = img.astype(np.int32)
    H, W = h - 6, w - 6
    c = v[3:h - 3, 3:w - 3]
    d = np.stack([v[3 + dy:3 + dy + H, 3 + dx:3 + dx + W] - c for (dx, dy) in RING])       # ring - centre, [16, H, W]
    best = np.full((H, W), -1 << 20, np.int32)
    for k in range(16):
        arc = d[[(k + i) % 16 for i in range(9)]]
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))      # brighter arc / darker arc
    score = np.zeros((h, w), np.int32)
    sc = best - 1                       # largest threshold at which the pixel is a corner
    score[3:h - 3, 3:w - 3] = np.where(best > t, sc, 0)
    if want_map:
        return score
    out = []
    for y in range(3, h - 3):
        row = score[y]
        for x in np.nonzero(row[3:w - 3])[0] + 3:
            s = row[x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 20), (2, 7), (3, 7), (4, 12), (5, 40)])
def test_fast_against_the_definition(oracle, synth, seed, t):
    from oracle_api import KP_DTYPE
    img = np.ascontiguousarray(synth.make_frame(300 + seed, 160, 120))
    out = np.zeros(20000, KP_DTYPE)
    n = oracle.lib.orb_oracle_fast(img.ctypes.data, 160, 120, 160, t, 1, out.ctypes.data, 20000)
    got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in out[:n]]
    ref = fast_definitional(img, t)
    assert len(ref) > 50, "the image should have corners at this threshold"
    assert got == ref
    # without suppression (cv::FAST(..., false) reports response 0): the corners at t are the pixels whose definitional score
    # is >= t, row-major -- i.e. the score does not depend on the threshold it was computed at
    smap = fast_definitional(img, 0, want_map=True)
    for th in (t, t + 9):
        n_all = oracle.lib.orb_oracle_fast(img.ctypes.data, 160, 120, 160, th, 0, out.ctypes.data, 20000)
        ys, xs = np.nonzero(smap >= th)
        keep = (smap[ys, xs] > 0) | (th == 0)
        assert [(int(k["x"]), int(k["y"])) for k in out[:n_all]] == list(zip(xs[keep].tolist(), ys[keep].tolist()))


def test_fast_on_a_sub_image_with_stride(oracle, synth):
    """the extractor calls FAST on cell sub-images of a level (stride = level width): same corners as a cropped copy"""
    from oracle_api import KP_DTYPE
    img = np.ascontiguousarray(synth.make_frame(310, 160, 120))
    out = np.zeros(5000, KP_DTYPE)
    x0, y0, cw, ch = 16, 16, 41, 41
    n = oracle.lib.orb_oracle_fast(img.ctypes.data + y0 * 160 + x0, cw, ch, 160, 7, 1, out.ctypes.data, 5000)
    got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in out[:n]]
    assert got == fast_definitional(img[y0:y0 + ch, x0:x0 + cw], 7) and len(got) > 3


# ---- float32 restatements (numpy float32 scalars: every operation rounds to float32, nothing is contracted) ----
F = np.float32


def fast_atan2_f32(y, x):
    """cv::fastAtan2, degrees"""
    s = F(57.29577951308232)            # (float)(180 / pi)
    p1, p3, p5, p7 = F(0.9997878412794807) * s, F(-0.3258083974640975) * s, F(0.1555786518463281) * s, F(-0.04432655554792128) * s
    eps = F(2.220446049250313e-16)      # (float)DBL_EPSILON
    ax, ay = F(abs(x)), F(abs(y))
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = F(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = F(180.0) - a
    if y < 0:
        a = F(360.0) - a
    return F(a)


def cv_round(v):
    """cvRound: round half to even"""
    return int(np.rint(np.float64(v)))


def ic_angle_definitional(img, x, y):
    cx, cy = cv_round(x), cv_round(y)
    m01 = m10 = 0
    for v in range(-15, 16):
        d = UMAX[abs(v)]
        rowv = img[cy + v, cx - d:cx + d + 1].astype(np.int64)
        m10 += int((np.arange(-d, d + 1) * rowv).sum())
        m01 += v * int(rowv.sum())
    return fast_atan2_f32(F(m01), F(m10))


def load_pattern():
    txt = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "orb_pattern_31.inc")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    vals = [int(t) for t in re.findall(r"-?\d+", txt)]
    assert len(vals) == 1024
    return np.array(vals, np.int32).reshape(256, 4)


def descriptor_definitional(blur, x, y, angle_deg, pattern):
    factor_pi = F(math.pi / F(180.0))               # (float)(CV_PI / 180.f)
    ang = F(angle_deg) * factor_pi
    a, b = F(math.cos(float(ang))), F(math.sin(float(ang)))
    cx, cy = cv_round(x), cv_round(y)

    def value(px, py):
        px, py = F(px), F(py)
        return int(blur[cy + cv_round(px * b + py * a), cx + cv_round(px * a - py * b)])
    bits = np.zeros(256, np.uint8)
    for i, (x0, y0, x1, y1) in enumerate(pattern):
        bits[i] = value(x0, y0) < value(x1, y1)
    return np.packbits(bits, bitorder="little")


def test_fast_atan2_restatement_matches_the_oracle(oracle):
    rs = np.random.RandomState(3)
    pts = [(0.0, 0.0), (0.0, 1.0), (1.0, 0.0), (-1.0, 0.0), (0.0, -1.0), (1.0, 1.0), (-1.0, -1.0), (3.0, -3.0)]
    pts += [(float(F(a)), float(F(b))) for a, b in rs.uniform(-1e5, 1e5, (3000, 2))]
    pts += [(float(a), float(b)) for a, b in rs.randint(-40000, 40000, (3000, 2))]
    oracle.lib.orb_oracle_fast_atan2.restype = __import__("ctypes").c_float
    for y, x in pts:
        assert F(oracle.lib.orb_oracle_fast_atan2(__import__("ctypes").c_float(y), __import__("ctypes").c_float(x))) == fast_atan2_f32(F(y), F(x)), (y, x)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_orientation_and_descriptors_against_the_definition(oracle, synth, seed):
    """every key point of a textured 160x120 frame: angle from the integer moments of the UNBLURRED level, descriptor bits from
    the blurred level, both restated above; the oracle's output order for a mono frame is the reversed level-major order"""
    img = synth.make_frame(320 + seed, 160, 120)
    ex = oracle.extractor(300, 1.2, 4, 20, 7)
    r, kps, desc = ex.extract(img, (0, 1000))
    assert r == 0 and len(kps) > 150
    pattern = load_pattern()
    j = 0
    n = len(kps)
    for level in range(4):
        lk = ex.level_keypoints(level)
        if len(lk) == 0:
            continue
        raw, blur = ex.level_image(level), ex.level_blurred(level)
        for k in lk:
            ang = ic_angle_definitional(raw, k["x"], k["y"])
            assert F(k["angle"]) == ang, (level, k)
            d = descriptor_definitional(blur, k["x"], k["y"], k["angle"], pattern)
            np.testing.assert_array_equal(desc[n - 1 - j], d, err_msg="level %d key point %r" % (level, k))
            assert kps[n - 1 - j]["angle"] == k["angle"] and kps[n - 1 - j]["octave"] == level
            j += 1
    assert j == n
