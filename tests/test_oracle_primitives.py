"""Known-answer tests that pin the oracle's restatement of the OpenCV primitives.

OpenCV is neither vendored nor installed, so each primitive is checked against its DEFINITION (written independently here in
numpy / pure Python), not against an OpenCV binary: "OpenCV-version parity unpinned".
"""
import math

import numpy as np
import pytest

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
          (-2, 2), (-1, 3)]


def test_cv_round_half_even(oracle_lib):
    L = oracle_lib.lib()
    for v, e in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2), (1e6 + 0.5, 1000000)]:
        assert L.yo_cv_round(v) == e


def test_reflect101(oracle_lib):
    L = oracle_lib.lib()
    n = 10
    assert [L.yo_reflect101(i, n) for i in range(-4, 14)] == [4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6, 5]


def _resize_ref(src, dw, dh):
    """cv::resize INTER_LINEAR 8U from the published algorithm, written with Python scalars."""
    sh, sw = src.shape
    sx_, sy_ = 1.0 / (dw / sw), 1.0 / (dh / sh)
    def coef(d, scale, n, clamp):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = math.floor(f)
        f = np.float32(f - np.float32(s))
        if clamp:
            if s < 0: f, s = np.float32(0), 0
            if s >= n - 1: f, s = np.float32(0), n - 1
        a0 = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))
        a1 = int(np.rint(f * np.float32(2048)))
        return s, a0, a1
    out = np.zeros((dh, dw), np.uint8)
    xs = [coef(x, sx_, sw, True) for x in range(dw)]
    for y in range(dh):
        s, b0, b1 = coef(y, sy_, sh, False)
        r0, r1 = min(max(s, 0), sh - 1), min(max(s + 1, 0), sh - 1)
        for x in range(dw):
            sx, a0, a1 = xs[x]
            sx1 = min(sx + 1, sw - 1)
            h0 = int(src[r0, sx]) * a0 + int(src[r0, sx1]) * a1
            h1 = int(src[r1, sx]) * a0 + int(src[r1, sx1]) * a1
            out[y, x] = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255
    return out


def test_resize_linear(oracle_lib):
    rng = np.random.default_rng(0)
    const = np.full((40, 50), 137, np.uint8)
    assert (oracle_lib.resize_linear_u8(const, 42, 33) == 137).all()
    img = rng.integers(0, 256, (48, 60), dtype=np.uint8)
    assert np.array_equal(oracle_lib.resize_linear_u8(img, 60, 48), img)            # scale 1: every coefficient is (2048, 0)
    for (dw, dh) in [(50, 40), (53, 41), (25, 20)]:
        assert np.array_equal(oracle_lib.resize_linear_u8(img, dw, dh), _resize_ref(img, dw, dh))
    ramp = np.tile(np.arange(60, dtype=np.uint8) * 4, (48, 1))
    out = oracle_lib.resize_linear_u8(ramp, 50, 40)
    assert (np.diff(out.astype(int), axis=1) >= 0).all() and (out[0] == out[-1]).all()


def _fast_score_definition(d16, thr):
    """Largest t for which the pixel is still a FAST-9/16 corner at threshold t (brute force), or None."""
    def corner(t):
        for s in range(16):
            arc = [d16[(s + k) % 16] for k in range(9)]
            if all(v > t for v in arc) or all(v < -t for v in arc):
                return True
        return False
    if not corner(thr):
        return None
    t = thr
    while corner(t + 1):
        t += 1
    return t


def test_corner_score_equals_definition(oracle_lib):
    rng = np.random.default_rng(1)
    hits = 0
    for _ in range(4000):
        base = rng.integers(-120, 120)
        d = (base + rng.integers(-30, 31, 16)).astype(np.int32)
        if rng.random() < 0.5:
            k = rng.integers(0, 16)
            for j in range(int(rng.integers(3, 9))):
                d[(k + j) % 16] = rng.integers(-10, 11)
        d = np.clip(d, -255, 255)
        ref = _fast_score_definition(list(d), 20)
        if ref is None:
            continue
        hits += 1
        d25 = np.concatenate([d, d[:9]]).astype(np.int32)
        assert oracle_lib.corner_score16(d25, 20) == ref
    assert hits > 500


def _patch(center, ring_vals):
    p = np.full((7, 7), center, np.uint8)
    for (dx, dy), v in zip(CIRCLE, ring_vals):
        p[3 + dy, 3 + dx] = v
    return p


def test_fast_hand_built_patches(oracle_lib):
    ring9 = [150] * 9 + [100] * 7
    kp = oracle_lib.fast9_16(_patch(100, ring9), 20, nms=True)
    assert len(kp) == 1 and (kp["x"][0], kp["y"][0]) == (3, 3)
    assert kp["response"][0] == 49 and kp["size"][0] == 7 and kp["angle"][0] == -1 and kp["class_id"][0] == -1
    assert len(oracle_lib.fast9_16(_patch(100, [150] * 8 + [100] * 8), 20)) == 0             # 8 contiguous: not a corner
    assert len(oracle_lib.fast9_16(_patch(100, [121] * 16), 20)) == 1                          # strict: 121 > 100 + 20
    assert len(oracle_lib.fast9_16(_patch(100, [120] * 16), 20)) == 0
    dark = oracle_lib.fast9_16(_patch(200, [100] * 3 + [60] * 10 + [100] * 3), 20)
    assert len(dark) == 1 and dark["response"][0] == 139                                     # best 9-arc lies inside the ten 60s: min(v - p) = 140 -> 140 - 1
    wrap = [150] * 5 + [100] * 7 + [150] * 4                                                  # arc wraps around index 0
    assert len(oracle_lib.fast9_16(_patch(100, wrap), 20)) == 1


def test_fast_nms_and_border(oracle_lib):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 52), dtype=np.uint8)
    all_k = oracle_lib.fast9_16(img, 20, nms=False)
    nms_k = oracle_lib.fast9_16(img, 20, nms=True)
    assert len(all_k) > len(nms_k) > 0
    assert all_k["x"].min() >= 3 and all_k["x"].max() <= 52 - 4 and all_k["y"].min() >= 3 and all_k["y"].max() <= 40 - 4
    score = np.zeros((40, 52), int)
    score[all_k["y"].astype(int), all_k["x"].astype(int)] = all_k["response"].astype(int)
    keep = []
    for y in range(3, 37):
        for x in range(3, 49):
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
            if s > 0 and (s > nb).all():
                keep.append((x, y, s))                                                        # row-major, strictly greater than 8 neighbours
    assert keep == list(zip(nms_k["x"].astype(int), nms_k["y"].astype(int), nms_k["response"].astype(int)))


def test_gaussian_kernel_and_blur(oracle_lib):
    k = oracle_lib.gauss_kernel_fixed(7, 2.0, 8)
    assert list(k) == [18, 34, 48, 56, 48, 34, 18] and k.sum() == 256
    const = np.full((20, 30), 201, np.uint8)
    assert (oracle_lib.gaussian_blur_7x7_s2(const) == 201).all()
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    out = oracle_lib.gaussian_blur_7x7_s2(imp).astype(int)
    ref = (255 * np.outer(k, k) + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], ref) and out.sum() == ref.sum()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (17, 23), dtype=np.uint8)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")                                      # numpy 'reflect' == BORDER_REFLECT_101
    h = sum(int(k[i]) * pad[:, i:i + 23] for i in range(7))
    v = sum(int(k[j]) * h[j:j + 17, :] for j in range(7))
    assert np.array_equal(oracle_lib.gaussian_blur_7x7_s2(img), ((v + 32768) >> 16).astype(np.uint8))


def test_fast_atan2(oracle_lib):
    for (y, x, e) in [(0, 1, 0), (1, 0, 90), (0, -1, 180), (-1, 0, 270)]:
        assert oracle_lib.fast_atan2(y, x) == pytest.approx(e, abs=0.02)
    assert oracle_lib.fast_atan2(0, 0) == 0
    rng = np.random.default_rng(4)
    ys, xs = rng.integers(-40000, 40000, 3000), rng.integers(-40000, 40000, 3000)
    err = 0
    for y, x in zip(ys, xs):
        a = oracle_lib.fast_atan2(float(y), float(x))
        assert 0 <= a <= 360
        t = math.degrees(math.atan2(y, x)) % 360
        err = max(err, min(abs(a - t), 360 - abs(a - t)))
    assert err < 0.3                                                                           # OpenCV documents ~0.3 degrees


def test_deterministic_trig_vs_libm(oracle_lib):
    """The steering trig contract (oracle/oracle_trig.h) against this container's libm and against float(cos(double))."""
    L = oracle_lib.lib()
    import ctypes as C
    nc, ns = C.c_long(), C.c_long()
    lo = np.float32(0).view(np.uint32).item()
    hi = np.float32(6.2831855).view(np.uint32).item()
    tot = L.yo_trig_mismatch_count(lo, hi, 997, C.byref(nc), C.byref(ns))                     # ~1.1 M samples over [0, 2 pi]
    assert tot > 1_000_000
    assert nc.value <= tot * 1e-3 and ns.value <= tot * 1e-3   # glibc cosf/sinf are < 1 ULP but not correctly rounded (measured here: ~4e-4 of inputs)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 6.2831855, 20000).astype(np.float32)
    c = np.array([L.yo_cosf_det(float(v)) for v in x], np.float32)
    s = np.array([L.yo_sinf_det(float(v)) for v in x], np.float32)
    assert np.array_equal(c, np.cos(x.astype(np.float64)).astype(np.float32))
    assert np.array_equal(s, np.sin(x.astype(np.float64)).astype(np.float32))


def test_constructor_tables(oracle_lib):
    t = oracle_lib.OrbExtractorOracle(1000, 1.2, 8, 20, 7).tables()
    assert list(t["per_level"]) == [217, 181, 151, 126, 105, 88, 73, 59]
    assert list(t["max_x"][:16]) == [0] * 11 + [26, 25, 22, 19, 15]                              # the reference's resize+push_back table
    assert list(oracle_lib.OrbExtractorOracle(2000, 1.2, 8, 20, 7).tables()["per_level"]) == [434, 362, 302, 252, 210, 175, 146, 119]
    assert t["scale"][3] == np.float32(1.2 ** 3 if False else np.float32(pow(np.float32(1.2), 3)))
