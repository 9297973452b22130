"""Oracle matcher (oracle/matcher_oracle.cpp): cross-checks against independent numpy statements of the same rules."""
import numpy as np

from helpers import bow_nodes, feature_vector, projection_queries, shifted_pair


def _scene(oracle_lib, idx=50, dx=5, dy=-3):
    ex = oracle_lib.OrbExtractorOracle(1000)
    a, b = shifted_pair(640, 480, idx, dx, dy)
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    return ka, da, kb, db, ex.tables()["scale"]


def test_descriptor_distance_is_popcount(oracle_lib):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    ref = np.unpackbits(a ^ b, axis=1).sum(axis=1)
    for i in range(500):
        assert oracle_lib.descriptor_distance(a[i], b[i]) == ref[i] == bin(int.from_bytes((a[i] ^ b[i]).tobytes(), "little")).count("1")
    assert oracle_lib.descriptor_distance(a[0], a[0]) == 0 and oracle_lib.descriptor_distance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_three_maxima_and_rot_bin(oracle_lib):
    import ctypes as C
    L = oracle_lib.lib()
    def tm(sizes):
        s = np.array(sizes, np.int32); o = np.zeros(3, np.int32)
        L.yo_three_maxima(s.ctypes.data_as(C.c_void_p), len(s), o.ctypes.data_as(C.c_void_p))
        return list(o)
    assert tm([0] * 30) == [-1, -1, -1]
    assert tm([5, 50, 0, 7] + [0] * 26) == [1, 3, 0]
    assert tm([100, 9, 8] + [0] * 27) == [0, -1, -1]          # second < 10 % of first
    assert tm([100, 10, 9] + [0] * 27) == [0, 1, -1]          # third < 10 %
    assert tm([3, 3, 3, 3] + [0] * 26) == [0, 1, 2]           # ties keep the earliest bins
    # factor = 1/30 (orbMatcher.cpp:78): only bins 0..12 are reachable
    bins = {L.yo_rot_bin(float(a), 0.0) for a in np.linspace(0, 359.99, 2000)}
    assert bins == set(range(13))
    assert L.yo_rot_bin(10.0, 350.0) == round((10.0 - 350.0 + 360.0) / 30.0)


def test_keypoints_in_area_matches_numpy_statement(oracle_lib):
    ka, da, kb, db, sf = _scene(oracle_lib)
    fo = oracle_lib.FrameOracle(kb, db, (0.0, 640.0, 0.0, 480.0))
    gw, gh = np.float32(64) / np.float32(640), np.float32(48) / np.float32(480)
    half_away = lambda v: np.trunc(v + np.copysign(np.float32(0.5), v)).astype(int)   # C round(): halves away from zero (245 * 0.1f == 24.5 exactly)
    lx = half_away((kb["x"] - np.float32(0)) * gw)
    ly = half_away((kb["y"] - np.float32(0)) * gh)                      # (uses minX for y too; minX == minY == 0 here)
    rng = np.random.default_rng(1)
    seen = 0
    for _ in range(40):
        x, y, r = np.float32(rng.uniform(0, 640)), np.float32(rng.uniform(0, 480)), np.float32(rng.uniform(5, 90))
        lo, hi = [(-1, -1), (0, 3), (2, -1), (1, 2)][int(rng.integers(0, 4))]
        x0, x1 = max(0, int(np.floor((x - r) * gw))), min(63, int(np.ceil((x + r) * gw)))
        y0, y1 = max(0, int(np.floor((y - r) * gh))), min(47, int(np.ceil((y + r) * gh)))
        ref = []
        for ix in range(x0, x1 + 1):
            for iy in range(y0, y1 + 1):
                for i in np.nonzero((lx == ix) & (ly == iy))[0]:
                    o = kb["octave"][i]
                    if (lo > 0 or hi >= 0) and (o < lo or (hi >= 0 and o < hi)):
                        continue
                    if abs(kb["x"][i] - x) > r and abs(kb["y"][i] - y) < r:     # the reference's per-axis test, frame.cpp:353
                        ref.append(i)
        got = fo.keypoints_in_area(x, y, r, lo, hi)
        assert list(got) == ref
        seen += len(ref)
    assert seen > 100


def test_projection_search_invariants(oracle_lib):
    ka, da, kb, db, sf = _scene(oracle_lib)
    fo = oracle_lib.FrameOracle(kb, db, (0.0, 640.0, 0.0, 480.0))
    for mode in (0, 1, 2):
        q = projection_queries(ka, sf, 5, -3, 15, mode, seed=mode)
        n, assigned, taken = fo.search_by_projection(mode, q, da, 0.9, False, orb_dist=100)
        hit = assigned >= 0
        assert n >= hit.sum() > 0                                      # the reference's count includes overwrites
        qs = assigned[hit]
        assert ((q["flags"][qs] & 1) == 1).all()                       # only valid queries match
        d = np.array([oracle_lib.descriptor_distance(da[qi], db[ti]) for qi, ti in zip(qs, np.nonzero(hit)[0])])
        assert (d <= 100).all() if mode != 1 else (d < 100).all()
        if mode == 2:
            assert (taken[hit] == 1).all()
        else:
            assert np.array_equal(taken[hit], ((q["flags"][qs] >> 1) & 1).astype(np.uint8))
    q = projection_queries(ka, sf, 5, -3, 15, 1, seed=9)
    q["flags"] = 0
    n, assigned, _ = fo.search_by_projection(1, q, da, 0.9, True)
    assert n == 0 and (assigned == -1).all()


def test_bow_search_invariants(oracle_lib):
    ka, da, kb, db, sf = _scene(oracle_lib)
    fa, fb = feature_vector(bow_nodes(da)), feature_vector(bow_nodes(db))
    va = np.ones(len(ka), np.uint8); vb = np.ones(len(kb), np.uint8)
    n3, o3 = oracle_lib.search_by_bow(3, ka, da, va, fa, kb, db, vb, fb, 0.7, False)
    n4, o4 = oracle_lib.search_by_bow(4, ka, da, va, fa, kb, db, vb, fb, 0.7, False)
    assert n3 == (o3 >= 0).sum() > 20 and n4 == (o4 >= 0).sum()
    # both modes accept the same pairs when every feature has a map point (only the output orientation differs)
    pairs3 = {(int(a), int(b)) for b, a in enumerate(o3) if a >= 0}
    pairs4 = {(int(a), int(b)) for a, b in enumerate(o4) if b >= 0}
    assert pairs3 == pairs4
    na, nb = bow_nodes(da), bow_nodes(db)
    for a, b in pairs3:
        assert na[a] == nb[b] and oracle_lib.descriptor_distance(da[a], db[b]) <= 50
    assert len({b for _, b in pairs3}) == len(pairs3)                   # a frame feature is matched once
    nz, oz = oracle_lib.search_by_bow(3, ka, da, np.zeros(len(ka), np.uint8), fa, kb, db, vb, fb, 0.7, True)
    assert nz == 0 and (oz == -1).all()


def _tri_inputs(oracle_lib, stereo_frac=0.5, mp_frac=0.3, seed=3, dy=-3):
    ka, da, kb, db, sf = _scene(oracle_lib, dx=11, dy=dy)
    rng = np.random.default_rng(seed)
    fa, fb = feature_vector(bow_nodes(da, 4)), feature_vector(bow_nodes(db, 4))
    mpa = (rng.random(len(ka)) < mp_frac).astype(np.uint8); mpb = (rng.random(len(kb)) < mp_frac).astype(np.uint8)
    ra = np.where(rng.random(len(ka)) < stereo_frac, ka["x"] - 5, -1).astype(np.float32)
    rb = np.where(rng.random(len(kb)) < stereo_frac, kb["x"] - 5, -1).astype(np.float32)
    # image B is image A moved by (11, dy): l = F^T x1 = (0, 1, -(y + dy)), i.e. the "epipolar line" of a pure image shift
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, -dy]], np.float32)
    sf = sf.astype(np.float32)
    return ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, sf, (sf * sf).astype(np.float32)


def test_triangulation_search_matches_independent_statement(oracle_lib):
    """OrbMatcher::searchForTriangulation (orbMatcher.cpp:463-565) against a second, numpy-float32 statement of
    the same rules: eligibility (no map point, stereo-only), `dist <= 50 && dist <= best` (ties: the LAST candidate wins), the
    epipole-distance exemption for stereo points, the float epipolar test with its squared denominator, one match per second feature."""
    f32 = np.float32
    for stereo_only, epi in ((False, (-900.0, 240.0)), (True, (320.0, 240.0)), (False, (320.0, 240.0))):
        ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, sf, sf2 = _tri_inputs(oracle_lib)
        n, out = oracle_lib.search_for_triangulation(ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, epi, sf, sf2, stereo_only, False)
        ex, ey = f32(epi[0]), f32(epi[1])
        exp = np.full(len(ka), -1, np.int64)
        matched = np.zeros(len(kb), bool)
        nodes_b = {int(i): fb[2][fb[1][k]:fb[1][k + 1]] for k, i in enumerate(fb[0])}
        for k, node in enumerate(fa[0]):
            if int(node) not in nodes_b:
                continue
            for i1 in fa[2][fa[1][k]:fa[1][k + 1]]:
                good_a = ra[i1] >= 0
                if mpa[i1] or (stereo_only and not good_a):
                    continue
                x1, y1 = f32(ka["x"][i1]), f32(ka["y"][i1])
                la = f32(f32(F[0, 0] * x1) + f32(F[1, 0] * y1)) + F[2, 0]
                lb = f32(f32(F[0, 1] * x1) + f32(F[1, 1] * y1)) + F[2, 1]
                lc = f32(f32(F[0, 2] * x1) + f32(F[1, 2] * y1)) + F[2, 2]
                best, best_i = 50, -1
                for i2 in nodes_b[int(node)]:
                    good_b = rb[i2] >= 0
                    if matched[i2] or mpb[i2] or (stereo_only and not good_b):
                        continue
                    d = int(np.unpackbits(da[i1] ^ db[i2]).sum())
                    if not (d <= 50 and d <= best):
                        continue
                    x2, y2 = f32(kb["x"][i2]), f32(kb["y"][i2])
                    far = float(f32(ex - x2)) ** 2 + float(f32(ey - y2)) ** 2 >= float(f32(f32(100) * sf[kb["octave"][i2]]))
                    if not (good_a or good_b or far):
                        continue
                    den = f32(f32(la * la) + f32(lb * lb))
                    if not den > 0:
                        continue
                    num = f32(f32(f32(la * x2) + f32(lb * y2)) + lc)
                    if float(f32(f32(num * num) / f32(den * den))) < 3.841 * float(sf2[kb["octave"][i2]]):
                        best, best_i = d, int(i2)
                if best_i >= 0:
                    matched[best_i] = True
                    exp[i1] = best_i
        assert np.array_equal(out, exp) and n == (exp >= 0).sum()
        assert len(set(exp[exp >= 0])) == (exp >= 0).sum()
    assert n > 10
    ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, sf, sf2 = _tri_inputs(oracle_lib)
    n2, out2 = oracle_lib.search_for_triangulation(ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, (-900.0, 240.0), sf, sf2, False, True)
    n1, out1 = oracle_lib.search_for_triangulation(ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, (-900.0, 240.0), sf, sf2, False, False)
    assert n2 <= n1 and np.all((out2 == out1) | (out2 == -1))            # the rotation histogram only removes matches


def test_fuse_search_matches_brute_force_statement(oracle_lib):
    """Search half of fuseByProjection (orbMatcher.cpp:682-745) against a brute-force statement built on the oracle's own
    getKeyPointsInArea (checked separately above): level window, float chi-square test, first minimum, <= 50."""
    f32 = np.float32
    ka, da, kb, db, sf = _scene(oracle_lib)
    sf = sf.astype(np.float32)
    inv_s2 = (f32(1.0) / (sf * sf)).astype(np.float32)
    rng = np.random.default_rng(9)
    right = np.where(rng.random(len(kb)) > 0.5, kb["x"] - 20 + rng.normal(0, 0.5, len(kb)), -1).astype(np.float32)
    fo = oracle_lib.FrameOracle(kb, db, (0.0, 640.0, 0.0, 480.0), right)
    q = projection_queries(ka, sf, 5, -3, 1.0, 1, seed=77, stereo=True)
    q["min_level"], q["max_level"] = -1, -1
    q["level"] = np.clip(ka["octave"] + rng.integers(-1, 2, len(q)), 0, 7)
    q["r"] = (f32(1.0) * sf[q["level"]]).astype(np.float32)
    n, best = fo.fuse_search(q, da, inv_s2)
    exp = np.full(len(q), -1, np.int64)
    for i in range(len(q)):
        if not (q["flags"][i] & 1):
            continue
        bd, bi = 256, -1
        for idx in fo.keypoints_in_area(float(q["u"][i]), float(q["v"][i]), float(q["r"][i])):
            lvl = int(kb["octave"][idx])
            mono = f32(float(f32(kb["x"][idx] - q["u"][i])) ** 2 + float(f32(kb["y"][idx] - q["v"][i])) ** 2)
            stereo = f32(float(mono) + float(f32(right[idx] - q["ur"][i])) ** 2)
            ok = (right[idx] >= 0 and float(f32(stereo * inv_s2[lvl])) <= 7.81) or (right[idx] < 0 and float(f32(mono * inv_s2[lvl])) <= 5.99)
            if not (q["level"][i] - 1 <= lvl <= q["level"][i] and ok):
                continue
            d = int(np.unpackbits(da[i] ^ db[idx]).sum())
            if d < bd:
                bd, bi = d, int(idx)
        if bd <= 50:
            exp[i] = bi
    assert np.array_equal(best, exp) and n == (exp >= 0).sum() > 10


def _distinctive_groups(seed=0):
    rng = np.random.default_rng(seed)
    groups = []
    for m in [1, 2, 3, 4, 5, 8, 13, 20, 20, 33, 64, 65, 130]:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        g = np.repeat(base[None], m, 0)
        flips = rng.random((m, 256)) < rng.uniform(0.02, 0.3)
        g = np.packbits(np.unpackbits(g, axis=1) ^ flips.astype(np.uint8), axis=1)
        if m >= 4 and seed % 2 == 0:
            g[m // 2] = g[0]      # duplicated rows: equal medians, the first row must win
        groups.append(g)
    return groups


def test_distinctive_descriptor_numpy_statement(oracle_lib):
    """mapPoint.cpp:191-213: least median of the sorted distance rows, median = sorted[(int)(0.5 m)], first winner kept."""
    for seed in range(4):
        for g in _distinctive_groups(seed):
            m = len(g)
            d = np.unpackbits(g[:, None, :] ^ g[None, :, :], axis=2).sum(axis=2)
            med = np.sort(d, axis=1)[:, int(0.5 * m)]
            assert oracle_lib.distinctive_descriptor(g) == int(np.argmin(med)), (seed, m)


def test_hamming_topk_chain_against_numpy(oracle_lib):
    """yo_hamming_topk = the best / second-best chain of orbMatcher.cpp:39-52.  Independent statement: stable argsort of the
    distances gives the first minimum and the first minimum of the rest; a distance of 256 never replaces the initial 256."""
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    t[7] = t[3]; t[50] = t[3]; q[0] = t[3]                      # ties at distance 0
    q[1] = 0; t[10] = 255                                       # distance 256
    offs = [0]
    cand = []
    for i in range(len(q)):
        n = int(rng.integers(0, 30)) if i != 1 else 1
        c = rng.integers(0, len(t), n) if i != 1 else np.array([10])
        cand += list(c); offs.append(len(cand))
    for co, ci in ((None, None), (np.array(offs, np.int32), np.array(cand, np.int32))):
        got = oracle_lib.hamming_topk(q, t, co, ci)
        for i in range(len(q)):
            lst = np.arange(len(t)) if ci is None else ci[co[i]:co[i + 1]]
            d = np.unpackbits(q[i][None] ^ t[lst], axis=1).sum(axis=1) if len(lst) else np.zeros(0, int)
            order = np.argsort(d, kind="stable")
            order = [r for r in order if d[r] < 256]
            exp = [256, -1, 256, -1, -1, -1]
            if len(order) > 0:
                exp[0], exp[1], exp[4] = int(d[order[0]]), int(lst[order[0]]), int(order[0])
            if len(order) > 1:
                exp[2], exp[3], exp[5] = int(d[order[1]]), int(lst[order[1]]), int(order[1])
            assert list(got[i].tolist()) == exp, (i, got[i], exp)
    assert oracle_lib.hamming_topk(q[:1], t)[0]["best_idx"] == 3 and oracle_lib.hamming_topk(q[:1], t)[0]["second_idx"] == 7
    assert oracle_lib.hamming_topk(q[1:2], t, np.array([0, 1], np.int32), np.array([10], np.int32))[0].tolist() == (256, -1, 256, -1, -1, -1)
