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
