"""Oracle checks for Frame::computeStereoMatches (reference src/frame.cpp:362-477).

Parity unpinned: the reference ships no fixture for this function and OpenCV is absent, so the restatement
(oracle/stereo_oracle.cpp) is checked against what the text of frame.cpp implies and against a numpy statement of the
per-keypoint steps."""
import numpy as np

from ydorbslam_amd.synth import synth_stereo_pair


def _pair(oracle_lib, index=0, n_features=800):
    from oracle.orb_oracle import OrbExtractorOracle
    left, right, drow = synth_stereo_pair(640, 480, index)
    el, er = OrbExtractorOracle(n_features), OrbExtractorOracle(n_features)
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    lv_l, lv_r = [], []
    for l in range(8):
        w, h, _ = el.level_dims(l)
        lv_l.append(el.level_padded(l)[19:19 + h, 19:19 + w])
        lv_r.append(er.level_padded(l)[19:19 + h, 19:19 + w])
    t = el.tables()
    return dict(kl=kl, dl=dl, kr=kr, dr=dr, lv_l=lv_l, lv_r=lv_r, scale=t["scale"], inv=t["inv_scale"], drow=drow)


def _numpy_one(P, k, s, bf, b):
    """One loop iteration of frame.cpp:388-462 for left keypoint k with descriptor / slot s: (completed, right_x, depth, sad)."""
    kl, kr = P["kl"], P["kr"]
    kp = kl[k]
    rows = P["lv_l"][0].shape[0]
    row = int(kp["y"])
    cand, in_row = [], False
    for j in range(len(kr)):
        r = np.float32(2.0) * P["scale"][kr[j]["octave"]]
        lo = int(max(np.floor(kr[j]["y"] - r), np.float32(0)))
        hi = min(np.ceil(kr[j]["y"] + r), np.float32(rows) - np.float32(1))
        if lo <= row <= hi:
            in_row = True
            if kp["octave"] - 1 <= kr[j]["octave"] <= kp["octave"] + 1 and kp["x"] - np.float32(bf / b) <= kr[j]["x"] <= kp["x"]:
                cand.append(j)
    if not in_row or not kp["x"] >= 0:
        return False, None, None, None
    best, bj = 256, 0
    for j in cand:
        d = int(np.unpackbits(P["dl"][s] ^ P["dr"][j]).sum())
        if d < best:
            best, bj = d, j
    if best >= 75:
        return True, None, None, None
    o = int(kp["octave"])
    inv = P["inv"][o]
    lx, ly, rx = (int(np.floor(np.float64(np.float32(v) * inv) + 0.5)) for v in (kp["x"], kp["y"], kr[bj]["x"]))   # round(): halves away from zero
    L, R = P["lv_l"][o].astype(np.int64), P["lv_r"][o].astype(np.int64)
    H, W = L.shape
    if ly - 5 < 0 or ly + 6 >= H or lx - 5 < 0 or lx + 6 >= W or rx < 0 or rx + 11 >= W or rx - 10 < 0:
        return False, None, None, None
    lp = L[ly - 5:ly + 6, lx - 5:lx + 6] - L[ly, lx]
    dists = []
    for i in range(-5, 6):
        rp = R[ly - 5:ly + 6, rx + i - 5:rx + i + 6] - R[ly, rx + i]
        dists.append(int(np.abs(lp - rp).sum()))
    sad, col = 256, 0
    for i, d in enumerate(dists):
        if d < sad:
            sad, col = d, i - 5
    if col in (-5, 5):
        return False, None, None, None
    d1, d2, d3 = dists[col + 4], dists[col + 5], dists[col + 6]
    den = 2.0 * (d1 + d3 - 2.0 * d2)
    delta = np.float32((d1 - d3) / den) if den != 0 else np.float32(np.nan if d1 == d3 else np.inf * np.sign(d1 - d3))
    if delta < -1 or delta > 1:
        return False, None, None, None
    brx = P["scale"][o] * ((np.float32(rx) + delta) + np.float32(col))
    disp = kp["x"] - brx
    if 0 <= disp < np.float32(bf / b):
        if disp <= 0:
            disp, brx = np.float32(0.01), np.float32(np.float64(kp["x"]) - 0.01)
        return True, brx, np.float32(bf) / disp, sad
    return True, None, None, None


def test_oracle_matches_numpy_statement(oracle_lib):
    P = _pair(oracle_lib, 0, 500)
    bf, b = 40.0, 0.1
    for by_kp in (True, False):
        rx, depth, kept, st = oracle_lib.stereo_matches(P["kl"], P["dl"], P["kr"], P["dr"], P["lv_l"], P["lv_r"], P["scale"], P["inv"], bf, b, by_kp)
        erx = np.full(len(P["kl"]), -1, np.float32)
        edp = np.full(len(P["kl"]), -1, np.float32)
        sads, s = [], 0
        for k in range(len(P["kl"])):
            if by_kp:
                s = k
            done, x, d, sad = _numpy_one(P, k, s, bf, b)
            if x is not None:
                erx[s], edp[s] = x, d
                sads.append(sad)
            s += int(done)
        if sads and sorted(sads)[len(sads) // 2] == 0:
            erx[edp > 0] = -2
            edp[edp > 0] = -2
        assert kept == len(sads) and st == 0
        assert np.array_equal(rx.view(np.uint32), erx.view(np.uint32)), by_kp
        assert np.array_equal(depth.view(np.uint32), edp.view(np.uint32)), by_kp


def test_per_keypoint_form_recovers_the_disparity(oracle_lib):
    P = _pair(oracle_lib, 1)
    bf, b = 40.0, 0.1
    rx, depth, kept, st = oracle_lib.stereo_matches(P["kl"], P["dl"], P["kr"], P["dr"], P["lv_l"], P["lv_r"], P["scale"], P["inv"], bf, b, True)
    ok = depth > 0
    assert kept == ok.sum() and kept > 100
    disp = P["kl"]["x"][ok] - rx[ok]
    true = P["drow"][P["kl"]["y"][ok].astype(int)]
    assert np.mean(np.abs(disp - true) < 1.5 * P["scale"][P["kl"]["octave"][ok]]) > 0.9
    assert np.allclose(depth[ok], np.float32(bf) / np.maximum(disp, np.float32(0.01)), rtol=1e-6)


def test_replay_equals_per_keypoint_form_up_to_the_first_skipped_keypoint(oracle_lib):
    """frame.cpp:462: `leftIdx` stays behind from the first keypoint that leaves the loop body early."""
    P = _pair(oracle_lib, 2)
    bf, b = 40.0, 0.1
    a = oracle_lib.stereo_matches(P["kl"], P["dl"], P["kr"], P["dr"], P["lv_l"], P["lv_r"], P["scale"], P["inv"], bf, b, True)
    r = oracle_lib.stereo_matches(P["kl"], P["dl"], P["kr"], P["dr"], P["lv_l"], P["lv_r"], P["scale"], P["inv"], bf, b, False)
    first_skip = next(k for k in range(len(P["kl"])) if not _numpy_one(P, k, k, bf, b)[0])
    assert first_skip > 0
    assert np.array_equal(a[1][:first_skip], r[1][:first_skip]) and np.array_equal(a[0][:first_skip], r[0][:first_skip])
    assert r[2] <= a[2]
