"""The C++ adapters of include/ydorb/*.hpp EXECUTED (not just parsed): tests/cpp_host/adapter_run.cpp instantiates
localBundleAdjustImpl (SURVEY 8a rows B0 graph assembly + B10 write-back, optimizer.cpp:138-283, :336-351),
searchByProjectionInLastAndCurrentFrame (orbMatcher.cpp:65-155), searchByBowInKeyFrameAndFrame (:303-379) and the OrbExtractor class
(orbExtractor.cpp:355-399 + the public m_v_imagePyramid) with stand-ins of Frame / KeyFrame / MapPoint / Map that carry real data,
runs them on the GPU and dumps what they did to those objects.  Here the reference's host-side logic is stated independently in
numpy (covisibility walk, vertex ids, edge filter, Converter round trips, float projections), the flat problem goes through the
direct C-ABI call (ctypes mirror), and the two must agree.  OpenCV / Eigen are absent: the harness compiles against the functional
test-only mocks of tests/cpu_harness/mockrt (fixed float arithmetic, documented there)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import bow_nodes, feature_vector

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp_host", "adapter_run.cpp")
F32 = np.float32


def _build(tmp):
    exe = os.path.join(tmp, "adapter_run")
    lib_dir = os.path.join(ROOT, "ydorbslam_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-ffp-contract=off", "-I" + os.path.join(ROOT, "tests", "cpu_harness", "mockrt"),
                           "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + lib_dir, "-l:libydorb.so", "-Wl,-rpath," + lib_dir])
    return exe


def _run(exe, what, blob, tmp):
    inp, out = os.path.join(tmp, what + ".in"), os.path.join(tmp, what + ".out")
    open(inp, "wb").write(blob)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(__import__("torch").__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    subprocess.check_call([exe, what, inp, out], env=env)
    return open(out, "rb").read()


def test_adapter_harness_builds_and_links(tmp_path):
    """CPU side: the three adapter headers instantiate against stand-ins with real storage and link against libydorb.so."""
    import ydorbslam_amd as y
    if not os.path.exists(y.library_path()):
        y.build_library()
    assert os.path.exists(_build(str(tmp_path)))


# ---- Converter (converter.cpp:12-38) as the adapter uses it: float 4x4 <-> (t, unit quaternion), the textbook formulas -------------
def _pose_to_se3quat(T):
    R = T[:3, :3].astype(np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)  # x y z w
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0], q[1], q[2] = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3], q[j], q[k] = (R[k, j] - R[j, k]) * t, (R[j, i] + R[i, j]) * t, (R[k, i] + R[i, k]) * t
    if q[3] < 0:
        q = -q
    q = q / np.sqrt(q @ q)
    return np.concatenate([T[:3, 3].astype(np.float64), q])


def _se3quat_to_pose(p):
    x, y, z, w = p[3:]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R.astype(np.float32)
    T[:3, 3] = p[:3].astype(np.float32)
    return T


def _camera_blob(fx, fy, cx, cy, base, bf, bounds):
    return np.array([fx, fy, cx, cy, base, bf, *bounds], np.float32).tobytes()


def _kp_blob(kps, desc, right):
    import ydorbslam_amd as y
    return np.ascontiguousarray(kps, y.KP_DTYPE).tobytes() + np.ascontiguousarray(desc, np.uint8).tobytes() + np.ascontiguousarray(right, np.float32).tobytes()


@pytest.mark.gpu
def test_local_bundle_adjust_adapter_executed(tmp_path):
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_ba_problem
    exe = _build(str(tmp_path))
    rng = np.random.default_rng(21)
    prob = synth_ba_problem(10, 500, 5, seed=21, outlier_frac=0.06, mono_frac=0.3)
    K, NP = len(prob["poses"]), len(prob["points"])
    ids = [0, 3, 4, 7, 8, 9, 12, 13, 15, 20]                    # keyframe index -> m_int_keyFrameID
    kf_bad = [False] * K; kf_bad[7] = True                     # a bad covisible keyframe: tagged, never a vertex (optimizer.cpp:146-151)
    kf0, conn = 8, [6, 7, 0, 9]                                # the current keyframe and its ordered covisible ones; index 0 has id 0 -> fixed (:191)
    mp_bad = np.zeros(NP, bool); mp_bad[rng.choice(NP, 12, replace=False)] = True
    inv_s2 = (1.0 / np.power(np.float32(1.2), 2 * np.arange(8))).astype(np.float32)
    poses32 = np.stack([_se3quat_to_pose(p) for p in prob["poses"]])
    # one keypoint per observation
    kf_kps = [[] for _ in range(K)]
    obs = [[] for _ in range(NP)]
    for e in range(len(prob["edge_pose"])):
        k, p = int(prob["edge_pose"][e]), int(prob["edge_point"][e])
        octv = int(rng.integers(0, 8))
        kf_kps[k].append((F32(prob["meas"][e, 0]), F32(prob["meas"][e, 1]), octv, F32(prob["meas"][e, 2]) if prob["meas"][e, 2] >= 0 else F32(-1), p))
        obs[p].append((k, len(kf_kps[k]) - 1))
    blob = _camera_blob(*[F32(v) for v in prob["camera"][:4]], F32(0.08), F32(prob["camera"][4]), (0, 640, 0, 480))
    blob += struct.pack("<4i", K, NP, kf0, len(conn))
    for k in range(K):
        n = len(kf_kps[k])
        kps = np.zeros(n, y.KP_DTYPE)
        kps["x"], kps["y"], kps["octave"] = [a[0] for a in kf_kps[k]], [a[1] for a in kf_kps[k]], [a[2] for a in kf_kps[k]]
        blob += struct.pack("<3i", ids[k], int(kf_bad[k]), n) + poses32[k].tobytes() + inv_s2.tobytes()
        blob += _kp_blob(kps, np.zeros((n, 32), np.uint8), np.array([a[3] for a in kf_kps[k]], np.float32))
        blob += np.array([a[4] for a in kf_kps[k]], np.int32).tobytes()
    blob += np.array(conn, np.int32).tobytes()
    pts32 = prob["points"].astype(np.float32)
    for p in range(NP):
        blob += struct.pack("<3i", p, int(mp_bad[p]), len(obs[p])) + pts32[p].tobytes() + np.array(obs[p], np.int32).reshape(-1, 2).tobytes()
    out = _run(exe, "ba", blob, str(tmp_path))

    # ---- the reference's host side, stated independently (optimizer.cpp:138-283) --------------------------------------------------
    tag = ids[kf0]
    local_tag = {kf0, *conn}
    local = [kf0] + [c for c in conn if not kf_bad[c]]                                  # :140-152
    local_mps, seen = [], set()
    for k in local:                                                                     # :153-162
        for a in kf_kps[k]:
            if not mp_bad[a[4]] and a[4] not in seen:
                seen.add(a[4]); local_mps.append(a[4])
    fixed_tag, fixed = set(), []
    for p in local_mps:                                                                 # :163-173 (order of a std::map keyed by pointers: free)
        for k, _ in obs[p]:
            if k not in local_tag and k not in fixed_tag:
                fixed_tag.add(k)
                if not kf_bad[k]:
                    fixed.append(k)
    pose_kf = [k for k in local if not kf_bad[k]] + fixed                               # :185-216
    is_fixed = [ids[k] == 0 for k in local if not kf_bad[k]] + [True] * len(fixed)
    max_id = max(ids[k] for k in pose_kf)
    pidx = {k: i for i, k in enumerate(pose_kf)}
    ep, eq, meas, info, edge_kf = [], [], [], [], []
    for i, p in enumerate(local_mps):                                                   # :225-283
        for k, idx in obs[p]:
            if kf_bad[k] or ids[k] > max_id or k not in pidx:
                continue
            a = kf_kps[k][idx]
            ep.append(pidx[k]); eq.append(i); meas.append([a[0], a[1], a[3] if a[3] >= 0 else -1.0]); info.append(inv_s2[a[2]]); edge_kf.append((k, idx))
    flat = dict(poses=np.stack([_pose_to_se3quat(poses32[k]) for k in pose_kf]), fixed=np.array(is_fixed, np.uint8),
                points=pts32[local_mps].astype(np.float64), edge_pose=np.array(ep, np.int32), edge_point=np.array(eq, np.int32),
                meas=np.array(meas, np.float64), info=np.array(info, np.float64),
                camera=np.array([F32(v) for v in prob["camera"]], np.float64))
    ref = y.Optimizer.local_bundle_adjust(flat)                                         # the direct C-ABI call on the same flat problem
    assert ref["outlier"].sum() > 5 and len(fixed) >= 2 and any(is_fixed[:len(local)])
    erased = {(edge_kf[e], local_mps[eq[e]]) for e in range(len(ep)) if ref["outlier"][e] and not mp_bad[local_mps[eq[e]]]}   # :336-341

    # ---- what the adapter did to the objects -----------------------------------------------------------------------------------
    off = 0
    n_local = len([k for k in local if not kf_bad[k]])
    for k in range(K):
        T = np.frombuffer(out, np.float32, 16, off).reshape(4, 4); off += 64
        ltag, ftag = struct.unpack_from("<2i", out, off); off += 8
        mpv = np.frombuffer(out, np.int32, len(kf_kps[k]), off); off += 4 * len(kf_kps[k])
        assert ltag == (tag if k in local_tag else 0) and ftag == (tag if k in fixed_tag else 0), "BA tags of keyframe %d" % k
        if k in pose_kf[:n_local]:                                                      # :342-346: local keyframes get the optimised pose
            want = _se3quat_to_pose(ref["poses"][pidx[k]])
            assert np.allclose(T, want, rtol=1e-5, atol=1e-6), "pose of local keyframe %d" % k
            if not is_fixed[pidx[k]]:
                assert not np.array_equal(T, poses32[k])
        else:
            assert np.array_equal(T, poses32[k]), "keyframe %d must keep its pose" % k
        for i, a in enumerate(kf_kps[k]):
            assert mpv[i] == (-1 if ((k, i), a[4]) in erased else a[4]), "matched map point %d of keyframe %d" % (i, k)
    for p in range(NP):
        X = np.frombuffer(out, np.float32, 3, off); off += 12
        upd, nob = struct.unpack_from("<2i", out, off); off += 8
        ob = np.frombuffer(out, np.int32, 2 * nob, off).reshape(-1, 2); off += 8 * nob
        if p in seen:                                                                    # :347-351
            assert upd == 1 and np.allclose(X, ref["points"][local_mps.index(p)].astype(np.float32), rtol=1e-5, atol=1e-6), "map point %d" % p
        else:
            assert upd == 0 and np.array_equal(X, pts32[p])
        assert sorted(map(tuple, ob.tolist())) == sorted(o for o in obs[p] if (o, p) not in erased), "observations of map point %d" % p
    assert off == len(out)


def _two_frames(w=640, h=480, nf=1000):
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_stream
    S = synth_stream(w, h, 2, seed=3, segment=2)
    ex = y.OrbExtractor(nf, 1.2, 8, 20, 7)
    (ka, da), (kb, db) = ex.extract(S["frames"][0]), ex.extract(S["frames"][1])
    return S, ex.tables()["scale"], ka, da, kb, db


@pytest.mark.gpu
@pytest.mark.parametrize("tz", [0.0, 0.5, -0.5])
def test_search_by_projection_last_current_adapter_executed(tmp_path, tz):
    """tz = 0 / +0.5 / -0.5 with a 0.08 baseline: the symmetric, forward and backward level windows of orbMatcher.cpp:79-80,95-101."""
    import ydorbslam_amd as y
    exe = _build(str(tmp_path))
    W, H = 640, 480
    S, sf, ka, da, kb, db = _two_frames(W, H)
    rng = np.random.default_rng(5)
    fx = fy = F32(500.0); cx, cy = F32((W - 1) * 0.5), F32((H - 1) * 0.5); base, bf = F32(0.08), F32(40.0)
    A = S["affine"][0].astype(np.float64)
    z0 = 5.0
    # a fronto-parallel plane at depth z0 seen by `last` = identity; `cur` = in-plane rotation + shift (+ tz): the image motion is A
    M = A[[0, 1, 3, 4]].reshape(2, 2)
    c = np.array([cx, cy], np.float64)
    sh = A[[2, 5]] - c + M @ c
    Tc = np.eye(4, dtype=np.float32)
    Tc[:2, :2] = M.astype(np.float32)
    Tc[:3, 3] = [sh[0] * z0 / fx, sh[1] * z0 / fy, tz]
    Tl = np.eye(4, dtype=np.float32)
    has = rng.random(len(ka)) < 0.7
    mp_of_last = np.full(len(ka), -1, np.int32); mp_of_last[has] = np.arange(has.sum())
    nmp = int(has.sum()) + 40
    pos = np.zeros((nmp, 3), np.float32)
    pos[:has.sum(), 0] = (ka["x"][has] - cx) / fx * F32(z0); pos[:has.sum(), 1] = (ka["y"][has] - cy) / fy * F32(z0); pos[:has.sum(), 2] = z0
    nobs = rng.integers(0, 4, nmp).astype(np.int32)
    mdesc = np.zeros((nmp, 32), np.uint8); mdesc[:has.sum()] = da[has]; mdesc[has.sum():] = rng.integers(0, 256, (40, 32))
    outl = (rng.random(len(ka)) < 0.1).astype(np.int32)
    mp_of_cur = np.full(len(kb), -1, np.int32)
    pre = rng.choice(len(kb), 60, replace=False)
    mp_of_cur[pre] = has.sum() + rng.integers(0, 40, 60)                                  # map points the current frame already holds
    right_b = np.where(rng.random(len(kb)) < 0.5, kb["x"] - F32(8.0), F32(-1)).astype(np.float32)
    blob = _camera_blob(fx, fy, cx, cy, base, bf, (0, W, 0, H)) + sf.astype(np.float32).tobytes() + struct.pack("<i", nmp)
    for i in range(nmp):
        blob += pos[i].tobytes() + struct.pack("<i", int(nobs[i])) + mdesc[i].tobytes()
    blob += Tl.tobytes() + struct.pack("<i", len(ka)) + _kp_blob(ka, da, np.full(len(ka), -1, np.float32)) + mp_of_last.tobytes() + outl.tobytes()
    blob += Tc.tobytes() + struct.pack("<i", len(kb)) + _kp_blob(kb, db, right_b) + mp_of_cur.tobytes() + np.zeros(len(kb), np.int32).tobytes()
    blob += struct.pack("<fi", 15.0, 1)
    out = _run(exe, "proj", blob, str(tmp_path))
    n_adapter = struct.unpack_from("<i", out, 0)[0]
    got = np.frombuffer(out, np.int32, len(kb), 4)

    # ---- orbMatcher.cpp:65-116 stated in float32, one rounded operation at a time ----------------------------------------------------
    def mv(R, v):   # 3x3 times 3-vector, products accumulated in float in ascending k (the mock cv::Mat's documented arithmetic)
        return np.array([F32(F32(F32(R[i, 0] * v[0]) + F32(R[i, 1] * v[1])) + F32(R[i, 2] * v[2])) for i in range(3)], np.float32)
    Rcw, tcw = Tc[:3, :3], Tc[:3, 3]
    twc = mv((-Rcw.T).astype(np.float32), tcw)
    tlc = (mv(Tl[:3, :3], twc) + Tl[:3, 3]).astype(np.float32)
    fwd, bwd = tlc[2] > base, -tlc[2] > base
    assert (fwd, bwd) == (tz < 0, tz > 0) or tz == 0.0                                     # (camera centre moves opposite to t)
    q = np.zeros(len(ka), y.QUERY_DTYPE)
    qd = np.zeros((len(ka), 32), np.uint8)
    for i in range(len(ka)):
        m = mp_of_last[i]
        if m < 0 or outl[i]:
            continue
        Xc = (mv(Rcw, pos[m]) + tcw).astype(np.float32)
        x, yy, z = Xc
        u, v = F32(F32(F32(fx * x) / z) + cx), F32(F32(F32(fy * yy) / z) + cy)
        if not (z >= 0.0 and 0 <= u < W and 0 <= v < H):
            continue
        o = int(ka["octave"][i])
        lv = (o, -1) if fwd else ((0, o) if bwd else (o - 1, o + 1))
        r = F32(F32(15.0) * sf[o])
        q[i] = (u, v, r, lv[0], lv[1], F32(u - F32(bf / z)), r, ka["angle"][i], o, 1 | (2 if nobs[m] > 0 else 0))
        qd[i] = mdesc[m]
    taken = np.array([1 if (mp_of_cur[i] >= 0 and nobs[mp_of_cur[i]] > 0) else 0 for i in range(len(kb))], np.uint8)
    n_ref, assigned, _ = y.OrbMatcher(0.9, True).search_by_projection(1, y.FrameView(kb, db, (0.0, float(W), 0.0, float(H)), right_b), q, qd, taken,
                                                                      np.full(len(kb), -2, np.int32))
    want = mp_of_cur.copy()
    for i in range(len(kb)):                                                               # :117-153 assignments and the histogram cull
        if assigned[i] >= 0:
            want[i] = mp_of_last[assigned[i]]
        elif assigned[i] == -1:
            want[i] = -1
    assert n_adapter == n_ref and n_ref > 10
    assert np.array_equal(got, want)
    assert (got != mp_of_cur).sum() >= 5        # (the count includes matches the histogram cull removed again, orbMatcher.cpp:138-153)


@pytest.mark.gpu
def test_search_by_bow_adapter_executed(tmp_path):
    import ydorbslam_amd as y
    exe = _build(str(tmp_path))
    _, _, ka, da, kb, db = _two_frames()
    rng = np.random.default_rng(9)
    has = rng.random(len(ka)) < 0.7
    mp_of_a = np.full(len(ka), -1, np.int32); mp_of_a[has] = np.arange(has.sum())
    nmp = int(has.sum())
    bad = (rng.random(nmp) < 0.1).astype(np.int32)
    fa, fb = feature_vector(bow_nodes(da)), feature_vector(bow_nodes(db))

    def fv_blob(fv):
        ids, start, feat = fv
        b = struct.pack("<i", len(ids))
        for i in range(len(ids)):
            f = np.asarray(feat[start[i]:start[i + 1]], np.uint32)
            b += struct.pack("<Ii", int(ids[i]), len(f)) + f.tobytes()
        return b
    blob = struct.pack("<i", nmp) + bad.tobytes()
    blob += struct.pack("<i", len(ka)) + _kp_blob(ka, da, np.full(len(ka), -1, np.float32)) + mp_of_a.tobytes() + fv_blob(fa)
    blob += struct.pack("<i", len(kb)) + _kp_blob(kb, db, np.full(len(kb), -1, np.float32)) + fv_blob(fb)
    blob += struct.pack("<fi", 0.7, 1)
    out = _run(exe, "bow", blob, str(tmp_path))
    n_adapter, n_out = struct.unpack_from("<2i", out, 0)
    got = np.frombuffer(out, np.int32, n_out, 8)
    valid = np.array([1 if (m >= 0 and not bad[m]) else 0 for m in mp_of_a], np.uint8)     # orbMatcher.cpp:322: has a map point that is not bad
    n_ref, o_ref = y.OrbMatcher(0.7, True).search_by_bow(3, ka, da, valid, y.FeatureVector(*fa), kb, db, None, y.FeatureVector(*fb))
    want = np.where(o_ref >= 0, mp_of_a[np.maximum(o_ref, 0)], -1)
    assert n_out == len(kb) and n_adapter == n_ref and n_ref > 20
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_extractor_adapter_executed(tmp_path):
    """YDORBSLAM::OrbExtractor (include/ydorb/orbExtractor.hpp): extractAndCompute + the public m_v_imagePyramid ROI views."""
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_frame
    exe = _build(str(tmp_path))
    w, h, nf = 752, 480, 1000
    img = synth_frame(w, h, 11)
    out = _run(exe, "extract", struct.pack("<3i", w, h, nf) + img.tobytes(), str(tmp_path))
    n = struct.unpack_from("<i", out, 0)[0]
    off = 4
    kps = np.frombuffer(out, y.KP_DTYPE, n, off); off += 28 * n
    desc = np.frombuffer(out, np.uint8, 32 * n, off).reshape(n, 32); off += 32 * n
    ex = y.OrbExtractor(nf, 1.2, 8, 20, 7)
    rk, rd = ex.extract(img)
    assert kps.tobytes() == rk.tobytes() and np.array_equal(desc, rd) and n > 900
    nl = struct.unpack_from("<i", out, off)[0]; off += 4
    assert nl == 8
    pyr = ex.read_pyramid()
    for l in range(nl):
        lw, lh = struct.unpack_from("<2i", out, off); off += 8
        roi = np.frombuffer(out, np.uint8, lw * lh, off).reshape(lh, lw); off += lw * lh
        assert np.array_equal(roi, pyr[l][19:19 + lh, 19:19 + lw]), "m_v_imagePyramid[%d]" % l
    levels, kpnum = struct.unpack_from("<2i", out, off); off += 8
    assert levels == 8 and kpnum == 8            # getKeyPointsNum() returns the level count in the reference too (orbExtractor.hpp:42)
    assert np.array_equal(np.frombuffer(out, np.float32, 8, off), ex.tables()["scale"])
