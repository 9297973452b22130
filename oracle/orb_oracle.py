"""ctypes front-end to oracle/liborb_oracle.so (CPU restatement of the reference hot path).

ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by ydorbslam_amd.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborb_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.yo_extractor_create.restype = C.c_void_p
        L.yo_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.yo_extractor_destroy.argtypes = [C.c_void_p]
        L.yo_extractor_set_libm_trig.argtypes = [C.c_void_p, C.c_int]
        L.yo_extractor_set_stale_pyramid.argtypes = [C.c_void_p, C.c_int]
        L.yo_extract.restype = C.c_int
        L.yo_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.yo_extractor_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.yo_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.yo_level_padded.restype = C.c_void_p
        L.yo_level_padded.argtypes = [C.c_void_p, C.c_int]
        L.yo_level_blurred.restype = C.c_void_p
        L.yo_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.yo_level_candidates.restype = C.c_int
        L.yo_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.yo_level_keypoints.restype = C.c_int
        L.yo_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.yo_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.yo_fast9_16.restype = C.c_int
        L.yo_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.yo_corner_score16.restype = C.c_int
        L.yo_corner_score16.argtypes = [C.c_void_p, C.c_int]
        L.yo_gaussian_blur_7x7_s2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.yo_gauss_kernel_fixed.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.yo_fast_atan2.restype = C.c_float
        L.yo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.yo_cv_round.restype = C.c_int
        L.yo_cv_round.argtypes = [C.c_float]
        L.yo_fuse_search.restype = C.c_int
        L.yo_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.yo_reflect101.restype = C.c_int
        L.yo_reflect101.argtypes = [C.c_int, C.c_int]
        L.yo_cosf_det.restype = C.c_float
        L.yo_cosf_det.argtypes = [C.c_float]
        L.yo_sinf_det.restype = C.c_float
        L.yo_sinf_det.argtypes = [C.c_float]
        L.yo_trig_mismatch_count.restype = C.c_long
        L.yo_trig_mismatch_count.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.yo_descriptor_distance.restype = C.c_int
        L.yo_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.yo_frame_create.restype = C.c_void_p
        L.yo_frame_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.yo_frame_destroy.argtypes = [C.c_void_p]
        L.yo_frame_keypoints_in_area.restype = C.c_int
        L.yo_frame_keypoints_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.yo_search_by_projection.restype = C.c_int
        L.yo_search_by_projection.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.yo_search_by_bow.restype = C.c_int
        L.yo_search_by_bow.argtypes = [C.c_int] + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] * 2 + [C.c_float, C.c_int, C.c_void_p]
        L.yo_three_maxima.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.yo_rot_bin.restype = C.c_int
        L.yo_rot_bin.argtypes = [C.c_float, C.c_float]
        L.yo_ba_solve.restype = C.c_int
        L.yo_ba_solve.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 12 + [C.c_int, C.c_void_p]
        L.yo_ba_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.yo_ba_jacobians.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.yo_ba_pose_oplus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.yo_ba_chol_solve.restype = C.c_int
        L.yo_ba_chol_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.yo_ba_huber.argtypes = [C.c_double, C.c_double, C.c_void_p]
        L.yo_distinctive_descriptor.restype = C.c_int
        L.yo_distinctive_descriptor.argtypes = [C.c_void_p, C.c_int]
        L.yo_bow_transform.restype = C.c_int
        L.yo_bow_transform.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.yo_stereo_matches.restype = C.c_int
        L.yo_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + \
            [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbExtractorOracle:
    """Mirror of YDORBSLAM::OrbExtractor (orbExtractor.hpp:31-74) on numpy arrays."""

    def __init__(self, n_features=1000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.n_levels = n_levels
        self.n_features = n_features
        self.h = self.L.yo_extractor_create(n_features, scale_factor, n_levels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.yo_extractor_destroy(self.h)
            self.h = None

    def set_libm_trig(self, on):
        self.L.yo_extractor_set_libm_trig(self.h, int(on))

    def set_stale_pyramid(self, on):
        self.L.yo_extractor_set_stale_pyramid(self.h, int(on))

    def tables(self):
        n = self.n_levels
        sf, isf, sf2, isf2 = (np.zeros(n, np.float32) for _ in range(4))
        per = np.zeros(n, np.int32)
        maxx = np.zeros(28, np.int32)
        self.L.yo_extractor_tables(self.h, _p(sf), _p(isf), _p(sf2), _p(isf2), _p(per), _p(maxx))
        return dict(scale=sf, inv_scale=isf, scale2=sf2, inv_scale2=isf2, per_level=per, max_x=maxx)

    def extract(self, img, cap=None):
        """extractAndCompute (orbExtractor.cpp:355): returns (keypoints[KP_DTYPE], descriptors[N,32] u8)."""
        img = np.ascontiguousarray(img, np.uint8)
        cap = cap or 4 * self.n_features + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.yo_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            raise RuntimeError("oracle: capacity exceeded")
        return kps[:n].copy(), desc[:n].copy()

    def level_dims(self, l):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        self.L.yo_level_dims(self.h, l, C.byref(w), C.byref(h), C.byref(s))
        return w.value, h.value, s.value

    def level_padded(self, l):
        w, h, s = self.level_dims(l)
        ptr = self.L.yo_level_padded(self.h, l)
        buf = (C.c_uint8 * ((h + 38) * s)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h + 38, s).copy()

    def level_blurred(self, l):
        w, h, _ = self.level_dims(l)
        ptr = self.L.yo_level_blurred(self.h, l)
        if not ptr:
            return None
        buf = (C.c_uint8 * (h * w)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h, w).copy()

    def level_candidates(self, l):
        n = self.L.yo_level_candidates(self.h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.yo_level_candidates(self.h, l, _p(out), n)
        return out[:n]

    def level_keypoints(self, l):
        n = self.L.yo_level_keypoints(self.h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.yo_level_keypoints(self.h, l, _p(out), n)
        return out[:n]


def resize_linear_u8(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().yo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def fast9_16(img, thr, nms=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, KP_DTYPE)
    n = lib().yo_fast9_16(_p(img), img.strides[0], img.shape[1], img.shape[0], thr, int(nms), _p(out), cap)
    return out[:n].copy()


def corner_score16(d25, thr):
    d = np.ascontiguousarray(d25, np.int32)
    return lib().yo_corner_score16(_p(d), thr)


def gaussian_blur_7x7_s2(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().yo_gaussian_blur_7x7_s2(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), out.strides[0])
    return out


def gauss_kernel_fixed(n, sigma, bits):
    out = np.zeros(n, np.int32)
    lib().yo_gauss_kernel_fixed(n, sigma, bits, _p(out))
    return out


def fast_atan2(y, x):
    return lib().yo_fast_atan2(float(y), float(x))


# ---------------------------------------------------------------------------------------------
# matcher oracle (oracle/matcher_oracle.cpp)
# ---------------------------------------------------------------------------------------------
QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                        ("ur", "<f4"), ("rs", "<f4"), ("angle", "<f4"), ("level", "<i4"), ("flags", "<i4")])


MATCH2_DTYPE = np.dtype([("best_dist", "<i4"), ("best_idx", "<i4"), ("second_dist", "<i4"), ("second_idx", "<i4"), ("best_rank", "<i4"),
                         ("second_rank", "<i4")])


def hamming_topk(q, t, cand_offsets=None, cand_idx=None):
    """Best / second-best chain of orbMatcher.cpp:39-52 over each query's candidate list (all targets when no lists are given)."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros(len(q), MATCH2_DTYPE)
    co = None if cand_offsets is None else np.ascontiguousarray(cand_offsets, np.int32)
    ci = None if cand_idx is None else np.ascontiguousarray(cand_idx, np.int32)
    f = lib().yo_hamming_topk
    f.restype = None
    f.argtypes = [C.c_void_p] + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3
    f(_p(q), len(q), _p(t), len(t), None if co is None else _p(co), None if ci is None or len(ci) == 0 else _p(ci), _p(out))
    return out


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().yo_descriptor_distance(_p(a), _p(b))


class FrameOracle:
    """POD stand-in for YDORBSLAM::Frame: keypoints, descriptors, right x, bounds, 64x48 grid (frame.cpp:249-264)."""

    def __init__(self, kps, desc, bounds, right_x=None):
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.right_x = None if right_x is None else np.ascontiguousarray(right_x, np.float32)
        self.n = len(self.kps)
        self.h = lib().yo_frame_create(_p(self.kps), self.n, _p(self.desc), None if self.right_x is None else _p(self.right_x), *[float(b) for b in bounds])

    def __del__(self):
        if getattr(self, "h", None):
            lib().yo_frame_destroy(self.h)
            self.h = None

    def keypoints_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(self.n, 1), np.int32)
        n = lib().yo_frame_keypoints_in_area(self.h, x, y, r, min_level, max_level, _p(out), len(out))
        return out[:n].copy()

    def fuse_search(self, queries, qdesc, inv_sigma2, max_dist=50):
        """Search half of OrbMatcher::fuseByProjection (orbMatcher.cpp:682-745): best[q] = keyframe feature index or -1."""
        q = np.ascontiguousarray(queries, QUERY_DTYPE); d = np.ascontiguousarray(qdesc, np.uint8)
        s2 = np.ascontiguousarray(inv_sigma2, np.float32)
        best = np.full(len(q), -1, np.int32)
        n = lib().yo_fuse_search(self.h, _p(q), _p(d), len(q), _p(s2), int(max_dist), _p(best))
        return n, best

    def search_by_projection(self, mode, queries, qdesc, ratio, check_orientation, taken=None, assigned=None, orb_dist=0):
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        taken = np.zeros(self.n, np.uint8) if taken is None else np.ascontiguousarray(taken, np.uint8).copy()
        assigned = np.full(self.n, -1, np.int32) if assigned is None else np.ascontiguousarray(assigned, np.int32).copy()
        n = lib().yo_search_by_projection(self.h, mode, _p(queries), _p(qdesc), len(queries), ratio, orb_dist, int(check_orientation), _p(taken), _p(assigned))
        return n, assigned, taken


def search_by_bow(mode, kps_a, desc_a, valid_a, fv_a, kps_b, desc_b, valid_b, fv_b, ratio, check_orientation):
    """fv_* = (node_ids u32 ascending, node_start i32, feat i32)."""
    ka = np.ascontiguousarray(kps_a, KP_DTYPE); kb = np.ascontiguousarray(kps_b, KP_DTYPE)
    da = np.ascontiguousarray(desc_a, np.uint8); db = np.ascontiguousarray(desc_b, np.uint8)
    va = np.ascontiguousarray(valid_a, np.uint8)
    vb = np.ones(len(kb), np.uint8) if valid_b is None else np.ascontiguousarray(valid_b, np.uint8)
    ia, sa, fa = (np.ascontiguousarray(fv_a[0], np.uint32), np.ascontiguousarray(fv_a[1], np.int32), np.ascontiguousarray(fv_a[2], np.int32))
    ib, sb, fb = (np.ascontiguousarray(fv_b[0], np.uint32), np.ascontiguousarray(fv_b[1], np.int32), np.ascontiguousarray(fv_b[2], np.int32))
    out = np.full(len(kb) if mode == 3 else len(ka), -1, np.int32)
    n = lib().yo_search_by_bow(mode, _p(ka), _p(da), len(ka), _p(va), _p(ia), _p(sa), len(ia), _p(fa), _p(kb), _p(db), len(kb), _p(vb), _p(ib), _p(sb), len(ib), _p(fb), ratio, int(check_orientation), _p(out))
    return n, out


def search_for_triangulation(kps_a, desc_a, has_mp_a, right_a, fv_a, kps_b, desc_b, has_mp_b, right_b, fv_b, F, epipole, sf_b, sf2_b,
                             stereo_only, check_orientation):
    """OrbMatcher::searchForTriangulation (orbMatcher.cpp:463-565).  F: 3x3 float32 (_fMatrix_first2second), epipole (x, y) float32,
    sf_b / sf2_b: the second keyframe's scale factors / their squares.  Returns (matchNum, out[first idx] = second idx or -1)."""
    ka = np.ascontiguousarray(kps_a, KP_DTYPE); kb = np.ascontiguousarray(kps_b, KP_DTYPE)
    da = np.ascontiguousarray(desc_a, np.uint8); db = np.ascontiguousarray(desc_b, np.uint8)
    ma = np.ascontiguousarray(has_mp_a, np.uint8); mb = np.ascontiguousarray(has_mp_b, np.uint8)
    ra = np.ascontiguousarray(right_a, np.float32); rb = np.ascontiguousarray(right_b, np.float32)
    ia, sa, fa = (np.ascontiguousarray(fv_a[0], np.uint32), np.ascontiguousarray(fv_a[1], np.int32), np.ascontiguousarray(fv_a[2], np.int32))
    ib, sb, fb = (np.ascontiguousarray(fv_b[0], np.uint32), np.ascontiguousarray(fv_b[1], np.int32), np.ascontiguousarray(fv_b[2], np.int32))
    Fm = np.ascontiguousarray(F, np.float32).reshape(9)
    s1 = np.ascontiguousarray(sf_b, np.float32); s2 = np.ascontiguousarray(sf2_b, np.float32)
    out = np.full(len(ka), -1, np.int32)
    L = lib()
    L.yo_search_for_triangulation.restype = C.c_int
    n = L.yo_search_for_triangulation(_p(ka), _p(da), len(ka), _p(ma), _p(ra), _p(ia), _p(sa), len(ia), _p(fa), _p(kb), _p(db), len(kb), _p(mb),
                                      _p(rb), _p(ib), _p(sb), len(ib), _p(fb), _p(Fm), C.c_float(float(epipole[0])), C.c_float(float(epipole[1])),
                                      _p(s1), _p(s2), int(stereo_only), int(check_orientation), _p(out))
    return n, out


# ---------------------------------------------------------------------------------------------
# local-BA oracle (oracle/ba_oracle.cpp)
# ---------------------------------------------------------------------------------------------
BA_OPTIONS_DTYPE = np.dtype([("iters1", "<i4"), ("iters2", "<i4"), ("chi2_mono", "<f8"), ("chi2_stereo", "<f8"),
                             ("delta_mono", "<f8"), ("delta_stereo", "<f8"), ("max_trials", "<i4"), ("flags", "<i4")])   # flags: 1 single stage (bundleAdjust), 2 no Huber


def ba_default_options(iters1=5, iters2=10):
    o = np.zeros(1, BA_OPTIONS_DTYPE)
    o["iters1"], o["iters2"] = iters1, iters2
    o["chi2_mono"], o["chi2_stereo"] = 5.991, 7.815                      # optimizer.cpp:296,306
    o["delta_mono"] = float(np.float32(np.sqrt(5.991)))                  # `const float monoDelta = sqrt(5.991)`, :223
    o["delta_stereo"] = float(np.float32(np.sqrt(7.815)))
    o["max_trials"] = 10                                                 # levenberg.cpp:50
    return o


def ba_global_options(iters=5, robust=True):
    """Optimizer::bundleAdjust / globalBundleAdjust (optimizer.cpp:7-137): one optimize(iters), Huber deltas sqrt(5.99) / sqrt(7.815)."""
    o = ba_default_options(iters, 0)
    o["delta_mono"] = float(np.float32(np.sqrt(5.99)))                   # `const float monoDelta = sqrt(5.99)`, :37
    o["flags"] = 1 | (0 if robust else 2)
    return o


def ba_solve(prob, options=None, stop=None):
    """prob: dict as ydorbslam_amd.synth.synth_ba_problem.  Returns dict(poses, points, outlier, log, trials)."""
    o = ba_default_options() if options is None else options
    poses = np.ascontiguousarray(prob["poses"], np.float64).copy()
    points = np.ascontiguousarray(prob["points"], np.float64).copy()
    fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
    ep = np.ascontiguousarray(prob["edge_pose"], np.int32); eq = np.ascontiguousarray(prob["edge_point"], np.int32)
    meas = np.ascontiguousarray(prob["meas"], np.float64); info = np.ascontiguousarray(prob["info"], np.float64)
    cam = np.ascontiguousarray(prob["camera"], np.float64)
    E = len(ep)
    outlier = np.zeros(E, np.uint8)
    log = np.zeros((64, 4), np.float64)
    nlog = C.c_int(0)
    stop_p = None if stop is None else _p(stop)
    trials = lib().yo_ba_solve(len(poses), len(points), E, _p(poses), _p(fixed), _p(points), _p(ep), _p(eq), _p(meas), _p(info), _p(cam),
                               stop_p, _p(o), _p(outlier), _p(log), 64, C.byref(nlog))
    return dict(poses=poses, points=points, outlier=outlier, log=log[:nlog.value].copy(), trials=trials)


def pose_optimize(prob):
    """Optimizer::optimizePose restated (oracle/ba_oracle.cpp::yo_pose_optimize).  prob: ydorbslam_amd.synth.synth_pose_problem dict.
    Returns dict(pose, outlier, inliers, chi2 (per episode), trials)."""
    pose = np.ascontiguousarray(prob["pose"], np.float64).copy()
    X = np.ascontiguousarray(prob["points"], np.float64); z = np.ascontiguousarray(prob["meas"], np.float64)
    w = np.ascontiguousarray(prob["info"], np.float64); cam = np.ascontiguousarray(prob["camera"], np.float64)
    E = len(w)
    outlier = np.zeros(max(E, 1), np.uint8)
    chi = np.zeros(4, np.float64)
    trials = C.c_int(0)
    L = lib()
    L.yo_pose_optimize.restype = C.c_int
    n = L.yo_pose_optimize(_p(pose), E, _p(X), _p(z), _p(w), _p(cam), _p(outlier), _p(chi), C.byref(trials))
    return dict(pose=pose, outlier=outlier[:E], inliers=n, chi2=chi, trials=trials.value)


def stereo_matches(kps_l, desc_l, kps_r, desc_r, levels_l, levels_r, scale, inv_scale, bf, b, index_by_keypoint=False):
    """Frame::computeStereoMatches (frame.cpp:362-477) for one pair.  levels_*: list of 2-D uint8 arrays (the level ROIs).
    Returns (right_x, depth, n_kept, status)."""
    L = lib()
    kl = np.ascontiguousarray(kps_l, KP_DTYPE); kr = np.ascontiguousarray(kps_r, KP_DTYPE)
    dl = np.ascontiguousarray(desc_l, np.uint8); dr = np.ascontiguousarray(desc_r, np.uint8)
    ll = [np.ascontiguousarray(a, np.uint8) for a in levels_l]; lr = [np.ascontiguousarray(a, np.uint8) for a in levels_r]
    nlv = len(ll)
    assert len(lr) == nlv and all(a.shape == c.shape for a, c in zip(ll, lr))
    pl = (C.c_void_p * nlv)(*[a.ctypes.data for a in ll]); pr = (C.c_void_p * nlv)(*[a.ctypes.data for a in lr])
    w = np.array([a.shape[1] for a in ll], np.int32); h = np.array([a.shape[0] for a in ll], np.int32)
    sl = np.array([a.strides[0] for a in ll], np.int32); sr = np.array([a.strides[0] for a in lr], np.int32)
    sc = np.ascontiguousarray(scale, np.float32); isc = np.ascontiguousarray(inv_scale, np.float32)
    rx = np.zeros(max(len(kl), 1), np.float32); depth = np.zeros(max(len(kl), 1), np.float32)
    st = C.c_int(0)
    kept = L.yo_stereo_matches(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), C.cast(pl, C.c_void_p), C.cast(pr, C.c_void_p), _p(w), _p(h),
                               _p(sl), _p(sr), nlv, _p(sc), _p(isc), float(bf), float(b), 1 if index_by_keypoint else 0, _p(rx), _p(depth),
                               C.byref(st))
    return rx[:len(kl)], depth[:len(kl)], kept, st.value


def distinctive_descriptor(desc):
    """MapPoint::computeDistinctiveDescriptors (mapPoint.cpp:191-213): index of the descriptor with the least median distance."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    return lib().yo_distinctive_descriptor(_p(d), len(d))


def bow_transform(tree, desc, levelsup=4, weighting=0, norm=1):
    """DBoW3::Vocabulary::transform (Vocabulary.cpp:752-824) for one descriptor set -> (bow_word, bow_value, fv_node, fv_start, fv_feat, status)."""
    L = lib()
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(d)
    cb = np.ascontiguousarray(tree["child_begin"], np.int32); ci = np.ascontiguousarray(tree["child_ids"], np.int32)
    nd = np.ascontiguousarray(tree["node_desc"], np.uint8); nw = np.ascontiguousarray(tree["node_weight"], np.float64)
    wd = np.ascontiguousarray(tree["node_word"], np.int32)
    bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64); fn = np.zeros(max(n, 1), np.int32)
    fs = np.zeros(n + 2, np.int32); ff = np.zeros(max(n, 1), np.int32)
    k1, k2 = C.c_int32(0), C.c_int32(0)
    st = L.yo_bow_transform(int(tree["levels"]), _p(cb), _p(ci), _p(nd), _p(nw), _p(wd), _p(d), n, int(levelsup), int(weighting), int(norm),
                            _p(bw), _p(bv), C.byref(k1), _p(fn), _p(fs), _p(ff), C.byref(k2))
    return bw[:k1.value], bv[:k1.value], fn[:k2.value], fs[:k2.value + 1], ff[:fs[k2.value]], st
