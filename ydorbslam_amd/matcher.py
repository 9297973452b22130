"""Python mirror of YDORBSLAM::OrbMatcher's search families (reference src/orbMatcher.hpp:24-66) over the C ABI."""
import ctypes as C
import numpy as np

from ._lib import YdBowSide, YdFeatureVector, YdFrameView, YdTriSide, YdorbError, check, lib
from .extractor import KP_DTYPE

QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                        ("ur", "<f4"), ("rs", "<f4"), ("angle", "<f4"), ("level", "<i4"), ("flags", "<i4")])
assert QUERY_DTYPE.itemsize == 40
MATCH2_DTYPE = np.dtype([("best_dist", "<i4"), ("best_idx", "<i4"), ("second_dist", "<i4"), ("second_idx", "<i4"), ("best_rank", "<i4"),
                         ("second_rank", "<i4")])

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # orbMatcher.cpp:7-9
FRAME_MAPPOINT, LAST_CURRENT, KEYFRAME_CURRENT, BOW_KEYFRAME_FRAME, BOW_TWO_KEYFRAMES = range(5)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FrameView:
    """The parts of YDORBSLAM::Frame the searches read (keypoints, descriptors, right x, image bounds)."""

    def __init__(self, kps, desc, bounds, right_x=None):
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.right_x = None if right_x is None else np.ascontiguousarray(right_x, np.float32)
        self.bounds = tuple(float(b) for b in bounds)  # min_x, max_x, min_y, max_y
        self.n = len(self.kps)

    def c(self):
        return YdFrameView(_p(self.kps), _p(self.desc), _p(self.right_x), self.n, *self.bounds)


class FeatureVector:
    """DBoW3::FeatureVector (std::map<node, vector<feature>>) as CSR arrays."""

    def __init__(self, node_ids, node_start, feat):
        self.node_ids = np.ascontiguousarray(node_ids, np.uint32)
        self.node_start = np.ascontiguousarray(node_start, np.int32)
        self.feat = np.ascontiguousarray(feat, np.int32)

    @staticmethod
    def from_nodes(node_of_feature):
        """Group features by node id; inside a node features keep ascending index (FeatureVector::addFeature appends)."""
        node_of_feature = np.asarray(node_of_feature)
        ids = np.unique(node_of_feature)
        order = np.argsort(node_of_feature, kind="stable")
        counts = np.array([(node_of_feature == i).sum() for i in ids], np.int64)
        start = np.concatenate([[0], np.cumsum(counts)])
        return FeatureVector(ids, start, order)

    def c(self):
        return YdFeatureVector(_p(self.node_ids), _p(self.node_start), _p(self.feat), len(self.node_ids))


class OrbMatcher:
    """OrbMatcher(ratio=0.6, checkOrientation=true) — orbMatcher.hpp:26."""

    def __init__(self, ratio=0.6, check_orientation=True, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.ydorb_matcher_create(device, C.byref(self._h)))
        self.ratio = float(ratio)
        self.check_orientation = bool(check_orientation)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ydorb_matcher_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def descriptor_distance(a, b):
        """static computeDescriptorsDistance (orbMatcher.cpp:11-23)."""
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return lib().ydorb_descriptor_distance(_p(a), _p(b))

    def descriptor_distance_rows(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros(len(a), np.int32)
        check(self._L.ydorb_descriptor_distance_rows(self._h, _p(a), _p(b), len(a), _p(out)))
        return out

    def keypoints_in_area(self, frame, x, y, r, min_level=-1, max_level=-1):
        """Frame::getKeyPointsInArea (frame.cpp:337-361)."""
        out = np.zeros(max(frame.n, 1), np.int32)
        n = C.c_int32(0)
        fv = frame.c()
        check(self._L.ydorb_frame_keypoints_in_area(self._h, C.byref(fv), x, y, r, min_level, max_level, _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def search_by_projection(self, mode, frame, queries, qdesc, taken=None, assigned=None, orb_dist=0):
        """modes 0..2 = the three searchByProjection* overloads.  Returns (n_matches, assigned, taken)."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        taken = np.zeros(frame.n, np.uint8) if taken is None else np.ascontiguousarray(taken, np.uint8).copy()
        assigned = np.full(frame.n, -1, np.int32) if assigned is None else np.ascontiguousarray(assigned, np.int32).copy()
        n = C.c_int32(0)
        fv = frame.c()
        check(self._L.ydorb_search_by_projection(self._h, mode, C.byref(fv), _p(queries), _p(qdesc), len(queries), self.ratio, orb_dist,
                                                 int(self.check_orientation), _p(taken), _p(assigned), C.byref(n)))
        return n.value, assigned, taken

    def search_by_bow(self, mode, kps_a, desc_a, valid_a, fv_a, kps_b, desc_b, valid_b, fv_b):
        """mode 3 = searchByBowInKeyFrameAndFrame, mode 4 = searchByBowInTwoKeyFrames.  Returns (n_matches, out)."""
        ka = np.ascontiguousarray(kps_a, KP_DTYPE); kb = np.ascontiguousarray(kps_b, KP_DTYPE)
        da = np.ascontiguousarray(desc_a, np.uint8); db = np.ascontiguousarray(desc_b, np.uint8)
        va = np.ascontiguousarray(valid_a, np.uint8)
        vb = None if valid_b is None else np.ascontiguousarray(valid_b, np.uint8)
        A = YdBowSide(_p(ka), _p(da), _p(va), len(ka), fv_a.c())
        B = YdBowSide(_p(kb), _p(db), _p(vb), len(kb), fv_b.c())
        out = np.full(len(kb) if mode == 3 else len(ka), -1, np.int32)
        n = C.c_int32(0)
        check(self._L.ydorb_search_by_bow(self._h, mode, C.byref(A), C.byref(B), self.ratio, int(self.check_orientation), _p(out), C.byref(n)))
        return n.value, out

    def fuse_search(self, frame, queries, qdesc, inv_sigma2, max_dist=50):
        """Search half of OrbMatcher::fuseByProjection (orbMatcher.cpp:682-745).  Returns (n_found, best[q] = feature index or -1)."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        s2 = np.ascontiguousarray(inv_sigma2, np.float32)
        best = np.full(max(len(queries), 1), -1, np.int32)
        n = C.c_int32(0)
        fv = frame.c()
        check(self._L.ydorb_window_search(self._h, C.byref(fv), _p(queries), _p(qdesc), len(queries), _p(s2), len(s2), int(max_dist), _p(best), C.byref(n)))
        return n.value, best[:len(queries)]

    def search_for_triangulation(self, kps_a, desc_a, has_mp_a, right_a, fv_a, kps_b, desc_b, has_mp_b, right_b, fv_b, F, epipole, sf_b, sf2_b,
                                 stereo_only=False):
        """OrbMatcher::searchForTriangulation (orbMatcher.cpp:463-565).  Returns (n_matches, out[first idx] = second idx or -1)."""
        ka = np.ascontiguousarray(kps_a, KP_DTYPE); kb = np.ascontiguousarray(kps_b, KP_DTYPE)
        da = np.ascontiguousarray(desc_a, np.uint8); db = np.ascontiguousarray(desc_b, np.uint8)
        ma = np.ascontiguousarray(has_mp_a, np.uint8); mb = np.ascontiguousarray(has_mp_b, np.uint8)
        ra = np.ascontiguousarray(right_a, np.float32); rb = np.ascontiguousarray(right_b, np.float32)
        Fm = np.ascontiguousarray(F, np.float32).reshape(9)
        s1 = np.ascontiguousarray(sf_b, np.float32); s2 = np.ascontiguousarray(sf2_b, np.float32)
        A = YdTriSide(_p(ka), _p(da), _p(ra), _p(ma), len(ka), fv_a.c())
        B = YdTriSide(_p(kb), _p(db), _p(rb), _p(mb), len(kb), fv_b.c())
        out = np.full(len(ka), -1, np.int32)
        n = C.c_int32(0)
        check(self._L.ydorb_search_for_triangulation(self._h, C.byref(A), C.byref(B), _p(Fm), float(epipole[0]), float(epipole[1]), _p(s1), _p(s2),
                                                     len(s1), int(stereo_only), int(self.check_orientation), _p(out), C.byref(n)))
        return n.value, out

    def distinctive_descriptors(self, groups):
        """MapPoint::computeDistinctiveDescriptors (mapPoint.cpp:169-218) for a list of [m_i, 32] uint8 arrays -> best index per group."""
        groups = [np.ascontiguousarray(g, np.uint8).reshape(-1, 32) for g in groups]
        offsets = np.zeros(len(groups) + 1, np.int32)
        offsets[1:] = np.cumsum([len(g) for g in groups])
        desc = np.concatenate(groups) if offsets[-1] else np.zeros((1, 32), np.uint8)
        best = np.zeros(max(len(groups), 1), np.int32)
        check(self._L.ydorb_distinctive_descriptors(self._h, _p(desc), _p(offsets), len(groups), _p(best)))
        return best[:len(groups)]

    def stereo_matches(self, left_ext, right_ext, kps_l, desc_l, n_l, kps_r, desc_r, n_r, bf, b, index_by_keypoint=False,
                       left_frames=(0, 1), right_frames=(0, 1)):
        """Frame::computeStereoMatches (frame.cpp:362-477) for a batch of rectified pairs.

        kps_*: [pairs, cap] KP_DTYPE, desc_*: [pairs, cap, 32], n_*: [pairs].  *_frames = (first_frame, frame_step) into the
        pyramids of the extractors' last call.  Returns (right_x[pairs, cap], depth[pairs, cap], n_kept[pairs], status[pairs])."""
        from ._lib import YdStereoSide, STEREO_INDEX_BY_KEYPOINT
        kl = np.ascontiguousarray(kps_l, KP_DTYPE); kr = np.ascontiguousarray(kps_r, KP_DTYPE)
        dl = np.ascontiguousarray(desc_l, np.uint8); dr = np.ascontiguousarray(desc_r, np.uint8)
        nl = np.ascontiguousarray(n_l, np.int32); nr = np.ascontiguousarray(n_r, np.int32)
        pairs, cap_l = kl.shape
        cap_r = kr.shape[1]
        assert kr.shape[0] == pairs and dl.shape == (pairs, cap_l, 32) and dr.shape == (pairs, cap_r, 32) and len(nl) == pairs == len(nr)
        A = YdStereoSide(left_ext._h, left_frames[0], left_frames[1], _p(kl), _p(dl), _p(nl), cap_l, 0)
        B = YdStereoSide(right_ext._h, right_frames[0], right_frames[1], _p(kr), _p(dr), _p(nr), cap_r, 0)
        rx = np.zeros((pairs, cap_l), np.float32); depth = np.zeros((pairs, cap_l), np.float32)
        kept = np.zeros(pairs, np.int32); status = np.zeros(pairs, np.int32)
        check(self._L.ydorb_stereo_matches(self._h, C.byref(A), C.byref(B), pairs, float(bf), float(b),
                                           STEREO_INDEX_BY_KEYPOINT if index_by_keypoint else 0, _p(rx), _p(depth), _p(kept), _p(status), None))
        return rx, depth, kept, status

    def stereo_matches_device(self, left_ext, right_ext, d_kps_l, d_desc_l, d_n_l, cap_l, d_kps_r, d_desc_r, d_n_r, cap_r, n_pairs, bf, b,
                              d_right_x, d_depth, d_kept=None, d_status=None, index_by_keypoint=False, left_frames=(0, 1),
                              right_frames=(0, 1), stream=None):
        """Device-resident form of stereo_matches (raw HBM addresses, asynchronous on `stream` or the matcher's stream)."""
        from ._lib import YdStereoSide, STEREO_INDEX_BY_KEYPOINT, STEREO_DEVICE_POINTERS
        A = YdStereoSide(left_ext._h, left_frames[0], left_frames[1], d_kps_l, d_desc_l, d_n_l, cap_l, 0)
        B = YdStereoSide(right_ext._h, right_frames[0], right_frames[1], d_kps_r, d_desc_r, d_n_r, cap_r, 0)
        check(self._L.ydorb_stereo_matches(self._h, C.byref(A), C.byref(B), n_pairs, float(bf), float(b),
                                           STEREO_DEVICE_POINTERS | (STEREO_INDEX_BY_KEYPOINT if index_by_keypoint else 0), d_right_x, d_depth,
                                           d_kept, d_status, stream))

    def match_consecutive_device(self, d_kps, d_desc, d_n, cap, n_frames, width, height, th, scale_factors, d_assigned, d_counts,
                                 d_affine=None, stream=None):
        sf = np.ascontiguousarray(scale_factors, np.float32)
        check(self._L.ydorb_match_consecutive_device(self._h, d_kps, d_desc, d_n, cap, n_frames, width, height, th, _p(sf), len(sf), d_affine,
                                                     int(self.check_orientation), d_assigned, d_counts, stream))

    def match_pairs_device(self, q_set, t_set, pairs, width, height, th, scale_factors, d_assigned, d_counts, d_affine=None, stream=None):
        """ydorb_match_pairs_device: q_set / t_set = (d_kps, d_desc, d_n, n_frames, cap) raw HBM addresses; pairs: [n_pairs, 2] host ints."""
        from ._lib import YdFrameSetDev
        sf = np.ascontiguousarray(scale_factors, np.float32)
        pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        Q, T = YdFrameSetDev(*q_set), YdFrameSetDev(*t_set)
        check(self._L.ydorb_match_pairs_device(self._h, C.byref(Q), C.byref(T), _p(pr), len(pr), width, height, th, _p(sf), len(sf), d_affine,
                                               int(self.check_orientation), d_assigned, d_counts, stream))

    def hamming_topk(self, q, t, cand_offsets=None, cand_idx=None):
        """ydorb_hamming_topk: brute-force top-2 of every query row against its candidate list (CSR) or, without lists, against all
        target rows.  Returns an array of MATCH2_DTYPE (best_dist, best_idx, second_dist, second_idx, best_rank, second_rank)."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros(len(q), MATCH2_DTYPE)
        co = None if cand_offsets is None else np.ascontiguousarray(cand_offsets, np.int32)
        ci = None if cand_idx is None else np.ascontiguousarray(cand_idx, np.int32)
        if ci is not None and len(ci) == 0:
            ci = np.zeros(1, np.int32)
        check(self._L.ydorb_hamming_topk(self._h, _p(q), len(q), _p(t), len(t), _p(co), _p(ci), _p(out)))
        return out

    def hamming_topk_device(self, d_qdesc, d_nq, d_tdesc, d_nt, cap, n_pairs, d_out, stream=None):
        check(self._L.ydorb_hamming_topk_device(self._h, d_qdesc, d_nq, d_tdesc, d_nt, cap, n_pairs, d_out, stream))

    def synchronize(self):
        check(self._L.ydorb_matcher_synchronize(self._h))

    def set_profiling(self, on=True):
        check(self._L.ydorb_matcher_set_profiling(self._h, int(on)))

    def stage_times(self):
        names = (C.c_char_p * 8)()
        ms = (C.c_float * 8)()
        n = C.c_int32(0)
        check(self._L.ydorb_matcher_stage_times(self._h, 8, names, ms, C.byref(n)))
        return {names[i].decode(): ms[i] for i in range(n.value)}
