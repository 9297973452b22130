"""Synthetic matching scenes shared by the oracle (CPU) tests and the GPU parity tests."""
import numpy as np

from ydorbslam_amd.synth import synth_frame

QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                        ("ur", "<f4"), ("rs", "<f4"), ("angle", "<f4"), ("level", "<i4"), ("flags", "<i4")])


def shifted_pair(w, h, idx, dx, dy):
    """Two views of one synthetic scene: B is A moved by (dx, dy) px plus fresh sensor noise."""
    big = synth_frame(w + 32, h + 32, idx)
    rng = np.random.default_rng(7000 + idx)
    a = big[16:16 + h, 16:16 + w]
    b = big[16 - dy:16 - dy + h, 16 - dx:16 - dx + w].astype(np.int16) + rng.integers(-2, 3, (h, w))
    return np.ascontiguousarray(a), np.clip(b, 0, 255).astype(np.uint8)


def projection_queries(kps_a, scale_factors, dx, dy, th, mode, seed, stereo=False):
    """Queries as the adapter would build them from frame A's keypoints (orbMatcher.cpp:30-31, 94-101, 182-183)."""
    rng = np.random.default_rng(seed)
    n = len(kps_a)
    q = np.zeros(n, QUERY_DTYPE)
    q["u"] = (kps_a["x"] + np.float32(dx) + rng.normal(0, 1.5, n)).astype(np.float32)
    q["v"] = (kps_a["y"] + np.float32(dy) + rng.normal(0, 1.5, n)).astype(np.float32)
    oct_ = kps_a["octave"]
    sf = scale_factors[oct_]
    if mode == 0:  # radius = th * (2.5 | 4.0), window radius*sf[level], levels (level-1, level)
        fac = np.where(rng.random(n) > 0.5, np.float32(2.5), np.float32(4.0))
        rad = (np.float32(th) * fac).astype(np.float32)
        q["r"] = (rad * sf).astype(np.float32)
        q["min_level"] = oct_ - 1
        q["max_level"] = oct_
    else:
        q["r"] = (np.float32(th) * sf).astype(np.float32)
        sel = rng.integers(0, 3, n) if mode == 1 else np.full(n, 2)
        q["min_level"] = np.where(sel == 0, oct_, np.where(sel == 1, 0, oct_ - 1))
        q["max_level"] = np.where(sel == 0, -1, np.where(sel == 1, oct_, oct_ + 1))
    q["rs"] = q["r"]
    q["ur"] = q["u"] - np.float32(20.0) if stereo else 0
    q["angle"] = kps_a["angle"]
    q["level"] = oct_
    valid = rng.random(n) > 0.1
    obs = rng.random(n) > 0.2
    q["flags"] = valid.astype(np.int32) | (obs.astype(np.int32) << 1)
    return q


def bow_nodes(desc, bits=5):
    """Stand-in for DBoW3's vocabulary (the blob is absent): node id from the leading descriptor bits."""
    return (desc[:, 0].astype(np.uint32) >> (8 - bits)) * 7 + 3  # non-contiguous ascending ids


def feature_vector(nodes):
    ids = np.unique(nodes)
    order = np.argsort(nodes, kind="stable").astype(np.int32)
    counts = np.array([(nodes == i).sum() for i in ids], np.int64)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return ids.astype(np.uint32), start, order
