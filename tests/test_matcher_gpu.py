"""GPU parity: HIP matcher (C ABI) vs the CPU oracle restatement of orbMatcher.cpp.  Bar: identical integers."""
import numpy as np
import pytest

from helpers import bow_nodes, feature_vector, projection_queries, shifted_pair

pytestmark = pytest.mark.gpu


# BASELINE.json configs 2, 3 and 4: the grid bounds, the number of keypoints per cell and the index ranges differ with the size
SIZES = [(640, 480, 1000), (1241, 376, 2000), (752, 480, 1000)]


@pytest.fixture(scope="module", params=SIZES, ids=["C2-640x480", "C3-1241x376-N2000", "C4-752x480"])
def scene(oracle_lib, request):
    from oracle.orb_oracle import OrbExtractorOracle
    w, h, nf = request.param
    ex = OrbExtractorOracle(nf)
    out = []
    for idx, (dx, dy) in enumerate([(5, -3), (-7, 4)]):
        a, b = shifted_pair(w, h, 50 + idx, dx, dy)
        ka, da = ex.extract(a)
        kb, db = ex.extract(b)
        out.append(dict(ka=ka, da=da, kb=kb, db=db, dx=dx, dy=dy, sf=ex.tables()["scale"], w=w, h=h,
                        bounds=(0.0, float(w), 0.0, float(h))))
    return out


def test_descriptor_distance(oracle_lib, scene):
    import ydorbslam_amd as y
    s = scene[0]
    n = min(len(s["ka"]), len(s["kb"]))
    m = y.OrbMatcher()
    gpu = m.descriptor_distance_rows(s["da"][:n], s["db"][:n])
    ref = np.unpackbits(s["da"][:n] ^ s["db"][:n], axis=1).sum(axis=1)
    assert np.array_equal(gpu, ref)
    for i in range(0, n, 97):
        assert y.OrbMatcher.descriptor_distance(s["da"][i], s["db"][i]) == ref[i] == oracle_lib.descriptor_distance(s["da"][i], s["db"][i])


def test_keypoints_in_area(oracle_lib, scene):
    import ydorbslam_amd as y
    s = scene[0]
    bounds = s["bounds"]
    fo = oracle_lib.FrameOracle(s["kb"], s["db"], bounds)
    fg = y.FrameView(s["kb"], s["db"], bounds)
    m = y.OrbMatcher()
    rng = np.random.default_rng(3)
    total = 0
    for _ in range(60):
        x, yy, r = float(rng.uniform(-20, s["w"] + 20)), float(rng.uniform(-20, s["h"] + 20)), float(rng.uniform(2, 120))
        lo, hi = [(-1, -1), (0, 3), (2, -1), (1, 2)][int(rng.integers(0, 4))]
        ref = fo.keypoints_in_area(np.float32(x), np.float32(yy), np.float32(r), lo, hi)
        got = m.keypoints_in_area(fg, np.float32(x), np.float32(yy), np.float32(r), lo, hi)
        assert np.array_equal(ref, got)
        total += len(ref)
    assert total > 200


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("stereo", [False, True])
def test_search_by_projection(oracle_lib, scene, mode, stereo):
    import ydorbslam_amd as y
    for si, s in enumerate(scene):
        bounds = s["bounds"]
        rng = np.random.default_rng(100 * mode + si)
        right = None
        if stereo:
            right = np.where(rng.random(len(s["kb"])) > 0.3, s["kb"]["x"] - 20 + rng.normal(0, 6, len(s["kb"])), -1).astype(np.float32)
        for th, ratio, check in [(15, 0.9, True), (3, 0.8, False), (7, 0.6, True)]:
            q = projection_queries(s["ka"], s["sf"], s["dx"], s["dy"], th, mode, seed=11 * mode + si, stereo=stereo)
            taken0 = (rng.random(len(s["kb"])) < 0.1).astype(np.uint8)
            assigned0 = np.where(taken0 > 0, 100000 + np.arange(len(taken0)), -1).astype(np.int32)
            fo = oracle_lib.FrameOracle(s["kb"], s["db"], bounds, right)
            n_ref, a_ref, t_ref = fo.search_by_projection(mode, q, s["da"], ratio, check, taken0, assigned0, orb_dist=64)
            m = y.OrbMatcher(ratio, check)
            n_gpu, a_gpu, t_gpu = m.search_by_projection(mode, y.FrameView(s["kb"], s["db"], bounds, right), q, s["da"], taken0, assigned0, orb_dist=64)
            assert n_gpu == n_ref
            assert np.array_equal(a_gpu, a_ref)
            assert np.array_equal(t_gpu, t_ref)
            if th == 15 and not stereo:
                assert n_ref > (50 if mode < 2 else 0)  # real matches exist (mode 2 keeps dist<=64; the |dx|>r test of frame.cpp:353 rarely offers true pairs)


@pytest.mark.parametrize("mode", [3, 4])
def test_search_by_bow(oracle_lib, scene, mode):
    import ydorbslam_amd as y
    for si, s in enumerate(scene):
        rng = np.random.default_rng(40 + si)
        fa, fb = feature_vector(bow_nodes(s["da"])), feature_vector(bow_nodes(s["db"]))
        va = (rng.random(len(s["ka"])) > 0.15).astype(np.uint8)
        vb = (rng.random(len(s["kb"])) > 0.15).astype(np.uint8)
        for ratio, check in [(0.7, True), (0.9, False), (0.75, True)]:
            n_ref, o_ref = oracle_lib.search_by_bow(mode, s["ka"], s["da"], va, fa, s["kb"], s["db"], vb, fb, ratio, check)
            m = y.OrbMatcher(ratio, check)
            n_gpu, o_gpu = m.search_by_bow(mode, s["ka"], s["da"], va, y.FeatureVector(*fa), s["kb"], s["db"], vb if mode == 4 else None,
                                           y.FeatureVector(*fb))
            assert n_gpu == n_ref
            assert np.array_equal(o_gpu, o_ref)
        assert n_ref > 20


@pytest.mark.parametrize("W,H,NF", SIZES)
def test_consecutive_device_matches_host_call(oracle_lib, W, H, NF):
    """The device-resident streaming search equals per-pair mode-1 searches built on the host (and so the oracle)."""
    import torch
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_frame
    F = 4
    base = synth_frame(W + 32, H + 32, 60)
    imgs = np.stack([np.ascontiguousarray(base[16 + 2 * i:16 + 2 * i + H, 16 + 3 * i:16 + 3 * i + W]) for i in range(F)])
    ex = y.OrbExtractor(NF, max_batch=F)
    cap = ex.max_keypoints
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(imgs).to(dev)
    d_kps = torch.zeros((F, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(F, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img.data_ptr(), W, H, W, W * H, F, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    ex.synchronize()
    m = y.OrbMatcher(0.9, True)
    d_assigned = torch.zeros((F - 1, cap), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(F - 1, dtype=torch.int32, device=dev)
    sf = ex.tables()["scale"]
    m.match_consecutive_device(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, F, W, H, 15.0, sf, d_assigned.data_ptr(), d_counts.data_ptr())
    m.synchronize()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(F, cap, 28).view(y.KP_DTYPE).reshape(F, cap)
    desc = d_desc.cpu().numpy()
    bounds = (0.0, float(W), 0.0, float(H))

    def check(pairs, assigned, counts, affine=None, floor=20):
        for c, (qf, tf) in enumerate(pairs):
            ka, kb = kps[qf, :n[qf]], kps[tf, :n[tf]]
            q = np.zeros(n[qf], y.QUERY_DTYPE)
            if affine is None:
                q["u"], q["v"] = ka["x"], ka["y"]
            else:   # single IEEE float operations in the kernel's order: (a0*x + a1*y) + a2
                A = affine[c].astype(np.float32)
                q["u"] = (A[0] * ka["x"] + A[1] * ka["y"]) + A[2]
                q["v"] = (A[3] * ka["x"] + A[4] * ka["y"]) + A[5]
            q["r"] = (np.float32(15.0) * sf[ka["octave"]]).astype(np.float32)
            q["min_level"], q["max_level"] = ka["octave"] - 1, ka["octave"] + 1
            q["angle"], q["level"] = ka["angle"], ka["octave"]
            q["flags"] = np.where((q["u"] >= 0) & (q["u"] < W) & (q["v"] >= 0) & (q["v"] < H), 3, 0)
            fo = oracle_lib.FrameOracle(kb, desc[tf, :n[tf]], bounds)
            n_ref, a_ref, _ = fo.search_by_projection(1, q, desc[qf, :n[qf]], 0.9, True)
            assert counts[c] == n_ref
            assert np.array_equal(assigned[c, :n[tf]], a_ref)
            assert n_ref > floor  # (few survive: frame.cpp:353 offers only |dx| > r candidates, then the 3-bin histogram cull)

    check([(f, f + 1) for f in range(F - 1)], d_assigned.cpu().numpy(), d_counts.cpu().numpy())
    # explicit pair list over two frame sets (the multi-GPU form: targets = an all-gathered set): any order, a frame used twice,
    # backward pairs, and a per-pair position prediction
    pairs = [(2, 0), (0, 3), (3, 1), (1, 2), (0, 1)]
    rng = np.random.default_rng(5)
    aff = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (len(pairs), 1))
    aff[:, 2] = rng.uniform(-6, 6, len(pairs)); aff[:, 5] = rng.uniform(-6, 6, len(pairs))
    aff[:, 1] = rng.uniform(-0.02, 0.02, len(pairs)); aff[:, 3] = -aff[:, 1]
    d_aff = torch.from_numpy(aff).to(dev)
    d_assigned2 = torch.zeros((len(pairs), cap), dtype=torch.int32, device=dev)
    d_counts2 = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    fs = (d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), F, cap)
    m.match_pairs_device(fs, fs, pairs, W, H, 15.0, sf, d_assigned2.data_ptr(), d_counts2.data_ptr(), d_aff.data_ptr())
    m.synchronize()
    check(pairs, d_assigned2.cpu().numpy(), d_counts2.cpu().numpy(), aff, floor=0)
    # queries and targets in different buffers (a copy of the set): same answers
    t_kps, t_desc, t_n = d_kps.clone(), d_desc.clone(), d_n.clone()
    d_assigned3 = torch.zeros_like(d_assigned2); d_counts3 = torch.zeros_like(d_counts2)
    m.match_pairs_device(fs, (t_kps.data_ptr(), t_desc.data_ptr(), t_n.data_ptr(), F, cap), pairs, W, H, 15.0, sf, d_assigned3.data_ptr(),
                         d_counts3.data_ptr(), d_aff.data_ptr())
    m.synchronize()
    assert torch.equal(d_assigned2, d_assigned3) and torch.equal(d_counts2, d_counts3)


@pytest.mark.parametrize("stereo_only", [False, True])
def test_search_for_triangulation(oracle_lib, scene, stereo_only):
    """OrbMatcher::searchForTriangulation (orbMatcher.cpp:463-565): eligibility flags, `<=` tie rule (last candidate
    wins), epipole-distance exemption, float epipolar test, rotation histogram — identical integers to the oracle."""
    import ydorbslam_amd as y
    total = 0
    for si, s in enumerate(scene):
        rng = np.random.default_rng(90 + si)
        ka, da, kb, db = s["ka"], s["da"], s["kb"], s["db"]
        fa, fb = feature_vector(bow_nodes(da, 4)), feature_vector(bow_nodes(db, 4))
        mpa = (rng.random(len(ka)) < 0.3).astype(np.uint8); mpb = (rng.random(len(kb)) < 0.3).astype(np.uint8)
        ra = np.where(rng.random(len(ka)) < 0.5, ka["x"] - 5, -1).astype(np.float32)
        rb = np.where(rng.random(len(kb)) < 0.5, kb["x"] - 5, -1).astype(np.float32)
        F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, -s["dy"]]], np.float32)      # epipolar "line" of a pure image shift: y' = y + dy
        sf = s["sf"].astype(np.float32); sf2 = (sf * sf).astype(np.float32)
        for epi, check in (((-900.0, 240.0), True), ((320.0, 240.0), False), ((100.5, 400.25), True)):
            n_ref, o_ref = oracle_lib.search_for_triangulation(ka, da, mpa, ra, fa, kb, db, mpb, rb, fb, F, epi, sf, sf2, stereo_only, check)
            m = y.OrbMatcher(0.6, check)
            n_gpu, o_gpu = m.search_for_triangulation(ka, da, mpa, ra, y.FeatureVector(*fa), kb, db, mpb, rb, y.FeatureVector(*fb), F, epi, sf, sf2,
                                                      stereo_only)
            assert n_gpu == n_ref
            assert np.array_equal(o_gpu, o_ref)
            total += n_ref
    assert total > 50


@pytest.mark.parametrize("stereo", [False, True])
def test_fuse_search(oracle_lib, scene, stereo):
    """Search half of OrbMatcher::fuseByProjection (orbMatcher.cpp:682-745): window without level check, explicit level
    window predicted-1 .. predicted, chi-square test on the float squared error, best distance <= 50 — identical indices to the oracle.
    (With the reference's th = 3 the `|dx| > r` test of frame.cpp:353 leaves no candidate that can pass the chi-square test; small
    radii are used here so that every branch is exercised.)"""
    import ydorbslam_amd as y
    total = 0
    for si, s in enumerate(scene):
        bounds = s["bounds"]
        rng = np.random.default_rng(300 + si)
        kb = s["kb"]
        right = np.where(rng.random(len(kb)) > 0.4, kb["x"] - 20 + rng.normal(0, 0.5, len(kb)), -1).astype(np.float32) if stereo else None
        sf = s["sf"].astype(np.float32)
        inv_s2 = (np.float32(1.0) / (sf * sf)).astype(np.float32)
        for th in (0.6, 1.0, 1.6):
            q = projection_queries(s["ka"], s["sf"], s["dx"], s["dy"], th, 1, seed=500 + si, stereo=stereo)
            q["min_level"], q["max_level"] = -1, -1
            q["level"] = np.clip(s["ka"]["octave"] + rng.integers(-1, 2, len(q)), 0, 7)
            q["r"] = (np.float32(th) * sf[q["level"]]).astype(np.float32)
            fo = oracle_lib.FrameOracle(kb, s["db"], bounds, right)
            n_ref, b_ref = fo.fuse_search(q, s["da"], inv_s2)
            m = y.OrbMatcher(0.6, True)
            n_gpu, b_gpu = m.fuse_search(y.FrameView(kb, s["db"], bounds, right), q, s["da"], inv_s2)
            assert n_gpu == n_ref and np.array_equal(b_gpu, b_ref)
            total += n_ref
    assert total > 20


def test_search_by_projection_in_sim_and_fuse_by_sim3(oracle_lib, scene):
    """Loop-closing variants: searchByProjectionInSim (orbMatcher.cpp:240-302) = mode 7 — window without level check,
    explicit level window, taken = already matched, <= 50, no histogram; fuseBySim3's search (:746-807) = the fuse search with the
    chi-square test disabled (zero inverse-sigma table).  Identical integers to the oracle."""
    import ydorbslam_amd as y
    total = 0
    for si, s in enumerate(scene):
        bounds = s["bounds"]
        rng = np.random.default_rng(700 + si)
        kb = s["kb"]
        for th in (0.8, 1.5, 4.0):
            q = projection_queries(s["ka"], s["sf"], s["dx"], s["dy"], th, 1, seed=800 + si)
            q["min_level"], q["max_level"] = -1, -1
            q["level"] = np.clip(s["ka"]["octave"] + rng.integers(-1, 2, len(q)), 0, 7)
            q["r"] = (np.float32(th) * s["sf"].astype(np.float32)[q["level"]]).astype(np.float32)
            taken0 = (rng.random(len(kb)) < 0.2).astype(np.uint8)
            assigned0 = np.where(taken0 > 0, 100000 + np.arange(len(taken0)), -1).astype(np.int32)
            fo = oracle_lib.FrameOracle(kb, s["db"], bounds, None)
            n_ref, a_ref, t_ref = fo.search_by_projection(7, q, s["da"], 0.0, False, taken0, assigned0)
            m = y.OrbMatcher(0.6, True)
            n_gpu, a_gpu, t_gpu = m.search_by_projection(7, y.FrameView(kb, s["db"], bounds, None), q, s["da"], taken0, assigned0)
            assert n_gpu == n_ref and np.array_equal(a_gpu, a_ref) and np.array_equal(t_gpu, t_ref)
            zero = np.zeros(8, np.float32)
            nf_ref, b_ref = fo.fuse_search(q, s["da"], zero)
            nf_gpu, b_gpu = m.fuse_search(y.FrameView(kb, s["db"], bounds, None), q, s["da"], zero)
            assert nf_gpu == nf_ref and np.array_equal(b_gpu, b_ref)
            nh_ref, h_ref = fo.fuse_search(q, s["da"], zero, 100)                  # one direction of searchBySim3 (:594-627): <= TH_HIGH
            nh_gpu, h_gpu = m.fuse_search(y.FrameView(kb, s["db"], bounds, None), q, s["da"], zero, 100)
            assert nh_gpu == nh_ref >= nf_ref and np.array_equal(h_gpu, h_ref)
            total += n_ref + nf_ref
    assert total > 50


def test_distinctive_descriptors_batch(oracle_lib):
    """MapPoint::computeDistinctiveDescriptors (mapPoint.cpp:169-218) for a batch of map points, including an empty one."""
    import ydorbslam_amd as y
    from test_oracle_matcher import _distinctive_groups
    groups = []
    for seed in range(6):
        groups += _distinctive_groups(seed)
    groups.insert(5, np.zeros((0, 32), np.uint8))
    got = y.OrbMatcher().distinctive_descriptors(groups)
    want = [oracle_lib.distinctive_descriptor(g) if len(g) else -1 for g in groups]
    assert list(got) == want
    assert len(y.OrbMatcher().distinctive_descriptors([])) == 0


def test_hamming_topk_brute_force(oracle_lib, scene):
    """ydorb_hamming_topk (north_star: brute-force 256-bit Hamming top-2 with wave min-reductions): the CSR-candidate form, the
    all-targets form and the device-resident all-pairs form all equal the oracle's best / second-best chain, ties included."""
    import torch
    import ydorbslam_amd as y
    s = scene[0]
    da, db = s["da"].copy(), s["db"].copy()
    db[5] = da[0]; db[9] = da[0]; db[700 % len(db)] = da[0]          # a three-way tie at distance 0 for query 0
    m = y.OrbMatcher()
    ref = oracle_lib.hamming_topk(da, db)
    got = m.hamming_topk(da, db)
    assert got.tobytes() == ref.tobytes()
    assert got[0]["best_idx"] == 5 and got[0]["second_idx"] == 9 and got[0]["best_dist"] == 0 == got[0]["second_dist"]
    rng = np.random.default_rng(3)
    sizes = rng.integers(0, 200, len(da)); sizes[3] = 0; sizes[4] = 1
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cand = rng.integers(0, len(db), int(offs[-1])).astype(np.int32)
    ref = oracle_lib.hamming_topk(da, db, offs, cand)
    got = m.hamming_topk(da, db, offs, cand)
    assert got.tobytes() == ref.tobytes()
    assert got[3].tolist() == (256, -1, 256, -1, -1, -1) and got[4]["second_idx"] == -1
    # device form: two (query frame, target frame) pairs in [pairs][cap][32] layout
    cap = max(len(da), len(db)) + 7
    dev = torch.device("cuda:0")
    qd = np.zeros((2, cap, 32), np.uint8); td = np.zeros((2, cap, 32), np.uint8)
    qd[0, :len(da)] = da; td[0, :len(db)] = db; qd[1, :len(db)] = db; td[1, :len(da)] = da
    nq = np.array([len(da), len(db)], np.int32); nt = np.array([len(db), len(da)], np.int32)
    d_q, d_t = torch.from_numpy(qd).to(dev), torch.from_numpy(td).to(dev)
    d_nq, d_nt = torch.from_numpy(nq).to(dev), torch.from_numpy(nt).to(dev)
    d_out = torch.zeros((2, cap, 6), dtype=torch.int32, device=dev)
    m.hamming_topk_device(d_q.data_ptr(), d_nq.data_ptr(), d_t.data_ptr(), d_nt.data_ptr(), cap, 2, d_out.data_ptr())
    m.synchronize()
    out = d_out.cpu().numpy()
    assert np.array_equal(out[0, :len(da)], oracle_lib.hamming_topk(da, db).view(np.int32).reshape(-1, 6))
    assert np.array_equal(out[1, :len(db)], oracle_lib.hamming_topk(db, da).view(np.int32).reshape(-1, 6))


def test_new_entry_points_on_empty_and_ragged_inputs(oracle_lib):
    """Edge cases of the round-2 entry points: no queries, no targets, empty candidate lists, frames without keypoints in a pair list,
    out-of-range pairs."""
    import torch
    import ydorbslam_amd as y
    m = y.OrbMatcher()
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (5, 32), dtype=np.uint8); t = rng.integers(0, 256, (7, 32), dtype=np.uint8)
    assert len(m.hamming_topk(np.zeros((0, 32), np.uint8), t)) == 0
    none = m.hamming_topk(q, np.zeros((0, 32), np.uint8))
    assert all(r.tolist() == (256, -1, 256, -1, -1, -1) for r in none)
    offs = np.array([0, 0, 3, 3, 4, 4], np.int32); cand = np.array([6, 0, 6, 2], np.int32)     # empty lists, a repeated candidate
    assert m.hamming_topk(q, t, offs, cand).tobytes() == oracle_lib.hamming_topk(q, t, offs, cand).tobytes()
    with pytest.raises(y.YdorbError):
        m.hamming_topk(q, t, np.array([1, 1, 1, 1, 1, 1], np.int32), cand)                      # offsets must start at 0
    dev = torch.device("cuda:0")
    cap, F = 64, 3
    kps = torch.zeros((F, cap, 7), dtype=torch.float32, device=dev); desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.tensor([0, 5, 0], dtype=torch.int32, device=dev)
    kps[1, :5, 0] = torch.arange(5, device=dev) * 20 + 30; kps[1, :5, 1] = 40
    asg = torch.full((2, cap), 7, dtype=torch.int32, device=dev); cnt = torch.full((2,), 9, dtype=torch.int32, device=dev)
    fs = (kps.data_ptr(), desc.data_ptr(), n.data_ptr(), F, cap)
    sf = np.power(np.float32(1.2), np.arange(8)).astype(np.float32)
    m.match_pairs_device(fs, fs, [(0, 1), (1, 2)], 640, 480, 15.0, sf, asg.data_ptr(), cnt.data_ptr())   # empty query frame; empty target frame
    m.synchronize()
    assert cnt.tolist() == [0, 0] and bool((asg == -1).all())
    with pytest.raises(y.YdorbError):
        m.match_pairs_device(fs, fs, [(0, 3)], 640, 480, 15.0, sf, asg.data_ptr(), cnt.data_ptr())


def test_back_to_back_calls_on_one_handle_without_synchronize():
    """Two device-resident searches issued back to back on ONE handle and ONE stream, no synchronize between them, different output
    buffers: the second call reuses the handle's scratch (queries, record pool, taken flags) behind the first call's resolve in stream
    order, so both outputs equal those of separately synchronized calls."""
    import torch
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_frame
    W, H, NF, F = 640, 480, 1000, 5
    base = synth_frame(W + 32, H + 32, 61)
    imgs = np.stack([np.ascontiguousarray(base[16 + 2 * i:16 + 2 * i + H, 16 + 3 * i:16 + 3 * i + W]) for i in range(F)])
    ex = y.OrbExtractor(NF, max_batch=F)
    cap = ex.max_keypoints
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(imgs).to(dev)
    d_kps = torch.zeros((F, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(F, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img.data_ptr(), W, H, W, W * H, F, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    ex.synchronize()
    sf = ex.tables()["scale"]
    main = torch.cuda.Stream(device=dev)
    m = y.OrbMatcher(0.9, True)
    ref_a = torch.zeros((F - 1, cap), dtype=torch.int32, device=dev)
    ref_c = torch.zeros(F - 1, dtype=torch.int32, device=dev)
    m.match_consecutive_device(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, F, W, H, 15.0, sf, ref_a.data_ptr(), ref_c.data_ptr(), stream=main.cuda_stream)
    m.synchronize()
    outs = [(torch.zeros_like(ref_a), torch.zeros_like(ref_c)) for _ in range(3)]
    for a, c in outs:
        m.match_consecutive_device(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, F, W, H, 15.0, sf, a.data_ptr(), c.data_ptr(), stream=main.cuda_stream)
    m.synchronize()
    assert int(ref_c.sum()) > 100
    for a, c in outs:
        assert torch.equal(a, ref_a) and torch.equal(c, ref_c)


@pytest.mark.parametrize("world", [2, 4])
def test_match_pairs_on_the_gathered_layouts_of_every_rank(world):
    """The multi-GPU match step on one GPU: the records of a stream of world * F frames laid out as the all-gathered set (rank-major,
    global frame g = t * world + rank at slot rank * F + t), every rank's pair list from parallel.round_robin_pairs, distinct query /
    target slots per pair and only F of the world * F frames searched (the matcher builds grids for those alone).  The union over
    ranks must equal the consecutive matching of the stream in its natural order."""
    import torch
    import ydorbslam_amd as y
    from ydorbslam_amd.parallel import round_robin_pairs
    from ydorbslam_amd.synth import stream_plan, stream_render
    W, H, F = 640, 480, 3
    G = world * F
    plan = stream_plan(W, H, G, seed=3, segment=G)
    imgs, _ = stream_render(plan, range(G))
    ex = y.OrbExtractor(1000, max_batch=G)
    cap = ex.max_keypoints
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(imgs).to(dev)
    kps = torch.zeros((G, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((G, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(G, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img.data_ptr(), W, H, W, W * H, G, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr())
    ex.synchronize()
    sf = ex.tables()["scale"]
    aff = torch.from_numpy(np.ascontiguousarray(plan["predicted"], np.float32)).to(dev)
    m = y.OrbMatcher(0.9, True)
    ref_a = torch.zeros((G - 1, cap), dtype=torch.int32, device=dev)
    ref_c = torch.zeros(G - 1, dtype=torch.int32, device=dev)
    m.match_consecutive_device(kps.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, G, W, H, 15.0, sf, ref_a.data_ptr(), ref_c.data_ptr(), d_affine=aff.data_ptr())
    m.synchronize()
    assert int(ref_c.sum()) > 50
    slot_of = torch.tensor([(g % world) * F + g // world for g in range(G)], device=dev)       # natural index -> gathered slot
    g_kps, g_desc, g_n = torch.zeros_like(kps), torch.zeros_like(desc), torch.zeros_like(n)
    g_kps[slot_of], g_desc[slot_of], g_n[slot_of] = kps, desc, n
    gs = (g_kps.data_ptr(), g_desc.data_ptr(), g_n.data_ptr(), G, cap)
    seen = set()
    for rank in range(world):
        pairs, pred = round_robin_pairs(rank, world, F)
        a = torch.zeros((len(pairs), cap), dtype=torch.int32, device=dev)
        c = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
        d_aff = aff[torch.from_numpy(pred).to(dev)].contiguous()
        mr = y.OrbMatcher(0.9, True)
        mr.match_pairs_device(gs, gs, pairs, W, H, 15.0, sf, a.data_ptr(), c.data_ptr(), d_aff.data_ptr())
        mr.synchronize()
        for i, g in enumerate(pred.tolist()):
            assert g not in seen
            seen.add(g)
            assert int(c[i]) == int(ref_c[g]) and torch.equal(a[i], ref_a[g]), (rank, g)
    assert seen == set(range(G - 1))
