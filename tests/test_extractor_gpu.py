"""GPU parity: HIP extractor (through the C ABI) vs the CPU oracle, stage by stage and end to end.

Bar: bit-exact for every byte / index / float bit (integer path; the float ops are single IEEE operations).
"""
import numpy as np
import pytest

from ydorbslam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu

CASES = [  # (w, h, n_features, frame index)
    (640, 480, 1000, 0),
    (640, 480, 1000, 7),
    (752, 480, 1000, 1),
    (1241, 376, 2000, 2),
    (321, 243, 500, 3),
    (131, 97, 300, 4),
]


def _pair(n_features, max_batch=1):
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    return y.OrbExtractor(n_features, 1.2, 8, 20, 7, max_batch=max_batch), OrbExtractorOracle(n_features, 1.2, 8, 20, 7)


def _same_kps(a, b):
    assert len(a) == len(b)
    for f in a.dtype.names:
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), "field %s differs" % f


@pytest.mark.parametrize("w,h,nf,idx", CASES)
def test_stages_and_output_bit_exact(oracle_lib, w, h, nf, idx):
    gpu, cpu = _pair(nf)
    img = synth_frame(w, h, idx)
    gk, gd = gpu.extract(img)
    ck, cd = cpu.extract(img)
    for l in range(8):
        cw, ch, _ = cpu.level_dims(l)
        gw, gh, _, _ = gpu.level_dims(l)
        assert (cw, ch) == (gw, gh)
        assert np.array_equal(gpu.read_level(l), cpu.level_padded(l)[:, :cw + 38]), "pyramid level %d" % l
        cb = cpu.level_blurred(l)
        if cb is not None:
            assert np.array_equal(gpu.debug_read(0, l), cb), "blurred level %d" % l
        gc, cc = gpu.debug_read(1, l), cpu.level_candidates(l)
        assert len(gc) == len(cc), "candidate count level %d: %d vs %d" % (l, len(gc), len(cc))
        for f in ("x", "y", "response"):
            assert np.array_equal(gc[f], cc[f]), "candidates level %d field %s" % (l, f)
        _same_kps(gpu.debug_read(2, l), cpu.level_keypoints(l))
    _same_kps(gk, ck)
    assert np.array_equal(gd, cd)
    quota = int(cpu.tables()["per_level"].sum())
    assert len(ck) <= quota
    if w >= 640:
        assert len(ck) >= 0.9 * quota  # synthetic frames must fill >= 90 % of the quotas


def test_batch_equals_single(oracle_lib):
    gpu, cpu = _pair(1000, max_batch=6)
    imgs = np.stack([synth_frame(640, 480, 20 + i) for i in range(6)])
    res = gpu.extract_batch(imgs)
    for f in range(6):
        ck, cd = cpu.extract(imgs[f])
        _same_kps(res[f][0], ck)
        assert np.array_equal(res[f][1], cd)


def test_repeat_call_uses_fresh_pyramid(oracle_lib):
    """Contract: first-call semantics (the reference re-extracts frame 1 on later calls, orbExtractor.cpp:612)."""
    gpu, cpu = _pair(1000)
    a, b = synth_frame(640, 480, 30), synth_frame(640, 480, 31)
    gpu.extract(a)
    gk, gd = gpu.extract(b)
    ck, cd = cpu.extract(b)
    _same_kps(gk, ck)
    assert np.array_equal(gd, cd)


def test_flat_and_empty_images(oracle_lib):
    gpu, cpu = _pair(1000)
    flat = np.full((480, 640), 128, np.uint8)
    gk, gd = gpu.extract(flat)
    assert len(gk) == 0 and gd.shape == (0, 32)
    gk, gd = gpu.extract(np.zeros((0, 0), np.uint8))
    assert len(gk) == 0


def test_strided_input(oracle_lib):
    import ctypes as C
    import ydorbslam_amd as y
    gpu, cpu = _pair(1000)
    big = synth_frame(752, 480, 40)
    view = big[:, 50:690]  # 640 wide, stride 752
    kps = np.zeros(gpu.max_keypoints, y.KP_DTYPE)
    desc = np.zeros((gpu.max_keypoints, 32), np.uint8)
    n = C.c_int32(0)
    rc = y.lib().ydorb_extract(gpu._h, view.ctypes.data_as(C.c_void_p), 640, 480, view.strides[0], kps.ctypes.data_as(C.c_void_p),
                               desc.ctypes.data_as(C.c_void_p), gpu.max_keypoints, C.byref(n))
    assert rc == 0
    ck, cd = cpu.extract(np.ascontiguousarray(view))
    _same_kps(kps[:n.value], ck)
    assert np.array_equal(desc[:n.value], cd)


@pytest.mark.parametrize("batched", [False, True])
def test_real_images_match_the_committed_goldens(batched):
    """The reference's four real 640x480 frames (thirdParty/DBow3/utils/images/image{0..3}.png): the HIP path against the committed oracle
    outputs (tests/golden/dbow3_image*_orb.npz, made by make_extractor_golden.py) - no oracle call here.  Real frames are sparse
    (~1.5 candidates per kept keypoint on level 0): the quad-tree's deep-tree form."""
    import os
    import ydorbslam_amd as y
    gold = [np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dbow3_image%d_orb.npz" % i)) for i in range(4)]
    if batched:
        ex = y.OrbExtractor(1000, 1.2, 8, 20, 7, max_batch=12, single_stream=True)
        res = ex.extract_batch(np.stack([g["image"] for g in gold] * 3))
    else:
        ex = y.OrbExtractor(1000, 1.2, 8, 20, 7)
        res = [ex.extract(g["image"]) for g in gold]
    for i, (k, d) in enumerate(res):
        g = gold[i % 4]
        _same_kps(k, g["keypoints"])
        assert np.array_equal(d, g["descriptors"]), i


@pytest.mark.parametrize("name", ["test_img1", "test_angles"])
def test_reference_test_images_match_the_committed_goldens(name):
    """The reference's own test images (test/data: 318x476 and 650x476) through the HIP path against the committed oracle outputs."""
    import os
    import ydorbslam_amd as y
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_%s_orb.npz" % name))
    k, d = y.OrbExtractor(int(g["n_features"]), 1.2, 8, 20, 7).extract(np.ascontiguousarray(g["image"]))
    _same_kps(k, g["keypoints"])
    assert np.array_equal(d, g["descriptors"])


def test_quadtree_flat_and_pass_kernels_agree(oracle_lib, monkeypatch):
    """The thinning has two device forms: the rank one (k_qt_fast: histogram pyramid, list positions from one scan, per-node maxima,
    quadtree_flat.h) and the pass one (quadtree_core.h, taken for units the rank form hands over).  Both must give the
    oracle's keypoints on dense frames, sparse frames (few candidates, nearly all kept) and the real image."""
    import os
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dbow3_image0_orb.npz"))
    frames = [synth_frame(640, 480, 11), g["image"], synth_frame(640, 480, 12).copy(), np.full((480, 640), 90, np.uint8)]
    frames[2][:240] = 128                                             # half-empty
    frames[3][100:150, 100:160] = synth_frame(640, 480, 13)[100:150, 100:160]   # one small textured patch
    imgs = np.stack(frames)
    flat = y.OrbExtractor(1000, 1.2, 8, 20, 7, max_batch=len(frames))
    monkeypatch.setenv("YDORB_QT_PASS", "1")
    passk = y.OrbExtractor(1000, 1.2, 8, 20, 7, max_batch=len(frames))
    monkeypatch.delenv("YDORB_QT_PASS")
    rf, rp = flat.extract_batch(imgs), passk.extract_batch(imgs)
    for f, img in enumerate(frames):
        ck, cd = OrbExtractorOracle(1000, 1.2, 8, 20, 7).extract(img)
        for (k, d) in (rf[f], rp[f]):
            _same_kps(k, ck)
            assert np.array_equal(d, cd)
        for l in range(8):
            assert flat.debug_read(3, l, f) == 0      # k_qt_fast itself finished every one of these units
            assert passk.debug_read(3, l, f) == 1


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (1241, 376, 2000), (752, 480, 1000), (131, 97, 300), (259, 203, 400), (1017, 333, 900)])
def test_pyramid_pads_written_by_level_kernels_and_by_border_launch(oracle_lib, monkeypatch, w, h, nf):
    """Two ways to the same padded pyramid (orbExtractor.cpp:612-621): the level kernels that write their own reflect-101 pad (default;
    levels with a side below 20 px fall back by themselves) and the interior-only kernels + one border launch (YDORB_PYR_FUSED=0).
    Every byte of every padded level, in a batch of frames, against the oracle."""
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    imgs = np.stack([synth_frame(w, h, 20 + i) for i in range(3)])
    fused = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=3)
    monkeypatch.setenv("YDORB_PYR_FUSED", "0")
    plain = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=3)
    monkeypatch.delenv("YDORB_PYR_FUSED")
    rf, rp = fused.extract_batch(imgs), plain.extract_batch(imgs)
    for f in range(3):
        cpu = OrbExtractorOracle(nf, 1.2, 8, 20, 7)
        ck, cd = cpu.extract(imgs[f])
        for l in range(8):
            cw, _, _ = cpu.level_dims(l)
            ref = cpu.level_padded(l)[:, :cw + 38]
            assert np.array_equal(fused.read_level(l, f), ref), "fused pads, frame %d level %d" % (f, l)
            assert np.array_equal(plain.read_level(l, f), ref), "border launch, frame %d level %d" % (f, l)
        for (k, d) in (rf[f], rp[f]):
            _same_kps(k, ck)
            assert np.array_equal(d, cd)


@pytest.mark.parametrize("kpw", ["1", "2", "4"])
def test_describe_kernel_keypoints_per_wave(oracle_lib, monkeypatch, kpw):
    """k_orient_describe_n<KPW> interleaves KPW keypoints in one wave (batch handles use 4, single-frame handles 1): the same
    keypoints, angles and descriptors whatever KPW is - also where a wave's slots straddle two levels or end past the last keypoint
    (the half-empty and the nearly empty frame)."""
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    frames = [synth_frame(640, 480, 41), synth_frame(640, 480, 42).copy(), np.full((480, 640), 90, np.uint8)]
    frames[1][:240] = 128
    frames[2][100:150, 100:160] = synth_frame(640, 480, 43)[100:150, 100:160]
    imgs = np.stack(frames)
    monkeypatch.setenv("YDORB_DESC_KPW", kpw)
    gpu = y.OrbExtractor(1000, 1.2, 8, 20, 7, max_batch=len(frames))
    monkeypatch.delenv("YDORB_DESC_KPW")
    res = gpu.extract_batch(imgs)
    for f, img in enumerate(frames):
        ck, cd = OrbExtractorOracle(1000, 1.2, 8, 20, 7).extract(img)
        _same_kps(res[f][0], ck)
        assert np.array_equal(res[f][1], cd)


@pytest.mark.parametrize("w,h,nf,sf,nl,thr", [
    (640, 480, 1500, 2.5, 3, 20),     # scale factor > 2: the resize kernel's byte-tap path
    (640, 480, 800, 2.0, 4, 12),      # exactly 2: still the 8-byte window path
    (752, 480, 1200, 1.5, 5, 30),
    (97, 61, 200, 1.2, 8, 20),        # smallest levels have no FAST cells and borders wider than the interior
    (1280, 720, 3000, 1.2, 8, 20),    # bigger than any BASELINE config: quotas above 600 per level, candidates beyond the LDS slots
    (661, 370, 4000, 1.5, 2, 12),     # two levels x 4000 features: the pass kernel's node tables no longer fit the LDS (HBM tables)
    (1145, 756, 2000, 2.3, 5, 12),    # > 8192 candidates on level 0 AND a quota too large for LDS node tables: hand-over to HBM tables
])
def test_other_pyramid_configurations(oracle_lib, w, h, nf, sf, nl, thr):
    """Constructor arguments other than the TUM/KITTI/EuRoC settings (orbExtractor.cpp:315-354 takes them freely)."""
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    gpu, cpu = y.OrbExtractor(nf, sf, nl, thr, 7), OrbExtractorOracle(nf, sf, nl, thr, 7)
    for idx in (0, 1):                # the second call re-sizes the quad-tree launches from the first call's candidate counts
        img = synth_frame(w, h, 20 + idx)
        gk, gd = gpu.extract(img)
        ck, cd = cpu.extract(img)
        for l in range(nl):
            cw, _, _ = cpu.level_dims(l)
            assert np.array_equal(gpu.read_level(l), cpu.level_padded(l)[:, :cw + 38]), "pyramid level %d" % l
        _same_kps(gk, ck)
        assert np.array_equal(gd, cd)


def test_huge_quota_pass_kernel_uses_hbm_node_tables(oracle_lib, monkeypatch):
    """With YDORB_QT_PASS=1 every unit goes through the pass kernel; a level whose quota needs more than the LDS for its node tables
    (here ~2700 features on level 0) must take them from HBM scratch and still give the oracle's keypoints."""
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    monkeypatch.setenv("YDORB_QT_PASS", "1")
    gpu = y.OrbExtractor(4000, 1.5, 2, 20, 7)
    monkeypatch.delenv("YDORB_QT_PASS")
    img = synth_frame(640, 480, 31)
    gk, gd = gpu.extract(img)
    ck, cd = OrbExtractorOracle(4000, 1.5, 2, 20, 7).extract(img)
    _same_kps(gk, ck)
    assert np.array_equal(gd, cd) and gpu.debug_read(3, 0) == 1


def test_device_path_reports_capacity_errors_once():
    """The device-resident entry point (ydorb_extract_batch_device, the one the headline bench drives) has no host read-back of its own,
    so the quad-tree kernels' capacity status must surface in ydorb_extractor_synchronize — and be cleared once reported, so that one
    overflow does not poison every later call.  Uniform noise gives ~10 % FAST corners: a 1200x1000 frame has > 65 535 candidates
    on level 0, the limit of the packed 16-bit candidate indices."""
    import torch
    import ydorbslam_amd as y
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (1000, 1200), dtype=np.uint8)
    ex = y.OrbExtractor(1000, 1.2, 8, 20, 7)
    with pytest.raises(y.YdorbError, match="65535"):
        ex.extract(noise)                                                    # host entry point: reported by the call itself
    good = synth_frame(1200, 1000, 5)
    k1, d1 = ex.extract(good)                                                # ... and not sticky
    assert len(k1) > 500
    cap = ex.max_keypoints
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(noise).to(dev)
    d_kps = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img.data_ptr(), 1200, 1000, 1200, 1200 * 1000, 1, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    with pytest.raises(y.YdorbError, match="65535"):
        ex.synchronize()
    ex.synchronize()                                                         # reported once
    d_img.copy_(torch.from_numpy(good).to(dev))
    ex.extract_batch_device(d_img.data_ptr(), 1200, 1000, 1200, 1200 * 1000, 1, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    ex.synchronize()
    n = int(d_n.item())
    assert n == len(k1)
    assert np.array_equal(d_desc[0, :n].cpu().numpy(), d1)


def test_read_pyramid_single_transfer_equals_level_reads():
    """ydorb_extractor_read_pyramid (all levels, one device-to-host transfer: the adapter's m_v_imagePyramid refresh) against the
    per-level reads that are themselves byte-compared with the oracle in test_stages_and_output_bit_exact."""
    import ydorbslam_amd as y
    ex = y.OrbExtractor(1000, max_batch=3)
    imgs = np.stack([synth_frame(752, 480, i) for i in range(3)])
    ex.extract_batch(imgs)
    for f in (0, 2):
        lv = ex.read_pyramid(f)
        assert len(lv) == 8
        for l in range(8):
            assert np.array_equal(lv[l], ex.read_level(l, f)), "frame %d level %d" % (f, l)


@pytest.mark.parametrize("w,h,nf,single", [(640, 480, 1000, False), (752, 480, 1000, False), (640, 480, 1000, True), (1241, 376, 2000, True)])
def test_batched_handle_paths_match_the_oracle(oracle_lib, w, h, nf, single):
    """A handle for more than 8 frames per call takes the throughput paths the headline bench runs: FAST cells launched in two level
    groups, one quad-tree launch (k_qt_fast) per group, four keypoints per wave in the descriptor kernel; `single` = a
    YDORB_EXTRACTOR_SINGLE_STREAM handle (every launch on the call's stream: what bench.py's lanes use).  Twelve frames of
    mixed content (textured, half empty, nearly empty) - twice, so that the second call runs with the retuned quad-tree footprints -
    against the oracle, frame by frame."""
    import ydorbslam_amd as y
    from oracle.orb_oracle import OrbExtractorOracle
    frames = [synth_frame(w, h, 50 + i) for i in range(12)]
    frames[3] = frames[3].copy(); frames[3][: h // 2] = 128
    frames[7] = np.full((h, w), 90, np.uint8); frames[7][100:150, 100:160] = synth_frame(w, h, 63)[100:150, 100:160]
    imgs = np.stack(frames)
    gpu = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=12, single_stream=single)
    gpu.extract_batch(imgs)
    res = gpu.extract_batch(imgs)
    for f, img in enumerate(frames):
        ck, cd = OrbExtractorOracle(nf, 1.2, 8, 20, 7).extract(img)
        _same_kps(res[f][0], ck)
        assert np.array_equal(res[f][1], cd)
