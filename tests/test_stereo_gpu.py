"""GPU parity: ydorb_stereo_matches (C ABI) vs oracle/stereo_oracle.cpp, the restatement of Frame::computeStereoMatches
(reference src/frame.cpp:362-477).  Bar: identical float bit patterns.  Parity unpinned (see the oracle's header)."""
import numpy as np
import pytest

from ydorbslam_amd.synth import synth_stereo_pair

pytestmark = pytest.mark.gpu

BF, B = 40.0, 0.1


def _oracle_pair(oracle_lib, left, right, n_features, kl=None, dl=None, kr=None, dr=None, by_kp=False):
    from oracle.orb_oracle import OrbExtractorOracle
    el, er = OrbExtractorOracle(n_features), OrbExtractorOracle(n_features)
    okl, odl = el.extract(left)
    okr, odr = er.extract(right)
    lv_l, lv_r = [], []
    for l in range(8):
        w, h, _ = el.level_dims(l)
        lv_l.append(el.level_padded(l)[19:19 + h, 19:19 + w])
        lv_r.append(er.level_padded(l)[19:19 + h, 19:19 + w])
    t = el.tables()
    kl = okl if kl is None else kl
    dl = odl if dl is None else dl
    kr = okr if kr is None else kr
    dr = odr if dr is None else dr
    return (okl, odl, okr, odr), oracle_lib.stereo_matches(kl, dl, kr, dr, lv_l, lv_r, t["scale"], t["inv_scale"], BF, B, by_kp)


def _pad(items, cap, dtype, tail=()):
    out = np.zeros((len(items), cap) + tuple(tail), dtype)
    for i, a in enumerate(items):
        out[i, :len(a)] = a
    return out


@pytest.mark.parametrize("by_kp", [False, True])
def test_batched_pairs_match_the_oracle(oracle_lib, by_kp):
    import ydorbslam_amd as y
    nf, pairs = 800, 3
    imgs, per = [], []
    for p in range(pairs):
        left, right, _ = synth_stereo_pair(640, 480, p)
        imgs += [left, right]
        per.append((left, right))
    ex = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=2 * pairs)
    res = ex.extract_batch(np.stack(imgs))
    cap = max(len(k) for k, _ in res)
    kl = _pad([res[2 * p][0] for p in range(pairs)], cap, y.KP_DTYPE)
    dl = _pad([res[2 * p][1] for p in range(pairs)], cap, np.uint8, (32,))
    kr = _pad([res[2 * p + 1][0] for p in range(pairs)], cap, y.KP_DTYPE)
    dr = _pad([res[2 * p + 1][1] for p in range(pairs)], cap, np.uint8, (32,))
    nl = np.array([len(res[2 * p][0]) for p in range(pairs)], np.int32)
    nr = np.array([len(res[2 * p + 1][0]) for p in range(pairs)], np.int32)
    m = y.OrbMatcher()
    rx, depth, kept, status = m.stereo_matches(ex, ex, kl, dl, nl, kr, dr, nr, BF, B, index_by_keypoint=by_kp, left_frames=(0, 2), right_frames=(1, 2))
    total = 0
    for p in range(pairs):
        (okl, _, okr, _), (orx, odepth, okept, ostatus) = _oracle_pair(oracle_lib, per[p][0], per[p][1], nf, by_kp=by_kp)
        assert len(okl) == nl[p] and len(okr) == nr[p]
        assert kept[p] == okept and status[p] == ostatus == 0
        assert np.array_equal(rx[p, :nl[p]].view(np.uint32), orx.view(np.uint32)), p
        assert np.array_equal(depth[p, :nl[p]].view(np.uint32), odepth.view(np.uint32)), p
        assert np.all(depth[p, nl[p]:] == -1)
        total += okept
    assert total > (300 if by_kp else 3)


def test_two_extractors_and_undefined_cases_are_reported(oracle_lib):
    """Left and right pyramids from two handles (as the reference's two extractors); a left keypoint below the image and a best
    right keypoint 8 px from the left edge exercise the two situations the reference leaves undefined."""
    import ydorbslam_amd as y
    nf = 600
    left, right, _ = synth_stereo_pair(640, 480, 5)
    el, er = y.OrbExtractor(nf), y.OrbExtractor(nf)
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    kl, kr = kl.copy(), kr.copy()
    lvl0 = np.flatnonzero((kl["octave"] == 0) & (kl["x"] > 60) & (kl["y"] > 30) & (kl["y"] < 440))
    a, c = int(lvl0[0]), int(lvl0[1])
    kl["y"][c] = 480.5                      # row index past the table
    fake = kr[:1].copy()
    fake["x"], fake["y"], fake["octave"] = 8.0, kl["y"][a], 0
    kr = np.concatenate([kr, fake])
    dr = np.concatenate([dr, dl[a:a + 1]])  # distance 0 to left keypoint a -> best right column at x = 8
    m = y.OrbMatcher()
    for by_kp in (True, False):
        rx, depth, kept, status = m.stereo_matches(el, er, kl[None], dl[None], [len(kl)], kr[None], dr[None], [len(kr)], BF, 0.05, index_by_keypoint=by_kp)
        from oracle.orb_oracle import OrbExtractorOracle
        ol, orr = OrbExtractorOracle(nf), OrbExtractorOracle(nf)
        ol.extract(left)
        orr.extract(right)
        lv_l = [ol.level_padded(l)[19:19 + ol.level_dims(l)[1], 19:19 + ol.level_dims(l)[0]] for l in range(8)]
        lv_r = [orr.level_padded(l)[19:19 + orr.level_dims(l)[1], 19:19 + orr.level_dims(l)[0]] for l in range(8)]
        t = ol.tables()
        orx, odepth, okept, ostatus = oracle_lib.stereo_matches(kl, dl, kr, dr, lv_l, lv_r, t["scale"], t["inv_scale"], BF, 0.05, by_kp)
        assert status[0] == ostatus and kept[0] == okept
        if by_kp:
            assert ostatus == 3
        assert np.array_equal(rx[0].view(np.uint32), orx.view(np.uint32))
        assert np.array_equal(depth[0].view(np.uint32), odepth.view(np.uint32))


def test_argument_checks():
    import ydorbslam_amd as y
    ex = y.OrbExtractor(300)
    m = y.OrbMatcher()
    k = np.zeros((1, 4), y.KP_DTYPE)
    d = np.zeros((1, 4, 32), np.uint8)
    with pytest.raises(y.YdorbError):   # no pyramid yet
        m.stereo_matches(ex, ex, k, d, [0], k, d, [0], BF, B)
    ex.extract(synth_stereo_pair(320, 240, 0)[0])
    with pytest.raises(y.YdorbError):   # frame outside the last call
        m.stereo_matches(ex, ex, k, d, [0], k, d, [0], BF, B, right_frames=(1, 1))
    rx, depth, kept, status = m.stereo_matches(ex, ex, k, d, [0], k, d, [0], BF, B)
    assert kept[0] == 0 and np.all(rx == -1) and np.all(depth == -1)


@pytest.mark.parametrize("w,h,nf", [(1241, 376, 2000), (752, 480, 1000)])
def test_baseline_stereo_configurations(oracle_lib, w, h, nf):
    """BASELINE.json configs 3 and 4 (KITTI-00-size and EuRoC-MH-size stereo): extraction of both images and the stereo
    association, bit-exact against the oracle in both index forms."""
    import ydorbslam_amd as y
    left, right, drow = synth_stereo_pair(w, h, 11)
    ex = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=2)
    (kl, dl), (kr, dr) = ex.extract_batch(np.stack([left, right]))
    m = y.OrbMatcher()
    for by_kp in (False, True):
        rx, depth, kept, status = m.stereo_matches(ex, ex, kl[None], dl[None], [len(kl)], kr[None], dr[None], [len(kr)], BF, B,
                                                   index_by_keypoint=by_kp, left_frames=(0, 1), right_frames=(1, 1))
        (okl, odl, okr, odr), (orx, odepth, okept, ostatus) = _oracle_pair(oracle_lib, left, right, nf, by_kp=by_kp)
        assert okl.tobytes() == kl.tobytes() and okr.tobytes() == kr.tobytes() and np.array_equal(odl, dl) and np.array_equal(odr, dr)
        assert kept[0] == okept and status[0] == ostatus == 0
        assert np.array_equal(rx[0].view(np.uint32), orx.view(np.uint32)) and np.array_equal(depth[0].view(np.uint32), odepth.view(np.uint32))
        if by_kp:
            ok = depth[0] > 0
            assert ok.sum() > 0.3 * len(kl)
            disp = kl["x"][ok] - rx[0][ok]
            assert np.mean(np.abs(disp - drow[kl["y"][ok].astype(int)]) < 1.5 * 1.2 ** kl["octave"][ok]) > 0.9


def test_index_chain_that_depends_on_the_descriptor_history(oracle_lib):
    """Every other left descriptor is noise and a third are copies of one right descriptor, so whether a keypoint reaches
    `leftIdx++` depends on which descriptor the lagging index hands it."""
    import ydorbslam_amd as y
    nf = 700
    rng = np.random.default_rng(3)
    pairs = []
    for p in range(2):
        left, right, _ = synth_stereo_pair(640, 480, 20 + p, disparities=(9, 9, 9))
        pairs.append((left, right))
    ex = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=4)
    res = ex.extract_batch(np.stack([im for pr in pairs for im in pr]))
    m = y.OrbMatcher()
    for p, (left, right) in enumerate(pairs):
        (kl, dl), (kr, dr) = res[2 * p], res[2 * p + 1]
        dl = dl.copy()
        dl[1::2] = rng.integers(0, 256, dl[1::2].shape, dtype=np.uint8)
        dl[::3] = dr[len(dr) // 2]
        _, (orx, odepth, okept, ostatus) = _oracle_pair(oracle_lib, left, right, nf, dl=dl)
        rx, depth, kept, status = m.stereo_matches(ex, ex, kl[None], dl[None], [len(kl)], kr[None], dr[None], [len(kr)], BF, B,
                                                   left_frames=(2 * p, 1), right_frames=(2 * p + 1, 1))
        assert kept[0] == okept and status[0] == ostatus
        assert np.array_equal(rx[0].view(np.uint32), orx.view(np.uint32)) and np.array_equal(depth[0].view(np.uint32), odepth.view(np.uint32))


def test_device_resident_form_equals_the_host_form():
    """extract_batch_device -> stereo with YDORB_STEREO_DEVICE_POINTERS: nothing leaves HBM between the two calls."""
    import torch
    import ydorbslam_amd as y
    pairs, W, H, nf = 3, 640, 480, 800
    imgs = np.stack([im for p in range(pairs) for im in synth_stereo_pair(W, H, 30 + p)[:2]])
    ex = y.OrbExtractor(nf, max_batch=2 * pairs)
    cap = ex.max_keypoints
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(imgs).to(dev)
    d_kps = torch.zeros((2 * pairs, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((2 * pairs, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(2 * pairs, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_img.data_ptr(), W, H, W, W * H, 2 * pairs, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    ex.synchronize()
    # de-interleave on the device: the stereo call takes [pairs][cap] arrays per side
    kl, kr = d_kps[0::2].contiguous(), d_kps[1::2].contiguous()
    dl, dr = d_desc[0::2].contiguous(), d_desc[1::2].contiguous()
    nl, nr = d_n[0::2].contiguous(), d_n[1::2].contiguous()
    torch.cuda.synchronize()
    m = y.OrbMatcher()
    to_kp = lambda t: t.cpu().numpy().view(np.uint8).reshape(pairs, cap, 28).view(y.KP_DTYPE).reshape(pairs, cap)
    for by_kp in (False, True):
        d_rx = torch.zeros((pairs, cap), dtype=torch.float32, device=dev)
        d_depth = torch.zeros((pairs, cap), dtype=torch.float32, device=dev)
        d_kept = torch.zeros(pairs, dtype=torch.int32, device=dev)
        d_status = torch.full((pairs,), 7, dtype=torch.int32, device=dev)
        m.stereo_matches_device(ex, ex, kl.data_ptr(), dl.data_ptr(), nl.data_ptr(), cap, kr.data_ptr(), dr.data_ptr(), nr.data_ptr(), cap, pairs, BF, B,
                                d_rx.data_ptr(), d_depth.data_ptr(), d_kept.data_ptr(), d_status.data_ptr(), index_by_keypoint=by_kp,
                                left_frames=(0, 2), right_frames=(1, 2))
        m.synchronize()
        rx, depth, kept, status = m.stereo_matches(ex, ex, to_kp(kl), dl.cpu().numpy(), nl.cpu().numpy(), to_kp(kr), dr.cpu().numpy(), nr.cpu().numpy(),
                                                   BF, B, index_by_keypoint=by_kp, left_frames=(0, 2), right_frames=(1, 2))
        assert np.array_equal(d_rx.cpu().numpy().view(np.uint32), rx.view(np.uint32))
        assert np.array_equal(d_depth.cpu().numpy().view(np.uint32), depth.view(np.uint32))
        assert np.array_equal(d_kept.cpu().numpy(), kept) and np.array_equal(d_status.cpu().numpy(), status) and kept.sum() > 0


def test_replay_with_and_without_row_lists(oracle_lib, monkeypatch):
    """The replay kernel takes a row's candidates from an index of the right keypoints sorted by the first row of their band (LDS) or
    scans every right keypoint's band (YDORB_STEREO_NO_ROW_LISTS forces the second form).  Same bits either way."""
    import ydorbslam_amd as y
    left, right, _ = synth_stereo_pair(640, 480, 41)
    ex = y.OrbExtractor(1000, max_batch=2)
    (kl, dl), (kr, dr) = ex.extract_batch(np.stack([left, right]))
    args = (ex, ex, kl[None], dl[None], [len(kl)], kr[None], dr[None], [len(kr)], BF, B)
    a = y.OrbMatcher().stereo_matches(*args, left_frames=(0, 1), right_frames=(1, 1))
    monkeypatch.setenv("YDORB_STEREO_NO_ROW_LISTS", "1")
    b = y.OrbMatcher().stereo_matches(*args, left_frames=(0, 1), right_frames=(1, 1))
    monkeypatch.delenv("YDORB_STEREO_NO_ROW_LISTS")
    _, (orx, odepth, okept, ostatus) = _oracle_pair(oracle_lib, left, right, 1000)
    for got in (a, b):
        assert got[2][0] == okept and got[3][0] == ostatus
        assert np.array_equal(got[0][0].view(np.uint32), orx.view(np.uint32)) and np.array_equal(got[1][0].view(np.uint32), odepth.view(np.uint32))


def test_every_right_keypoint_on_the_top_level(oracle_lib):
    """Right keypoints that all claim the top level have the widest bands (17 rows each instead of ~8 on average): every row's window of the
    sorted index is then as long as it can get for this keypoint density.  Still the reference's result."""
    import ydorbslam_amd as y
    left, right, _ = synth_stereo_pair(640, 480, 43)
    ex = y.OrbExtractor(1000, max_batch=2)
    (kl, dl), (kr, dr) = ex.extract_batch(np.stack([left, right]))
    kr2 = kr.copy()
    kr2["octave"] = 7
    got = y.OrbMatcher().stereo_matches(ex, ex, kl[None], dl[None], [len(kl)], kr2[None], dr[None], [len(kr2)], BF, B, left_frames=(0, 1), right_frames=(1, 1))
    _, (orx, odepth, okept, ostatus) = _oracle_pair(oracle_lib, left, right, 1000, kl=kl, dl=dl, kr=kr2, dr=dr)
    assert got[2][0] == okept and got[3][0] == ostatus
    assert np.array_equal(got[0][0].view(np.uint32), orx.view(np.uint32)) and np.array_equal(got[1][0].view(np.uint32), odepth.view(np.uint32))


def test_right_keypoints_crowded_into_a_few_rows(oracle_lib):
    """More than 128 index entries in one row's window (the first trip of the candidate scan takes two per lane): the right keypoints are
    moved onto three image rows, the left ones onto the same rows, so every step walks several trips.  Same bits as the oracle and the scan form."""
    import ydorbslam_amd as y
    left, right, _ = synth_stereo_pair(640, 480, 44)
    ex = y.OrbExtractor(1000, max_batch=2)
    (kl, dl), (kr, dr) = ex.extract_batch(np.stack([left, right]))
    kl2, kr2 = kl.copy(), kr.copy()
    kr2["y"] = (200 + 40 * (np.arange(len(kr2)) % 3)).astype(np.float32) + (kr2["y"] - np.floor(kr2["y"]))
    kl2["y"] = (200 + 40 * (np.arange(len(kl2)) % 3)).astype(np.float32) + (kl2["y"] - np.floor(kl2["y"]))
    args = (ex, ex, kl2[None], dl[None], [len(kl2)], kr2[None], dr[None], [len(kr2)], BF, B)
    got = y.OrbMatcher().stereo_matches(*args, left_frames=(0, 1), right_frames=(1, 1))
    _, (orx, odepth, okept, ostatus) = _oracle_pair(oracle_lib, left, right, 1000, kl=kl2, dl=dl, kr=kr2, dr=dr)
    assert got[2][0] == okept and got[3][0] == ostatus
    assert np.array_equal(got[0][0].view(np.uint32), orx.view(np.uint32)) and np.array_equal(got[1][0].view(np.uint32), odepth.view(np.uint32))
