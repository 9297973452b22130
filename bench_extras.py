"""Secondary sections of bench.py (one GPU only): what ONE call of the drop-in costs host to host, the brute-force Hamming figure,
and the SURVEY.md 8(f) "next" rows (optimizePose, computeStereoMatches, Vocabulary::transform, computeDistinctiveDescriptors), each with
its CPU-oracle baseline on a bounded sample.  `run` fills `out` and returns the state `cpu_baselines` needs."""
import time

import numpy as np

INT_VALU_PEAK = 256 * 4 * 32 * 2.4e9   # integer-VALU peak for the brute-force figure: 256 CUs x 4 SIMDs x 32 lanes per cycle at 2.4 GHz


def _med_ms(fn, n=15):
    fn()
    ts = []
    for _ in range(n):
        t_ = time.perf_counter(); fn(); ts.append(time.perf_counter() - t_)
    return float(np.median(ts) * 1e3)


def single_call(ctx, y, S, out):
    """what ONE call of the drop-in sees, host to host (frame.cpp:129, tracking.cpp:456, localMapping.cpp:140)"""
    W, H, NFEAT, sf = S.W, S.H, S.NFEAT, S.sf
    ex1 = y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=ctx.local_rank)
    one = S.imgs[3]
    sc = {"extract_ms": _med_ms(lambda: ex1.extract(one)),
          "extract_plus_pyramid_download_ms": _med_ms(lambda: (ex1.extract(one), ex1.read_pyramid())),
          "note": "median host-to-host wall time of one call: ydorb_extract = H2D 307 KB + ~20 launches + D2H 60 KB; the adapter's "
                  "m_v_imagePyramid refresh adds one 1.3 MB device-to-host copy + host repacking (ydorb_extractor_read_pyramid)"}
    ka_, da_ = ex1.extract(S.imgs[3]); kb_, db_ = ex1.extract(S.imgs[4])
    q1 = np.zeros(len(ka_), y.QUERY_DTYPE)
    q1["u"], q1["v"] = ka_["x"], ka_["y"]
    q1["r"] = (np.float32(15.0) * sf[ka_["octave"]]).astype(np.float32)
    q1["min_level"], q1["max_level"] = ka_["octave"] - 1, ka_["octave"] + 1
    q1["angle"], q1["level"], q1["flags"] = ka_["angle"], ka_["octave"], 3
    fv1 = y.FrameView(kb_, db_, (0.0, float(W), 0.0, float(H)))
    m1 = y.OrbMatcher(0.9, True, device=ctx.local_rank)
    sc["search_by_projection_ms"] = _med_ms(lambda: m1.search_by_projection(1, fv1, q1, da_))
    sc["keypoints_per_s_one_frame_at_a_time"] = len(ka_) / ((sc["extract_ms"] + sc["search_by_projection_ms"]) * 1e-3)
    if "ba" in out:
        sc["ba_solve_ms"] = out["ba"]["ms_per_solve"]
        from ydorbslam_amd.synth import synth_pose_problem
        pp1 = [synth_pose_problem(400, seed=100)]
        sc["pose_optimize_ms"] = _med_ms(lambda: y.Optimizer.optimize_poses(pp1))
    out["single_call"] = sc


def run(ctx, y, S, out, want):
    torch, dev, args = ctx.torch, ctx.dev, ctx.args
    W, H, NFEAT, F, cap = S.W, S.H, S.NFEAT, S.F, S.cap
    st = {}
    if want("single_call"):
        single_call(ctx, y, S, out)
    if not want("rest"):
        return st
    d_desc = S.own[0][S.off_desc:S.off_desc + S.F * cap * 32].view(S.F, cap, 32)
    d_n = S.counts(0)
    # ---- brute-force N x M Hamming top-2 (north_star; SURVEY 8d secondary figure, against the integer-VALU peak) -------------------
    NB_ = min(F - 1, 255)
    mb = y.OrbMatcher(device=ctx.local_rank)
    d_best = torch.zeros((NB_, cap, 6), dtype=torch.int32, device=dev)
    call = lambda: mb.hamming_topk_device(d_desc.data_ptr(), d_n.data_ptr(), d_desc[1:].data_ptr(), d_n[1:].data_ptr(), cap, NB_, d_best.data_ptr())
    call()
    torch.cuda.synchronize()
    tb_ = time.perf_counter()
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    tb_ = (time.perf_counter() - tb_) / 5
    nn_ = d_n.cpu().numpy().astype(np.int64)
    npairs_ = float((nn_[:NB_] * nn_[1:NB_ + 1]).sum())
    out["match_bruteforce"] = {"metric": "all-pairs 256-bit Hamming top-2, frame t vs frame t+1", "frame_pairs_per_call": NB_,
                               "value": npairs_ / tb_ / 1e9, "unit": "G descriptor pairs/s", "ms_per_call": tb_ * 1e3,
                               "lane_ops_per_pair": 16, "int_valu_peak_Gops": INT_VALU_PEAK / 1e9,
                               "frac_of_int_valu_peak": npairs_ * 16 / tb_ / INT_VALU_PEAK}
    if args.no_ba:
        return st
    # ---- pose-only optimisation (Optimizer::optimizePose): a batch of frames per launch -------------------
    from ydorbslam_amd.synth import synth_pose_problem, synth_stereo_pair, synth_vocabulary
    NPF = 256
    pprobs = [synth_pose_problem(400, seed=100 + i) for i in range(NPF)]
    y.Optimizer.optimize_poses(pprobs)
    tp = time.perf_counter()
    for _ in range(5):
        pres = y.Optimizer.optimize_poses(pprobs)
    tp = (time.perf_counter() - tp) / 5
    out["pose_optimize"] = {"metric": "optimizePose frames/sec (400 correspondences per frame, 4 episodes x 10 LM iterations)",
                            "frames_per_launch": NPF, "value": NPF / tp, "unit": "frames/s", "ms_per_launch": tp * 1e3,
                            "mean_inliers": float(np.mean([r_["inliers"] for r_ in pres]))}
    st["pprobs"] = pprobs
    # ---- stereo association (Frame::computeStereoMatches): a batch of rectified pairs per call ----------------
    NSP, SDIST = 64, 8
    spairs = [synth_stereo_pair(W, H, i) for i in range(SDIST)]
    sex = y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, max_batch=2 * NSP)
    sres = sex.extract_batch(np.stack([spairs[p % SDIST][i] for p in range(NSP) for i in (0, 1)]))
    scap = max(len(k_) for k_, _ in sres)
    skl = np.zeros((NSP, scap), y.KP_DTYPE); skr = np.zeros((NSP, scap), y.KP_DTYPE)
    sdl = np.zeros((NSP, scap, 32), np.uint8); sdr = np.zeros((NSP, scap, 32), np.uint8)
    snl = np.zeros(NSP, np.int32); snr = np.zeros(NSP, np.int32)
    for p in range(NSP):
        (ka_, da_), (kb_, db_) = sres[2 * p], sres[2 * p + 1]
        skl[p, :len(ka_)], sdl[p, :len(ka_)], snl[p] = ka_, da_, len(ka_)
        skr[p, :len(kb_)], sdr[p, :len(kb_)], snr[p] = kb_, db_, len(kb_)
    sm = y.OrbMatcher()
    out["stereo"] = {"metric": "computeStereoMatches pairs/sec (640x480, 1000 features per image, host keypoints in, depth out)", "pairs_per_call": NSP}
    for name, by_kp in (("reference_replay", False), ("index_by_keypoint", True)):
        sm.stereo_matches(sex, sex, skl, sdl, snl, skr, sdr, snr, 40.0, 0.1, by_kp, (0, 2), (1, 2))
        ts = time.perf_counter()
        for _ in range(5):
            sout = sm.stereo_matches(sex, sex, skl, sdl, snl, skr, sdr, snr, 40.0, 0.1, by_kp, (0, 2), (1, 2))
        ts = (time.perf_counter() - ts) / 5
        out["stereo"][name] = {"value": NSP / ts, "unit": "pairs/s", "ms_per_call": ts * 1e3, "measurements_per_pair": float(np.mean(sout[2]))}
    st["spairs"] = spairs
    # ---- vocabulary transform (DBoW3::Vocabulary::transform): BowVector + FeatureVector per frame ----------------
    vtree = synth_vocabulary(10, 5, seed=1)   # k = 10 like the ORB vocabulary, one level less (L = 6 would be 35 MB of synthetic nodes)
    voc = y.Vocabulary(vtree)
    NBF = min(256, F)
    hn = d_n[:NBF].cpu().numpy()
    hdesc = d_desc[:NBF].cpu().numpy()
    bdescs = [hdesc[f, :hn[f]] for f in range(NBF)]
    voc.transform(bdescs, 3)
    tv = time.perf_counter()
    for _ in range(3):
        bout = voc.transform(bdescs, 3)
    tv = (time.perf_counter() - tv) / 3
    out["bow_transform"] = {"metric": "Vocabulary::transform frames/sec (1000 descriptors per frame, k=10 L=5 synthetic tree, levelsup 3; host descriptors in, host vectors out)",
                            "frames_per_call": NBF, "tree_nodes": int(len(vtree["node_word"])), "value": NBF / tv, "unit": "frames/s",
                            "ms_per_call": tv * 1e3, "mean_words_per_frame": float(np.mean([len(b[0]) for b in bout]))}
    st["vtree"], st["bdescs"] = vtree, bdescs
    # ---- distinctive descriptors (MapPoint::computeDistinctiveDescriptors): a batch of map points per call -----------
    rngd = np.random.default_rng(5)
    NMP = 50000
    pool_d = np.concatenate(bdescs[:64])
    sizes = rngd.integers(2, 21, NMP)
    groups_d = [pool_d[a:a + m_] for a, m_ in zip(rngd.integers(0, len(pool_d) - 21, NMP), sizes)]
    mm_d = y.OrbMatcher()
    mm_d.distinctive_descriptors(groups_d[:100])
    td = time.perf_counter()
    best_d = mm_d.distinctive_descriptors(groups_d)
    td = time.perf_counter() - td
    out["distinctive_descriptors"] = {"metric": "computeDistinctiveDescriptors map points/sec (2-20 observations each, one batched call, host in / host out incl. Python packing)",
                                      "points_per_call": NMP, "value": NMP / td, "unit": "points/s", "ms_per_call": td * 1e3}
    st["groups_d"], st["best_d"] = groups_d, best_d
    return st


def cpu_baselines(ctx, S, out, st):
    """CPU-oracle baselines of the sections above (rank 0, bounded samples)."""
    if "pprobs" not in st:
        return
    from oracle.orb_oracle import OrbExtractorOracle
    from oracle.orb_oracle import pose_optimize as oracle_pose_optimize
    NFEAT = S.NFEAT
    tpc = time.perf_counter()
    for i in range(32):
        oracle_pose_optimize(st["pprobs"][i])
    tpc = (time.perf_counter() - tpc) / 32
    out["pose_optimize"]["cpu_baseline"] = {"value": 1.0 / tpc, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "32 of the same frames"}
    from oracle.orb_oracle import stereo_matches as oracle_stereo
    oel, oer = OrbExtractorOracle(NFEAT, 1.2, 8, 20, 7), OrbExtractorOracle(NFEAT, 1.2, 8, 20, 7)
    tsc, nsc = 0.0, 0
    for p in range(4):
        kl_, dl_ = oel.extract(st["spairs"][p][0]); kr_, dr_ = oer.extract(st["spairs"][p][1])
        lvl_ = [oel.level_padded(l)[19:19 + oel.level_dims(l)[1], 19:19 + oel.level_dims(l)[0]] for l in range(8)]
        lvr_ = [oer.level_padded(l)[19:19 + oer.level_dims(l)[1], 19:19 + oer.level_dims(l)[0]] for l in range(8)]
        tb_ = oel.tables()
        for by_kp in (False, True):
            t0_ = time.perf_counter()
            oracle_stereo(kl_, dl_, kr_, dr_, lvl_, lvr_, tb_["scale"], tb_["inv_scale"], 40.0, 0.1, by_kp)
            tsc += time.perf_counter() - t0_; nsc += 1
    out["stereo"]["cpu_baseline"] = {"value": nsc / tsc, "unit": "pairs/s", "cores": 1, "kind": "port",
                                     "sample": "4 of the same pairs, both index forms, association only (pyramids and keypoints given)"}
    from oracle.orb_oracle import bow_transform as oracle_bow
    tvc = time.perf_counter()
    for f in range(16):
        oracle_bow(st["vtree"], st["bdescs"][f], 3, 0, 1)
    tvc = (time.perf_counter() - tvc) / 16
    out["bow_transform"]["cpu_baseline"] = {"value": 1.0 / tvc, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "16 of the same frames"}
    from oracle.orb_oracle import distinctive_descriptor as oracle_dd
    tdc = time.perf_counter()
    ok_d = all(oracle_dd(st["groups_d"][i]) == st["best_d"][i] for i in range(5000))
    tdc = (time.perf_counter() - tdc) / 5000
    out["distinctive_descriptors"]["cpu_baseline"] = {"value": 1.0 / tdc, "unit": "points/s", "cores": 1, "kind": "port",
                                                      "sample": "5000 of the same points (results equal: %s)" % ok_d}
