#!/usr/bin/env python3
"""Headline benchmark: ORB extract+match Mkeypoints/s (BASELINE.json config 2: 640x480 mono stream, 1000 features/frame)
and local-BA LM iterations/s (config 5: 100 keyframes x 10 000 points) on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic frames that already sit in HBM:
  ydorb_extract_batch_device (pyramid -> FAST cells -> quad-tree -> blur -> orientation + rBRIEF)
  + ydorb_match_consecutive_device (grid build -> candidate distances -> ordered resolve) over the F-1 frame pairs.
N > 1: one process per GPU (torch.distributed / RCCL); frames shard across ranks (weak scaling); the frame pair that
straddles two ranks is matched after an all-gather of the boundary frames' keypoints + descriptors.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch stream; `cpu_baseline` times the
CPU oracle (a port of the reference algorithm, single thread) on a bounded sample of the same frames.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT = 640, 480, 1000
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
FP64_VEC_PEAK = 78.6e12    # FLOP/s, MI355X FP64 vector datasheet figure (the in-container guide lists no FP64 number)


def algorithmic_bytes_extract(w, h, n):
    """image read + padded pyramid written (public output) + keypoints/descriptors written."""
    tot = w * h + n * 60
    for l in range(8):
        inv = np.float32(pow(np.float32(1.2), -l))
        wl, hl = int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))
        tot += (wl + 38) * (hl + 38)
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=512, help="frames per step and per GPU")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic frames (tiled to --frames)")
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the CPU-oracle baseline sample")
    ap.add_argument("--extractors", type=int, default=2, help="extractor handles (each with its own stream) the frames of a step are split over")
    ap.add_argument("--ba-threads", type=int, default=8, help="host threads of the concurrent local-BA figure (the library pools 8 contexts per device)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    # YDORB_BENCH_BACKEND=gloo + YDORB_BENCH_ONE_GPU=1: rehearse the N > 1 logic with every rank on cuda:0 and the collectives
    # staged through host memory (RCCL refuses two ranks on one device).  The driver's real runs use RCCL ("nccl").
    backend = os.environ.get("YDORB_BENCH_BACKEND", "nccl")
    if os.environ.get("YDORB_BENCH_ONE_GPU"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    staged = world > 1 and backend != "nccl"

    def all_reduce_(t, op):
        if staged:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def all_gather_into(out, inp):
        if staged:
            parts = [torch.zeros_like(inp, device="cpu") for _ in range(world)]
            dist.all_gather(parts, inp.cpu())
            out.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(out, inp)

    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_frame, synth_ba_problem

    F = args.frames
    distinct = [synth_frame(W, H, rank * 1000 + i) for i in range(min(args.distinct, F))]
    imgs = np.stack([distinct[i % len(distinct)] for i in range(F)])
    # The frames of a step are split over `--extractors` handles, each with its own stream (like the reference's two extractor
    # objects for stereo): while one handle sits in its latency-bound quad-tree stage the other's pyramid/FAST kernels fill the chip.
    NEX = max(1, min(args.extractors, F // 8))
    parts = [(i * F // NEX, (i + 1) * F // NEX) for i in range(NEX)]
    exs = [y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=b - a) for a, b in parts]
    ex = exs[0]
    mt = y.OrbMatcher(0.9, True, device=local_rank)
    cap = ex.max_keypoints
    sf = ex.tables()["scale"]
    d_img = torch.from_numpy(imgs).to(dev)
    # Two output sets + two explicit streams: extraction of step k+1 (stream A) overlaps the matching of step k (stream B).
    # (The default stream's handle is 0, which the C ABI reads as "use the handle's own stream": always pass real streams.)
    d_kps = [torch.zeros((F, cap, 7), dtype=torch.float32, device=dev) for _ in range(2)]
    d_desc = [torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_n = [torch.zeros(F, dtype=torch.int32, device=dev) for _ in range(2)]
    d_assigned = [torch.zeros((F - 1, cap), dtype=torch.int32, device=dev) for _ in range(2)]
    d_counts = [torch.zeros(F - 1, dtype=torch.int32, device=dev) for _ in range(2)]
    sAs, sB = [torch.cuda.Stream(device=dev) for _ in range(NEX)], torch.cuda.Stream(device=dev)
    ev_extracted = [[torch.cuda.Event() for _ in range(NEX)] for _ in range(2)]
    ev_matched = [torch.cuda.Event() for _ in range(2)]
    for e in ev_matched:
        e.record(sB)
    mts = [mt, y.OrbMatcher(0.9, True, device=local_rank)]   # one matcher (own scratch) per output set
    step_no = [0]
    # cross-rank boundary pair (only N > 1): all-gather of [keypoints | descriptors] of each rank's last frame
    if world > 1:
        rec = cap * (28 + 32) + 4
        send = torch.zeros(rec, dtype=torch.uint8, device=dev)
        gathered = torch.zeros(world * rec, dtype=torch.uint8, device=dev)
        b_kps = torch.zeros((2, cap, 7), dtype=torch.float32, device=dev)
        b_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        b_n = torch.zeros(2, dtype=torch.int32, device=dev)
        b_assigned = torch.zeros((1, cap), dtype=torch.int32, device=dev)
        b_counts = torch.zeros(1, dtype=torch.int32, device=dev)
        mt2 = y.OrbMatcher(0.9, True, device=local_rank)

    def step():
        b = step_no[0] & 1
        step_no[0] += 1
        for i, (f0, f1) in enumerate(parts):
            sA = sAs[i]
            sA.wait_event(ev_matched[b])          # the matcher that last read this output set is done
            exs[i].extract_batch_device(d_img[f0].data_ptr(), W, H, W, W * H, f1 - f0, d_kps[b][f0].data_ptr(), d_desc[b][f0].data_ptr(), cap,
                                        d_n[b][f0:].data_ptr(), sA.cuda_stream)
            ev_extracted[b][i].record(sA)
            sB.wait_event(ev_extracted[b][i])
        mts[b].match_consecutive_device(d_kps[b].data_ptr(), d_desc[b].data_ptr(), d_n[b].data_ptr(), cap, F, W, H, 15.0, sf,
                                        d_assigned[b].data_ptr(), d_counts[b].data_ptr(), None, sB.cuda_stream)
        if world > 1:
            with torch.cuda.stream(sB):
                send[:cap * 28] = d_kps[b][F - 1].view(torch.uint8).reshape(-1)
                send[cap * 28:cap * 60] = d_desc[b][F - 1].reshape(-1)
                send[cap * 60:] = d_n[b][F - 1:F].view(torch.uint8)
                all_gather_into(gathered, send)
                prev = (rank - 1) % world
                g = gathered[prev * rec:(prev + 1) * rec]
                b_kps[0] = g[:cap * 28].view(torch.float32).reshape(cap, 7)
                b_desc[0] = g[cap * 28:cap * 60].reshape(cap, 32)
                b_n[0:1] = g[cap * 60:].view(torch.int32)
                b_kps[1], b_desc[1], b_n[1:2] = d_kps[b][0], d_desc[b][0], d_n[b][0:1]
                mt2.match_consecutive_device(b_kps.data_ptr(), b_desc.data_ptr(), b_n.data_ptr(), cap, 2, W, H, 15.0, sf, b_assigned.data_ptr(),
                                             b_counts.data_ptr(), None, sB.cuda_stream)
        ev_matched[b].record(sB)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # per-stage device time is taken LIVE in the timed region: every handle records HIP events around its stages on the stream it
    # is launched on (the library reads a launch's events at the next launch if they have completed, never waiting)
    for h_ in exs:
        h_.set_profiling(True)
    for m_ in mts:
        m_.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    for m_ in mts:
        m_.synchronize()  # surfaces a record-pool overflow, if any (and reads the last launch's stage events)
    live = {}
    for h_ in exs:
        for k_, v_ in h_.stage_times().items():
            live.setdefault(k_, []).append(v_)
    for m_ in mts:
        for k_, v_ in m_.stage_times().items():
            if v_ > 0:
                live.setdefault(k_, []).append(v_)
    live = {k_: float(np.mean(v_)) for k_, v_ in live.items() if len(v_)}
    for h_ in exs:
        h_.set_profiling(False)
    for m_ in mts:
        m_.set_profiling(False)
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    kp_local = int(d_n[0].sum().item())
    matched_local = int(d_counts[0].sum().item())
    kp_all = torch.tensor([kp_local], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce_(t_all, dist.ReduceOp.MAX)
        all_reduce_(kp_all, dist.ReduceOp.SUM)
    dt = float(t_all.item())
    kp_total = float(kp_all.item())
    value = kp_total * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # ---- extract only (both figures): the same launches without the matcher, after the timed region ------------------
    def extract_step():
        for i, (f0, f1) in enumerate(parts):
            exs[i].extract_batch_device(d_img[f0].data_ptr(), W, H, W, W * H, f1 - f0, d_kps[0][f0].data_ptr(), d_desc[0][f0].data_ptr(), cap,
                                        d_n[0][f0:].data_ptr(), sAs[i].cuda_stream)
    extract_step()
    torch.cuda.synchronize()
    te = time.perf_counter()
    for _ in range(max(args.steps // 2, 2)):
        extract_step()
    torch.cuda.synchronize()
    te = (time.perf_counter() - te) / max(args.steps // 2, 2)
    extract_only = {"value": kp_local / te / 1e6 * world, "unit": "Mkeypoints/s", "ms_per_step": te * 1e3,
                    "note": "rank 0's clock; extraction launches only, same handles and streams"}

    # ---- roofline of the dominant kernel: per-stage device time, HIP events on the launch stream -------------------
    FL = parts[0][1] - parts[0][0]   # frames per extractor launch in the timed run
    ex2 = y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=FL)
    ex2.set_profiling(True)
    mt.set_profiling(True)
    for _ in range(5):
        ex2.extract_batch(imgs[:FL])
    for _ in range(5):
        mt.match_consecutive_device(d_kps[0].data_ptr(), d_desc[0].data_ptr(), d_n[0].data_ptr(), cap, F, W, H, 15.0, sf, d_assigned[0].data_ptr(),
                                    d_counts[0].data_ptr())
        mt.synchronize()
    mt.match_consecutive_device(d_kps[0].data_ptr(), d_desc[0].data_ptr(), d_n[0].data_ptr(), cap, F, W, H, 15.0, sf, d_assigned[0].data_ptr(),
                                d_counts[0].data_ptr())
    mt.synchronize()
    isolated = dict(ex2.stage_times())       # the same stages with nothing else on the GPU (one handle, no overlap): for reference
    isolated.update(mt.stage_times())
    mt.set_profiling(False)
    stages = dict(isolated)
    stages.update(live)                      # the roofline uses the live durations
    dom = max(stages, key=stages.get)
    A_frame = algorithmic_bytes_extract(W, H, NFEAT)
    n_kp_frame = kp_local / F
    pyr_pad = A_frame - W * H - NFEAT * 60
    # algorithmic bytes of each kernel per frame (DESIGN.md "Kernels"): what it must read + write once
    kbytes = {
        "pyramid": W * H + pyr_pad,                       # read image, write padded pyramid
        "fast_cells": pyr_pad,                            # read every pyramid pixel once (candidates are << 1 %)
        "quadtree_after_blur": 8 * 4200 * 4,                         # read ~4.2 k packed candidates per level (measured average), write keypoints
        "blur": 2 * (A_frame - W * H - NFEAT * 60),       # read pyramid, write blurred levels
        "orient_describe": int(n_kp_frame) * (60 + 2 * 1849),  # 43x43 patch of the level and of the blurred level + 60 B out
        "grid_build": int(n_kp_frame) * (28 + 4),
        "gather_distances": int(n_kp_frame) * (40 + 32 + 24 * 36),  # query + descriptor + ~24 candidates x (32 B descriptor + 4 B record)
        "resolve": int(n_kp_frame) * 24 * 4,
    }
    t_dom = stages[dom] * 1e-3
    launch_frames = F if dom in ("grid_build", "gather_distances", "resolve") else FL
    achieved = kbytes.get(dom, A_frame) * launch_frames / t_dom if t_dom > 0 else 0.0
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01f_pmc_hbm_traffic.csv: separate
    # FETCH_SIZE / WRITE_SIZE runs of the same kernels at 128 frames per launch; FETCH under-counts this 4-byte access pattern
    # by 1.33x, calibrated on k_pyr_level0's known read size; WRITE_SIZE is exact) scaled to this run's frames per launch.
    traffic = None
    try:
        import csv
        stage_kernels = {"pyramid": ("k_pyr_level0", "k_pyr_resize", "k_pyr_borders"), "fast_cells": ("k_fast_cells",), "blur": ("k_blur",),
                         "quadtree_after_blur": ("k_quadtree_flat", "k_quadtree"), "orient_describe": ("k_orient_describe",)}
        rows = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01f_pmc_hbm_traffic.csv")))}
        if dom in stage_kernels:
            per_frame = sum(float(rows[k]["fetch_MB_per_frame_raw"]) * 1.33 + float(rows[k]["write_MB_per_frame"]) for k in stage_kernels[dom])
            traffic = per_frame * 1e6 * (F / NEX)
    except Exception:  # noqa: BLE001
        traffic = None
    # what actually bounds the dominant kernel: share of the launch's SIMD cycles spent issuing vector instructions
    # (4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), committed SQ pass profiles/r01f_pmc_sq_valu.csv)
    valu_busy = None
    try:
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01f_pmc_sq_valu.csv"))):
            if dom in stage_kernels and r["kernel"] == stage_kernels[dom][0]:
                valu_busy = float(r["VALU_busy_pct_of_SIMD_cycles"]) / 100.0
    except Exception:  # noqa: BLE001
        valu_busy = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic, "valu_busy_frac": valu_busy,
                "pipeline_achieved_GBps": A_frame * F * args.steps / dt / 1e9, "pipeline_frac": A_frame * F * args.steps / dt / HBM_PEAK,
                "stage_ms_per_launch": {k: round(v, 4) for k, v in stages.items()},
                "stage_ms_per_launch_isolated": {k: round(v, 4) for k, v in isolated.items()},
                "frames_per_extract_launch": FL, "frames_per_match_launch": F}

    out = {"metric": "ORB extract+match Mkeypoints/sec", "value": value, "unit": "Mkeypoints/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": "TUM-fr1-size 640x480 mono stream, 1000 feat/frame, extract + consecutive-frame searchByProjection",
                      "frames_per_step_per_gpu": F, "extractor_handles": NEX, "distinct_frames": len(distinct), "keypoints_per_frame": n_kp_frame,
                      "matches_per_pair": matched_local / max(F - 1, 1), "parallelism": "frames sharded x%d" % world},
           "roofline": roofline}

    # ---- local BA (config 5) ------------------------------------------------------------------------------------------
    out["extract_only"] = extract_only
    if not args.no_ba:
        prob = synth_ba_problem(100, 10000, 8, seed=1)
        if world > 1:  # shard landmarks (and their edges) across ranks; every rank holds all poses
            from ydorbslam_amd.parallel import shard_ba_problem
            prob, _, _ = shard_ba_problem(prob, rank, world)
            comm = torch.zeros(640 * 641 + 4096, dtype=torch.float64, device=dev)  # >= n*n + n doubles, n = 6*K rounded up to 32

            def allreduce(user, d_buf, count, op):
                try:
                    all_reduce_(comm[:count], dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
                    torch.cuda.synchronize()
                    return 0
                except Exception:  # noqa: BLE001
                    return 1
            kw = dict(allreduce=allreduce, comm_tensor_ptr=comm.data_ptr(), comm_doubles=comm.numel(), rank=rank, world=world)
        else:
            kw = {}
        opt = y.Optimizer.default_options(device=local_rank)
        y.Optimizer.local_bundle_adjust(prob, opt, **kw)  # warm-up (allocations, code objects)
        barrier()
        t0 = time.perf_counter()
        reps = 3
        trials = 0
        for _ in range(reps):
            r = y.Optimizer.local_bundle_adjust(prob, opt, **kw)
            trials += r["trials"]
        barrier()
        tb = time.perf_counter() - t0
        tb_all = torch.tensor([tb], dtype=torch.float64, device=dev)
        if world > 1:
            all_reduce_(tb_all, dist.ReduceOp.MAX)
        tb = float(tb_all.item())
        flops_schur = 89.9e6  # Schur part of one LM trial at C5 / 8 obs
        ms = r["ms"]
        out["ba"] = {"metric": "local-BA LM iterations/sec (100 KF x 10k points, 8 obs/point)", "value": trials / tb, "unit": "it/s",
                     "lm_trials_per_solve": r["trials"], "ms_per_solve": tb / reps * 1e3, "final_chi2": float(r["log"][-1, 0]),
                     "device_ms_per_solve": {k: round(float(v), 3) for k, v in ms.items()},
                     "schur_fp64_frac": (flops_schur * r["trials"] / (ms["schur"] * 1e-3) / FP64_VEC_PEAK) if ms["schur"] > 0 else None,
                     "scaling": "strong (landmarks sharded, all-reduce of the reduced camera system)" if world > 1 else "single GPU"}
        if world == 1:
            # Additional figure: several independent local-BA problems at once, one host thread each (ctypes drops the
            # GIL; the library keeps a pool of per-device contexts).  One solve is a latency chain that leaves the GPU mostly
            # idle, so concurrent maps / sessions overlap almost freely.
            NT = args.ba_threads
            probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(2 * NT)]
            y.Optimizer.local_bundle_adjust_batch(probs[:NT], opt, NT)
            tcc = time.perf_counter()
            bres = y.Optimizer.local_bundle_adjust_batch(probs, opt, NT)
            tcc = time.perf_counter() - tcc
            out["ba"]["concurrent"] = {"problems": len(probs), "in_flight": NT, "value": sum(b_["trials"] for b_ in bres) / tcc, "unit": "it/s (aggregate)",
                                       "note": "ydorb_ba_solve_batch: independent copies of the same C5 problem, one library thread and one pooled context each"}

    # ---- pose-only optimisation (Optimizer::optimizePose): a batch of frames per launch -------------------
    if not args.no_ba and world == 1:
        from ydorbslam_amd.synth import synth_pose_problem
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

    # ---- stereo association (Frame::computeStereoMatches): a batch of rectified pairs per call ----------------
    if not args.no_ba and world == 1:
        from ydorbslam_amd.synth import synth_stereo_pair
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
        out["stereo"] = {"metric": "computeStereoMatches pairs/sec (640x480, 1000 features per image, host keypoints in, depth out)",
                         "pairs_per_call": NSP}
        for name, by_kp in (("reference_replay", False), ("index_by_keypoint", True)):
            sm.stereo_matches(sex, sex, skl, sdl, snl, skr, sdr, snr, 40.0, 0.1, by_kp, (0, 2), (1, 2))
            ts = time.perf_counter()
            for _ in range(5):
                sout = sm.stereo_matches(sex, sex, skl, sdl, snl, skr, sdr, snr, 40.0, 0.1, by_kp, (0, 2), (1, 2))
            ts = (time.perf_counter() - ts) / 5
            out["stereo"][name] = {"value": NSP / ts, "unit": "pairs/s", "ms_per_call": ts * 1e3, "measurements_per_pair": float(np.mean(sout[2]))}

    # ---- vocabulary transform (DBoW3::Vocabulary::transform): BowVector + FeatureVector per frame ----------------
    if not args.no_ba and world == 1:
        from ydorbslam_amd.synth import synth_vocabulary
        vtree = synth_vocabulary(10, 5, seed=1)   # k = 10 like the ORB vocabulary, one level less (L = 6 would be 35 MB of synthetic nodes)
        voc = y.Vocabulary(vtree)
        NBF = min(256, F)
        hn = d_n[0][:NBF].cpu().numpy()
        hdesc = d_desc[0][:NBF].cpu().numpy()
        bdescs = [hdesc[f, :hn[f]] for f in range(NBF)]
        voc.transform(bdescs, 3)
        tv = time.perf_counter()
        for _ in range(3):
            bout = voc.transform(bdescs, 3)
        tv = (time.perf_counter() - tv) / 3
        out["bow_transform"] = {"metric": "Vocabulary::transform frames/sec (1000 descriptors per frame, k=10 L=5 synthetic tree, levelsup 3; host descriptors in, host vectors out)",
                                "frames_per_call": NBF, "tree_nodes": int(len(vtree["node_word"])), "value": NBF / tv, "unit": "frames/s",
                                "ms_per_call": tv * 1e3, "mean_words_per_frame": float(np.mean([len(b[0]) for b in bout]))}

    # ---- distinctive descriptors (MapPoint::computeDistinctiveDescriptors): a batch of map points per call -----------
    if not args.no_ba and world == 1:
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

    # ---- CPU baseline: the oracle (port of the reference algorithm), one thread, bounded sample -------------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle.orb_oracle import FrameOracle, OrbExtractorOracle, QUERY_DTYPE, ba_solve
        oex = OrbExtractorOracle(NFEAT, 1.2, 8, 20, 7)
        ncpu = min(args.cpu_frames, F)
        tc = time.perf_counter()
        prev = None
        nk = 0
        for i in range(ncpu):
            k, d = oex.extract(imgs[i])
            nk += len(k)
            if prev is not None:
                pk, pd = prev
                q = np.zeros(len(pk), QUERY_DTYPE)
                q["u"], q["v"] = pk["x"], pk["y"]
                q["r"] = (np.float32(15.0) * sf[pk["octave"]]).astype(np.float32)
                q["min_level"], q["max_level"] = pk["octave"] - 1, pk["octave"] + 1
                q["angle"], q["level"], q["flags"] = pk["angle"], pk["octave"], 3
                FrameOracle(k, d, (0.0, float(W), 0.0, float(H))).search_by_projection(1, q, pd, 0.9, True)
            prev = (k, d)
        tc = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": nk / tc / 1e6, "unit": "Mkeypoints/s", "cores": 1, "kind": "port",
                               "sample": "%d of the same 640x480 frames, extract + consecutive match, oracle (C++ -O2), 1 thread" % ncpu}
        if not args.no_ba:
            pb = synth_ba_problem(100, 10000, 8, seed=1)
            tcb = time.perf_counter()
            rb = ba_solve(pb)
            tcb = time.perf_counter() - tcb
            out["ba"]["cpu_baseline"] = {"value": rb["trials"] / tcb, "unit": "it/s", "cores": 1, "kind": "port",
                                         "sample": "one full localBundleAdjust schedule (%d LM trials) on the same problem" % rb["trials"]}
            out["ba"]["vs_cpu"] = out["ba"]["value"] / out["ba"]["cpu_baseline"]["value"]
            from oracle.orb_oracle import pose_optimize as oracle_pose_optimize
            tpc = time.perf_counter()
            for i in range(32):
                oracle_pose_optimize(pprobs[i])
            tpc = (time.perf_counter() - tpc) / 32
            out["pose_optimize"]["cpu_baseline"] = {"value": 1.0 / tpc, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "32 of the same frames"}
            from oracle.orb_oracle import stereo_matches as oracle_stereo
            oel, oer = OrbExtractorOracle(NFEAT, 1.2, 8, 20, 7), OrbExtractorOracle(NFEAT, 1.2, 8, 20, 7)
            tsc, nsc = 0.0, 0
            for p in range(4):
                kl_, dl_ = oel.extract(spairs[p][0]); kr_, dr_ = oer.extract(spairs[p][1])
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
                oracle_bow(vtree, bdescs[f], 3, 0, 1)
            tvc = (time.perf_counter() - tvc) / 16
            out["bow_transform"]["cpu_baseline"] = {"value": 1.0 / tvc, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "16 of the same frames"}
            from oracle.orb_oracle import distinctive_descriptor as oracle_dd
            tdc = time.perf_counter()
            ok_d = all(oracle_dd(groups_d[i]) == best_d[i] for i in range(5000))
            tdc = (time.perf_counter() - tdc) / 5000
            out["distinctive_descriptors"]["cpu_baseline"] = {"value": 1.0 / tdc, "unit": "points/s", "cores": 1, "kind": "port",
                                                              "sample": "5000 of the same points (results equal: %s)" % ok_d}
        out["vs_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
