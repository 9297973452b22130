#!/usr/bin/env python3
"""Headline benchmark: ORB extract+match Mkeypoints/s (BASELINE.json config 2: 640x480 mono stream, 1000 features/frame)
and local-BA LM iterations/s (config 5: 100 keyframes x 10 000 points) on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic frames that already sit in HBM:
  ydorb_extract_batch_device (pyramid -> FAST cells -> quad-tree -> blur -> orientation + rBRIEF)
  + ydorb_match_pairs_device (grid build -> candidate distances -> ordered resolve): every owned frame is searched for the
  keypoints of its predecessor in the stream (searchByProjectionInLastAndCurrentFrame rules, th = 15), predicted with the known
  inter-frame motion (all frames distinct, frame t+1 = frame t after a small roll / shift).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (fresh child processes, before
anything touches a GPU); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  N > 1: one process per GPU
(torch.distributed / RCCL).  The frames of the stream are dealt round-robin (global frame g lives on rank g % N), so EVERY
consecutive pair straddles two GPUs: per step each rank all-gathers every rank's [keypoints | descriptors | count] records of
the step and then matches, locally against the gathered set, the pairs whose later frame it owns.  Per-GPU
work is fixed as N grows (weak scaling); the gather grows with N.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch stream; `cpu_baseline` times the CPU
oracle (a port of the reference algorithm) on a bounded sample of the same frames, on one thread and on all host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT = 640, 480, 1000
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
FP64_VEC_PEAK = 78.6e12    # FLOP/s, MI355X FP64 vector / matrix datasheet figure (the in-container guide lists no FP64 number)
# integer-VALU peak for the brute-force figure: 256 CUs x 4 SIMDs x 32 lanes per cycle at the 2.4 GHz maximum clock
INT_VALU_PEAK = 256 * 4 * 32 * 2.4e9
PARITY_NOTE = ("bit-exact vs the build's CPU restatement of the reference algorithm (oracle/); oracle unpinned for extractor / "
               "matcher / stereo / BoW: OpenCV absent, the reference holds no fixture; BA solver pinned by g2o's own linear-system vector only")


def algorithmic_bytes_extract(w, h, n):
    """image read + padded pyramid written (public output) + keypoints/descriptors written."""
    import numpy as np
    tot = w * h + n * 60
    for l in range(8):
        inv = np.float32(pow(np.float32(1.2), -l))
        wl, hl = int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))
        tot += (wl + 38) * (hl + 38)
    return tot


def launch_ranks(n):
    """--gpus N without a launcher: start N ranks as fresh children (this parent never touches a GPU) and relay rank 0's line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    try:   # a container's CPU share (cgroup v2 cpu.max = "<quota> <period>" or "max <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            usable = max(1, min(usable, int(round(int(q) / int(per)))))
    except (OSError, ValueError):
        try:   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                usable = max(1, min(usable, int(round(q / per))))
        except (OSError, ValueError):
            pass
    return model, os.cpu_count() or 1, usable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=512, help="frames per step and per GPU")
    ap.add_argument("--segment", type=int, default=64, help="frames per synthetic scene (a new scene is a cut)")
    ap.add_argument("--cpu-frames", type=int, default=96, help="frames of the 1-thread CPU-oracle sample")
    ap.add_argument("--extractors", type=int, default=1,
                    help="extractor handles (each with its own stream) the frames of a step are split over; 1 is fastest with the matcher beside it "
                         "(4 hardware queues: more streams share queues and serialise): 163 Mkeypoints/s against 158 with 2 handles")
    ap.add_argument("--alternate", type=int, default=int(os.environ.get("YDORB_BENCH_ALTERNATE", "2")),
                    help="N >= 2: N lanes (extractor handle + matcher + stream each) take consecutive STEPS, a step's extraction and matching back to "
                         "back on its lane's stream, no events between lanes; 0: one handle, matcher on a second stream")
    ap.add_argument("--ba-threads", type=int, default=64, help="problems of the lock-step batched local-BA figure (ydorb_ba_solve_batch)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="", help="with extras: run only these sections (comma list of single_call, pcie, config3, config4, rest)")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-call / PCIe-inclusive / config 3 / config 4 / brute-force sections")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)
        return

    # Only the JSON line may reach stdout: RCCL prints its own warnings there (e.g. "Missing iommu=pt"), so everything any library
    # writes to file descriptor 1 goes to stderr from here on and the result line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
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
    # YDORB_BENCH_FORCE_DIST=1: initialise the process group and run the collectives even at world size 1 (an RCCL rehearsal of the
    # N > 1 code path - in-place all-gather, reductions, barrier - on a one-GPU box)
    force_dist = world == 1 and bool(os.environ.get("YDORB_BENCH_FORCE_DIST"))
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
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

    def all_gather_inplace(full, mine):
        """full: [world * k, ...] tensor whose slice [rank*k, (rank+1)*k) is `mine` (the extractor wrote it there): no packing copy."""
        if staged:
            parts = [torch.zeros_like(mine, device="cpu") for _ in range(world)]
            dist.all_gather(parts, mine.cpu())
            full.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(full, mine)

    import ydorbslam_amd as y
    from ydorbslam_amd.synth import stream_plan, stream_render, synth_ba_problem

    # The working streams are created FIRST, before any handle creates streams of its own: the device runs 4 hardware queues and HIP deals
    # streams onto them in creation order, so the first four land on four different queues (profiles/r02d_queue_overlap.txt).
    early_streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    F = args.frames
    G = F * world                                   # frames of the global stream per step
    # every frame distinct; frame g+1 = frame g after a small known motion (scene cut every --segment frames).
    # Round-robin ownership: global frame g = t * world + rank is this rank's frame t.
    plan = stream_plan(W, H, G, seed=0, segment=args.segment)
    own = [t * world + rank for t in range(F)]
    imgs, _ = stream_render(plan, own)
    NEX = max(1, min(args.extractors, F // 8))
    parts = [(i * F // NEX, (i + 1) * F // NEX) for i in range(NEX)]
    LANES = max(2, args.alternate)
    # YDORB_BENCH_RESOLVE_STREAM=1: the serial resolve of a step runs on a side stream, the lane goes on with its next step at once; a
    # lane then alternates between two output sets (+ matchers), so the resolve of step k and the extraction of step k + LANES never share buffers
    side_resolve = bool(int(os.environ.get("YDORB_BENCH_RESOLVE_STREAM", "0"))) and args.alternate > 0
    NSET = LANES * (2 if side_resolve else 1)      # output sets
    single = bool(int(os.environ.get("YDORB_BENCH_SINGLE_STREAM", "0")))
    if args.alternate:
        NEX, parts = LANES, [(0, F)]
        exs = [y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=F, single_stream=single) for _ in range(LANES)]
    else:
        exs = [y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=b - a) for a, b in parts]
    ex = exs[0]
    cap = ex.max_keypoints
    sf = ex.tables()["scale"]
    d_img = torch.from_numpy(imgs).to(dev)
    # Two output sets + explicit streams: extraction of step k+1 overlaps the matching of step k.  An output set is laid out as the
    # GATHERED set [world * F][cap]; this rank's extractors write straight into its slice.
    # (The default stream's handle is 0, which the C ABI reads as "use the handle's own stream": always pass real streams.)
    g_kps = [torch.zeros((G, cap, 7), dtype=torch.float32, device=dev) for _ in range(NSET)]
    g_desc = [torch.zeros((G, cap, 32), dtype=torch.uint8, device=dev) for _ in range(NSET)]
    g_n = [torch.zeros(G, dtype=torch.int32, device=dev) for _ in range(NSET)]
    lo = rank * F
    d_kps = [t[lo:lo + F] for t in g_kps]
    d_desc = [t[lo:lo + F] for t in g_desc]
    d_n = [t[lo:lo + F] for t in g_n]
    # pairs (query = predecessor in the global stream, target = an owned frame), as indices into the gathered (rank-major) set
    from ydorbslam_amd.parallel import round_robin_pairs
    pairs, pred_idx = round_robin_pairs(rank, world, F)
    pair_aff = plan["predicted"][pred_idx]
    NPAIR = len(pairs)
    d_aff = torch.from_numpy(np.ascontiguousarray(pair_aff, np.float32)).to(dev)
    d_assigned = [torch.zeros((NPAIR, cap), dtype=torch.int32, device=dev) for _ in range(NSET)]
    d_counts = [torch.zeros(NPAIR, dtype=torch.int32, device=dev) for _ in range(NSET)]
    side_mode = int(os.environ.get("YDORB_BENCH_SIDE_STREAMS", "0")) if args.alternate == 2 else 0   # 1-3: caller-provided quad-tree side streams (no gain measured: 186-191 against 191-198 Mkeypoints/s)
    if side_mode:
        # two lanes + one caller-provided quad-tree side stream per lane = the four queues
        sAs, sB = early_streams[:2], early_streams[4]
        for i_, h_ in enumerate(exs):
            if side_mode == 1:     # one side stream per lane
                h_.set_side_streams([early_streams[2 + i_].cuda_stream])
            elif side_mode == 2:   # two side streams shared by both lanes
                h_.set_side_streams([early_streams[2].cuda_stream, early_streams[3].cuda_stream])
            else:                  # 3: two side streams per lane, the second pair on the queues of the first (creation order 6, 7 -> queues of 2, 3)
                h_.set_side_streams([early_streams[2 + 4 * i_].cuda_stream, early_streams[3 + 4 * i_].cuda_stream])
    else:
        sAs, sB = [torch.cuda.Stream(device=dev) for _ in range(NEX)], torch.cuda.Stream(device=dev)
    ev_extracted = [[torch.cuda.Event() for _ in range(NEX)] for _ in range(NSET)]
    ev_matched = [torch.cuda.Event() for _ in range(NSET)]
    for e in ev_matched:
        e.record(sB)
    mts = [y.OrbMatcher(0.9, True, device=local_rank) for _ in range(NSET)]   # one matcher (own scratch) per output set
    step_no = [0]

    s_res = torch.cuda.Stream(device=dev) if side_resolve else None
    ev_res = [torch.cuda.Event() for _ in range(NSET)]
    if side_resolve:
        for m_ in mts:
            m_.set_resolve_stream(s_res.cuda_stream)
        for e_ in ev_res:
            e_.record(s_res)

    def step_alternate():
        # output set b belongs to handle b and stream b: step k's extraction AND matching run on stream k & 1, back to back; the two
        # streams overlap freely (no events between them), so one stream's latency-bound kernels run beside the other's busy ones
        k_ = step_no[0]
        step_no[0] += 1
        lane = k_ % LANES
        b = lane + LANES * ((k_ // LANES) & 1) if side_resolve else lane       # output set (and matcher) of this step
        sA = sAs[lane]
        if side_resolve:
            sA.wait_event(ev_res[b])       # the resolve that last read this set (two of the lane's steps ago) is done
        exs[lane].extract_batch_device(d_img.data_ptr(), W, H, W, W * H, F, d_kps[b].data_ptr(), d_desc[b].data_ptr(), cap, d_n[b].data_ptr(), sA.cuda_stream)
        with torch.cuda.stream(sA):
            if world > 1 or force_dist:
                all_gather_inplace(g_kps[b], d_kps[b])
                all_gather_inplace(g_desc[b], d_desc[b])
                all_gather_inplace(g_n[b], d_n[b])
            gs = (g_kps[b].data_ptr(), g_desc[b].data_ptr(), g_n[b].data_ptr(), G, cap)
            mts[b].match_pairs_device(gs, gs, pairs, W, H, 15.0, sf, d_assigned[b].data_ptr(), d_counts[b].data_ptr(), d_aff.data_ptr(), sA.cuda_stream)
        if side_resolve:
            ev_res[b].record(s_res)

    def step():
        if args.alternate:
            return step_alternate()
        b = step_no[0] & 1
        step_no[0] += 1
        for i, (f0, f1) in enumerate(parts):
            sA = sAs[i]
            sA.wait_event(ev_matched[b])          # the matcher (and the gather) that last used this output set is done
            exs[i].extract_batch_device(d_img[f0].data_ptr(), W, H, W, W * H, f1 - f0, d_kps[b][f0].data_ptr(), d_desc[b][f0].data_ptr(), cap,
                                        d_n[b][f0:].data_ptr(), sA.cuda_stream)
            ev_extracted[b][i].record(sA)
            sB.wait_event(ev_extracted[b][i])
        with torch.cuda.stream(sB):
            if world > 1 or force_dist:   # all-gather of every rank's records of the step, three large collectives, no packing
                all_gather_inplace(g_kps[b], d_kps[b])
                all_gather_inplace(g_desc[b], d_desc[b])
                all_gather_inplace(g_n[b], d_n[b])
            gs = (g_kps[b].data_ptr(), g_desc[b].data_ptr(), g_n[b].data_ptr(), G, cap)
            mts[b].match_pairs_device(gs, gs, pairs, W, H, 15.0, sf, d_assigned[b].data_ptr(), d_counts[b].data_ptr(), d_aff.data_ptr(), sB.cuda_stream)
        ev_matched[b].record(sB)

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
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
    for h_ in exs:
        h_.synchronize()  # surfaces a quad-tree capacity status, if any
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
    if world > 1 or force_dist:
        all_reduce_(t_all, dist.ReduceOp.MAX)
        all_reduce_(kp_all, dist.ReduceOp.SUM)
    dt = float(t_all.item())
    kp_total = float(kp_all.item())
    value = kp_total * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # ---- extract only (both figures): the same launches without the matcher, after the timed region ------------------
    def extract_step():
        if args.alternate:
            b = step_no[0] % LANES
            step_no[0] += 1
            exs[b].extract_batch_device(d_img.data_ptr(), W, H, W, W * H, F, d_kps[b].data_ptr(), d_desc[b].data_ptr(), cap, d_n[b].data_ptr(), sAs[b].cuda_stream)
            return
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
    mt = mts[0]
    ex2 = y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=FL)
    ex2.set_profiling(True)
    mt.set_profiling(True)
    for _ in range(5):
        ex2.extract_batch(imgs[:FL])
    gs0 = (g_kps[0].data_ptr(), g_desc[0].data_ptr(), g_n[0].data_ptr(), G, cap)
    for _ in range(6):
        mt.match_pairs_device(gs0, gs0, pairs, W, H, 15.0, sf, d_assigned[0].data_ptr(), d_counts[0].data_ptr(), d_aff.data_ptr())
        mt.synchronize()
    isolated = dict(ex2.stage_times())       # the same stages with nothing else on the GPU (one handle, no overlap): for reference
    isolated.update(mt.stage_times())
    mt.set_profiling(False)
    del ex2
    stages = dict(isolated)
    stages.update(live)                      # the roofline uses the live durations
    ext_stages = ("pyramid", "fast_cells", "blur", "orient_describe")
    dom = max((k_ for k_ in stages if k_ in ext_stages), key=stages.get)   # the dominant extractor kernel (see the note at stage_kernels)
    A_frame = algorithmic_bytes_extract(W, H, NFEAT)
    n_kp_frame = kp_local / F
    pyr_pad = A_frame - W * H - NFEAT * 60
    # algorithmic bytes of each kernel per frame (DESIGN.md "Kernels"): what it must read + write once
    kbytes = {
        "pyramid": W * H + pyr_pad,                       # read image, write padded pyramid
        "fast_cells": pyr_pad,                            # read every pyramid pixel once (candidates are << 1 %)
        "quadtree_after_blur": 8 * 4200 * 4,              # read ~4.2 k packed candidates per level (measured average), write keypoints
        "blur": 2 * pyr_pad,                              # read pyramid, write blurred levels
        "orient_describe": int(n_kp_frame) * (60 + 2 * 1849),  # 43x43 patch of the level and of the blurred level + 60 B out
        "grid_build": int(n_kp_frame) * (28 + 4),
        "gather_distances": int(n_kp_frame) * (40 + 32 + 24 * 36),  # query + descriptor + ~24 candidates x (32 B descriptor + 4 B record)
        "resolve": int(n_kp_frame) * 24 * 4,
    }
    t_dom = stages[dom] * 1e-3
    launch_frames = NPAIR if dom in ("grid_build", "gather_distances", "resolve") else FL
    achieved = kbytes.get(dom, A_frame) * launch_frames / t_dom if t_dom > 0 else 0.0
    # HBM bytes per launch of the dominant kernel: NOT measured in this run (PMC counters need rocprofv3 around the process).  They
    # come from the committed counter passes of the same kernels (separate FETCH_SIZE / WRITE_SIZE runs; FETCH under-counts this
    # 4-byte access pattern by 1.33x, calibrated on k_pyr_level0's known read size; WRITE_SIZE is exact), scaled to this run's frames
    # per launch; `traffic_source` names the file.
    traffic, valu_busy, traffic_source = None, None, None
    # a stage's kernels = the rows of the counter CSV whose name starts with one of these (k_pyr_level0_f / k_pyr_resize_f, or the
    # interior-only kernels + k_pyr_borders of a plan that cannot fuse its pads; k_orient_describe or k_orient_describe_n)
    stage_kernels = {"pyramid": ("k_pyr_",), "fast_cells": ("k_fast_cells",), "blur": ("k_blur",),
                     "quadtree_after_blur": ("k_quadtree",), "orient_describe": ("k_orient_describe",)}
    # The roofline line names the dominant EXTRACTOR kernel: the stage timers of the two pipelines overlap, and of the matcher's stages
    # the resolve is serial by definition (one wave per pair) - neither is a bandwidth figure.
    try:
        import csv
        for tag in ("r02d", "r02", "r01f"):
            pth = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.csv" % tag)
            if not os.path.exists(pth):
                continue
            rows = {r["kernel"]: r for r in csv.DictReader(open(pth))}
            mine = [k for k in rows if dom in stage_kernels and k.startswith(stage_kernels[dom])]
            if mine:
                per_frame = sum(float(rows[k]["fetch_MB_per_frame_raw"]) * float(rows[k].get("fetch_correction") or 1.33) + float(rows[k]["write_MB_per_frame"])
                                for k in mine)
                traffic = per_frame * 1e6 * FL
                traffic_source = ("profiles/%s_pmc_hbm_traffic.csv (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the extractor alone, FETCH calibrated "
                                  "on k_pyr_level0's known read size, per frame, scaled to %d frames per launch; not collected in this run)" % (tag, FL))
            pth2 = os.path.join(ROOT, "profiles", "%s_pmc_sq_valu.csv" % tag)
            if os.path.exists(pth2):
                for r in csv.DictReader(open(pth2)):
                    if dom in stage_kernels and r["kernel"].startswith(stage_kernels[dom]) and valu_busy is None:
                        valu_busy = float(r["VALU_busy_pct_of_SIMD_cycles"]) / 100.0
            if traffic is not None:
                break
    except Exception:  # noqa: BLE001
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                # the same kernel with nothing else on the GPU (one handle, one stream; HIP events, measured after the timed region)
                "achieved_alone": (kbytes.get(dom, A_frame) * launch_frames / (isolated[dom] * 1e-3) / 1e9) if isolated.get(dom) else None,
                "frac_alone": (kbytes.get(dom, A_frame) * launch_frames / (isolated[dom] * 1e-3) / HBM_PEAK) if isolated.get(dom) else None,
                "traffic": traffic, "traffic_source": traffic_source, "valu_busy_frac": valu_busy,
                "valu_busy_source": traffic_source and traffic_source.replace("hbm_traffic", "sq_valu"),
                "pipeline_achieved_GBps": A_frame * F * args.steps / dt / 1e9, "pipeline_frac": A_frame * F * args.steps / dt / HBM_PEAK,
                "stage_note": "a stage = every launch of its kernels in one extractor call over frames_per_extract_launch frames (fast_cells: two launches, "
                              "level 0 then levels 1-7; pyramid: one launch per level), HIP events on the launch stream IN the pipelined run, where a "
                              "second lane's kernels share the GPU (isolated = the same call with nothing else running); profiles/*_512frame_launches.txt "
                              "has rocprofv3's per-launch durations of the same command",
                "stage_ms_per_launch": {k: round(v, 4) for k, v in stages.items()},
                "stage_ms_per_launch_isolated": {k: round(v, 4) for k, v in isolated.items()},
                "frames_per_extract_launch": FL, "pairs_per_match_launch": NPAIR}

    out = {"metric": "ORB extract+match Mkeypoints/sec", "value": value, "unit": "Mkeypoints/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic", "parity": "partial", "parity_note": PARITY_NOTE,
           "config": {"workload": "TUM-fr1-size 640x480 mono stream, 1000 feat/frame, extract + consecutive-frame searchByProjection (th 15); frames resident in HBM",
                      "frames_per_step_per_gpu": F, "extractor_handles": NEX, "distinct_frames": F * world, "frames_per_scene": args.segment,
                      "pipelining": ("%d lanes (extractor handle + matcher + stream each) take consecutive steps; a step's extraction and matching run back to back "
                                     "on its lane's stream, no events between lanes" % NEX) if args.alternate else
                                    "one extractor handle; the matcher of step k on a second stream under the extraction of step k + 1",
                      "motion": "per frame: roll within +-3 deg, shift within +-8 px (bounded walk); prediction = true motion + N(0,1.5^2) px on the translation",
                      "keypoints_per_frame": n_kp_frame, "matches_per_pair": matched_local / max(NPAIR, 1),
                      "parallelism": ("frames dealt round-robin x%d, all-gather of [kp|desc|n] per step, local match against the gathered set" % world)
                      if world > 1 else "1 GPU"},
           "roofline": roofline}
    out["extract_only"] = extract_only

    # ---- local BA (config 5) ------------------------------------------------------------------------------------------
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
        flops_chol = 72.7e6   # (6K)^3/3 + 2(6K)^2, same table
        # the per-phase breakdown comes from one more solve with YDORB_BA_PHASE_TIMES (its event pairs cost ~8 % of a solve: not in the timed ones)
        ms = y.Optimizer.local_bundle_adjust(prob, y.Optimizer.default_options(device=local_rank, phase_times=True), **kw)["ms"]
        out["ba"] = {"metric": "local-BA LM iterations/sec (100 KF x 10k points, 8 obs/point)", "value": trials / tb, "unit": "it/s",
                     "lm_trials_per_solve": r["trials"], "ms_per_solve": tb / reps * 1e3, "final_chi2": float(r["log"][-1, 0]),
                     "device_ms_per_solve": {k: round(float(v), 3) for k, v in ms.items()},
                     "schur_fp64_frac": (flops_schur * r["trials"] / (ms["schur"] * 1e-3) / FP64_VEC_PEAK) if ms["schur"] > 0 else None,
                     "solve_fp64_frac": (flops_chol * r["trials"] / (ms["solve"] * 1e-3) / FP64_VEC_PEAK) if ms["solve"] > 0 else None,
                     "fp64_note": "device_ms_per_solve.schur covers k_dinv + k_bd + k_bs + k_schur_pairs, .solve the Cholesky chain + both substitutions; the kernels' own rocprof durations are in profiles/",
                     "parity_note": "vs the oracle's restatement of g2o (unpinned end to end; dense solver pinned by g2o's linear_solver_test vector, tol 1e-6)",
                     "scaling": "strong (landmarks sharded, all-reduce of the reduced camera system)" if world > 1 else "single GPU"}
        if world == 1:
            # Additional figure: independent local-BA problems solved in lock step (ydorb_ba_solve_batch: one launch per
            # phase for all problems, blockIdx.z = problem).  A single solve is a latency chain; the batch is throughput-bound.
            NT = args.ba_threads
            probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(NT)]
            y.Optimizer.local_bundle_adjust_batch(probs, opt, NT)
            tcc = time.perf_counter()
            bres = y.Optimizer.local_bundle_adjust_batch(probs, opt, NT)
            tcc = time.perf_counter() - tcc
            out["ba"]["concurrent"] = {"problems": len(probs), "in_flight": NT, "value": sum(b_["trials"] for b_ in bres) / tcc, "unit": "it/s (aggregate)",
                                       "ms_per_batch": tcc * 1e3,
                                       "note": "ydorb_ba_solve_batch, lock-step batch: independent copies of the same C5 problem, every result bit-identical to its single solve"}

    extras = world == 1 and not args.no_extras
    only = [x for x in args.only.split(",") if x]
    want = lambda name: extras and (not only or name in only)
    pprobs = spairs = vtree = bdescs = groups_d = best_d = None

    # ---- what ONE call of the drop-in sees, host to host (frame.cpp:129, tracking.cpp:456, localMapping.cpp:140) ---------------
    if want('single_call'):
        def med_ms(fn, n=15):
            fn()
            ts = []
            for _ in range(n):
                t_ = time.perf_counter(); fn(); ts.append(time.perf_counter() - t_)
            return float(np.median(ts) * 1e3)
        ex1 = y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank)
        one = imgs[3]
        sc = {"extract_ms": med_ms(lambda: ex1.extract(one)),
              "extract_plus_pyramid_download_ms": med_ms(lambda: (ex1.extract(one), ex1.read_pyramid())),
              "note": "median host-to-host wall time of one call: ydorb_extract = H2D 307 KB + ~20 launches + D2H 60 KB; the adapter's "
                      "m_v_imagePyramid refresh adds one 1.3 MB device-to-host copy + host repacking (ydorb_extractor_read_pyramid)"}
        ka_, da_ = ex1.extract(imgs[3]); kb_, db_ = ex1.extract(imgs[4])
        q1 = np.zeros(len(ka_), y.QUERY_DTYPE)
        q1["u"], q1["v"] = ka_["x"], ka_["y"]
        q1["r"] = (np.float32(15.0) * sf[ka_["octave"]]).astype(np.float32)
        q1["min_level"], q1["max_level"] = ka_["octave"] - 1, ka_["octave"] + 1
        q1["angle"], q1["level"], q1["flags"] = ka_["angle"], ka_["octave"], 3
        fv1 = y.FrameView(kb_, db_, (0.0, float(W), 0.0, float(H)))
        m1 = y.OrbMatcher(0.9, True, device=local_rank)
        sc["search_by_projection_ms"] = med_ms(lambda: m1.search_by_projection(1, fv1, q1, da_))
        sc["keypoints_per_s_one_frame_at_a_time"] = len(ka_) / ((sc["extract_ms"] + sc["search_by_projection_ms"]) * 1e-3)
        if not args.no_ba:
            sc["ba_solve_ms"] = out["ba"]["ms_per_solve"]
            from ydorbslam_amd.synth import synth_pose_problem
            pp1 = [synth_pose_problem(400, seed=100)]
            sc["pose_optimize_ms"] = med_ms(lambda: y.Optimizer.optimize_poses(pp1))
        out["single_call"] = sc
        del ex1

    # ---- PCIe-inclusive: pinned host frames in, host keypoints / descriptors / matches out (SURVEY 8d "incl. H2D/D2H") -------------
    if want('pcie'):
        CH = 4                                              # chunks per step: copy of chunk c+1 overlaps the extraction of chunk c
        cf = F // CH
        h_img = torch.from_numpy(imgs).pin_memory()
        h_kps = torch.zeros((F, cap, 7), dtype=torch.float32).pin_memory()
        h_desc = torch.zeros((F, cap, 32), dtype=torch.uint8).pin_memory()
        h_n = torch.zeros(F, dtype=torch.int32).pin_memory()
        h_assigned = torch.zeros((NPAIR, cap), dtype=torch.int32).pin_memory()
        h_counts = torch.zeros(NPAIR, dtype=torch.int32).pin_memory()
        p_img = [torch.zeros_like(d_img) for _ in range(2)]   # device image buffers: the upload of step k+1 runs under the extraction of step k
        # two handles of their own, alternating over the chunks: the copy of chunk c+1 and the extraction of chunk c overlap
        pexs = [y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=local_rank, max_batch=cf) for _ in range(2)]
        psts = [torch.cuda.Stream(device=dev) for _ in range(2)]
        s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        ev_in = [torch.cuda.Event() for _ in range(CH)]   # (re-recorded per step: a stream waits for the latest record at the time of the wait call)
        ev_ex = [torch.cuda.Event() for _ in range(2 * CH)]
        ev_pm = [torch.cuda.Event() for _ in range(2)]       # matcher + read-back of an output set done
        for e_ in ev_ex:
            e_.record(sB)
        for e_ in ev_pm:
            e_.record(sB)
        pstep = [0]

        def pcie_step():
            b = pstep[0] & 1                                 # output set of this step: the matcher of step k reads set b while step k+1 fills the other
            pstep[0] += 1
            gsb = (g_kps[b].data_ptr(), g_desc[b].data_ptr(), g_n[b].data_ptr(), G, cap)
            for c in range(CH):
                a_, b_ = c * cf, (c + 1) * cf
                with torch.cuda.stream(s_in):
                    s_in.wait_event(ev_ex[b * CH + c])       # the extraction that last read this chunk of this image buffer (two steps ago) is done
                    p_img[b][a_:b_].copy_(h_img[a_:b_], non_blocking=True)
                    ev_in[c].record(s_in)
                sA = psts[c % 2]
                sA.wait_event(ev_in[c])
                sA.wait_event(ev_pm[b])                      # the matcher that last read this output set (two steps ago) is done
                pexs[c % 2].extract_batch_device(p_img[b][a_].data_ptr(), W, H, W, W * H, cf, d_kps[b][a_].data_ptr(), d_desc[b][a_].data_ptr(), cap,
                                                 d_n[b][a_:].data_ptr(), sA.cuda_stream)
                ev_ex[b * CH + c].record(sA)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_ex[b * CH + c])
                    h_kps[a_:b_].copy_(d_kps[b][a_:b_], non_blocking=True)
                    h_desc[a_:b_].copy_(d_desc[b][a_:b_], non_blocking=True)
                    h_n[a_:b_].copy_(d_n[b][a_:b_], non_blocking=True)
                sB.wait_event(ev_ex[b * CH + c])
            mts[b].match_pairs_device(gsb, gsb, pairs, W, H, 15.0, sf, d_assigned[b].data_ptr(), d_counts[b].data_ptr(), d_aff.data_ptr(), sB.cuda_stream)
            with torch.cuda.stream(sB):
                h_assigned.copy_(d_assigned[b], non_blocking=True)
                h_counts.copy_(d_counts[b], non_blocking=True)
            ev_pm[b].record(sB)
        if F % CH == 0:
            pcie_step(); pcie_step(); torch.cuda.synchronize()
            n_p = max(args.steps // 2, 3)
            tp = time.perf_counter()
            for _ in range(n_p):
                pcie_step()
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp) / n_p
            bytes_in, bytes_out = F * W * H, F * cap * 60 + F * 4 + NPAIR * cap * 4 + NPAIR * 4
            out["pcie_inclusive"] = {"value": float(h_n.sum().item()) / tp / 1e6, "unit": "Mkeypoints/s", "ms_per_step": tp * 1e3,
                                     "host_to_device_MB_per_step": bytes_in / 1e6, "device_to_host_MB_per_step": bytes_out / 1e6,
                                     "note": "pinned host frames -> H2D on a copy stream (%d chunks per step into one of two device image buffers, overlapped with the extraction of "
                                             "earlier chunks) -> extract into one of two output sets -> match -> keypoints, descriptors, counts and match lists back to "
                                             "pinned host memory; every buffer reuse ordered by events; never the headline value" % CH}
        del h_img, h_kps, h_desc, h_assigned, p_img, pexs

    # ---- configs 3 and 4: stereo streams (extract L + R, computeStereoMatches, consecutive left-frame search) ----------------------
    if extras and (not only or "config3" in only or "config4" in only):
        def stereo_config(w, h, nf, n_pairs, label):
            pl = stream_plan(w, h, n_pairs, seed=7, segment=32)
            L_, R_ = stream_render(pl, range(n_pairs), stereo=True)
            diL, diR = torch.from_numpy(L_).to(dev), torch.from_numpy(R_).to(dev)
            mk = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
            prs = np.array([(i, i + 1) for i in range(n_pairs - 1)], np.int32)
            daf = torch.from_numpy(np.ascontiguousarray(pl["predicted"], np.float32)).to(dev)
            # NSETS lanes, each a complete set (extractor pair, outputs, matchers) with ONE stream: a step's two extractions, its association
            # (the serial replay of frame.cpp:391-462) and its left-frame search (ordered resolve) run back to back on the lane's stream,
            # consecutive steps go to consecutive lanes, no events between lanes.  The two latency chains of a step (~5 ms at 2000
            # features) then hide behind the extractions of the other lanes.  The handles are single-stream (YDORB_EXTRACTOR_SINGLE_STREAM):
            # with side streams, 5+ streams share the device's 4 hardware queues and a stream that lands behind a chain stalls.
            NSETS = int(os.environ.get("YDORB_BENCH_STEREO_SETS", "4"))
            single = os.environ.get("YDORB_BENCH_STEREO_SINGLE_STREAM", "1") != "0"
            sets = []
            for _ in range(NSETS):
                S_ = dict(xL=y.OrbExtractor(nf, 1.2, 8, 20, 7, device=local_rank, max_batch=n_pairs, single_stream=single),
                          xR=y.OrbExtractor(nf, 1.2, 8, 20, 7, device=local_rank, max_batch=n_pairs, single_stream=single))
                scap = S_["xL"].max_keypoints
                S_.update(kL=mk(n_pairs, scap, 7), kR=mk(n_pairs, scap, 7), dL=mk(n_pairs, scap, 32, dt=torch.uint8), dR=mk(n_pairs, scap, 32, dt=torch.uint8),
                          nL=mk(n_pairs, dt=torch.int32), nR=mk(n_pairs, dt=torch.int32), rx=mk(n_pairs, scap), dp=mk(n_pairs, scap), kept=mk(n_pairs, dt=torch.int32),
                          asg=mk(n_pairs - 1, scap, dt=torch.int32), cnt=mk(n_pairs - 1, dt=torch.int32), sm=y.OrbMatcher(device=local_rank),
                          mm=y.OrbMatcher(0.9, True, device=local_rank), st=torch.cuda.Stream(device=dev))
                sets.append(S_)
            scap = sets[0]["xL"].max_keypoints
            ssf = sets[0]["xL"].tables()["scale"]
            kstep = [0]

            def one(full=True):
                S_ = sets[kstep[0] % NSETS]
                st = S_["st"].cuda_stream
                kstep[0] += 1
                S_["xL"].extract_batch_device(diL.data_ptr(), w, h, w, w * h, n_pairs, S_["kL"].data_ptr(), S_["dL"].data_ptr(), scap, S_["nL"].data_ptr(), st)
                S_["xR"].extract_batch_device(diR.data_ptr(), w, h, w, w * h, n_pairs, S_["kR"].data_ptr(), S_["dR"].data_ptr(), scap, S_["nR"].data_ptr(), st)
                if full:
                    S_["sm"].stereo_matches_device(S_["xL"], S_["xR"], S_["kL"].data_ptr(), S_["dL"].data_ptr(), S_["nL"].data_ptr(), scap, S_["kR"].data_ptr(),
                                                   S_["dR"].data_ptr(), S_["nR"].data_ptr(), scap, n_pairs, 40.0, 0.1, S_["rx"].data_ptr(), S_["dp"].data_ptr(),
                                                   S_["kept"].data_ptr(), None, False, (0, 1), (0, 1), st)
                    fs_ = (S_["kL"].data_ptr(), S_["dL"].data_ptr(), S_["nL"].data_ptr(), n_pairs, scap)
                    S_["mm"].match_pairs_device(fs_, fs_, prs, w, h, 15.0, ssf, S_["asg"].data_ptr(), S_["cnt"].data_ptr(), daf.data_ptr(), st)
            res = {}
            for name, full in (("extract_stereo_match", True), ("extract_only", False)):
                for _ in range(NSETS):
                    one(full)
                torch.cuda.synchronize()
                reps_ = max(args.steps // 2, 4)
                t_ = time.perf_counter()
                for _ in range(reps_):
                    one(full)
                torch.cuda.synchronize()
                t_ = (time.perf_counter() - t_) / reps_
                res[name] = {"value": float(sets[0]["nL"].sum().item() + sets[0]["nR"].sum().item()) / t_ / 1e6, "unit": "Mkeypoints/s", "ms_per_step": t_ * 1e3}
            for S_ in sets:
                S_["xL"].synchronize(); S_["xR"].synchronize(); S_["sm"].synchronize(); S_["mm"].synchronize()
            A_ = algorithmic_bytes_extract(w, h, nf)
            res.update({"workload": label, "stereo_pairs_per_step": n_pairs, "keypoints_per_image": float(sets[0]["nL"].float().mean().item()),
                        "stereo_measurements_per_pair": float(sets[0]["kept"].float().mean().item()), "matches_per_left_pair": float(sets[0]["cnt"].float().mean().item()),
                        "algorithmic_bytes_per_image": A_, "pipelining": "%d lanes (handle pair + matchers + one stream each) take consecutive steps; a step's extraction, association and search run back to back on its lane" % NSETS,
                        "pipeline_frac_of_hbm_peak": A_ * 2 * n_pairs / (res["extract_stereo_match"]["ms_per_step"] * 1e-3) / HBM_PEAK})
            del sets
            return res
        if not only or "config3" in only:
            out["config3"] = stereo_config(1241, 376, 2000, 128, "KITTI-00-size 1241x376 stereo, 2000 feat/image: extract L+R, computeStereoMatches (as the reference writes it), consecutive left-frame search")
        if not only or "config4" in only:
            out["config4"] = stereo_config(752, 480, 1000, 64, "EuRoC-MH-size 752x480 stereo batch, 1000 feat/image, one GPU's share: extract L+R, computeStereoMatches, consecutive left-frame search")

    # ---- brute-force N x M Hamming top-2 (north_star; SURVEY 8d secondary figure, against the integer-VALU peak) -------------------
    if want("rest") and hasattr(y.OrbMatcher, "hamming_topk_device"):
        NB_ = min(F - 1, 255)
        mb = y.OrbMatcher(device=local_rank)
        d_best = torch.zeros((NB_, cap, 6), dtype=torch.int32, device=dev)
        mb.hamming_topk_device(d_desc[0].data_ptr(), d_n[0].data_ptr(), d_desc[0][1:].data_ptr(), d_n[0][1:].data_ptr(), cap, NB_, d_best.data_ptr())
        torch.cuda.synchronize()
        tb_ = time.perf_counter()
        for _ in range(5):
            mb.hamming_topk_device(d_desc[0].data_ptr(), d_n[0].data_ptr(), d_desc[0][1:].data_ptr(), d_n[0][1:].data_ptr(), cap, NB_, d_best.data_ptr())
        torch.cuda.synchronize()
        tb_ = (time.perf_counter() - tb_) / 5
        nn_ = d_n[0].cpu().numpy().astype(np.int64)
        npairs_ = float((nn_[:NB_] * nn_[1:NB_ + 1]).sum())
        out["match_bruteforce"] = {"metric": "all-pairs 256-bit Hamming top-2, frame t vs frame t+1", "frame_pairs_per_call": NB_,
                                   "value": npairs_ / tb_ / 1e9, "unit": "G descriptor pairs/s", "ms_per_call": tb_ * 1e3,
                                   "lane_ops_per_pair": 16, "int_valu_peak_Gops": INT_VALU_PEAK / 1e9,
                                   "frac_of_int_valu_peak": npairs_ * 16 / tb_ / INT_VALU_PEAK}

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

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores, bounded sample ---------------------------
    # Rank 0 only; at N > 1 it runs after every timed region (the other ranks wait at the final barrier), on a smaller sample.
    if rank == 0 and not args.no_cpu:
        from concurrent.futures import ThreadPoolExecutor
        from oracle.orb_oracle import FrameOracle, OrbExtractorOracle, QUERY_DTYPE, ba_solve
        model, ncpu_all, ncpu = cpu_info()

        def cpu_chunk(frames, affs):
            """extract every frame and search each one for its predecessor's keypoints, like the GPU step; returns keypoints processed"""
            oex = OrbExtractorOracle(NFEAT, 1.2, 8, 20, 7)
            prev, nk = None, 0
            for i in range(len(frames)):
                k, d = oex.extract(frames[i])
                nk += len(k)
                if prev is not None:
                    pk, pd = prev
                    A = affs[i - 1].astype(np.float32)
                    q = np.zeros(len(pk), QUERY_DTYPE)
                    q["u"] = (A[0] * pk["x"] + A[1] * pk["y"]) + A[2]
                    q["v"] = (A[3] * pk["x"] + A[4] * pk["y"]) + A[5]
                    q["r"] = (np.float32(15.0) * sf[pk["octave"]]).astype(np.float32)
                    q["min_level"], q["max_level"] = pk["octave"] - 1, pk["octave"] + 1
                    q["angle"], q["level"] = pk["angle"], pk["octave"]
                    q["flags"] = np.where((q["u"] >= 0) & (q["u"] < W) & (q["v"] >= 0) & (q["v"] < H), 3, 0)
                    FrameOracle(k, d, (0.0, float(W), 0.0, float(H))).search_by_projection(1, q, pd, 0.9, True)
                prev = (k, d)
            return nk
        if world == 1:
            c_frames, c_affs = imgs, plan["predicted"]
        else:   # this rank's frames are every world-th frame of the stream: render a contiguous piece for the CPU sample
            c_frames, _ = stream_render(plan, range(min(args.cpu_frames, 48)))
            c_affs = plan["predicted"]
        n1 = min(args.cpu_frames if world == 1 else 32, len(c_frames))
        tc = time.perf_counter()
        nk1 = cpu_chunk(c_frames[:n1], c_affs)
        tc = time.perf_counter() - tc
        per = max(4, min(16, len(c_frames) // max(ncpu, 1)))
        chunks = [(i * per, (i + 1) * per) for i in range(ncpu) if (i + 1) * per <= len(c_frames)]
        tca = time.perf_counter()
        with ThreadPoolExecutor(max_workers=max(len(chunks), 1)) as pool:   # ctypes releases the GIL inside the oracle
            nka = sum(pool.map(lambda ab: cpu_chunk(c_frames[ab[0]:ab[1]], c_affs[ab[0]:]), chunks))
        tca = time.perf_counter() - tca
        out["cpu_baseline"] = {"value": nk1 / tc / 1e6, "unit": "Mkeypoints/s", "cores": 1, "kind": "port",
                               "sample": "%d of the same 640x480 frames, extract + consecutive match, oracle (C++ -O2), 1 thread" % n1,
                               "cpu_model": model, "host_cores": ncpu_all, "usable_cores": ncpu,
                               "all_cores": {"value": nka / tca / 1e6, "unit": "Mkeypoints/s", "cores": len(chunks),
                                             "sample": "%d threads x %d consecutive frames each (frame-parallel; the reference itself uses <= 2 extractor threads, frame.cpp:84-85)" % (len(chunks), per)}}
        out["vs_cpu"] = value / out["cpu_baseline"]["value"]
        out["vs_cpu_all_cores"] = value / out["cpu_baseline"]["all_cores"]["value"]
        if not args.no_ba:
            pb = synth_ba_problem(100, 10000, 8, seed=1)
            tcb = time.perf_counter()
            rb = ba_solve(pb)
            tcb = time.perf_counter() - tcb
            out["ba"]["cpu_baseline"] = {"value": rb["trials"] / tcb, "unit": "it/s", "cores": 1, "kind": "port", "cpu_model": model,
                                         "sample": "one full localBundleAdjust schedule (%d LM trials) on the same problem" % rb["trials"]}
            nbp = min(ncpu, 16)
            tcb2 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nbp) as pool:
                tr_ = sum(r_["trials"] for r_ in pool.map(lambda _: ba_solve(pb), range(nbp)))
            tcb2 = time.perf_counter() - tcb2
            out["ba"]["cpu_baseline"]["all_cores"] = {"value": tr_ / tcb2, "unit": "it/s (aggregate)", "cores": nbp,
                                                      "sample": "%d copies of the problem, one solve per core (problem-parallel; g2o itself is single-threaded here)" % nbp}
            out["ba"]["vs_cpu"] = out["ba"]["value"] / out["ba"]["cpu_baseline"]["value"]
        if not args.no_ba and world == 1:
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
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
