#!/usr/bin/env python3
"""Headline benchmark: ORB extract+match Mkeypoints/s (BASELINE.json config 2: 640x480 mono stream, 1000 features/frame) and
local-BA LM iterations/s (config 5: 100 keyframes x 10 000 points) on N MI355X of one node; configs 3 and 4 (stereo streams) as
first-class entries of the same line.

The stream the reference processes comes from host memory, one frame at a time (src/frame.cpp:84-87,129 from src/tracking.cpp:96-137),
and SURVEY.md 8(d) times "the batch incl. H2D/D2H".  So the headline `value` is the H2D/D2H-INCLUSIVE figure:
  pinned host frames -> hipMemcpyAsync into a ring of device image buffers (copy-in stream)
  -> ydorb_extract_batch_device (pyramid -> FAST cells -> quad-tree -> blur -> orientation + rBRIEF)
  -> [N > 1: all-gather of the ranks' records on a communication stream]
  -> ydorb_match_pairs_device (grid build -> candidate distances -> ordered resolve): every frame is searched for the keypoints of its
     predecessor in the stream (searchByProjectionInLastAndCurrentFrame rules, th = 15), predicted with the known inter-frame motion
  -> keypoints, descriptors, counts and match lists back to pinned host memory (copy-out stream).
`kernel_pipeline` is the same work with the frames already resident in HBM and the results left there (what rounds 1-2 reported as
`value`), `pcie` is the measured host link next to it, `roofline` the dominant kernel against the HBM roof.

One "step" = --substeps launches of --frames frames each (default 8 x 2048 = 16384 frames), dealt to the lanes in turn (inclusive: three lanes + one
stream that only uploads; resident: four lanes); the timed
region is exactly --steps steps, bracketed by barrier + synchronize, and it is repeated --repeats times: `value` is the median,
`repeats` holds min / max.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (fresh child processes, before
anything touches a GPU); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  N > 1: one process per GPU (torch.distributed /
RCCL), weak scaling.  --exchange allgather (default): frames dealt round-robin, so EVERY consecutive pair straddles two GPUs and each
rank all-gathers every rank's [keypoints | descriptors | count] records per launch; --exchange neighbour: contiguous
frame shards, only the shard's boundary frame is exchanged.

Prints ONE JSON line (rank 0).  `cpu_baseline` times the CPU oracle (a port of the reference algorithm, built -O3 like the reference)
on a bounded sample of the same frames, on one thread and on all usable host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
FP64_PEAK = 78.6e12        # FLOP/s, MI355X FP64 vector / matrix datasheet figure (the in-container guide lists no FP64 number)
PCIE_SPEC = 63.0e9         # B/s, MI355X_MICROARCH.md "Host link: PCIe Gen5 x16, 63 GB/s (spec)"
PARITY_NOTE = ("bit-exact vs the build's CPU restatement of the reference algorithm (oracle/); oracle unpinned for extractor / "
               "matcher / stereo / BoW: OpenCV absent, the reference holds no fixture; BA solver pinned by g2o's own linear-system vector only")
PMC_TAGS = ("r03", "r02d", "r02", "r01f")   # committed counter passes, newest first (profiles/<tag>_pmc_*.csv)


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


def prefer_image_rocm():
    """libydorb.so is built against the image's ROCm (/opt/rocm, what a C++ host of the reference links), but a process that imports torch first
    runs on the HIP / ROCr runtime bundled in the PyTorch wheel (torch/lib, an older release): same sonames, first one loaded wins.  Under the
    bundled runtime an upload and a read-back in flight together slow each other (tools/ubench/duplex_runtimes.sh: 256 MB up + 64 MB down
    52.7 + 13.2 GB/s against 57.0 + 14.2 on the image's runtime; torch copies both ways 28.5 + 28.5 against 48.5 + 48.5).  So the bench loads
    the image's runtime before torch (YDORB_BENCH_SYSTEM_ROCM=0: leave the order alone).  N > 1 keeps PyTorch's own stack unless
    YDORB_BENCH_SYSTEM_ROCM=1 asks otherwise: the RCCL build in the wheel was made for the runtime beside it, and the combination with the
    image's runtime could only be rehearsed at world size 1 on this pool (tools/rehearse_multigpu.sh).  Returns what was loaded, for the JSON line."""
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if os.environ.get("YDORB_BENCH_SYSTEM_ROCM", "0" if multi else "1") != "1" or "torch" in sys.modules:
        return None
    import ctypes
    root = os.environ.get("ROCM_PATH", "/opt/rocm")
    libs = [os.path.join(root, "lib", n) for n in ("libhsa-runtime64.so.1", "libamdhip64.so.7")]
    if not all(os.path.exists(l_) for l_ in libs):
        return None
    try:
        for l_ in libs:
            ctypes.CDLL(l_, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None
    return os.path.realpath(libs[1])


class Ctx:
    """Process-wide state: arguments, rank / world, device, the collectives (RCCL, or gloo staged through the host for rehearsals)."""

    def __init__(self, args):
        self.hip_runtime = prefer_image_rocm()
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # YDORB_BENCH_BACKEND=gloo + YDORB_BENCH_ONE_GPU=1: rehearse the N > 1 logic with every rank on cuda:0 and the collectives staged
        # through host memory (RCCL refuses two ranks on one device).  The driver's real runs use RCCL ("nccl").
        self.backend = os.environ.get("YDORB_BENCH_BACKEND", "nccl")
        if os.environ.get("YDORB_BENCH_ONE_GPU"):
            self.local_rank = 0
        # YDORB_BENCH_FORCE_DIST=1: process group + collectives even at world size 1 (an RCCL rehearsal of the N > 1 code path on one GPU)
        self.force_dist = self.world == 1 and bool(os.environ.get("YDORB_BENCH_FORCE_DIST"))
        if self.force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        self.dev = torch.device("cuda", self.local_rank)
        self.distributed = self.world > 1 or self.force_dist
        if self.distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.local_rank)
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        torch.cuda.set_device(self.dev)
        self.staged = self.world > 1 and self.backend != "nccl"

    def all_reduce_(self, t, op):
        if self.staged:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=op)

    def all_gather_into(self, full, mine):
        """full: [world * k, ...], mine: [k, ...] (may be this rank's own slice of `full`: in place, no packing copy)."""
        if self.staged:
            parts = [self.torch.zeros_like(mine, device="cpu") for _ in range(self.world)]
            self.dist.all_gather(parts, mine.cpu())
            full.copy_(self.torch.cat(parts))
        else:
            self.dist.all_gather_into_tensor(full, mine)

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if not self.distributed:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.all_reduce_(t, self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v):
        if not self.distributed:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.all_reduce_(t, self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, fn, steps, warmup, repeats):
        """`repeats` timed regions of exactly `steps` calls of fn, each bracketed by barrier + synchronize on both sides; MAX over ranks.
        Returns the list of region durations in seconds."""
        for _ in range(warmup):
            fn()
        out = []
        for _ in range(repeats):
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            self.barrier()
            out.append(self.max_over_ranks(time.perf_counter() - t0))
        return out


def stats(per_repeat):
    import numpy as np
    a = np.asarray(per_repeat, np.float64)
    return {"n": int(len(a)), "median": float(np.median(a)), "min": float(a.min()), "max": float(a.max())}


# ---------------------------------------------------------------------------------------------------------------------------------
# Host link: pinned hipMemcpyAsync H2D, D2H and both at once.  The link is full duplex (tools/ubench/duplex.hip: 48.5 GB/s each way together,
# 57 + 14 for an upload with a quarter-size read-back) - on the image's ROCm runtime; under the runtime bundled with PyTorch the two directions
# slow each other (28.5 each from torch copies), see prefer_image_rocm().  The pipeline's roof is its upload bytes / the one-way rate.
# ---------------------------------------------------------------------------------------------------------------------------------
def pcie_link(ctx, mb=160, reps=5):
    torch = ctx.torch
    n = mb << 20
    h_a, h_b = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    d_a, d_b = torch.empty(n, dtype=torch.uint8, device=ctx.dev), torch.zeros(n, dtype=torch.uint8, device=ctx.dev)
    s1, s2 = torch.cuda.Stream(device=ctx.dev), torch.cuda.Stream(device=ctx.dev)

    def run(h2d, d2h):
        def issue():
            if h2d:
                with torch.cuda.stream(s1):
                    d_a.copy_(h_a, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_b.copy_(d_b, non_blocking=True)
        issue(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            issue()
        torch.cuda.synchronize()
        return n * (int(h2d) + int(d2h)) * reps / (time.perf_counter() - t0) / 1e9
    r = {"h2d_GBps": run(True, False), "d2h_GBps": run(False, True), "h2d_plus_d2h_GBps": run(True, True), "transfer_MB": mb,
         "spec_GBps": PCIE_SPEC / 1e9}
    # the pipeline's own mix: ~5 bytes in per byte out, two uploads in flight (two lanes copy at a time)
    n5 = n // 5
    s3 = torch.cuda.Stream(device=ctx.dev)
    d_c = torch.empty(n, dtype=torch.uint8, device=ctx.dev)

    def mixed():
        with torch.cuda.stream(s1):
            d_a.copy_(h_a, non_blocking=True)
        with torch.cuda.stream(s3):
            d_c.copy_(h_a, non_blocking=True)
        with torch.cuda.stream(s2):
            h_b[:n5].copy_(d_b[:n5], non_blocking=True)
            h_b[n5:2 * n5].copy_(d_b[n5:2 * n5], non_blocking=True)
    mixed(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        mixed()
    torch.cuda.synchronize()
    r["mixed_5to1_GBps"] = (2 * n + 2 * n5) * reps / (time.perf_counter() - t0) / 1e9
    r["peak_GBps"] = max(r["h2d_GBps"], r["d2h_GBps"])
    r["note"] = ("pinned hipMemcpyAsync of %d MB, this process on this box; peak_GBps = the better one-way rate; h2d_plus_d2h = one stream per direction, "
                 "sum of both; mixed_5to1 = two uploads + read-backs of a fifth of the bytes, sum.  Both exceed the one-way rate only on the image's "
                 "ROCm runtime (config.hip_runtime): the link is full duplex, the runtime bundled with PyTorch serialises the two directions" % mb)
    del h_a, h_b, d_a, d_b
    return r


# ---------------------------------------------------------------------------------------------------------------------------------
# Config 2: the mono stream
# ---------------------------------------------------------------------------------------------------------------------------------
class MonoStream:
    W, H, NFEAT = 640, 480, 1000
    # 4 streams = the inclusive pipeline's three lanes + upload stream; N > 1: 6 (five lanes + upload stream, on the process's 8 hardware queues) -
    # every launch of a lane then also waits for its exchange, and the other lanes' launches are what hides that wait (N = 1: 4 and 6 measure the same)
    LANES = int(os.environ.get("YDORB_BENCH_LANES", "6" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "4"))
    # single-stream handles (YDORB_EXTRACTOR_SINGLE_STREAM: quad-tree launches on the lane's stream): the device runs 4 hardware queues, so 4
    # streams keep every lane on its own queue; handles with side streams share queues with the copies and with each other.  Resident
    # figures (kernel_pipeline, extract_only): 4 lanes of one stream each take consecutive launches.  Inclusive pipeline: see UPLOAD_STREAM.
    # Measured, inclusive / resident Mkeypoints/s (tools/bench_copy_sweep.sh, 512 frames per launch): 2 lanes with side-stream handles 93 /
    # 194, copy-in / copy-out streams + 2 single-stream lanes 140 / 197, 4 single-stream lanes with the copies in the lane 151 / 199.
    SINGLE = bool(int(os.environ.get("YDORB_BENCH_SINGLE_STREAM", "1")))
    # N > 1: the exchange on a communication stream of the bench's own (1), or issued from the lane's stream (0: the collective itself still runs
    # on RCCL's internal stream and the lane waits for it).  One more stream means the lanes share hardware queues again: with an RCCL
    # process group at world size 1 on one GPU, 94 against 146 Mkeypoints/s inclusive (tools/rehearse_multigpu.sh) - off by default.
    COMM_STREAM = bool(int(os.environ.get("YDORB_BENCH_COMM_STREAM", "0")))
    # Inclusive pipeline, 1 (default): the LAST lane's stream carries nothing but the uploads, queued back to back into a ring of image buffers
    # (the link is the bottleneck of the inclusive figure: it must never wait for a lane), the other lanes take the launches - a lane waits
    # for its frames' event, the upload stream for the buffer's release event - and read back in their own stream order.  0: every lane
    # uploads its own frames in stream order (no events; the link idles whenever no lane is in its upload phase).  Same box, alternating
    # (tools/bench_upload_ab.sh, image ROCm runtime): 172.7 / 175.3 against 165.7 / 162.1 Mkeypoints/s; a FIFTH stream for the uploads
    # shares a hardware queue with a lane: 121 - 138.
    UPLOAD_STREAM = bool(int(os.environ.get("YDORB_BENCH_UPLOAD_STREAM", "1")))

    def __init__(self, ctx, y, exchange=None):
        import numpy as np
        from ydorbslam_amd.synth import stream_plan, stream_render
        from ydorbslam_amd.parallel import ring_pairs, frame_shard
        torch, args = ctx.torch, ctx.args
        self.ctx, self.y = ctx, y
        W, H, F, world, rank, dev = self.W, self.H, args.frames, ctx.world, ctx.rank, ctx.dev
        self.F, self.G = F, F * world
        self.mode = (exchange or args.exchange) if ctx.distributed else "none"
        neighbour = self.mode == "neighbour"
        # every frame distinct; frame g+1 = frame g after a small known motion (scene cut every --segment frames).
        self.plan = stream_plan(W, H, self.G, seed=0, segment=args.segment)
        if neighbour:           # contiguous shards: global frame g = rank * F + t
            lo_g, _ = frame_shard(self.G, rank, world)
            own = list(range(lo_g, lo_g + F))
        else:                   # round-robin: global frame g = t * world + rank
            own = [t * world + rank for t in range(F)]
        self.imgs, _ = stream_render(self.plan, own)
        # The lane streams are created FIRST, before any handle creates streams of its own: the device runs 4 hardware queues and HIP deals
        # streams onto them in creation order (profiles/r02d_queue_overlap.txt).
        self.lane_streams = [torch.cuda.Stream(device=dev) for _ in range(self.LANES)]
        self.s_comm = torch.cuda.Stream(device=dev) if ctx.distributed and self.COMM_STREAM else None
        self.s_up = self.lane_streams[-1] if self.UPLOAD_STREAM and self.LANES >= 2 else None
        self.INC_LANES = self.LANES - 1 if self.s_up is not None else self.LANES   # lanes that take the inclusive pipeline's launches
        self.exs = [y.OrbExtractor(self.NFEAT, 1.2, 8, 20, 7, device=ctx.local_rank, max_batch=F, single_stream=self.SINGLE) for _ in range(self.LANES)]
        self.cap = cap = self.exs[0].max_keypoints
        self.sf = self.exs[0].tables()["scale"]
        self.NSET = NSET = self.LANES            # one output set (and one image buffer) per lane
        # A rank's records of a launch are ONE slab [keypoints FS x cap x 28 B | descriptors FS x cap x 32 B | counts FS x 4 B]; an output set
        # holds `world` slabs, rank r's at slab r, and this rank's extractor writes straight into its own (no packing copy).  With the
        # frames dealt round-robin, every frame's predecessor lives on rank - 1 (mod world): the queries of a launch are rank - 1's slab,
        # the targets the rank's own.  FS = F, or F + 1 with the neighbour exchange (slot F = the previous rank's boundary frame).
        FS = F + 1 if neighbour else F
        self.FS = FS
        self.off_desc = FS * cap * 28
        self.off_n = self.off_desc + FS * cap * 32
        self.slab = (self.off_n + FS * 4 + 255) // 256 * 256
        self.sets = [torch.zeros(world * self.slab, dtype=torch.uint8, device=dev) for _ in range(NSET)]
        self.own = [t[rank * self.slab:(rank + 1) * self.slab] for t in self.sets]
        self.prev_rank = (rank - 1) % world
        if neighbour:
            prs = [(t - 1, t) for t in range(1, F)]
            pred = [lo_g + t - 1 for t in range(1, F)]
            if rank > 0:      # the pair that straddles the shard boundary: query = rank - 1's last frame (slot F), target = own frame 0
                prs.append((F, 0)); pred.append(lo_g - 1)
            self.pairs, pred_idx = np.array(prs, np.int32).reshape(-1, 2), np.array(pred, np.int64)
            self.q_rank = rank
            self.rec = cap * 60 + 4
            self.recs = [torch.zeros((world, self.rec), dtype=torch.uint8, device=dev) for _ in range(NSET)]
            self.myrec = [torch.zeros(self.rec, dtype=torch.uint8, device=dev) for _ in range(NSET)]
        else:
            self.pairs, pred_idx = ring_pairs(rank, world, F)
            self.q_rank = self.prev_rank
        self.NPAIR = NPAIR = len(self.pairs)
        self.d_aff = torch.from_numpy(np.ascontiguousarray(self.plan["predicted"][pred_idx], np.float32)).to(dev)
        self.d_assigned = [torch.zeros((NPAIR, cap), dtype=torch.int32, device=dev) for _ in range(NSET)]
        self.d_counts = [torch.zeros(NPAIR, dtype=torch.int32, device=dev) for _ in range(NSET)]
        self.mts = [y.OrbMatcher(0.9, True, device=ctx.local_rank) for _ in range(NSET)]   # one matcher (own scratch) per output set
        # host side of the inclusive pipeline: pinned frames, one device image buffer per lane, pinned result sets
        self.h_img = torch.from_numpy(self.imgs).pin_memory()
        self.NIMG = self.INC_LANES + 2 if self.s_up is not None else self.LANES      # upload stream: a ring it may run two launches ahead in
        self.d_img = [torch.empty_like(self.h_img, device=dev) for _ in range(self.NIMG)]
        self.ev_in, self.ev_free = [torch.cuda.Event() for _ in range(self.NIMG)], [None] * self.NIMG
        self.d_img[0].copy_(self.h_img)
        self.h_out = [[torch.zeros_like(t, device="cpu").pin_memory() for t in (self.own[b], self.d_assigned[b], self.d_counts[b])] for b in range(NSET)]
        ev = lambda n_: [torch.cuda.Event() for _ in range(n_)]
        self.ev_x, self.ev_g = ev(NSET), ev(NSET)
        self.k = 0
        self.up_trace = [] if self.s_up is not None and os.environ.get("YDORB_BENCH_UPLOAD_TRACE") else None   # diagnostic: (start, end) events of every upload
        self.bytes_in = F * W * H
        self.bytes_out = sum(t.numel() * t.element_size() for t in self.h_out[0])

    def ptrs(self, b, r):
        """(d_kps, d_desc, d_n, n_frames, cap) of rank r's slab in output set b"""
        base = self.sets[b].data_ptr() + r * self.slab
        return (base, base + self.off_desc, base + self.off_n, self.FS, self.cap)

    def counts(self, b, host=False):
        """this rank's keypoint counts of output set b (device tensor, or the pinned host copy the inclusive pipeline delivered)"""
        t = self.h_out[b][0] if host else self.own[b]
        return t[self.off_n:self.off_n + 4 * self.F].view(self.ctx.torch.int32)

    def exchange(self, b, sA):
        """N > 1: the exchange of launch k; the lane's matcher waits for it, the other lanes extract launches k + 1 .. k + 3 meanwhile."""
        ctx, torch, dist = self.ctx, self.ctx.torch, self.ctx.dist
        sc = self.s_comm if self.s_comm is not None else sA
        if sc is not sA:
            self.ev_x[b].record(sA)
            sc.wait_event(self.ev_x[b])
        with torch.cuda.stream(sc):
            if self.mode == "allgather":     # all-gather of every rank's records of the launch: ONE collective, in place
                ctx.all_gather_into(self.sets[b].view(ctx.world, self.slab), self.own[b].view(1, self.slab))
            elif self.mode == "ring":        # only what the matcher reads: the previous rank's slab (one send, one receive per rank)
                if ctx.world > 1:
                    prev = self.sets[b][self.prev_rank * self.slab:(self.prev_rank + 1) * self.slab]
                    if ctx.staged:
                        hs, hr = self.own[b].cpu(), torch.empty(self.slab, dtype=torch.uint8)
                        for r_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, hs, (ctx.rank + 1) % ctx.world), dist.P2POp(dist.irecv, hr, self.prev_rank)]):
                            r_.wait()
                        prev.copy_(hr)
                    else:
                        for r_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, self.own[b], (ctx.rank + 1) % ctx.world), dist.P2POp(dist.irecv, prev, self.prev_rank)]):
                            r_.wait()
            else:                            # neighbour: every rank's LAST frame record (60 KB + 4 B); the previous rank's goes to slot F
                F, cap, o = self.F, self.cap, self.own[b]
                m = self.myrec[b]
                m[:cap * 28].copy_(o[(F - 1) * cap * 28:F * cap * 28])
                m[cap * 28:cap * 60].copy_(o[self.off_desc + (F - 1) * cap * 32:self.off_desc + F * cap * 32])
                m[cap * 60:].copy_(o[self.off_n + 4 * (F - 1):self.off_n + 4 * F])
                ctx.all_gather_into(self.recs[b], m.view(1, self.rec))
                p = self.recs[b][self.prev_rank]
                o[F * cap * 28:(F + 1) * cap * 28].copy_(p[:cap * 28])
                o[self.off_desc + F * cap * 32:self.off_desc + (F + 1) * cap * 32].copy_(p[cap * 28:cap * 60])
                o[self.off_n + 4 * F:self.off_n + 4 * F + 4].copy_(p[cap * 60:])
        if sc is not sA:
            self.ev_g[b].record(sc)
            sA.wait_event(self.ev_g[b])

    def launch(self, inclusive, match=True):
        """One launch of F frames on lane k % LANES (its stream, image buffer, output set and matcher): [upload] -> extraction ->
        [exchange] -> matching -> [read-back], in stream order."""
        torch, W, H, F = self.ctx.torch, self.W, self.H, self.F
        k = self.k
        self.k += 1
        lane = b = k % (self.INC_LANES if inclusive else self.LANES)
        sA = self.lane_streams[lane]
        r = k % self.NIMG if self.s_up is not None else lane
        img = self.d_img[r if inclusive else 0]
        if inclusive and self.s_up is not None:   # uploads queue back to back on their own stream; a lane waits for its frames only
            if self.ev_free[r] is not None:
                self.s_up.wait_event(self.ev_free[r])
            with torch.cuda.stream(self.s_up):
                if self.up_trace is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.s_up)
                img.copy_(self.h_img, non_blocking=True)
                if self.up_trace is not None:
                    e1.record(self.s_up)
                    self.up_trace.append((e0, e1))
            self.ev_in[r].record(self.s_up)
            sA.wait_event(self.ev_in[r])
        elif inclusive:
            with torch.cuda.stream(sA):
                img.copy_(self.h_img, non_blocking=True)
        own = self.ptrs(b, self.ctx.rank)
        self.exs[lane].extract_batch_device(img.data_ptr(), W, H, W, W * H, F, own[0], own[1], self.cap, own[2], sA.cuda_stream)
        if inclusive and self.s_up is not None:
            if self.ev_free[r] is None:
                self.ev_free[r] = torch.cuda.Event()
            self.ev_free[r].record(sA)
        if not match:
            return
        if self.ctx.distributed:
            self.exchange(b, sA)
        self.mts[b].match_pairs_device(self.ptrs(b, self.q_rank), own, self.pairs, W, H, 15.0, self.sf, self.d_assigned[b].data_ptr(),
                                       self.d_counts[b].data_ptr(), self.d_aff.data_ptr(), sA.cuda_stream)
        if inclusive:
            with torch.cuda.stream(sA):
                for h_, d_ in zip(self.h_out[b], (self.own[b], self.d_assigned[b], self.d_counts[b])):
                    h_.copy_(d_, non_blocking=True)

    def step(self, inclusive, match=True):
        for _ in range(self.ctx.args.substeps):
            self.launch(inclusive, match)

    def set_profiling(self, on):
        for h_ in self.exs:
            h_.set_profiling(on)
        for m_ in self.mts:
            m_.set_profiling(on)

    def stage_times(self):
        import numpy as np
        live = {}
        for h_ in self.exs:
            for k_, v_ in h_.stage_times().items():
                live.setdefault(k_, []).append(v_)
        for m_ in self.mts:
            for k_, v_ in m_.stage_times().items():
                if v_ > 0:
                    live.setdefault(k_, []).append(v_)
        return {k_: float(np.mean(v_)) for k_, v_ in live.items() if len(v_)}

    def synchronize(self):
        for h_ in self.exs:
            h_.synchronize()   # surfaces a quad-tree capacity status, if any
        for m_ in self.mts:
            m_.synchronize()   # surfaces a record-pool overflow, if any


def counter_traffic(dom, frames_per_launch):
    """HBM bytes per launch of the dominant kernel: NOT measured in this run (PMC counters need rocprofv3 around the process).  They come
    from the committed counter passes of the same kernels (separate FETCH_SIZE / WRITE_SIZE runs; FETCH under-counts this 4-byte access
    pattern by 1.33x, calibrated on k_pyr_level0's known read size; WRITE_SIZE is exact), scaled to this run's frames per launch."""
    stage_kernels = {"pyramid": ("k_pyr_",), "fast_cells": ("k_fast_cells",), "blur": ("k_blur",),
                     "quadtree_after_blur": ("k_quadtree", "k_qt_"), "orient_describe": ("k_orient_describe",)}
    traffic = valu_busy = source = None
    try:
        import csv
        for tag in PMC_TAGS:
            pth = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.csv" % tag)
            if not os.path.exists(pth):
                continue
            rows = {r["kernel"]: r for r in csv.DictReader(open(pth))}
            mine = [k for k in rows if dom in stage_kernels and k.startswith(stage_kernels[dom])]
            if mine:
                per_frame = sum(float(rows[k]["fetch_MB_per_frame_raw"]) * float(rows[k].get("fetch_correction") or 1.33) + float(rows[k]["write_MB_per_frame"])
                                for k in mine)
                traffic = per_frame * 1e6 * frames_per_launch
                source = ("profiles/%s_pmc_hbm_traffic.csv (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the extractor alone, FETCH calibrated "
                          "on k_pyr_level0's known read size, per frame, scaled to %d frames per launch; not collected in this run)" % (tag, frames_per_launch))
            pth2 = os.path.join(ROOT, "profiles", "%s_pmc_sq_valu.csv" % tag)
            if os.path.exists(pth2):
                for r in csv.DictReader(open(pth2)):
                    if dom in stage_kernels and r["kernel"].startswith(stage_kernels[dom]) and valu_busy is None:
                        valu_busy = float(r["VALU_busy_pct_of_SIMD_cycles"]) / 100.0
            if traffic is not None:
                break
    except Exception:  # noqa: BLE001
        pass
    return traffic, valu_busy, source


def extract_roofline(w, h, nfeat, n_kp_frame, stages, isolated, frames_per_launch, pairs_per_launch, pipe_bytes_per_s):
    """Roofline object of the dominant EXTRACTOR kernel (the stage timers of the lanes overlap, and of the matcher's stages the resolve
    is serial by definition - neither is a bandwidth figure)."""
    A_frame = algorithmic_bytes_extract(w, h, nfeat)
    pyr_pad = A_frame - w * h - nfeat * 60
    kbytes = {   # algorithmic bytes of each kernel per frame (DESIGN.md "Kernels"): what it must read + write once
        "pyramid": w * h + pyr_pad,                       # read image, write padded pyramid
        "fast_cells": pyr_pad,                            # read every pyramid pixel once (candidates are << 1 %)
        "quadtree_after_blur": 8 * 4200 * 4,              # read ~4.2 k packed candidates per level (measured average), write keypoints
        "blur": 2 * pyr_pad,                              # read pyramid, write blurred levels
        "orient_describe": int(n_kp_frame) * (60 + 2 * 1849),  # 43x43 patch of the level and of the blurred level + 60 B out
    }
    ext_stages = ("pyramid", "fast_cells", "blur", "orient_describe")
    dom = max((k_ for k_ in stages if k_ in ext_stages), key=stages.get)
    t_dom = stages[dom] * 1e-3
    achieved = kbytes[dom] * frames_per_launch / t_dom if t_dom > 0 else 0.0
    alone = (kbytes[dom] * frames_per_launch / (isolated[dom] * 1e-3)) if isolated and isolated.get(dom) else None
    traffic, valu_busy, source = counter_traffic(dom, frames_per_launch)
    return {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
            "achieved_alone": alone / 1e9 if alone else None, "frac_alone": alone / HBM_PEAK if alone else None,
            "traffic": traffic, "traffic_source": source, "valu_busy_frac": valu_busy,
            "valu_busy_source": source and source.replace("hbm_traffic", "sq_valu"),
            "algorithmic_bytes_per_frame": kbytes[dom], "pipeline_algorithmic_bytes_per_frame": A_frame,
            "pipeline_achieved_GBps": pipe_bytes_per_s / 1e9, "pipeline_frac": pipe_bytes_per_s / HBM_PEAK,
            "stage_note": "a stage = every launch of its kernels in one extractor call over frames_per_extract_launch frames, HIP events on the launch "
                          "stream IN the resident-frames pipelined run, where a second lane's kernels share the GPU (isolated = the same call with nothing "
                          "else running); profiles/*_512frame_launches.txt has rocprofv3's per-launch durations of the same command",
            "stage_ms_per_launch": {k: round(v, 4) for k, v in stages.items()},
            "stage_ms_per_launch_isolated": {k: round(v, 4) for k, v in (isolated or {}).items()},
            "frames_per_extract_launch": frames_per_launch, "pairs_per_match_launch": pairs_per_launch}


def mono_section(ctx, y, link):
    """Config 2.  Returns (top-level keys of the JSON line, the MonoStream for later sections)."""
    args, torch = ctx.args, ctx.torch
    S = MonoStream(ctx, y)
    W, H, NFEAT, F, world = S.W, S.H, S.NFEAT, S.F, ctx.world
    launches = args.steps * args.substeps
    # ---- the contract's number: H2D/D2H-inclusive, exactly --steps steps per timed region -----------------------------------------
    t_inc = ctx.timed(lambda: S.step(True), args.steps, args.warmup, args.repeats)
    S.synchronize()
    if S.up_trace:   # diagnostic (YDORB_BENCH_UPLOAD_TRACE): duration of the uploads and the gaps between consecutive ones, last region
        tr = S.up_trace[-launches:]
        dur = [a_.elapsed_time(b_) for a_, b_ in tr]
        gap = [tr[i][1].elapsed_time(tr[i + 1][0]) for i in range(len(tr) - 1)]
        q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
        sys.stderr.write("upload trace: %d uploads, duration ms min %.2f median %.2f p90 %.2f max %.2f; gap ms median %.3f p90 %.3f max %.3f; sum dur %.1f sum gap %.1f\n"
                         % (len(tr), min(dur), q(dur, .5), q(dur, .9), max(dur), q(gap, .5), q(gap, .9), max(gap), sum(dur), sum(gap)))
        S.up_trace = None
    kp_local = int(S.counts(0, host=True).sum().item())   # from the pinned host copy the pipeline delivered
    matched_local = int(S.h_out[0][2].sum().item())
    if kp_local != int(S.counts(0).sum().item()) or kp_local <= 0:
        raise RuntimeError("the read-back of the inclusive pipeline does not match the device results")
    kp_total = ctx.sum_over_ranks(float(kp_local))
    rate = lambda dt: kp_total * launches / dt / 1e6
    inc = stats([rate(t) for t in t_inc])
    dt_inc = float(sorted(t_inc)[len(t_inc) // 2])
    # ---- the same work with the frames resident in HBM and the results left there ---------------------------------------------------
    S.set_profiling(True)     # per-stage device time LIVE in the timed regions (HIP events on the launch stream, read at the next launch, never waited for)
    t_res = ctx.timed(lambda: S.step(False), args.steps, args.warmup, args.repeats)
    S.synchronize()
    live = S.stage_times()
    S.set_profiling(False)
    res = stats([rate(t) for t in t_res])
    dt_res = float(sorted(t_res)[len(t_res) // 2])
    # ---- extraction only (both figures) -------------------------------------------------------------------------------------
    t_ext = ctx.timed(lambda: S.step(False, match=False), max(args.steps // 2, 2), 1, 3)
    ext = stats([kp_total * max(args.steps // 2, 2) * args.substeps / t / 1e6 for t in t_ext])
    # ---- the stages with nothing else on the GPU (one handle, one stream) -----------------------------------------------------------------
    ex2 = y.OrbExtractor(NFEAT, 1.2, 8, 20, 7, device=ctx.local_rank, max_batch=F)
    ex2.set_profiling(True)
    mt = S.mts[0]
    mt.set_profiling(True)
    for _ in range(5):
        ex2.extract_batch(S.imgs)
    for _ in range(6):
        mt.match_pairs_device(S.ptrs(0, S.q_rank), S.ptrs(0, ctx.rank), S.pairs, W, H, 15.0, S.sf, S.d_assigned[0].data_ptr(), S.d_counts[0].data_ptr(), S.d_aff.data_ptr())
        mt.synchronize()
    isolated = dict(ex2.stage_times())
    isolated.update(mt.stage_times())
    mt.set_profiling(False)
    del ex2
    stages = dict(isolated)
    stages.update(live)
    A_frame = algorithmic_bytes_extract(W, H, NFEAT)
    roofline = extract_roofline(W, H, NFEAT, kp_local / F, stages, isolated, F, S.NPAIR, A_frame * F * launches / dt_res)
    pcie = dict(link)
    pcie.update({"host_to_device_MB_per_launch": S.bytes_in / 1e6, "device_to_host_MB_per_launch": S.bytes_out / 1e6,
                 "achieved_GBps": S.bytes_in * launches / dt_inc / 1e9, "achieved_d2h_GBps": S.bytes_out * launches / dt_inc / 1e9,
                 "frac": S.bytes_in * launches / dt_inc / (link["peak_GBps"] * 1e9),
                 "frac_note": "upload bytes of the inclusive pipeline per second / peak_GBps (the one-way rate of pinned hipMemcpyAsync measured in this "
                              "process): the frames going in are 5x the results coming back and the link is full duplex, so the upload direction is the "
                              "roof - %.1f Mkeypoints/s at peak_GBps for this workload" % (kp_total / (S.bytes_in / (link["peak_GBps"] * 1e9)) / 1e6 / 1.0)})
    par = "1 GPU"
    if world > 1:
        par = {"neighbour": "contiguous frame shards x%d, all-gather of each rank's boundary frame record (60 KB) per launch, local match",
               "ring": "frames dealt round-robin x%d, every rank sends its [kp|desc|n] slab of the launch to rank + 1 (the only rank that reads it), local match",
               "allgather": "frames dealt round-robin x%d, ONE all-gather of every rank's [kp|desc|n] slab per launch, "
                            "local match of the owned frames against rank - 1's slab"}[S.mode] % world
    variants = None
    if ctx.distributed and world > 1 or (ctx.force_dist and os.environ.get("YDORB_BENCH_VARIANTS")):
        # the other two exchange forms, measured in the same run (3 repeats each): value stays the all-gather form north_star names
        variants = {}
        def quick(Sx):
            ti = ctx.timed(lambda: Sx.step(True), args.steps, args.warmup, 3)
            tr = ctx.timed(lambda: Sx.step(False), args.steps, args.warmup, 3)
            Sx.synchronize()
            kpx = ctx.sum_over_ranks(float(int(Sx.counts(0).sum().item())))
            med = lambda ts: float(sorted(ts)[len(ts) // 2])
            return {"value": kpx * launches / med(ti) / 1e6, "kernel_pipeline": kpx * launches / med(tr) / 1e6, "unit": "Mkeypoints/s",
                    "exchanged_MB_per_rank_per_launch": (Sx.slab * (world - 1) if Sx.mode == "allgather" else Sx.slab if Sx.mode == "ring" else Sx.rec * world) / 1e6}
        main_mode = S.mode
        for mode in ("allgather", "ring"):
            if mode != main_mode and main_mode != "neighbour":
                S.mode = mode
                variants[mode] = quick(S)
        S.mode = main_mode
        other = "neighbour" if main_mode != "neighbour" else "allgather"
        S2 = MonoStream(ctx, y, exchange=other)
        variants[other] = quick(S2)
        if other == "allgather":
            S2.mode = "ring"
            variants["ring"] = quick(S2)
        del S2
    out = {"metric": "ORB extract+match Mkeypoints/sec", "value": inc["median"], "unit": "Mkeypoints/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt_inc / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic", "parity": "partial", "parity_note": PARITY_NOTE,
           "value_definition": "H2D/D2H-inclusive (SURVEY.md 8(d): 'GPU timed ... around the batch incl. H2D/D2H'; the reference's frames come from host "
                               "memory, frame.cpp:84-87,129): pinned host frames in, keypoints / descriptors / match lists back in pinned host memory. "
                               "kernel_pipeline is the same work with the frames resident in HBM (the figure rounds 1-2 reported as value).",
           "repeats": {"n": inc["n"], "min": inc["min"], "max": inc["max"], "median": inc["median"], "timed_region_s": dt_inc,
                       "note": "each repeat = exactly --steps steps bracketed by barrier + synchronize; value = median"},
           "config": {"workload": "TUM-fr1-size 640x480 mono stream, 1000 feat/frame, extract + consecutive-frame searchByProjection (th 15); "
                                  "pinned host frames in, results back in pinned host memory",
                      "frames_per_step_per_gpu": F * args.substeps, "frames_per_launch": F, "launches_per_step": args.substeps,
                      "distinct_frames": F * world, "frames_per_scene": args.segment,
                      "pipelining": ("%d lanes (extractor handle + matcher + output set + ONE stream each) take consecutive launches: extraction, matching and "
                                     "read-back of a launch in its lane's stream order; the uploads run back to back on a stream of their own (the 4th "
                                     "hardware queue) into a ring of %d image buffers, a lane waits for its frames' event only.  kernel_pipeline / "
                                     "extract_only: %d lanes, no copies" % (S.INC_LANES, S.NIMG, S.LANES)) if S.s_up is not None else
                                    ("%d lanes (extractor handle + matcher + image buffer + output set + ONE stream each) take consecutive launches: upload, "
                                     "extraction, matching and read-back of a launch in its lane's stream order, the lanes overlap each other" % S.LANES),
                      "motion": "per frame: roll within +-3 deg, shift within +-8 px (bounded walk); prediction = true motion + N(0,1.5^2) px on the translation",
                      "keypoints_per_frame": kp_local / F, "matches_per_pair": matched_local / max(S.NPAIR, 1),
                      "matches_note": "what the reference's rules accept: ~770 of 1000 keypoints of a frame re-appear within 2.5 px x scale in the next one "
                                      "(median Hamming distance 35), but Frame::getKeyPointsInArea keeps only candidates with |dx| > r inside the "
                                      "window's grid cells (frame.cpp:353, reproduced bit for bit) and the rotation histogram uses the factor 1/30 "
                                      "(orbMatcher.cpp:78): ~120 pass the distance test, ~50 the rotation check",
                      "parallelism": par,
                      "hip_runtime": ctx.hip_runtime or "the one bundled with PyTorch (torch/lib)",
                      "hardware_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))},
           "kernel_pipeline": {"value": res["median"], "unit": "Mkeypoints/s", "ms_per_step": dt_res / args.steps * 1e3, "min": res["min"], "max": res["max"],
                               "n": res["n"], "timed_region_s": dt_res,
                               "note": "frames resident in HBM, results left in HBM; the same launches, lanes and streams without the copies"},
           "extract_only": {"value": ext["median"], "unit": "Mkeypoints/s", "min": ext["min"], "max": ext["max"],
                            "note": "extraction launches only (frames resident), same handles and streams"},
           "pcie": pcie, "roofline": roofline}
    if variants is not None:
        out["exchange"] = {"mode": S.mode, "MB_per_rank_per_launch": (S.slab * (world - 1) if S.mode == "allgather" else S.slab if S.mode == "ring" else S.rec * world) / 1e6,
                           "variants": variants,
                           "note": "value / kernel_pipeline above use `mode`; variants = the other exchange forms in the same run (3 repeats): ring = round-robin "
                                   "frames, each rank sends its slab to rank + 1 only; neighbour = contiguous frame shards, only the boundary frame travels"}
    return out, S


# ---------------------------------------------------------------------------------------------------------------------------------
# Configs 3 and 4: stereo streams (extract L + R, computeStereoMatches, consecutive left-frame search), inclusive and resident
# ---------------------------------------------------------------------------------------------------------------------------------
def stereo_config(ctx, y, link, w, h, nf, n_pairs, label, repeats, tile_default=1, copy_default="lane", sets_default=4):
    import numpy as np
    from ydorbslam_amd.synth import stream_plan, stream_render
    torch, dev, args = ctx.torch, ctx.dev, ctx.args
    # `tile` launches' worth of the stream per launch: the two serial chains of a launch (stereo replay, ordered resolve) take the same time
    # for 128 pairs as for 512 (one wave per pair), so larger launches hide them better behind the other lanes' extractions
    # (measured, inclusive / resident Mkeypoints/s with 4 lanes - tools/bench_stereo_sweep.sh: config 3 at 128 / 256 pairs per launch 133 / 193 ->
    # 186 / 205; config 4 at 64 / 128 / 256 / 512 pairs per launch 79 / 138 -> 93 / 171 -> 124 / 190 -> 139 / 194; with the upload stream,
    # tools/bench_stereo_tile.sh: config 3 at 256 / 512 / 1024 pairs 206 / 226 / 225 inclusive, config 4 at 512 / 1024 pairs 146 / 154.5)
    tile = max(1, int(os.environ.get("YDORB_BENCH_STEREO_TILE", str(tile_default))))
    distinct = n_pairs
    pl = stream_plan(w, h, distinct, seed=7, segment=32)
    L_, R_ = stream_render(pl, range(distinct), stereo=True)
    if tile > 1:
        L_, R_ = np.concatenate([L_] * tile), np.concatenate([R_] * tile)
        n_pairs = distinct * tile
    hL, hR = torch.from_numpy(L_).pin_memory(), torch.from_numpy(R_).pin_memory()
    mk = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
    prs = np.array([(i, i + 1) for i in range(n_pairs - 1)], np.int32)
    pred_ = np.ascontiguousarray(pl["predicted"], np.float32)
    if tile > 1:   # the pair that joins two copies of the stream is a scene cut (identity prediction)
        ident = np.array([[1, 0, 0, 0, 1, 0]], np.float32)
        pred_ = np.concatenate([pred_] + [np.concatenate([ident, pred_])] * (tile - 1))
    daf = torch.from_numpy(pred_).to(dev)
    # NSETS lanes, each a complete set (extractor pair, outputs, matchers) with ONE stream: a launch's two extractions, its association (the
    # serial replay of frame.cpp:391-462) and its left-frame search (ordered resolve) run back to back on the lane's stream, consecutive
    # launches go to consecutive lanes, no events between lanes: the two latency chains of a launch hide behind the extractions of the other
    # lanes.  The handles are single-stream (YDORB_EXTRACTOR_SINGLE_STREAM): with side streams, 5+ streams share the device's 4 hardware
    # queues and a stream that lands behind a chain stalls.
    NSETS = int(os.environ.get("YDORB_BENCH_STEREO_SETS", str(sets_default)))
    # YDORB_BENCH_COPY: "lane" = every lane uploads its own pairs in stream order; "upload" = the last lane's stream only uploads, back to back
    # into a ring of image pairs, the other lanes take the launches (MonoStream.UPLOAD_STREAM); "streams" = a copy-in and a copy-out stream
    # beside the lanes (round 2's form).  Config 3 is bound by its kernels and needs many lanes to hide its two serial chains: with the
    # process's 8 hardware queues (GPU_MAX_HW_QUEUES, main()) 7 lanes + the upload stream 207 (202-211) against 4 lanes with in-lane copies
    # 184-193 (138-212), 3 lanes + upload stream 167; config 4 is bound by the link: 3 lanes + upload stream 146, in-lane 138
    # (tools/bench_c3_ab.sh, tools/bench_hwq_sweep.sh, profiles/r03_sweeps.txt).
    copy_mode = os.environ.get("YDORB_BENCH_COPY", copy_default)
    in_lane, up_mode = copy_mode == "lane", copy_mode == "upload" and NSETS >= 2
    INC = NSETS - 1 if up_mode else NSETS                     # lanes that take the inclusive pipeline's launches
    RING = INC + 2 if up_mode else NSETS if in_lane else NSETS + 1
    lanes = [torch.cuda.Stream(device=dev) for _ in range(NSETS)]
    if up_mode:
        s_in = lanes[-1]
        s_out = None
    else:
        s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ring = [(torch.empty_like(hL, device=dev), torch.empty_like(hR, device=dev)) for _ in range(RING)]
    ring[0][0].copy_(hL); ring[0][1].copy_(hR)
    ev = lambda n_: [torch.cuda.Event() for _ in range(n_)]
    ev_in, ev_free, ev_done, ev_out = ev(RING), ev(RING), ev(NSETS), ev(NSETS)
    free_used = [False] * RING
    sets = []
    for i in range(NSETS):
        S_ = dict(xL=y.OrbExtractor(nf, 1.2, 8, 20, 7, device=ctx.local_rank, max_batch=n_pairs, single_stream=True),
                  xR=y.OrbExtractor(nf, 1.2, 8, 20, 7, device=ctx.local_rank, max_batch=n_pairs, single_stream=True))
        scap = S_["xL"].max_keypoints
        S_.update(kL=mk(n_pairs, scap, 7), kR=mk(n_pairs, scap, 7), dL=mk(n_pairs, scap, 32, dt=torch.uint8), dR=mk(n_pairs, scap, 32, dt=torch.uint8),
                  nL=mk(n_pairs, dt=torch.int32), nR=mk(n_pairs, dt=torch.int32), rx=mk(n_pairs, scap), dp=mk(n_pairs, scap), kept=mk(n_pairs, dt=torch.int32),
                  asg=mk(n_pairs - 1, scap, dt=torch.int32), cnt=mk(n_pairs - 1, dt=torch.int32), sm=y.OrbMatcher(device=ctx.local_rank),
                  mm=y.OrbMatcher(0.9, True, device=ctx.local_rank), st=lanes[i])
        S_["outs"] = [S_[k_] for k_ in ("kL", "kR", "dL", "dR", "nL", "nR", "rx", "dp", "kept", "asg", "cnt")]
        S_["h_outs"] = [torch.zeros_like(t, device="cpu").pin_memory() for t in S_["outs"]]
        sets.append(S_)
    scap = sets[0]["xL"].max_keypoints
    ssf = sets[0]["xL"].tables()["scale"]
    kstep = [0]
    parts = os.environ.get("YDORB_BENCH_STEREO_PARTS", "stereo,match").split(",")   # diagnostic: leave the association or the search out

    def one(inclusive, full=True):
        k = kstep[0]
        kstep[0] += 1
        b = k % (INC if inclusive else NSETS)
        S_ = sets[b]
        sA, st = S_["st"], S_["st"].cuda_stream
        dL_, dR_ = ring[0]
        r = k % RING
        if inclusive and in_lane:
            dL_, dR_ = ring[b]
            with torch.cuda.stream(sA):
                dL_.copy_(hL, non_blocking=True)
                dR_.copy_(hR, non_blocking=True)
        elif inclusive:
            dL_, dR_ = ring[r]
            if not up_mode or free_used[r]:
                s_in.wait_event(ev_free[r])
            with torch.cuda.stream(s_in):
                dL_.copy_(hL, non_blocking=True)
                dR_.copy_(hR, non_blocking=True)
            ev_in[r].record(s_in)
            sA.wait_event(ev_in[r])
            if not up_mode:
                sA.wait_event(ev_out[b])
        S_["xL"].extract_batch_device(dL_.data_ptr(), w, h, w, w * h, n_pairs, S_["kL"].data_ptr(), S_["dL"].data_ptr(), scap, S_["nL"].data_ptr(), st)
        S_["xR"].extract_batch_device(dR_.data_ptr(), w, h, w, w * h, n_pairs, S_["kR"].data_ptr(), S_["dR"].data_ptr(), scap, S_["nR"].data_ptr(), st)
        if inclusive and not in_lane:
            ev_free[r].record(sA)
            free_used[r] = True
        if full and "stereo" in parts:
            S_["sm"].stereo_matches_device(S_["xL"], S_["xR"], S_["kL"].data_ptr(), S_["dL"].data_ptr(), S_["nL"].data_ptr(), scap, S_["kR"].data_ptr(),
                                           S_["dR"].data_ptr(), S_["nR"].data_ptr(), scap, n_pairs, 40.0, 0.1, S_["rx"].data_ptr(), S_["dp"].data_ptr(),
                                           S_["kept"].data_ptr(), None, False, (0, 1), (0, 1), st)
        if full and "match" in parts:
            fs_ = (S_["kL"].data_ptr(), S_["dL"].data_ptr(), S_["nL"].data_ptr(), n_pairs, scap)
            S_["mm"].match_pairs_device(fs_, fs_, prs, w, h, 15.0, ssf, S_["asg"].data_ptr(), S_["cnt"].data_ptr(), daf.data_ptr(), st)
        if inclusive and (in_lane or up_mode):
            with torch.cuda.stream(sA):
                for h_, d_ in zip(S_["h_outs"], S_["outs"]):
                    h_.copy_(d_, non_blocking=True)
        elif inclusive:
            ev_done[b].record(sA)
            s_out.wait_event(ev_done[b])
            with torch.cuda.stream(s_out):
                for h_, d_ in zip(S_["h_outs"], S_["outs"]):
                    h_.copy_(d_, non_blocking=True)
            ev_out[b].record(s_out)
    # launches per timed region: about one second of work (calibrated on a short run)
    for _ in range(2 * NSETS):
        one(False)
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    for _ in range(2 * NSETS):
        one(False)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t_) / (2 * NSETS)
    n_launch = max(4 * NSETS, int(np.ceil(args.region_s / max(per, 1e-4))))
    res = {}
    for S_ in sets:
        S_["xL"].set_profiling(True)
    kp = None
    for name, inclusive, full, rep in (("inclusive", True, True, repeats), ("kernel_pipeline", False, True, repeats), ("extract_only", False, False, 3)):
        ts = ctx.timed(lambda: one(inclusive, full), n_launch, NSETS, rep)
        if kp is None:
            kp = float(sets[0]["nL"].sum().item() + sets[0]["nR"].sum().item())
        st_ = stats([kp * n_launch / t / 1e6 for t in ts])
        med_t = float(sorted(ts)[len(ts) // 2])
        res[name] = {"value": st_["median"], "unit": "Mkeypoints/s", "ms_per_launch": med_t / n_launch * 1e3, "min": st_["min"], "max": st_["max"], "n": st_["n"],
                     "timed_region_s": med_t, "launches_per_region": n_launch}
        if name == "kernel_pipeline":
            live = {}
            for S_ in sets:
                for k_, v_ in S_["xL"].stage_times().items():
                    live.setdefault(k_, []).append(v_)
            live = {k_: float(np.mean(v_)) for k_, v_ in live.items()}
            for S_ in sets:
                S_["xL"].set_profiling(False)
    for S_ in sets:
        S_["xL"].synchronize(); S_["xR"].synchronize(); S_["sm"].synchronize(); S_["mm"].synchronize()
    A_ = algorithmic_bytes_extract(w, h, nf)
    kp_img = float(sets[0]["nL"].float().mean().item())
    bytes_in = 2 * n_pairs * w * h
    bytes_out = sum(t.numel() * t.element_size() for t in sets[0]["h_outs"])
    t_inc = res["inclusive"]["ms_per_launch"] * 1e-3
    t_res = res["kernel_pipeline"]["ms_per_launch"] * 1e-3
    out = {"workload": label, "value": res["inclusive"]["value"], "unit": "Mkeypoints/s", "inclusive": res["inclusive"], "kernel_pipeline": res["kernel_pipeline"],
           "extract_only": res["extract_only"], "stereo_pairs_per_launch": n_pairs, "distinct_stereo_pairs": distinct, "keypoints_per_image": kp_img,
           "stereo_measurements_per_pair": float(sets[0]["kept"].float().mean().item()), "matches_per_left_pair": float(sets[0]["cnt"].float().mean().item()),
           "pipelining": "%d lanes (handle pair + matchers + one stream each) take consecutive launches; a launch's extractions, association, search and "
                         "read-back run back to back on its lane; copies: %s (ring of %d device image pairs).  kernel_pipeline / extract_only: %d lanes, no copies"
                         % (INC, "the uploads back to back on a stream of their own" if up_mode else "uploads in the lane too" if in_lane else
                            "a copy-in and a copy-out stream", RING, NSETS),
           "pcie": {"host_to_device_MB_per_launch": bytes_in / 1e6, "device_to_host_MB_per_launch": bytes_out / 1e6,
                    "achieved_GBps": bytes_in / t_inc / 1e9, "achieved_d2h_GBps": bytes_out / t_inc / 1e9, "peak_GBps": link["peak_GBps"],
                    "frac": bytes_in / t_inc / (link["peak_GBps"] * 1e9)},
           "roofline": extract_roofline(w, h, nf, kp_img, live, None, n_pairs, n_pairs - 1, A_ * 2 * n_pairs / t_res)}
    del sets
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Config 5: local BA
# ---------------------------------------------------------------------------------------------------------------------------------
def ba_section(ctx, y):
    from ydorbslam_amd.synth import synth_ba_problem
    torch, args, world, rank = ctx.torch, ctx.args, ctx.world, ctx.rank
    prob = synth_ba_problem(100, 10000, 8, seed=1)
    kw = {}
    if world > 1:  # shard landmarks (and their edges) across ranks; every rank holds all poses
        from ydorbslam_amd.parallel import shard_ba_problem
        prob, _, _ = shard_ba_problem(prob, rank, world)
        comm = torch.zeros(640 * 641 + 4096, dtype=torch.float64, device=ctx.dev)  # >= n*n + n doubles, n = 6*K rounded up to 32

        def allreduce(user, d_buf, count, op):
            try:
                ctx.all_reduce_(comm[:count], ctx.dist.ReduceOp.MAX if op == 1 else ctx.dist.ReduceOp.SUM)
                torch.cuda.synchronize()
                return 0
            except Exception:  # noqa: BLE001
                return 1
        kw = dict(allreduce=allreduce, comm_tensor_ptr=comm.data_ptr(), comm_doubles=comm.numel(), rank=rank, world=world)
    opt = y.Optimizer.default_options(device=ctx.local_rank)
    y.Optimizer.local_bundle_adjust(prob, opt, **kw)  # warm-up (allocations, code objects)
    reps, per_rep, r = max(3, args.ba_reps), [], None
    for _ in range(args.repeats):
        ctx.barrier()
        t0 = time.perf_counter()
        trials = 0
        for _ in range(reps):
            r = y.Optimizer.local_bundle_adjust(prob, opt, **kw)
            trials += r["trials"]
        ctx.barrier()
        per_rep.append(trials / ctx.max_over_ranks(time.perf_counter() - t0))
    st_ = stats(per_rep)
    flops_schur = 89.9e6  # Schur part of one LM trial at C5 / 8 obs
    flops_chol = 72.7e6   # (6K)^3/3 + 2(6K)^2, same table
    # the per-phase breakdown comes from one more solve with YDORB_BA_PHASE_TIMES (its event pairs cost ~8 % of a solve: not in the timed ones)
    ms = y.Optimizer.local_bundle_adjust(prob, y.Optimizer.default_options(device=ctx.local_rank, phase_times=True), **kw)["ms"]
    ba = {"metric": "local-BA LM iterations/sec (100 KF x 10k points, 8 obs/point)", "value": st_["median"], "unit": "it/s", "min": st_["min"], "max": st_["max"],
          "n": st_["n"], "solves_per_repeat": reps, "lm_trials_per_solve": r["trials"], "ms_per_solve": r["trials"] / st_["median"] * 1e3,
          "final_chi2": float(r["log"][-1, 0]), "device_ms_per_solve": {k: round(float(v), 3) for k, v in ms.items()},
          "schur_fp64_frac": (flops_schur * r["trials"] / (ms["schur"] * 1e-3) / FP64_PEAK) if ms["schur"] > 0 else None,
          "solve_fp64_frac": (flops_chol * r["trials"] / (ms["solve"] * 1e-3) / FP64_PEAK) if ms["solve"] > 0 else None,
          "fp64_note": "device_ms_per_solve.schur covers k_dinv + k_bd + k_bs + k_schur_pairs, .solve the Cholesky chain + both substitutions; the kernels' own "
                       "rocprof durations are in profiles/.  The Schur mapping uses a 6x6 corner of each 16x16 MFMA tile: 14 % of the FP64 matrix peak is its ceiling",
          "parity_note": "vs the oracle's restatement of g2o (unpinned end to end; dense solver pinned by g2o's linear_solver_test vector, tol 1e-6)",
          "scaling": "strong (landmarks sharded, all-reduce of the reduced camera system)" if world > 1 else "single GPU"}
    if world == 1:
        # Additional figure: independent local-BA problems solved in lock step (ydorb_ba_solve_batch: one launch per phase for all
        # problems, blockIdx.z = problem).  A single solve is a latency chain; the batch is throughput-bound.
        NT = args.ba_threads
        probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(NT)]
        y.Optimizer.local_bundle_adjust_batch(probs, opt, NT)
        rates = []
        for _ in range(max(3, min(args.repeats, 5))):
            tcc = time.perf_counter()
            bres = y.Optimizer.local_bundle_adjust_batch(probs, opt, NT)
            tcc = time.perf_counter() - tcc
            rates.append(sum(b_["trials"] for b_ in bres) / tcc)
        sb = stats(rates)
        ba["concurrent"] = {"problems": len(probs), "in_flight": NT, "value": sb["median"], "unit": "it/s (aggregate)", "min": sb["min"], "max": sb["max"], "n": sb["n"],
                            "ms_per_batch": sum(b_["trials"] for b_ in bres) / sb["median"] * 1e3,
                            "note": "ydorb_ba_solve_batch, lock-step batch: independent copies of the same C5 problem, every result bit-identical to its single solve"}
        y.Optimizer.release(ctx.local_rank)   # the batch's pooled scratch (~2.5 GB) goes back before the stream pipeline allocates
    return ba


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline of the headline: the oracle (port of the reference algorithm) on the host cores, bounded sample.  Rank 0 only.
# ---------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline_section(ctx, S, out):
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle.orb_oracle import FrameOracle, OrbExtractorOracle, QUERY_DTYPE, ba_solve
    from ydorbslam_amd.synth import stream_render, synth_ba_problem
    args, world = ctx.args, ctx.world
    W, H, NFEAT, sf = S.W, S.H, S.NFEAT, S.sf
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
        c_frames, c_affs = S.imgs, S.plan["predicted"]
    else:   # this rank's frames need not be consecutive in the stream: render a contiguous piece for the CPU sample
        c_frames, _ = stream_render(S.plan, range(min(args.cpu_frames, 48)))
        c_affs = S.plan["predicted"]
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
                           "sample": "%d of the same 640x480 frames, extract + consecutive match, oracle (C++ -O3 like the reference's build), 1 thread" % n1,
                           "cpu_model": model, "host_cores": ncpu_all, "usable_cores": ncpu,
                           "all_cores": {"value": nka / tca / 1e6, "unit": "Mkeypoints/s", "cores": len(chunks),
                                         "sample": "%d threads x %d consecutive frames each (frame-parallel; the reference itself uses <= 2 extractor threads, frame.cpp:84-85)" % (len(chunks), per)}}
    out["vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    out["vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores"]["value"]
    if "ba" in out:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=10, help="timed regions of exactly --steps steps each; value = median")
    ap.add_argument("--frames", type=int, default=2048, help="frames per launch and per GPU (measured, inclusive / resident Mkeypoints/s, uploads on their own "
                                                            "stream: 512 frames 158 / 248, 1024 frames 174 / 250, 2048 frames 180 / 250: longer uploads keep the link busier)")
    ap.add_argument("--substeps", type=int, default=8, help="launches per step (a step = substeps x frames frames: 20 steps are then a timed region of > 1 s)")
    ap.add_argument("--region-s", type=float, default=1.0, help="target length of the timed regions of the config 3 / config 4 sections")
    ap.add_argument("--segment", type=int, default=64, help="frames per synthetic scene (a new scene is a cut)")
    ap.add_argument("--exchange", choices=("allgather", "ring", "neighbour"), default="allgather",
                    help="N > 1: round-robin frames + all-gather of every rank's records (SURVEY 8e); round-robin + only the previous rank's records "
                         "(point-to-point); or contiguous shards + boundary frame only")
    ap.add_argument("--cpu-frames", type=int, default=96, help="frames of the 1-thread CPU-oracle sample")
    ap.add_argument("--ba-threads", type=int, default=64, help="problems of the lock-step batched local-BA figure (ydorb_ba_solve_batch)")
    ap.add_argument("--ba-reps", type=int, default=3, help="solves per timed repeat of the single local-BA figure")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="", help="with extras: run only these sections (comma list of single_call, config3, config4, rest)")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-call / config 3 / config 4 / brute-force / next-row sections")
    args = ap.parse_args()

    # The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one run in turn.  The
    # mono pipeline needs four (its default), config 3's eight lanes need eight.  Read when the runtime loads, so set before anything else.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)
        return

    # Only the JSON line may reach stdout: RCCL prints its own warnings there (e.g. "Missing iommu=pt"), so everything any library
    # writes to file descriptor 1 goes to stderr from here on and the result line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    ctx = Ctx(args)
    import ydorbslam_amd as y
    # Config 5 first: with the stream pipeline's handles resident (25 GB of batch buffers for four 2048-frame lanes) the same solve takes
    # 8.8 ms instead of 7.9 (device time, not host time: the solver's pooled buffers are then carved out of a crowded address space)
    ba = None if args.no_ba else ba_section(ctx, y)
    link = pcie_link(ctx)
    out, S = mono_section(ctx, y, link)
    if ba is not None:
        out["ba"] = ba

    extras = ctx.world == 1 and not args.no_extras
    only = [x for x in args.only.split(",") if x]
    want = lambda name: extras and (not only or name in only)
    if want("config3"):
        out["config3"] = stereo_config(ctx, y, link, 1241, 376, 2000, 128, "KITTI-00-size 1241x376 stereo, 2000 feat/image: extract L+R, computeStereoMatches "
                                       "(as the reference writes it), consecutive left-frame search; pinned host frames in, results back in pinned host memory",
                                       max(3, args.repeats // 2), tile_default=4, copy_default="upload", sets_default=8)
    if want("config4"):
        out["config4"] = stereo_config(ctx, y, link, 752, 480, 1000, 64, "EuRoC-MH-size 752x480 stereo batch, 1000 feat/image, one GPU's share: extract L+R, "
                                       "computeStereoMatches, consecutive left-frame search; pinned host frames in, results back in pinned host memory",
                                       max(3, args.repeats // 2), tile_default=16, copy_default="upload")
    state = None
    if extras:
        import bench_extras
        state = bench_extras.run(ctx, y, S, out, want)
    if ctx.rank == 0 and not args.no_cpu:
        cpu_baseline_section(ctx, S, out)
        if state is not None:
            import bench_extras
            bench_extras.cpu_baselines(ctx, S, out, state)
    if ctx.rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if ctx.distributed:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
