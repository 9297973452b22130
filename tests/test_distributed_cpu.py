"""N > 1 host logic on CPU: 2 ranks over gloo (torch.distributed), the same slicing / packing / reduction pattern the GPU
path runs over RCCL.  No HIP kernel runs here; per-rank BA terms come from the oracle's residual so the all-reduce
combination can be checked against the unsharded value."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ydorbslam_amd import parallel as par
    from ydorbslam_amd.synth import synth_ba_problem
    from oracle import orb_oracle as o
    import ctypes as C
    res = {}
    # 1. frame sharding covers the stream exactly once, in order
    spans = [par.frame_shard(37, r, world) for r in range(world)]
    res["spans"] = spans
    # 2. boundary exchange: all-gather of each rank's last-frame record
    cap = 64
    rng = np.random.default_rng(100 + rank)
    kps = rng.integers(0, 256, cap * 28, dtype=np.uint8); desc = rng.integers(0, 256, (cap, 32), dtype=np.uint8); n = 40 + rank
    rec = torch.from_numpy(par.pack_boundary(kps, desc, n, cap))
    gathered = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(gathered, rec)
    prev = (rank - 1) % world
    pk, pd, pn = par.unpack_boundary(gathered[prev].numpy(), cap)
    prng = np.random.default_rng(100 + prev)
    res["boundary_ok"] = bool(np.array_equal(pk, prng.integers(0, 256, cap * 28, dtype=np.uint8)) and
                              np.array_equal(pd, prng.integers(0, 256, (cap, 32), dtype=np.uint8)) and pn == 40 + prev)
    # 3. BA landmark sharding: sum over ranks of the per-rank chi2 == chi2 of the whole problem
    prob = synth_ba_problem(6, 120, 4, seed=3)
    sub, keep, ke = par.shard_ba_problem(prob, rank, world)
    L = o.lib()
    def chi2(p):
        tot = 0.0
        e = np.zeros(3)
        for i in range(len(p["edge_pose"])):
            pose = np.ascontiguousarray(p["poses"][p["edge_pose"][i]]); X = np.ascontiguousarray(p["points"][p["edge_point"][i]])
            z = np.ascontiguousarray(p["meas"][i])
            L.yo_ba_residual(pose.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), int(z[2] >= 0),
                             np.ascontiguousarray(p["camera"]).ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p))
            tot += p["info"][i] * float(e @ e)
        return tot
    part = torch.tensor([chi2(sub), float(len(sub["points"])), float(len(sub["edge_pose"]))], dtype=torch.float64)
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    res["chi2_sum"], res["n_pts"], res["n_edges"] = part.tolist()
    res["chi2_full"] = chi2(prob)
    res["n_pts_full"], res["n_edges_full"] = len(prob["points"]), len(prob["edge_pose"])
    mx = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    res["max"] = mx.item()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        res = out[r]
        assert res["spans"] == [(0, 19), (19, 37)]
        assert res["boundary_ok"]
        assert res["n_pts"] == res["n_pts_full"] and res["n_edges"] == res["n_edges_full"]
        assert abs(res["chi2_sum"] - res["chi2_full"]) <= 1e-9 * res["chi2_full"]
        assert res["max"] == world
