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
    # 4. SURVEY 8(e) match exchange: frames dealt round-robin, all-gather of every rank's [keypoints | descriptors | count] records
    #    (each rank's outputs ARE its slice of the gathered set), then a local match of the pairs whose later frame the rank owns.
    #    The records are real (oracle extraction of a small stream); the union over ranks must equal the consecutive matching of the
    #    whole stream done by one process.
    from ydorbslam_amd.synth import stream_plan, stream_render
    Wd, Hd, Fr = 320, 240, 3
    G = Fr * world
    plan = stream_plan(Wd, Hd, G, seed=5, segment=G)
    oex = o.OrbExtractorOracle(300)
    cap2 = int(oex.tables()["per_level"].sum())
    sfo = oex.tables()["scale"]
    KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    g_kps = torch.zeros((G, cap2, 28), dtype=torch.uint8); g_desc = torch.zeros((G, cap2, 32), dtype=torch.uint8); g_n = torch.zeros(G, dtype=torch.int32)
    own = [t * world + rank for t in range(Fr)]
    frames, _ = stream_render(plan, own)
    for t in range(Fr):
        k, d = oex.extract(frames[t])
        g_kps[rank * Fr + t, :len(k)] = torch.from_numpy(k.view(np.uint8).reshape(-1, 28).copy())
        g_desc[rank * Fr + t, :len(k)] = torch.from_numpy(d.copy())
        g_n[rank * Fr + t] = len(k)
    for full in (g_kps, g_desc, g_n):
        mine = full[rank * Fr:(rank + 1) * Fr]
        dist.all_gather_into_tensor(full, mine.clone())

    def match(kq, dq, kt, dt_, A):
        q = np.zeros(len(kq), o.QUERY_DTYPE)
        A = A.astype(np.float32)
        q["u"] = (A[0] * kq["x"] + A[1] * kq["y"]) + A[2]; q["v"] = (A[3] * kq["x"] + A[4] * kq["y"]) + A[5]
        q["r"] = (np.float32(15.0) * sfo[kq["octave"]]).astype(np.float32)
        q["min_level"], q["max_level"] = kq["octave"] - 1, kq["octave"] + 1
        q["angle"], q["level"] = kq["angle"], kq["octave"]
        q["flags"] = np.where((q["u"] >= 0) & (q["u"] < Wd) & (q["v"] >= 0) & (q["v"] < Hd), 3, 0)
        n_, a_, _ = o.FrameOracle(kt, dt_, (0.0, float(Wd), 0.0, float(Hd))).search_by_projection(1, q, dq, 0.9, True)
        return n_, a_.tolist()

    def rec(i):
        n_ = int(g_n[i])
        return g_kps[i, :n_].numpy().reshape(-1).view(KP).copy(), g_desc[i, :n_].numpy().copy()
    pairs, pred = par.round_robin_pairs(rank, world, Fr)
    res["pairs"] = pairs.tolist()
    res["match"] = {int(p): match(*rec(qi), *rec(ti), plan["predicted"][p]) for (qi, ti), p in zip(pairs, pred)}
    if rank == 0:   # the same stream, one process, consecutive pairs
        allf, _ = stream_render(plan, range(G))
        recs = [oex.extract(f) for f in allf]
        res["match_single"] = {g: match(*recs[g], *recs[g + 1], plan["predicted"][g]) for g in range(G - 1)}
    # 5. the ordering bench.py runs: the exchange of launch k is in flight (communication stream there, async collectives here) while
    #    launch k + 1 is being extracted into ANOTHER output set; launch 1 shows the stream's frames in reverse order of ownership
    #    (a different set of records), so a gather that landed in the wrong set or was matched too early gives different matches.
    sets = [(g_kps, g_desc, g_n), (torch.zeros_like(g_kps), torch.zeros_like(g_desc), torch.zeros_like(g_n))]
    works = []
    def issue(b):
        return [dist.all_gather_into_tensor(full, full[rank * Fr:(rank + 1) * Fr].clone(), async_op=True) for full in sets[b]]
    for full in sets[0]:
        full[:rank * Fr] = 0; full[(rank + 1) * Fr:] = 0          # forget what step 4 gathered: only the own slice is known
    works.append(issue(0))                                        # launch 0's exchange starts ...
    frames1, _ = stream_render(plan, [G - 1 - g for g in own])    # ... while launch 1 is "extracted" (the stream mirrored in time)
    for t in range(Fr):
        k, d = oex.extract(frames1[t])
        sets[1][0][rank * Fr + t, :len(k)] = torch.from_numpy(k.view(np.uint8).reshape(-1, 28).copy())
        sets[1][1][rank * Fr + t, :len(k)] = torch.from_numpy(d.copy())
        sets[1][2][rank * Fr + t] = len(k)
    for w_ in works[0]:
        w_.wait()
    works.append(issue(1))
    res["match_overlapped"] = {int(p): match(*rec(qi), *rec(ti), plan["predicted"][p]) for (qi, ti), p in zip(pairs, pred)}   # set 0 again
    for w_ in works[1]:
        w_.wait()
    res["set1_counts"] = sets[1][2].tolist()
    # 6. neighbour exchange (bench.py --exchange neighbour): contiguous shards, only each rank's LAST frame record travels; the set a
    #    rank matches on is [own Fr frames | every rank's boundary frame]
    lo_g, hi_g = par.frame_shard(G, rank, world)
    framesc, _ = stream_render(plan, range(lo_g, hi_g))
    n_kps = torch.zeros((Fr + world, cap2, 28), dtype=torch.uint8); n_desc = torch.zeros((Fr + world, cap2, 32), dtype=torch.uint8)
    n_n = torch.zeros(Fr + world, dtype=torch.int32)
    for t in range(Fr):
        k, d = oex.extract(framesc[t])
        n_kps[t, :len(k)] = torch.from_numpy(k.view(np.uint8).reshape(-1, 28).copy()); n_desc[t, :len(k)] = torch.from_numpy(d.copy()); n_n[t] = len(k)
    for full in (n_kps, n_desc, n_n):
        dist.all_gather_into_tensor(full[Fr:], full[Fr - 1:Fr].clone())

    def recn(i):
        n_ = int(n_n[i])
        return n_kps[i, :n_].numpy().reshape(-1).view(KP).copy(), n_desc[i, :n_].numpy().copy()
    npairs = [(t - 1, t, lo_g + t - 1) for t in range(1, Fr)] + ([(Fr + rank - 1, 0, lo_g - 1)] if rank > 0 else [])
    res["match_neighbour"] = {int(p): match(*recn(qi), *recn(ti), plan["predicted"][p]) for qi, ti, p in npairs}
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
    # every consecutive pair of the stream is matched exactly once, by the owner of its later frame, with the single-process answer
    merged = {}
    for r in range(world):
        for g, v in out[r]["match"].items():
            assert g not in merged and (g + 1) % world == r
            merged[g] = v
    single = out[0]["match_single"]
    assert sorted(merged) == sorted(single) == list(range(world * 3 - 1))
    # the overlapped ordering gives the same matches, and launch 1's gathered set is complete on every rank
    for r in range(world):
        assert out[r]["match_overlapped"] == out[r]["match"]
        assert out[r]["set1_counts"] == out[0]["set1_counts"] and min(out[r]["set1_counts"]) > 0
    # neighbour exchange: every consecutive pair exactly once, with the single-process answer
    nb = {}
    for r in range(world):
        for g, v in out[r]["match_neighbour"].items():
            assert g not in nb
            nb[g] = v
    assert sorted(nb) == sorted(single)
    for g in single:
        assert nb[g] == single[g], g
    for g in single:
        assert merged[g][0] == single[g][0] and merged[g][1] == single[g][1], "pair %d" % g
    assert sum(v[0] for v in single.values()) > 0
    assert out[1]["pairs"][0] == [0, 3]        # rank 1's frame 0 (global 1) is searched for rank 0's frame 0 (global 0)


def test_ring_pairs_are_the_round_robin_pairs_in_local_indices():
    """parallel.ring_pairs: with the frames dealt round-robin every query of a rank lives on rank - 1 (mod world); the pairs must be
    those of round_robin_pairs (indices into the rank-major gathered set), and over all ranks every consecutive pair exactly once."""
    from ydorbslam_amd import parallel as par
    for world in (1, 2, 3, 8):
        F = 5
        seen = []
        for rank in range(world):
            gp, gpred = par.round_robin_pairs(rank, world, F)
            lp, lpred = par.ring_pairs(rank, world, F)
            assert np.array_equal(gpred, lpred) and len(gp) == len(lp)
            prev = (rank - 1) % world
            assert np.array_equal(gp[:, 0], prev * F + lp[:, 0]) and np.array_equal(gp[:, 1], rank * F + lp[:, 1])
            seen += lpred.tolist()
        assert sorted(seen) == list(range(world * F - 1))
