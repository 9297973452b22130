"""Multi-GPU host logic (one process per GPU, torch.distributed over RCCL): how the hot path shards.

* extract + match: frames are independent units.  The frames of a stream are dealt ROUND-ROBIN (global frame g lives on rank
  g % world as local frame g // world), so extraction needs no collective and every consecutive pair straddles two ranks.  Per
  step each rank all-gathers every rank's [keypoints 28 B | descriptors 32 B | count] records (three large collectives straight
  out of the extractor's output buffers, which ARE this rank's slice of the gathered set: no packing copy), then matches, locally
  against the gathered set, the pairs whose later frame it owns (`round_robin_pairs`); greedy acceptance stays with that owner.
  (`frame_shard` / `pack_boundary`: the contiguous-chunk alternative, where only the chunk boundary frame is exchanged.)
* local BA: landmarks (and their edges) shard across ranks, every rank holds all poses; per LM trial one all-reduce(sum)
  of the reduced camera system (+ small reductions of chi2 / scale), see ydorbslam_amd/csrc/ba_solver.hip.
Nothing here computes on the hot path; it only slices inputs and plans index lists.
"""
import numpy as np


def frame_shard(n_frames, rank, world):
    """[begin, end) of the frames rank owns (contiguous chunks keep consecutive-frame pairs local)."""
    base, rem = divmod(n_frames, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def round_robin_pairs(rank, world, frames_per_rank):
    """Pairs (query, target) this rank matches, as indices into the gathered set laid out rank-major ([world][frames_per_rank]):
    target = an owned frame t (global g = t*world + rank), query = its predecessor g-1 in the stream, which lives on rank
    (g-1) % world.  Returns (pairs [n,2] int32, pred [n] = g-1, the index of the pair's motion / affine in the stream)."""
    F = frames_per_rank
    pairs, pred = [], []
    for t in range(F):
        g = t * world + rank
        if g == 0:
            continue
        pr, pt = (g - 1) % world, (g - 1) // world
        pairs.append((pr * F + pt, rank * F + t))
        pred.append(g - 1)
    return np.array(pairs, np.int32).reshape(-1, 2), np.array(pred, np.int64)


def ring_pairs(rank, world, frames_per_rank):
    """The same pairs as round_robin_pairs, as (query, target) LOCAL frame indices: with the frames dealt round-robin every owned frame's
    predecessor lives on rank - 1 (mod world), so the queries of a launch are that ONE rank's records - local index t for rank > 0
    (g - 1 = t * world + rank - 1), t - 1 for rank 0 (g - 1 = (t - 1) * world + world - 1; global frame 0 has no predecessor).
    Returns (pairs [n,2] int32, pred [n] = g - 1)."""
    F = frames_per_rank
    pairs, pred = [], []
    for t in range(F):
        g = t * world + rank
        if g == 0:
            continue
        pairs.append((t if rank > 0 else t - 1, t))
        pred.append(g - 1)
    return np.array(pairs, np.int32).reshape(-1, 2), np.array(pred, np.int64)


def boundary_record_bytes(cap):
    return cap * (28 + 32) + 4


def pack_boundary(kps_bytes, desc, n, cap):
    """kps_bytes: uint8[cap*28], desc: uint8[cap,32], n: int -> uint8 record."""
    rec = np.zeros(boundary_record_bytes(cap), np.uint8)
    rec[:cap * 28] = np.asarray(kps_bytes, np.uint8).reshape(-1)[:cap * 28]
    rec[cap * 28:cap * 60] = np.asarray(desc, np.uint8).reshape(-1)[:cap * 32]
    rec[cap * 60:] = np.array([n], np.int32).view(np.uint8)
    return rec


def unpack_boundary(rec, cap):
    rec = np.asarray(rec, np.uint8)
    return rec[:cap * 28].copy(), rec[cap * 28:cap * 60].reshape(cap, 32).copy(), int(rec[cap * 60:].view(np.int32)[0])


def shard_ba_problem(prob, rank, world):
    """Landmark l belongs to rank l % world; its edges follow it; poses are replicated."""
    keep = np.arange(len(prob["points"])) % world == rank
    remap = np.cumsum(keep) - 1
    ke = keep[prob["edge_point"]]
    sub = dict(prob)
    sub["points"] = prob["points"][keep]
    sub["edge_pose"] = prob["edge_pose"][ke]
    sub["edge_point"] = remap[prob["edge_point"][ke]].astype(np.int32)
    sub["meas"], sub["info"] = prob["meas"][ke], prob["info"][ke]
    if "truth_points" in prob:
        sub["truth_points"] = prob["truth_points"][keep]
    return sub, keep, ke
