"""Deterministic synthetic inputs: frames, warps, BA problems.  numpy only."""
import os
import numpy as np


def _noise_octave(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw), dtype=np.float32)
    ys = (np.arange(h, dtype=np.float32) + 0.5) / cell
    xs = (np.arange(w, dtype=np.float32) + 0.5) / cell
    y0 = np.floor(ys).astype(np.int32); x0 = np.floor(xs).astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def synth_frame(w=640, h=480, index=0, n_rects=None, n_blobs=200):
    """uint8 frame: 3 octaves of bilinear noise + rectangles + Gaussian blobs + N(0,2^2) noise; seed = 1000+index."""
    rng = np.random.default_rng(1000 + index)
    if n_rects is None:
        n_rects = {640: 400, 752: 600, 1241: 800}.get(w, max(100, w * h // 768))
    img = 0.5 * _noise_octave(rng, h, w, 64) + 0.3 * _noise_octave(rng, h, w, 16) + 0.2 * _noise_octave(rng, h, w, 4)
    img = (img - img.min()) / max(float(img.max() - img.min()), 1e-6) * 255.0
    for _ in range(n_rects):
        rw, rh = int(rng.integers(4, 40)), int(rng.integers(4, 40))
        x, y = int(rng.integers(0, w - rw)), int(rng.integers(0, h - rh))
        img[y:y + rh, x:x + rw] = np.clip(img[y:y + rh, x:x + rw] + float(rng.choice([-1.0, 1.0])) * float(rng.uniform(30, 90)), 0, 255)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(n_blobs):
        cx, cy, s = float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(1.5, 4.0))
        amp = float(rng.choice([-1.0, 1.0])) * float(rng.uniform(40, 100))
        r = int(4 * s) + 1
        x0, x1, y0, y1 = max(0, int(cx) - r), min(w, int(cx) + r + 1), max(0, int(cy) - r), min(h, int(cy) + r + 1)
        if x1 > x0 and y1 > y0:
            img[y0:y1, x0:x1] += amp * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * s * s))
    img = img + rng.normal(0, 2.0, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_stereo_pair(w=640, h=480, index=0, disparities=(7, 15, 26), noise=1.5):
    """Rectified pair: the right image is the left one moved left by a per-band disparity (three horizontal bands), plus
    independent N(0, noise^2) on the right.  Returns (left, right, disparity_of_row[h])."""
    rng = np.random.default_rng(7000 + index)
    pad = int(max(disparities)) + 1
    base = synth_frame(w + pad, h, 300 + index).astype(np.float32)
    left = base[:, :w]
    right = np.empty_like(left)
    drow = np.zeros(h, np.int32)
    bands = np.linspace(0, h, len(disparities) + 1).astype(int)
    for b, d in enumerate(disparities):
        right[bands[b]:bands[b + 1]] = base[bands[b]:bands[b + 1], d:d + w]
        drow[bands[b]:bands[b + 1]] = d
    right = right + rng.normal(0, noise, right.shape)
    return left.astype(np.uint8), np.clip(np.rint(right), 0, 255).astype(np.uint8), drow


def synth_batch(w, h, n, start=0):
    return np.stack([synth_frame(w, h, start + i) for i in range(n)])


# ---------------------------------------------------------------------------------------------
# Synthetic local-BA problem: K cameras on a line, P points, O observations each.
# ---------------------------------------------------------------------------------------------
def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def _rot_to_quat(Rm):
    w = np.sqrt(max(0.0, 1 + Rm[0, 0] + Rm[1, 1] + Rm[2, 2])) / 2
    x = (Rm[2, 1] - Rm[1, 2]) / (4 * w)
    y = (Rm[0, 2] - Rm[2, 0]) / (4 * w)
    z = (Rm[1, 0] - Rm[0, 1]) / (4 * w)
    return np.array([x, y, z, w])


def synth_ba_problem(n_kf=100, n_pts=10000, n_obs=8, seed=1, stereo=True, n_fixed=1, outlier_frac=0.0, mono_frac=0.0):
    """Returns dict(poses[K,7] (t,q) f64, fixed[K] u8, points[P,3] f64, edge_pose, edge_point i32, meas[E,3] f64,
    info[E] f64, camera (fx,fy,cx,cy,bf), truth_poses, truth_points).  Poses are world->camera (T_c2w in the reference's naming)."""
    rng = np.random.default_rng(seed)
    fx = fy = 500.0
    cx, cy, bf = 320.0, 240.0, 40.0
    K, P, O = n_kf, n_pts, min(n_obs, n_kf)
    cam_x = -0.05 * np.arange(K)  # camera centres along -x, identity rotation -> t = -C
    truth_poses = np.zeros((K, 7)); truth_poses[:, 6] = 1.0; truth_poses[:, 0] = -cam_x
    k0 = rng.integers(0, K, P)
    pts = np.stack([cam_x[k0] + rng.uniform(-3, 3, P), rng.uniform(-2, 2, P), rng.uniform(3, 9, P)], axis=1)
    ep, eq, meas = [], [], []
    for p in range(P):
        cams = rng.choice(K, O, replace=False)
        cams.sort()
        for k in cams:
            Xc = pts[p] + truth_poses[k, :3]
            u = fx * Xc[0] / Xc[2] + cx + rng.normal(0, 0.5)
            v = fy * Xc[1] / Xc[2] + cy + rng.normal(0, 0.5)
            ur = u - bf / Xc[2] if (stereo and rng.random() >= mono_frac) else -1.0
            if rng.random() < outlier_frac:
                u += rng.normal(0, 30); v += rng.normal(0, 30)
            ep.append(k); eq.append(p); meas.append((u, v, ur))
    E = len(ep)
    poses = truth_poses.copy()
    for k in range(n_fixed, K):
        d = rng.normal(0, 0.005, 6)
        Rm = _so3_exp(d[:3])
        poses[k, :3] = Rm @ truth_poses[k, :3] + d[3:]
        poses[k, 3:] = _rot_to_quat(Rm)
    fixed = np.zeros(K, np.uint8); fixed[:n_fixed] = 1
    points = pts + rng.normal(0, 0.03, (P, 3))
    return dict(poses=poses, fixed=fixed, points=points, edge_pose=np.array(ep, np.int32), edge_point=np.array(eq, np.int32),
                meas=np.array(meas, np.float64), info=np.ones(E, np.float64), camera=np.array([fx, fy, cx, cy, bf], np.float64),
                truth_poses=truth_poses, truth_points=pts)


def synth_pose_problem(n_pts=400, seed=1, outlier_frac=0.1, mono_frac=0.3, pose_noise=0.02):
    """One Tracking-style pose-only problem (Optimizer::optimizePose, optimizer.cpp:358-501): map points in front of the camera
    (float32 world coordinates, as getPosInWorld returns them), pixel measurements with 1-px noise on a random octave's sigma,
    some gross outliers, and a perturbed start pose.  Returns dict(pose[7], points[E,3], meas[E,3], info[E], camera[5], truth_pose)."""
    rng = np.random.default_rng(seed)
    fx = fy = 500.0
    cx, cy, bf = 320.0, 240.0, 40.0
    w = rng.normal(0, 0.3, 3)
    Rt = _so3_exp(w)
    tt = rng.normal(0, 0.5, 3)
    Xc = np.stack([rng.uniform(-3, 3, n_pts), rng.uniform(-2, 2, n_pts), rng.uniform(2, 10, n_pts)], axis=1)
    Xw = ((Xc - tt) @ Rt).astype(np.float32).astype(np.float64)        # X_c = R X_w + t
    Xc = Xw @ Rt.T + tt
    octave = rng.integers(0, 8, n_pts)
    sigma = 1.2 ** octave
    u = fx * Xc[:, 0] / Xc[:, 2] + cx + rng.normal(0, 1, n_pts) * sigma
    v = fy * Xc[:, 1] / Xc[:, 2] + cy + rng.normal(0, 1, n_pts) * sigma
    ur = np.where(rng.random(n_pts) >= mono_frac, u - bf / Xc[:, 2], -1.0)
    bad = rng.random(n_pts) < outlier_frac
    u = u + bad * rng.normal(0, 40, n_pts)
    v = v + bad * rng.normal(0, 40, n_pts)
    meas = np.stack([u, v, ur], axis=1).astype(np.float32).astype(np.float64)   # cv::KeyPoint / rightX are floats
    info = (1.0 / (sigma * sigma)).astype(np.float32).astype(np.float64)       # m_v_invScaleFactorSquares is a float table
    d = rng.normal(0, pose_noise, 6)
    Rn = _so3_exp(d[:3]) @ Rt
    pose = np.concatenate([Rn @ np.zeros(3) + _so3_exp(d[:3]) @ tt + d[3:], _rot_to_quat(Rn)])
    truth = np.concatenate([tt, _rot_to_quat(Rt)])
    return dict(pose=pose, points=Xw, meas=meas, info=info, camera=np.array([fx, fy, cx, cy, bf], np.float64), truth_pose=truth,
                is_gross_outlier=bad)


# ---------------------------------------------------------------------------------------------
# Synthetic vocabulary tree (the reference's ORB vocabulary is a missing blob): k children per inner node, L levels, node
# descriptors = parent's descriptor with a level-dependent share of flipped bits (so that a descent is decided by real distance
# differences, with occasional ties), idf-like positive weights with a share of stopped (zero-weight) words, and optionally
# branches that end early.  Node ids in breadth-first order as DBoW3 assigns them while clustering; word ids in leaf creation order.
# ---------------------------------------------------------------------------------------------
def synth_vocabulary(k=10, levels=3, seed=0, early_leaf_frac=0.0, stopped_frac=0.05):
    rng = np.random.default_rng(9000 + seed)
    desc = [rng.integers(0, 256, 32, dtype=np.uint8)]
    depth = [0]
    children = [[]]
    frontier = [0]
    for lvl in range(1, levels + 1):
        nxt = []
        for u in frontier:
            if lvl > 1 and rng.random() < early_leaf_frac:
                continue                      # u stays a leaf above the bottom level
            kk = k if rng.random() > 0.2 else int(rng.integers(2, k + 1))
            for _ in range(kk):
                flips = rng.random(256) < 0.5 / (lvl + 0.5)
                d = np.packbits(np.unpackbits(desc[u]) ^ flips.astype(np.uint8))
                desc.append(d); depth.append(lvl); children.append([])
                children[u].append(len(desc) - 1)
                nxt.append(len(desc) - 1)
        frontier = nxt
    n = len(desc)
    child_begin = np.zeros(n + 1, np.int32)
    child_ids = []
    for u in range(n):
        order = list(children[u])
        if len(order) > 2 and rng.random() < 0.3:
            order = [order[i] for i in rng.permutation(len(order))]   # Node::children need not be ascending
        child_ids += order
        child_begin[u + 1] = len(child_ids)
    word = np.full(n, 0, np.int32)
    weight = np.zeros(n, np.float64)
    leaves = [u for u in range(n) if child_begin[u + 1] == child_begin[u]]
    for wid, u in enumerate(leaves):
        word[u] = wid
        weight[u] = 0.0 if rng.random() < stopped_frac else float(np.log(1.0 + rng.uniform(0.5, 200.0)))
    return dict(levels=levels, child_begin=child_begin, child_ids=np.array(child_ids, np.int32), node_desc=np.stack(desc),
                node_weight=weight, node_word=word, n_words=len(leaves))


# ---------------------------------------------------------------------------------------------
# Synthetic frame STREAM (SURVEY.md §8(d) "synthetic matching workload"): every frame distinct, frame t+1 = frame t seen after a
# small known motion (roll within +-3 deg and shift within +-8 px per step), so that consecutive-frame matching has true
# correspondences and a known position prediction.
# ---------------------------------------------------------------------------------------------
def _sample_bilinear(canvas, xs, ys):
    x0 = np.floor(xs).astype(np.int32); y0 = np.floor(ys).astype(np.int32)
    fx = (xs - x0).astype(np.float32); fy = (ys - y0).astype(np.float32)
    np.clip(x0, 0, canvas.shape[1] - 2, out=x0); np.clip(y0, 0, canvas.shape[0] - 2, out=y0)
    a = canvas[y0, x0]; b = canvas[y0, x0 + 1]; c = canvas[y0 + 1, x0]; d = canvas[y0 + 1, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def stream_plan(w, h, n_frames, seed=0, segment=64, pred_sigma=1.5):
    """Motion of a stream of n_frames frames (cheap; every rank of a multi-GPU run computes the plan of the WHOLE stream and renders
    only the frames it owns).  Every `segment` frames share one scene; frame t views it through a similarity (rotation theta_t about the
    image centre, shift T_t): a bounded random walk with steps |dtheta| <= 3 deg, |dT| <= 8 px (rng seed 2000 + seed).  A new segment
    is a scene cut.  Returns dict(pose [F,3] (theta, tx, ty), affine [F-1,6] f32: the 2x3 map taking a pixel of frame t to its place in
    frame t+1 (identity across a cut), predicted [F-1,6] f32: the same with the translation off by N(0, pred_sigma^2) px (a motion
    model's prediction error), cut [F-1] bool, plus the arguments)."""
    rng = np.random.default_rng(2000 + seed)
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    poses = np.zeros((n_frames, 3))
    aff = np.zeros((max(n_frames - 1, 0), 6), np.float32)
    pred = np.zeros_like(aff)
    cut = np.zeros(max(n_frames - 1, 0), bool)
    lim = np.array([np.deg2rad(9.0), 24.0, 24.0])
    pose = None
    for t in range(n_frames):
        if t % segment == 0:
            new = np.array([rng.uniform(-lim[0], lim[0]) * 0.3, rng.uniform(-8, 8), rng.uniform(-8, 8)])
            is_cut = True
        else:
            step = np.array([np.deg2rad(rng.uniform(-3, 3)), rng.uniform(-8, 8), rng.uniform(-8, 8)])
            new = pose + step
            over = np.abs(new) > lim                      # reflect at the bounds: the walk stays inside the canvas margin
            new[over] = pose[over] - step[over]
            is_cut = False
        if t > 0:
            if is_cut:
                A = np.array([1, 0, 0, 0, 1, 0], np.float64)
            else:  # p' = c + R1^T (R0 (p - c) + T0 - T1)
                c0, s0, c1, s1 = np.cos(pose[0]), np.sin(pose[0]), np.cos(new[0]), np.sin(new[0])
                R0 = np.array([[c0, -s0], [s0, c0]]); R1 = np.array([[c1, -s1], [s1, c1]])
                M = R1.T @ R0
                cvec = np.array([cx, cy])
                off = cvec + R1.T @ (pose[1:] - new[1:]) - M @ cvec
                A = np.array([M[0, 0], M[0, 1], off[0], M[1, 0], M[1, 1], off[1]])
            aff[t - 1] = A
            P = A.copy()
            if not is_cut:
                P[2] += rng.normal(0, pred_sigma); P[5] += rng.normal(0, pred_sigma)
            pred[t - 1] = P
            cut[t - 1] = is_cut
        pose = new
        poses[t] = pose
    return dict(w=w, h=h, n_frames=n_frames, seed=seed, segment=segment, pose=poses, affine=aff, predicted=pred, cut=cut)


def stream_render(plan, indices, stereo=False, disparities=(7, 15, 26), margin=112, workers=None):
    """Render the frames `indices` of a planned stream: the segment's scene `synth_frame(w + 2*margin, h + 2*margin, seed*1000 + s)`
    sampled bilinearly through the frame's similarity, plus N(0,1) sensor noise (rng seeded per frame, so the pixels of frame t do not
    depend on which process renders it).  Returns (frames [n,h,w] u8, right [n,h,w] u8 or None); the rectified right view shows the
    same rows with the scene moved left by a per-band disparity, with independent N(0,1.5^2) noise."""
    w, h, seed, segment = plan["w"], plan["h"], plan["seed"], plan["segment"]
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    indices = list(indices)
    frames = np.empty((len(indices), h, w), np.uint8)
    right = np.empty((len(indices), h, w), np.uint8) if stereo else None
    if workers is None:
        workers = min(8, os.cpu_count() or 1)
    if workers > 1 and len(indices) >= 4 * workers:   # contiguous chunks on threads (numpy releases the GIL); same pixels: the noise is seeded per frame
        from concurrent.futures import ThreadPoolExecutor
        step = -(-len(indices) // (2 * workers))
        chunks = [(i0, indices[i0:i0 + step]) for i0 in range(0, len(indices), step)]
        with ThreadPoolExecutor(workers) as pool:
            for (i0, idx), (fr, rt) in zip(chunks, pool.map(lambda c: stream_render(plan, c[1], stereo, disparities, margin, workers=1), chunks)):
                frames[i0:i0 + len(idx)] = fr
                if stereo:
                    right[i0:i0 + len(idx)] = rt
        return frames, right
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    xx -= cx; yy -= cy
    bands = np.linspace(0, h, len(disparities) + 1).astype(int)
    drow = np.zeros((h, 1), np.float32)
    for b, d in enumerate(disparities):
        drow[bands[b]:bands[b + 1]] = d
    canvases = {}
    for i, t in enumerate(indices):
        sg = t // segment
        if sg not in canvases:
            canvases = {sg: synth_frame(w + 2 * margin, h + 2 * margin, seed * 1000 + sg).astype(np.float32)}   # keep one
        canvas = canvases[sg]
        pose = plan["pose"][t]
        rng = np.random.default_rng([3000 + seed, t])
        c_, s_ = np.float32(np.cos(pose[0])), np.float32(np.sin(pose[0]))
        xs = c_ * xx - s_ * yy + np.float32(cx + pose[1] + margin)
        ys = s_ * xx + c_ * yy + np.float32(cy + pose[2] + margin)
        img = _sample_bilinear(canvas, xs, ys) + rng.normal(0, 1.0, (h, w)).astype(np.float32)
        frames[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if stereo:
            img = _sample_bilinear(canvas, xs + c_ * drow, ys + s_ * drow) + rng.normal(0, 1.5, (h, w)).astype(np.float32)
            right[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return frames, right


def synth_stream(w=640, h=480, n_frames=64, seed=0, segment=64, stereo=False, disparities=(7, 15, 26), margin=112, pred_sigma=1.5):
    """stream_plan + stream_render of every frame: dict(frames, right, affine, predicted, cut)."""
    plan = stream_plan(w, h, n_frames, seed, segment, pred_sigma)
    frames, right = stream_render(plan, range(n_frames), stereo, disparities, margin)
    return dict(frames=frames, right=right, affine=plan["affine"], predicted=plan["predicted"], cut=plan["cut"])
