"""Deterministic synthetic inputs: frames, warps, BA problems.  numpy only."""
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


def synth_batch(w, h, n, start=0):
    return np.stack([synth_frame(w, h, start + i) for i in range(n)])
