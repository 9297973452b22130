"""Which quad-tree kernel (flat / pass) handled each (frame, level) unit, and flat-vs-pass output equality on the GPU."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_frame

shapes = [(640, 480, 1000), (1241, 376, 2000), (752, 480, 1200), (320, 240, 500)]
for (w, h, nf) in shapes:
    imgs = np.stack([synth_frame(w, h, i) for i in range(8)])
    imgs[6, : h // 2] = 128            # half-empty frame: sparse levels
    imgs[5] = 90; imgs[5, 100:140, 100:150] = synth_frame(w, h, 3)[100:140, 100:150]   # one small textured patch
    ex = y.OrbExtractor(nf, max_batch=8)
    r1 = ex.extract_batch(imgs)
    first = sum(ex.debug_read(3, l, f) for l in range(8) for f in range(8))
    ex.synchronize(); r1 = ex.extract_batch(imgs); r1 = ex.extract_batch(imgs)   # re-sized from the counts of the earlier calls
    kinds = [[ex.debug_read(3, l, f) for l in range(8)] for f in range(8)]
    os.environ["YDORB_QT_PASS"] = "1"
    ex2 = y.OrbExtractor(nf, max_batch=8)
    del os.environ["YDORB_QT_PASS"]
    r2 = ex2.extract_batch(imgs)
    same = all(a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) for a, b in zip(r1, r2))
    print((w, h, nf), "flat==pass:", same, "pass units on the first call:", first, "n:", [len(a[0]) for a in r1], "pass-kernel units (frame, level):", [(f, l) for f in range(8) for l in range(8) if kinds[f][l]],
          "candidates:", [[len(ex.debug_read(1, l, f)) for l in range(8)] for f in (0, 5, 6)])
