"""Stage-by-stage diagnostic of the HIP extractor vs the oracle (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from oracle.orb_oracle import OrbExtractorOracle
from ydorbslam_amd.synth import synth_frame

w, h, nf = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (640, 480, 1000)))
img = synth_frame(w, h, 0)
gpu = y.OrbExtractor(nf); cpu = OrbExtractorOracle(nf)
t = time.time(); gk, gd = gpu.extract(img); print("gpu extract s", time.time() - t, len(gk))
t = time.time(); ck, cd = cpu.extract(img); print("cpu extract s", time.time() - t, len(ck))
for l in range(8):
    cw, ch, _ = cpu.level_dims(l)
    gp, cp = gpu.read_level(l), cpu.level_padded(l)[:, :cw + 38]
    gb, cb = gpu.debug_read(0, l), cpu.level_blurred(l)
    gc, cc = gpu.debug_read(1, l), cpu.level_candidates(l)
    g2, c2 = gpu.debug_read(2, l), cpu.level_keypoints(l)
    same_c = len(gc) == len(cc) and all(np.array_equal(gc[f], cc[f]) for f in ("x", "y", "response"))
    same_k = len(g2) == len(c2) and all(np.array_equal(g2[f], c2[f]) for f in ("x", "y", "response"))
    same_a = len(g2) == len(c2) and np.array_equal(g2["angle"].view(np.uint32), c2["angle"].view(np.uint32))
    print("L%d pyr=%s (%d diff) blur=%s cand=%s (%d/%d) qt=%s (%d/%d) angle=%s" % (
        l, np.array_equal(gp, cp), int((gp != cp).sum()), cb is None or np.array_equal(gb, cb), same_c, len(gc), len(cc),
        same_k, len(g2), len(c2), same_a))
    if not same_a and len(g2) == len(c2):
        bad = np.nonzero(g2["angle"] != c2["angle"])[0][:5]
        print("   angle diffs", [(float(g2["angle"][i]), float(c2["angle"][i])) for i in bad])
print("final kps equal:", len(gk) == len(ck) and all(np.array_equal(gk[f].view(np.uint32), ck[f].view(np.uint32)) for f in gk.dtype.names))
print("desc equal:", gd.shape == cd.shape and np.array_equal(gd, cd), "differing rows:", int((gd != cd).any(axis=1).sum()) if gd.shape == cd.shape else -1)
gpu.set_profiling(True)
for _ in range(5): gpu.extract(img)
print(gpu.stage_times())
