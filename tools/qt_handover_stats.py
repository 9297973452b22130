"""Which (frame, level) units of the bench's synthetic frames does k_qt_fast hand over to the pass kernel (k_quadtree_list), per configuration?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import stream_plan, stream_render
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for name, w, h, nf, stereo in (("config2 640x480 1000", 640, 480, 1000, False), ("config3 1241x376 2000", 1241, 376, 2000, True), ("config4 752x480 1000", 752, 480, 1000, True)):
    plan = stream_plan(w, h, F, seed=7 if stereo else 0, segment=32 if stereo else 64)   # the bench's streams
    r = stream_render(plan, list(range(F)), stereo=stereo)
    imgs = r[0]
    ex = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=F, single_stream=True)
    ex.extract_batch(imgs)
    ex.extract_batch(imgs)     # second call: the per-level ITEMS were retuned from the first
    print(name)
    for l in range(8):
        form = [ex.debug_read(3, l, f) for f in range(F)]
        n = [len(ex.debug_read(1, l, f)) for f in range(F)]
        print("  level %d: %3d of %d units handed over; candidates min %5d mean %7.1f max %5d" % (l, sum(form), F, min(n), np.mean(n), max(n)))
