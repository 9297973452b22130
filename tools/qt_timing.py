"""Phase timestamps (100 MHz wall clock) of k_quadtree_flat for frame 0 of a batch — needs a -DQT_FLAT_TIMING build (YDORB_LIB)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_frame
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
imgs = np.stack([synth_frame(640, 480, i % 16) for i in range(F)])
ex = y.OrbExtractor(1000, max_batch=F)
for _ in range(3):
    ex.extract_batch(imgs)
for l in range(8):
    out = np.zeros(16, np.int64); w = C.c_size_t(0)
    y._lib.check(ex._L.ydorb_extractor_debug_read(ex._h, 4, 0, l, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(w)))
    n = len(ex.debug_read(1, l, 0))
    t = (out - out[0]) / 100.0
    if n > 1024:
        print("L%d n=%d  copy %.1f | keys+hist %.1f | pyramid %.1f | reduce+P %.1f | rank keys %.1f | sort %.1f | scatter %.1f | starts+keys %.1f | best %.1f  = %.1f us" % (
            l, n, t[1], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[5]-t[4], t[6]-t[5], t[7]-t[6], t[8]-t[7], t[9]-t[8], t[9]))
    else:
        print("L%d n=%d  copy %.1f | keys+pairs %.1f | P+R %.1f | rank pairs %.1f | scatter %.1f | starts+keys %.1f | best %.1f = %.1f us" % (
            l, n, t[1], t[10]-t[1], t[11]-t[10], t[12]-t[11], t[7]-t[12], t[8]-t[7], t[9]-t[8], t[9]))
