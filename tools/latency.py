"""Single-frame latency of the host-pointer entry points (what the drop-in adapters call once per frame / per search)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_frame
img = synth_frame(640, 480, 0); img2 = synth_frame(640, 480, 1)
ex = y.OrbExtractor(1000)
for _ in range(20): ex.extract(img)
N = 200
t = time.perf_counter()
for i in range(N): k, d = ex.extract(img if i & 1 else img2)
dt = (time.perf_counter() - t) / N
print("ydorb_extract 640x480 N=1000: %.3f ms per frame (H2D + kernels + D2H, host clock)" % (dt * 1e3))
ex.set_profiling(True)
for i in range(50): ex.extract(img)
print({k: round(v, 4) for k, v in ex.stage_times().items()})
