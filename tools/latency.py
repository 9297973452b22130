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

# one searchByProjectionInLastAndCurrentFrame-style call (mode 1): queries = the previous frame's keypoints
from ydorbslam_amd.matcher import FrameView, OrbMatcher, QUERY_DTYPE
k0, d0 = ex.extract(img); k1, d1 = ex.extract(img2)
sf = 1.2 ** np.arange(8, dtype=np.float32)
q = np.zeros(len(k0), QUERY_DTYPE)
q["u"], q["v"] = k0["x"], k0["y"]
q["r"] = (np.float32(15.0) * sf[k0["octave"]]).astype(np.float32)
q["min_level"], q["max_level"] = k0["octave"] - 1, k0["octave"] + 1
q["angle"], q["level"], q["flags"] = k0["angle"], k0["octave"], 3
m = OrbMatcher(0.9, True)
fv = FrameView(k1, d1, (0.0, 640.0, 0.0, 480.0))
for _ in range(10): m.search_by_projection(1, fv, q, d0)
t = time.perf_counter()
for _ in range(N): nm, asg, tk = m.search_by_projection(1, fv, q, d0)
dt = (time.perf_counter() - t) / N
print("ydorb_search_by_projection mode 1, %d queries x %d keypoints: %.3f ms per call (%d matches)" % (len(q), len(k1), dt * 1e3, nm))
m.set_profiling(True)
for _ in range(20): m.search_by_projection(1, fv, q, d0)
print({k: round(v, 4) for k, v in m.stage_times().items()})

# Optimizer::optimizePose: one frame, and a batch
from ydorbslam_amd.synth import synth_pose_problem
pp = [synth_pose_problem(400, seed=i) for i in range(64)]
y.Optimizer.optimize_poses(pp[:1]); y.Optimizer.optimize_poses(pp)
t = time.perf_counter()
for _ in range(50): y.Optimizer.optimize_poses(pp[:1])
d1 = (time.perf_counter() - t) / 50
t = time.perf_counter()
for _ in range(20): y.Optimizer.optimize_poses(pp)
d64 = (time.perf_counter() - t) / 20
print("ydorb_pose_optimize, 400 correspondences: %.3f ms for one frame, %.3f ms for 64 frames (%.1f us per frame)" % (d1 * 1e3, d64 * 1e3, d64 / 64 * 1e6))
from oracle import orb_oracle as oo
t = time.perf_counter()
for i in range(20): oo.pose_optimize(pp[i])
print("oracle (1 CPU thread): %.3f ms per frame" % ((time.perf_counter() - t) / 20 * 1e3))
