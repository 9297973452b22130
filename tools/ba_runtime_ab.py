"""One local-BA solve (config 5) timed under the process's HIP runtime: `torch` as argument = import torch first (the runtime bundled in
the wheel), else the image's ROCm.  GPU_MAX_HW_QUEUES comes from the environment."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "torch" in sys.argv:
    import torch
    torch.zeros(1, device="cuda")
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
p = synth_ba_problem(100, 10000, 8, seed=1)
o = y.Optimizer.default_options()
for _ in range(3):
    y.Optimizer.local_bundle_adjust(p, o)
ts = []
for _ in range(30):
    t = time.perf_counter(); r = y.Optimizer.local_bundle_adjust(p, o); ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e3
print("%-6s queues %s: ms min %.2f median %.2f max %.2f -> %.0f it/s (median)" % ("torch" if "torch" in sys.argv else "image", os.environ.get("GPU_MAX_HW_QUEUES", "default"),
      ts.min(), np.median(ts), ts.max(), r["trials"] / np.median(ts) * 1e3))
probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(64)]
y.Optimizer.local_bundle_adjust_batch(probs, o, 64)
rs = []
for _ in range(5):
    t = time.perf_counter(); res = y.Optimizer.local_bundle_adjust_batch(probs, o, 64); dt = time.perf_counter() - t
    rs.append(sum(b["trials"] for b in res) / dt)
print("       64 problems in lock step: %.0f it/s median (%.0f - %.0f)" % (np.median(rs), min(rs), max(rs)))
