import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(N)]
opt = y.Optimizer.default_options()
y.Optimizer.local_bundle_adjust(probs[0], opt)
t = time.perf_counter(); r = y.Optimizer.local_bundle_adjust(probs[0], opt); t = time.perf_counter() - t
print("single: %.2f ms, %.0f it/s" % (t * 1e3, r["trials"] / t))
for g in (1, 4, 8, 16, 32, 64):
    y.Optimizer.local_bundle_adjust_batch(probs[:g], opt, g)
    t = time.perf_counter(); res = y.Optimizer.local_bundle_adjust_batch(probs, opt, g); t = time.perf_counter() - t
    print("group %2d: %7.1f ms for %d problems -> %.0f it/s" % (g, t * 1e3, N, sum(b["trials"] for b in res) / t), flush=True)
