import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
p = synth_ba_problem(100, 10000, 8, seed=1)
y.Optimizer.local_bundle_adjust(p)
y.Optimizer.local_bundle_adjust(p)

t = time.perf_counter()
r = y.Optimizer.local_bundle_adjust(p)
print("wall ms", (time.perf_counter() - t) * 1e3, "trials", len(r["log"]))
