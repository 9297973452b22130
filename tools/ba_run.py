import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
p = synth_ba_problem(100, 10000, 8, seed=1)
o = y.Optimizer.default_options(phase_times=len(sys.argv) > 1)
y.Optimizer.local_bundle_adjust(p, o)
t = time.time(); r = y.Optimizer.local_bundle_adjust(p, o); dt = time.time() - t
print("trials", r["trials"], "ms", round(dt * 1e3, 2), "it/s", round(r["trials"] / dt, 1), {k: round(v, 2) for k, v in r["ms"].items()})
