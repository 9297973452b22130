import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(N)]
opt = y.Optimizer.default_options()
y.Optimizer.local_bundle_adjust_batch(probs, opt, N)
sys.stderr.write("==== second call\n")
t = time.perf_counter(); res = y.Optimizer.local_bundle_adjust_batch(probs, opt, N); t = time.perf_counter() - t
sys.stderr.write("python wall %.1f ms\n" % (t * 1e3))
