import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
rng = np.random.default_rng(0)
outs = []
for n in (32, 96, 608, 1024):
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    b = rng.standard_normal(n)
    x, ok = y.Optimizer.dense_solve(A, b)
    outs.append(x)
    print(n, ok, float(np.abs(A @ x - b).max()))
A = -np.eye(64); x, ok = y.Optimizer.dense_solve(A, np.ones(64)); print("indefinite ok flag:", ok)
np.save(sys.argv[1], np.concatenate(outs))
p = synth_ba_problem(100, 10000, 8, seed=1)
y.Optimizer.local_bundle_adjust(p)
t = time.perf_counter()
for _ in range(3):
    r = y.Optimizer.local_bundle_adjust(p)
print("ms per solve", (time.perf_counter() - t) / 3 * 1e3, "trials", r["trials"], "chi2", r["log"][-1, 0], r["ms"])
np.save(sys.argv[1] + ".poses", r["poses"])
