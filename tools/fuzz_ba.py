"""BA-only differential sweep (GPU vs oracle) printing the parameters of every disagreement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from oracle import orb_oracle as oo
from ydorbslam_amd.synth import synth_ba_problem
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n = bad = 0
while time.time() - t0 < budget:
    args = (int(rng.integers(3, 40)), int(rng.integers(30, 1500)), int(rng.integers(2, 8)))
    kw = dict(seed=int(rng.integers(0, 1 << 30)), outlier_frac=float(rng.choice([0, 0.05, 0.2])), mono_frac=float(rng.choice([0, 0.3, 1.0])), n_fixed=int(rng.integers(1, 3)))
    prob = synth_ba_problem(*args, **kw)
    r = oo.ba_solve(prob); p = y.Optimizer.local_bundle_adjust(prob)
    n += 1
    ok_out = np.array_equal(p["outlier"], r["outlier"]); ok_len = len(p["log"]) == len(r["log"])
    ok_chi = ok_len and np.allclose(p["log"][:, 0], r["log"][:, 0], rtol=1e-6)
    ok_pose = np.allclose(p["poses"].astype(np.float32), r["poses"].astype(np.float32), rtol=1e-4, atol=1e-6)
    if not (ok_out and ok_len and ok_chi and ok_pose):
        bad += 1
        print("MISMATCH", args, kw, "outlier", ok_out, "(diff %d of %d)" % (int((p["outlier"] != r["outlier"]).sum()), len(r["outlier"])), "len", ok_len, len(p["log"]), len(r["log"]), "chi", ok_chi, "pose", ok_pose, flush=True)
        m = min(len(p["log"]), len(r["log"]))
        print("   gpu   ", np.array2string(p["log"][:m, 0], precision=9), p["log"][:m, 2])
        print("   oracle", np.array2string(r["log"][:m, 0], precision=9), r["log"][:m, 2])
print("ba fuzz: %d problems, %d mismatches in %.0f s" % (n, bad, time.time() - t0))
