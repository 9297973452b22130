import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from oracle import orb_oracle as oo
from ydorbslam_amd.synth import synth_ba_problem
cases = [((33, 512, 2), dict(seed=255056479, outlier_frac=0.2, mono_frac=1.0, n_fixed=1)),
         ((31, 36, 2), dict(seed=636061016, outlier_frac=0.0, mono_frac=1.0, n_fixed=1)),
         ((32, 35, 2), dict(seed=407056446, outlier_frac=0.2, mono_frac=1.0, n_fixed=1))]
for a, kw in cases:
    prob = synth_ba_problem(*a, **kw)
    r = oo.ba_solve(prob); p = y.Optimizer.local_bundle_adjust(prob)
    m = min(len(p["log"]), len(r["log"]))
    rel = np.abs(p["log"][:m, 0] - r["log"][:m, 0]) / np.maximum(np.abs(r["log"][:m, 0]), 1e-300)
    print(os.environ.get("YDORB_LIB", "default")[-24:], a, "len", len(p["log"]), len(r["log"]), "max rel chi2 diff %.2e" % rel.max(), "trials", p["log"][:m, 2].tolist() == r["log"][:m, 2].tolist())
