import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
probs = [synth_ba_problem(8 + 3 * i, 200 + 150 * i, 4 + i % 3, seed=20 + i, outlier_frac=0.05) for i in range(3)]
single = [y.Optimizer.local_bundle_adjust(p) for p in probs]
for nb in (1, 3):
    batch = y.Optimizer.local_bundle_adjust_batch(probs[:nb])
    for i, (a, b) in enumerate(zip(single, batch)):
        print("B=%d prob %d trials %d/%d iters %d/%d logs equal %s poses maxdiff %.3e points maxdiff %.3e outlier equal %s" % (
            nb, i, a["trials"], b["trials"], a["iterations"], b["iterations"], a["log"].shape == b["log"].shape and np.array_equal(a["log"], b["log"]),
            np.abs(a["poses"] - b["poses"]).max(), np.abs(a["points"] - b["points"]).max(), np.array_equal(a["outlier"], b["outlier"])))
        if a["log"].shape == b["log"].shape:
            d = np.abs(a["log"] - b["log"]).max(axis=1)
            print("   log row diffs", d[:16])
        else:
            print(a["log"][:, [0, 2, 3]].tolist()); print(b["log"][:, [0, 2, 3]].tolist())
