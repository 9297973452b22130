import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from oracle import orb_oracle as oo
from ydorbslam_amd.synth import synth_pose_problem
for i, (n, o, m) in enumerate([(400, 0.1, 0.3), (1000, 0.05, 0.0), (150, 0.3, 1.0), (9, 0.0, 0.5)]):
    p = synth_pose_problem(n, seed=50 + i, outlier_frac=o, mono_frac=m)
    g = y.Optimizer.optimize_poses([p])[0]; r = oo.pose_optimize(p)
    print(n, "trials", g["trials"], r["trials"], "inl", g["inliers"], r["inliers"], "outlier diff", int((g["outlier"] != r["outlier"]).sum()))
    print("  chi2 gpu", g["chi2"]); print("  chi2 cpu", r["chi2"]); print("  pose diff", np.abs(g["pose"] - r["pose"]).max())
