"""Per-phase clocks of k_chol_step (needs a -DCHOL_TIMING build in YDORB_LIB); prints to stderr from the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
rng = np.random.default_rng(0)
n = 600
M = rng.normal(size=(n, n))
A = M @ M.T + n * np.eye(n)
b = rng.normal(size=n)
for _ in range(2):
    x, ok = y.Optimizer.dense_solve(A, b)
print("ok", ok, "residual", np.abs(A @ x - b).max())
