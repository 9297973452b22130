"""Device time of ydorb_stereo_matches (reference replay form) alone: config-3-size pairs (1241x376, 2000 features), 128 pairs per call, and one pair per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ydorbslam_amd as y
from ydorbslam_amd.synth import stream_plan, stream_render
w, h, nf, n_pairs = 1241, 376, 2000, 128
dev = torch.device("cuda", 0)
pl = stream_plan(w, h, n_pairs, seed=7, segment=32)
L_, R_ = stream_render(pl, range(n_pairs), stereo=True)
diL, diR = torch.from_numpy(L_).to(dev), torch.from_numpy(R_).to(dev)
mk = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
xL = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=n_pairs, single_stream=True); xR = y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=n_pairs, single_stream=True)
cap = xL.max_keypoints
kL, kR, dL, dR = mk(n_pairs, cap, 7), mk(n_pairs, cap, 7), mk(n_pairs, cap, 32, dt=torch.uint8), mk(n_pairs, cap, 32, dt=torch.uint8)
nL, nR, rx, dp, kept = mk(n_pairs, dt=torch.int32), mk(n_pairs, dt=torch.int32), mk(n_pairs, cap), mk(n_pairs, cap), mk(n_pairs, dt=torch.int32)
sm = y.OrbMatcher()
st = torch.cuda.Stream(device=dev)
xL.extract_batch_device(diL.data_ptr(), w, h, w, w * h, n_pairs, kL.data_ptr(), dL.data_ptr(), cap, nL.data_ptr(), st.cuda_stream)
xR.extract_batch_device(diR.data_ptr(), w, h, w, w * h, n_pairs, kR.data_ptr(), dR.data_ptr(), cap, nR.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
for np_ in (n_pairs, 1):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t = time.perf_counter()
        sm.stereo_matches_device(xL, xR, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(), cap, kR.data_ptr(), dR.data_ptr(), nR.data_ptr(), cap, np_, 40.0, 0.1,
                                 rx.data_ptr(), dp.data_ptr(), kept.data_ptr(), None, False, (0, 1), (0, 1), st.cuda_stream)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%s: %3d pairs per call: %.3f ms median (min %.3f); kept per pair %.1f; keypoints per left image %.0f"
          % (os.environ.get("YDORB_LIB", "default"), np_, np.median(ts[2:]) * 1e3, min(ts) * 1e3, kept[:np_].float().mean().item(), nL.float().mean().item()))
