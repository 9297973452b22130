"""Where does the host block while it enqueues a stereo step?  Times every library call of the lane pipeline of bench.py's config 3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ydorbslam_amd as y
from ydorbslam_amd.synth import stream_plan, stream_render
w, h, nf, n_pairs, NSETS = 1241, 376, 2000, 128, 4
dev = torch.device("cuda", 0)
pl = stream_plan(w, h, n_pairs, seed=7, segment=32)
L_, R_ = stream_render(pl, range(n_pairs), stereo=True)
diL, diR = torch.from_numpy(L_).to(dev), torch.from_numpy(R_).to(dev)
mk = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
prs = np.array([(i, i + 1) for i in range(n_pairs - 1)], np.int32)
daf = torch.from_numpy(np.ascontiguousarray(pl["predicted"], np.float32)).to(dev)
sets = []
for _ in range(NSETS):
    S = dict(xL=y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=n_pairs, single_stream=True), xR=y.OrbExtractor(nf, 1.2, 8, 20, 7, max_batch=n_pairs, single_stream=True))
    cap = S["xL"].max_keypoints
    S.update(kL=mk(n_pairs, cap, 7), kR=mk(n_pairs, cap, 7), dL=mk(n_pairs, cap, 32, dt=torch.uint8), dR=mk(n_pairs, cap, 32, dt=torch.uint8),
             nL=mk(n_pairs, dt=torch.int32), nR=mk(n_pairs, dt=torch.int32), rx=mk(n_pairs, cap), dp=mk(n_pairs, cap), kept=mk(n_pairs, dt=torch.int32),
             asg=mk(n_pairs - 1, cap, dt=torch.int32), cnt=mk(n_pairs - 1, dt=torch.int32), sm=y.OrbMatcher(), mm=y.OrbMatcher(0.9, True), st=torch.cuda.Stream(device=dev))
    sets.append(S)
ssf = sets[0]["xL"].tables()["scale"]
T = {"xL": [], "xR": [], "stereo": [], "match": []}
def one(k):
    S = sets[k % NSETS]; st = S["st"].cuda_stream
    t0 = time.perf_counter()
    S["xL"].extract_batch_device(diL.data_ptr(), w, h, w, w * h, n_pairs, S["kL"].data_ptr(), S["dL"].data_ptr(), cap, S["nL"].data_ptr(), st)
    t1 = time.perf_counter()
    S["xR"].extract_batch_device(diR.data_ptr(), w, h, w, w * h, n_pairs, S["kR"].data_ptr(), S["dR"].data_ptr(), cap, S["nR"].data_ptr(), st)
    t2 = time.perf_counter()
    S["sm"].stereo_matches_device(S["xL"], S["xR"], S["kL"].data_ptr(), S["dL"].data_ptr(), S["nL"].data_ptr(), cap, S["kR"].data_ptr(), S["dR"].data_ptr(),
                                  S["nR"].data_ptr(), cap, n_pairs, 40.0, 0.1, S["rx"].data_ptr(), S["dp"].data_ptr(), S["kept"].data_ptr(), None, False, (0, 1), (0, 1), st)
    t3 = time.perf_counter()
    fs = (S["kL"].data_ptr(), S["dL"].data_ptr(), S["nL"].data_ptr(), n_pairs, cap)
    S["mm"].match_pairs_device(fs, fs, prs, w, h, 15.0, ssf, S["asg"].data_ptr(), S["cnt"].data_ptr(), daf.data_ptr(), st)
    t4 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
for k in range(NSETS):
    one(k)
torch.cuda.synchronize()
t = time.perf_counter()
for k in range(16):
    r = one(NSETS + k)
    print("step %2d  xL %.3f  xR %.3f  stereo %.3f  match %.3f ms   (t = %.2f ms)" % (k, *(1e3 * v for v in r), 1e3 * (time.perf_counter() - t)))
torch.cuda.synchronize()
print("total %.2f ms per step" % (1e3 * (time.perf_counter() - t) / 16))
