import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_ba_problem
for NT in (1, 2, 4, 6, 8):
    probs = [synth_ba_problem(100, 10000, 8, seed=1) for _ in range(NT)]
    for p in probs[:1]:
        y.Optimizer.local_bundle_adjust(p)
    res = [0] * NT
    def work(i):
        n = 0
        for _ in range(4):
            n += y.Optimizer.local_bundle_adjust(probs[i])["trials"]
        res[i] = n
    th = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
    t = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    t = time.perf_counter() - t
    print(NT, "threads:", round(sum(res) / t), "it/s")
