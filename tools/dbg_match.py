import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_frame
F = int(sys.argv[1]); W, H = 640, 480
dev = torch.device("cuda:0")
distinct = [synth_frame(W, H, i) for i in range(16)]
imgs = np.stack([distinct[i % 16] for i in range(F)])
ex = y.OrbExtractor(1000, max_batch=F); mt = y.OrbMatcher(0.9, True)
cap = ex.max_keypoints; sf = ex.tables()["scale"]
d_img = torch.from_numpy(imgs).to(dev)
d_kps = torch.zeros((F, cap, 7), dtype=torch.float32, device=dev); d_desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev)
d_n = torch.zeros(F, dtype=torch.int32, device=dev); d_as = torch.zeros((F - 1, cap), dtype=torch.int32, device=dev); d_c = torch.zeros(F - 1, dtype=torch.int32, device=dev)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); st = ts.cuda_stream
for step in range(3):
    ex.extract_batch_device(d_img.data_ptr(), W, H, W, W * H, F, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), st)
    mt.match_consecutive_device(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, F, W, H, 15.0, sf, d_as.data_ptr(), d_c.data_ptr(), None, st)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy(); c = d_c.cpu().numpy()
    print("step", step, "n", n.min(), n.max(), "counts", c.min(), c.max(), "first", c[:4], "last", c[-4:])
    try:
        mt.synchronize(); print("  sync ok")
    except Exception as e:
        print("  ", e)
