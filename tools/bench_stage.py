"""Print per-stage device times of the extract + match pipeline (HIP events) for the library named by YDORB_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ydorbslam_amd as y
from ydorbslam_amd.synth import synth_frame
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
imgs = np.stack([synth_frame(640, 480, i % 32) for i in range(F)])
ex = y.OrbExtractor(1000, max_batch=F, single_stream=bool(int(os.environ.get("YDORB_STAGE_SINGLE", "1"))))   # single-stream handle: what bench.py's lanes use
ex.extract_batch(imgs)
ex.set_profiling(True)
for _ in range(int(os.environ.get("YDORB_STAGE_REPS", "5"))):   # a few hundred: the sustained-clock figure
    ex.extract_batch(imgs)
print(os.environ.get("YDORB_LIB", "default"), {k: round(v, 4) for k, v in ex.stage_times().items()})
