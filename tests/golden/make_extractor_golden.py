#!/usr/bin/env python3
"""Golden vectors of the extractor oracle on a REAL 640x480 gray frame and on seeded synthetic frames.

Input: thirdParty/DBow3/utils/images/image{0..3}.png of the reference (data files its DBoW3 tests use; read with Pillow).
No reference binary exists for the extractor (OpenCV is absent), so these vectors pin the ORACLE against
regressions — "OpenCV-version parity unpinned" — and give the GPU parity tests a fixed real-image case.
"""
import hashlib, json, pathlib, sys
import numpy as np
from PIL import Image
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.orb_oracle import OrbExtractorOracle
from ydorbslam_amd.synth import synth_frame

for i in range(4):
    img = np.array(Image.open("/root/reference/thirdParty/DBow3/utils/images/image%d.png" % i).convert("L"))
    assert img.shape == (480, 640)
    k, d = OrbExtractorOracle(1000, 1.2, 8, 20, 7).extract(img)
    np.savez_compressed(pathlib.Path(__file__).with_name("dbow3_image%d_orb.npz" % i), image=img, keypoints=k, descriptors=d)
    print("image%d: %d keypoints" % (i, len(k)))
# the reference's own test images (test/data: other sizes than 640x480; gray conversion by Pillow, the pixels travel in the fixture)
for name, path, nf in (("test_img1", "/root/reference/test/data/img1.png", 500), ("test_angles", "/root/reference/test/data/same-picture-different-angles.jpg", 1000)):
    img = np.array(Image.open(path).convert("L"))
    k, d = OrbExtractorOracle(nf, 1.2, 8, 20, 7).extract(img)
    np.savez_compressed(pathlib.Path(__file__).with_name("ref_%s_orb.npz" % name), image=img, keypoints=k, descriptors=d, n_features=nf)
    print("%s %s: %d keypoints" % (name, img.shape, len(k)))
h = {}
for (w, hh, nf, idx) in [(640, 480, 1000, 0), (752, 480, 1000, 1), (1241, 376, 2000, 2), (321, 243, 500, 3)]:
    kk, dd = OrbExtractorOracle(nf, 1.2, 8, 20, 7).extract(synth_frame(w, hh, idx))
    h["%dx%d_n%d_i%d" % (w, hh, nf, idx)] = {"n": int(len(kk)), "kps_sha256": hashlib.sha256(kk.tobytes()).hexdigest(),
                                             "desc_sha256": hashlib.sha256(dd.tobytes()).hexdigest()}
pathlib.Path(__file__).with_name("synthetic_orb_hashes.json").write_text(json.dumps(h, indent=1))
print(len(k), h)
