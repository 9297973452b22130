"""The C ABI driven by a plain C++ host (tests/cpp_host/host_roundtrip.cpp: no Python, no OpenCV, no torch) — the situation of the
reference's own translation units once they forward to the library.  Its output must equal what the ctypes mirror returns."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp_host", "host_roundtrip.cpp")


def _build(tmp):
    exe = os.path.join(tmp, "host_roundtrip")
    lib_dir = os.path.join(ROOT, "ydorbslam_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + lib_dir, "-l:libydorb.so",
                           "-Wl,-rpath," + lib_dir])
    return exe


def test_cpp_host_builds_and_links(tmp_path):
    """CPU side: the header is valid C++ for a plain host compiler and every symbol the program uses resolves against libydorb.so."""
    import ydorbslam_amd as y
    if not os.path.exists(y.library_path()):
        y.build_library()
    assert os.path.exists(_build(str(tmp_path)))


@pytest.mark.gpu
def test_cpp_host_equals_ctypes_mirror(tmp_path):
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_frame
    exe = _build(str(tmp_path))
    w, h = 640, 480
    img = synth_frame(w, h, 77)
    raw, out = str(tmp_path / "in.raw"), str(tmp_path / "out.bin")
    img.tofile(raw)
    env = dict(os.environ)
    torch_lib = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")   # the same HIP runtime the Python path binds to
    env["LD_LIBRARY_PATH"] = torch_lib + ":" + env.get("LD_LIBRARY_PATH", "")
    subprocess.check_call([exe, raw, str(w), str(h), out], env=env)
    buf = open(out, "rb").read()
    na, nb, nm, cap = np.frombuffer(buf, np.int32, 4)
    off = 16
    ka = np.frombuffer(buf, y.KP_DTYPE, na, off); off += 28 * na
    da = np.frombuffer(buf, np.uint8, 32 * na, off).reshape(na, 32); off += 32 * na
    kb = np.frombuffer(buf, y.KP_DTYPE, nb, off); off += 28 * nb
    db = np.frombuffer(buf, np.uint8, 32 * nb, off).reshape(nb, 32); off += 32 * nb
    assigned = np.frombuffer(buf, np.int32, nb, off)
    # the same through the ctypes mirror
    moved = np.roll(img, (2, 3), (0, 1))
    ex = y.OrbExtractor(1000, 1.2, 8, 20, 7)
    pka, pda = ex.extract(img)
    pkb, pdb = ex.extract(moved)
    assert ka.tobytes() == pka.tobytes() and np.array_equal(da, pda) and kb.tobytes() == pkb.tobytes() and np.array_equal(db, pdb)
    sf = ex.tables()["scale"]
    q = np.zeros(len(pka), y.QUERY_DTYPE)
    q["u"], q["v"] = pka["x"], pka["y"]
    q["r"] = (np.float32(15.0) * sf[pka["octave"]]).astype(np.float32)
    q["min_level"], q["max_level"] = pka["octave"] - 1, pka["octave"] + 1
    q["angle"], q["level"], q["flags"] = pka["angle"], pka["octave"], 3
    n, a, _ = y.OrbMatcher(0.9, True).search_by_projection(1, y.FrameView(pkb, pdb, (0.0, float(w), 0.0, float(h)), None), q, pda)
    assert n == nm and np.array_equal(a, assigned) and n > 20


@pytest.mark.gpu
def test_cpp_host_local_bundle_adjust(tmp_path):
    """Optimizer::localBundleAdjust's flat problem solved from C++; the solver is bit-reproducible, so the result equals the ctypes call's."""
    import ydorbslam_amd as y
    from ydorbslam_amd.synth import synth_ba_problem
    exe = _build(str(tmp_path))
    p = synth_ba_problem(12, 600, 6, seed=9, outlier_frac=0.05)
    K, P, E = len(p["poses"]), len(p["points"]), len(p["edge_pose"])
    inp, out = str(tmp_path / "prob.bin"), str(tmp_path / "ba.bin")
    with open(inp, "wb") as f:
        f.write(np.array([K, P, E], np.int32).tobytes())
        for key, dt in (("poses", np.float64), ("fixed", np.uint8), ("points", np.float64), ("edge_pose", np.int32), ("edge_point", np.int32),
                        ("meas", np.float64), ("info", np.float64), ("camera", np.float64)):
            f.write(np.ascontiguousarray(p[key], dt).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(__import__("torch").__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    subprocess.check_call([exe, "ba", inp, out], env=env)
    buf = open(out, "rb").read()
    trials, iters = np.frombuffer(buf, np.int32, 2)
    poses = np.frombuffer(buf, np.float64, K * 7, 8).reshape(K, 7)
    points = np.frombuffer(buf, np.float64, P * 3, 8 + 56 * K).reshape(P, 3)
    outlier = np.frombuffer(buf, np.uint8, E, 8 + 56 * K + 24 * P)
    r = y.Optimizer.local_bundle_adjust(p)
    assert trials == r["trials"] and iters == r["iterations"]
    assert poses.tobytes() == r["poses"].tobytes() and points.tobytes() == r["points"].tobytes() and np.array_equal(outlier, r["outlier"])
