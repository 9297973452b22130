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
