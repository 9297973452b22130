#!/usr/bin/env python3
"""Extract the known-answer linear system of the reference's own solver test into a JSON fixture.

Source (data only, read in the build container): /root/reference/thirdParty/g2o/unit_test/solver/sparse_system_helper.cpp
  sparseMatrixString()  -> 12x12 blocks of 3x3 ("BLOCK : r c" + 9 numbers), upper block triangle
  createTestVectorB()   -> b (36)
  createTestVectorX()   -> x (36), the solution linear_solver_test.cpp:69-83 expects with isApprox(1e-6)
LinearSolverEigen factorises the UPPER triangle (linear_solver_eigen.h), so A = triu(M) + triu(M,1)^T.
"""
import json, re, pathlib
import numpy as np

src = pathlib.Path("/root/reference/thirdParty/g2o/unit_test/solver/sparse_system_helper.cpp").read_text()
body = src[src.index("sparseMatrixString()"):src.index("denseInverseMatrixString()")]
lines = re.findall(r'aux << "([^"]*)"', body)
M = np.zeros((36, 36))
i = 0
while i < len(lines):
    if lines[i].startswith("BLOCK"):
        r, c = (int(v) for v in lines[i].split(":")[1].split())
        blk = np.array([[float(v) for v in lines[i + 1 + k].split()] for k in range(3)])
        M[3 * r:3 * r + 3, 3 * c:3 * c + 3] = blk
        i += 4
    else:
        i += 1
A = np.triu(M) + np.triu(M, 1).T
def vec(fn):
    seg = src[src.index(fn):]
    seg = seg[:seg.index("return result")]
    return [float(v) for v in re.findall(r"result\(idx\+\+\) = ([-0-9.e+]+);", seg)]
b, x = vec("createTestVectorB()"), vec("createTestVectorX()")
assert len(b) == 36 and len(x) == 36
assert np.allclose(np.linalg.solve(A, b), x, rtol=1e-5)
out = pathlib.Path(__file__).with_name("g2o_linear_system.json")
out.write_text(json.dumps({"source": "thirdParty/g2o/unit_test/solver/sparse_system_helper.cpp", "A": A.tolist(), "b": b, "x": x}))
print("wrote", out, "cond", np.linalg.cond(A))
