// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Dependency-free CPU restatement (double precision) of the arithmetic that
// Optimizer::localBundleAdjust (reference src/optimizer.cpp:138-352) drives through the vendored g2o:
//   residuals / Jacobians  thirdParty/g2o/g2o/types/sba/types_six_dof_expmap.{h,cpp} (:208-213, :269-274, :289-403)
//   SE3 algebra            thirdParty/g2o/g2o/types/slam3d/se3quat.h (:100-115, :212-257, :280-285), se3_ops.hpp
//   quadratic form, Huber  core/base_binary_edge.hpp:64-136, core/robust_kernel_impl.cpp:65-78, core/base_edge.h:94-97,143-149
//   Schur + back-subst.    core/block_solver.hpp:315-444 (setLambda/restoreDiagonal :525-565)
//   Levenberg-Marquardt    core/optimization_algorithm_levenberg.cpp:57-173
//   active sets, optimize  core/sparse_optimizer.cpp:63-116, 208-280, 366-440
//   two-stage schedule     src/optimizer.cpp:284-334 (optimize(5), chi2 / depth cull, optimize(10), final outlier list)
// on a flat problem (poses as SE3Quat t,q; points; edges), i.e. what the C ABI's YdBaProblem carries.
//
// PARITY STATUS: "parity unpinned" against g2o end-to-end.  g2o's sources need g2o/config.h, which only
// its CMake configure step generates (thirdParty/g2o/config.h.in), so under this build's rules the
// reference BA stack is unbuildable here and no g2o-generated golden vector exists.  What pins this file:
// the reference's own known-answer linear system (thirdParty/g2o/unit_test/solver/linear_solver_test.cpp +
// test_helper/sparse_system_helper.cpp, committed as tests/golden/g2o_linear_system.json), the
// numeric-vs-analytic Jacobian check of unit_test/test_helper/evaluate_jacobian.h (1e-6), SE3 identities,
// and recovery of the ground truth on the ba_demo-style synthetic problem (tests/test_oracle_ba.py).
// The reduced system is solved with a dense LL^T instead of Eigen's SimplicialLLT + AMD ordering: same
// solution up to rounding.  g2o sums H/b in edge-creation order (sparse_optimizer.cpp:497) and the reference
// creates edges in std::map<shared_ptr<KeyFrame>> (heap address) order, so its own FP64 sums are not
// reproducible run to run; BA parity is tolerance-based by nature.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

typedef double R;

struct V3 { R x, y, z; };
struct Q4 { R x, y, z, w; };
struct M3 { R m[3][3]; };

inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 scale(V3 a, R s) { return {a.x * s, a.y * s, a.z * s}; }
// Eigen quaternion * vector: v + w*uv + vec x uv, uv = 2 vec x v
inline V3 qrot(Q4 q, V3 v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = add(uv, uv);
  return add(add(v, scale(uv, q.w)), cross(qv, uv));
}
inline Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline M3 qToR(Q4 q) {  // Eigen toRotationMatrix
  const R tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const R twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y,
          tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz; r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = 1 - (txx + tyy);
  return r;
}
inline Q4 rToQ(const M3& a) {  // Eigen Quaternion(Matrix3)
  Q4 q;
  R t = a.m[0][0] + a.m[1][1] + a.m[2][2];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (a.m[2][1] - a.m[1][2]) * t;
    q.y = (a.m[0][2] - a.m[2][0]) * t;
    q.z = (a.m[1][0] - a.m[0][1]) * t;
  } else {
    int i = 0;
    if (a.m[1][1] > a.m[0][0]) i = 1;
    if (a.m[2][2] > a.m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0);
    R c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (a.m[k][j] - a.m[j][k]) * t;
    c[j] = (a.m[j][i] + a.m[i][j]) * t;
    c[k] = (a.m[k][i] + a.m[i][k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}
inline void qnormalize(Q4& q) {  // SE3Quat::normalizeRotation, se3quat.h:280-285
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const R n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
inline M3 mmul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline M3 skew(V3 v) { return M3{{{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}}}; }
inline V3 mvec(const M3& a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}

struct Pose { V3 t; Q4 q; };

// SE3Quat::exp, se3quat.h:218-257 (update = [omega, upsilon])
Pose se3Exp(const R* u) {
  V3 omega{u[0], u[1], u[2]}, ups{u[3], u[4], u[5]};
  const R theta = sqrt(omega.x * omega.x + omega.y * omega.y + omega.z * omega.z);
  const M3 Om = skew(omega), Om2 = mmul(Om, Om);
  M3 Rm, V;
  R a, b, c;
  if (theta < 0.00001) { a = 1; b = 0.5; c = 1. / 6.; }
  else { a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta); c = (theta - sin(theta)) / pow(theta, 3); }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const R I = i == j ? 1.0 : 0.0;
      if (theta < 0.00001) { Rm.m[i][j] = I + Om.m[i][j] + 0.5 * Om2.m[i][j]; V.m[i][j] = I + 0.5 * Om.m[i][j] + c * Om2.m[i][j]; }
      else { Rm.m[i][j] = I + a * Om.m[i][j] + b * Om2.m[i][j]; V.m[i][j] = I + b * Om.m[i][j] + c * Om2.m[i][j]; }
    }
  Pose p;
  p.q = rToQ(Rm);
  qnormalize(p.q);
  p.t = mvec(V, ups);
  return p;
}
// VertexSE3Expmap::oplusImpl: exp(update) * estimate, se3quat.h:100-106
Pose poseOplus(const Pose& T, const R* u) {
  Pose e = se3Exp(u), r;
  r.t = add(e.t, qrot(e.q, T.t));
  r.q = qmul(e.q, T.q);
  qnormalize(r.q);
  return r;
}

struct Cam { R fx, fy, cx, cy, bf; };

// residual: computeError of EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ
inline void residual(const Pose& T, V3 X, const R* z, bool stereo, const Cam& c, R* e, R* depth) {
  const V3 p = add(qrot(T.q, X), T.t);
  *depth = p.z;
  if (stereo) {
    const R invz = 1.0f / p.z;
    const R u = p.x * invz * c.fx + c.cx, v = p.y * invz * c.fy + c.cy;
    const float bf = (float)c.bf;  // `const float &bf` parameter of cam_project, types_six_dof_expmap.cpp:335
    e[0] = z[0] - u; e[1] = z[1] - v; e[2] = z[2] - (u - bf * invz);
  } else {
    e[0] = z[0] - (p.x / p.z * c.fx + c.cx);
    e[1] = z[1] - (p.y / p.z * c.fy + c.cy);
    e[2] = 0;
  }
}
// linearizeOplus: A = d e / d X (D x 3), B = d e / d xi (D x 6); types_six_dof_expmap.cpp:289-325, 357-403
inline void jacobians(const Pose& T, V3 X, bool stereo, const Cam& c, R A[3][3], R B[3][6]) {
  const V3 p = add(qrot(T.q, X), T.t);
  const M3 Rm = qToR(T.q);
  const R x = p.x, y = p.y, z = p.z, z2 = z * z;
  if (stereo) {
    for (int j = 0; j < 3; j++) {
      A[0][j] = -c.fx * Rm.m[0][j] / z + c.fx * x * Rm.m[2][j] / z2;
      A[1][j] = -c.fy * Rm.m[1][j] / z + c.fy * y * Rm.m[2][j] / z2;
      A[2][j] = A[0][j] - c.bf * Rm.m[2][j] / z2;
    }
  } else {
    R tmp[2][3] = {{c.fx, 0, -x / z * c.fx}, {0, c.fy, -y / z * c.fy}};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        R s = 0;
        for (int k = 0; k < 3; k++) s += (-1. / z * tmp[i][k]) * Rm.m[k][j];
        A[i][j] = s;
      }
    A[2][0] = A[2][1] = A[2][2] = 0;
  }
  B[0][0] = x * y / z2 * c.fx; B[0][1] = -(1 + (x * x / z2)) * c.fx; B[0][2] = y / z * c.fx;
  B[0][3] = -1. / z * c.fx; B[0][4] = 0; B[0][5] = x / z2 * c.fx;
  B[1][0] = (1 + y * y / z2) * c.fy; B[1][1] = -x * y / z2 * c.fy; B[1][2] = -x / z * c.fy;
  B[1][3] = 0; B[1][4] = -1. / z * c.fy; B[1][5] = y / z2 * c.fy;
  if (stereo) {
    B[2][0] = B[0][0] - c.bf * y / z2; B[2][1] = B[0][1] + c.bf * x / z2; B[2][2] = B[0][2];
    B[2][3] = B[0][3]; B[2][4] = 0; B[2][5] = B[0][5] - c.bf / z2;
  } else {
    for (int j = 0; j < 6; j++) B[2][j] = 0;
  }
}
// RobustKernelHuber::robustify, robust_kernel_impl.cpp:65-78
inline void huber(R e, R delta, R* rho0, R* rho1) {
  const R dsqr = delta * delta;
  if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
  else { const R s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

inline bool inv3(const R* d, R* o) {  // symmetric 3x3 [xx,xy,xz,yy,yz,zz] -> inverse (Eigen's cofactor inverse)
  const R a = d[0], b = d[1], c = d[2], e = d[3], f = d[4], i = d[5];
  const R c00 = e * i - f * f, c01 = c * f - b * i, c02 = b * f - c * e;
  const R det = a * c00 + b * c01 + c * c02;
  const R id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a * i - c * c) * id; o[4] = (b * c - a * f) * id; o[5] = (a * e - b * b) * id;
  return true;
}

// dense LL^T of an n x n SPD matrix (row-major, lower part used), in place; false if not positive definite
bool cholesky(std::vector<R>& a, int n) {
  for (int j = 0; j < n; j++) {
    R d = a[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
    if (!(d > 0)) return false;
    d = sqrt(d);
    a[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      R s = a[(size_t)i * n + j];
      const R* ri = &a[(size_t)i * n];
      const R* rj = &a[(size_t)j * n];
      for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
      a[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}
void cholSolve(const std::vector<R>& l, int n, const R* b, R* x) {
  for (int i = 0; i < n; i++) {
    R s = b[i];
    for (int k = 0; k < i; k++) s -= l[(size_t)i * n + k] * x[k];
    x[i] = s / l[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    R s = x[i];
    for (int k = i + 1; k < n; k++) s -= l[(size_t)k * n + i] * x[k];
    x[i] = s / l[(size_t)i * n + i];
  }
}

struct Options {
  int32_t iters1, iters2;
  double chi2Mono, chi2Stereo, deltaMono, deltaStereo;
  int32_t maxTrials, flags;   // flags: 1 = single stage (Optimizer::bundleAdjust, optimizer.cpp:7-137), 2 = no Huber kernels (_bIsRobust false)
};
struct IterLog { double chi2, lambda; int32_t trials, stage; };

struct Solver {
  int K, P, E;
  std::vector<Pose> poses;
  std::vector<uint8_t> fixed;
  std::vector<V3> pts;
  const int32_t *ePose, *ePoint;
  const double *eMeas, *eInfo;
  Cam cam;
  const volatile uint8_t* stop;
  Options opt;
  std::vector<uint8_t> level, robust;  // per edge: level 1 = excluded, robust kernel on/off
  std::vector<R> err;                  // per edge, as last computed (stale for inactive edges, like g2o's _error)
  std::vector<IterLog> log;
  int totalTrials = 0;

  bool stopped() const { return stop && *stop; }
  bool isStereo(int e) const { return eMeas[3 * e + 2] >= 0; }  // rightX < 0 -> mono edge, optimizer.cpp:239
  R chi2Of(int e) const {
    const R* r = &err[3 * e];
    return eInfo[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  }
  void computeErrors(const std::vector<int>& act) {
    for (int e : act) {
      R d;
      residual(poses[ePose[e]], pts[ePoint[e]], &eMeas[3 * e], isStereo(e), cam, &err[3 * e], &d);
    }
  }
  R robustChi2(const std::vector<int>& act) const {
    R chi = 0;
    for (int e : act) {
      const R c = chi2Of(e);
      if (robust[e]) { R r0, r1; huber(c, isStereo(e) ? opt.deltaStereo : opt.deltaMono, &r0, &r1); chi += r0; }
      else chi += c;
    }
    return chi;
  }

  // one optimize(iterations) call: sparse_optimizer.cpp:366 + optimization_algorithm_levenberg.cpp:57
  void optimize(int iterations, int stage) {
    std::vector<int> act;
    for (int e = 0; e < E; e++) if (!level[e]) act.push_back(e);
    if (act.empty()) return;
    // index mapping: free poses with an active edge first (by index), then points with an active edge
    std::vector<int> poseIdx(K, -1), ptIdx(P, -1), poseOf, ptOf;
    {
      std::vector<uint8_t> pu(K, 0), qu(P, 0);
      for (int e : act) { pu[ePose[e]] = 1; qu[ePoint[e]] = 1; }
      for (int k = 0; k < K; k++) if (pu[k] && !fixed[k]) { poseIdx[k] = (int)poseOf.size(); poseOf.push_back(k); }
      for (int p = 0; p < P; p++) if (qu[p]) { ptIdx[p] = (int)ptOf.size(); ptOf.push_back(p); }
    }
    const int nP = (int)poseOf.size(), nL = (int)ptOf.size(), sp = 6 * nP;
    // edges of each point (active), ordered by pose index like the columns of _HplCCS
    std::vector<std::vector<int>> ptEdges(nL);
    for (int e : act) ptEdges[ptIdx[ePoint[e]]].push_back(e);
    for (auto& v : ptEdges) std::sort(v.begin(), v.end(), [&](int a, int b) { return poseIdx[ePose[a]] < poseIdx[ePose[b]]; });

    std::vector<R> Hpp((size_t)sp * sp), Hll((size_t)6 * nL), Hpl((size_t)18 * E), b((size_t)sp + 3 * nL), x((size_t)sp + 3 * nL);
    std::vector<R> Hs, bs(sp), coef(sp), Dinv((size_t)6 * nL);
    R lambda = 0, ni = 2;
    for (int it = 0; it < iterations && !stopped(); it++) {
      computeErrors(act);
      R currentChi = robustChi2(act);
      // buildSystem, block_solver.hpp:463-521
      std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
      for (int e : act) {
        const bool st = isStereo(e);
        const int D = st ? 3 : 2;
        R A[3][3], B[3][6];
        jacobians(poses[ePose[e]], pts[ePoint[e]], st, cam, A, B);
        R w = eInfo[e], rho1 = 1;
        if (robust[e]) { R r0; huber(chi2Of(e), st ? opt.deltaStereo : opt.deltaMono, &r0, &rho1); }
        const R W = rho1 * w;  // robustInformation: rho'[1] * information (first order), base_edge.h:143-149
        R omr[3];
        for (int d = 0; d < 3; d++) omr[d] = -w * err[3 * e + d] * rho1;
        const int li = ptIdx[ePoint[e]], pi = poseIdx[ePose[e]];
        R* hl = &Hll[(size_t)6 * li];
        int t = 0;
        for (int r = 0; r < 3; r++)
          for (int c = r; c < 3; c++, t++) { R s = 0; for (int d = 0; d < D; d++) s += A[d][r] * W * A[d][c]; hl[t] += s; }
        for (int r = 0; r < 3; r++) { R s = 0; for (int d = 0; d < D; d++) s += A[d][r] * omr[d]; b[sp + 3 * li + r] += s; }
        if (pi >= 0) {
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) { R s = 0; for (int d = 0; d < D; d++) s += B[d][r] * W * B[d][c]; Hpp[(size_t)(6 * pi + r) * sp + 6 * pi + c] += s; }
          for (int r = 0; r < 6; r++) { R s = 0; for (int d = 0; d < D; d++) s += B[d][r] * omr[d]; b[6 * pi + r] += s; }
          R* hpl = &Hpl[(size_t)18 * e];  // 6x3 = B^T W A
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 3; c++) { R s = 0; for (int d = 0; d < D; d++) s += B[d][r] * W * A[d][c]; hpl[r * 3 + c] = s; }
        }
      }
      if (it == 0) {  // computeLambdaInit, levenberg.cpp:150-164
        R mx = 0;
        for (int i = 0; i < sp; i++) mx = std::max(fabs(Hpp[(size_t)i * sp + i]), mx);
        for (int l = 0; l < nL; l++) { mx = std::max(fabs(Hll[6 * l]), mx); mx = std::max(fabs(Hll[6 * l + 3]), mx); mx = std::max(fabs(Hll[6 * l + 5]), mx); }
        lambda = 1e-5 * mx;
        ni = 2;
      }
      R rho = 0;
      int qmax = 0;
      do {
        const std::vector<Pose> posesBak = poses;  // push()
        const std::vector<V3> ptsBak = pts;
        // Schur complement with lambda on both diagonals, block_solver.hpp:334-400
        Hs = Hpp;
        for (int i = 0; i < sp; i++) Hs[(size_t)i * sp + i] += lambda;
        std::fill(coef.begin(), coef.end(), 0.0);
        for (int l = 0; l < nL; l++) {
          R d[6];
          for (int t = 0; t < 6; t++) d[t] = Hll[6 * l + t];
          d[0] += lambda; d[3] += lambda; d[5] += lambda;
          R* di = &Dinv[6 * l];
          inv3(d, di);
          const R bl[3] = {b[sp + 3 * l], b[sp + 3 * l + 1], b[sp + 3 * l + 2]};
          const R db[3] = {di[0] * bl[0] + di[1] * bl[1] + di[2] * bl[2], di[1] * bl[0] + di[3] * bl[1] + di[4] * bl[2],
                           di[2] * bl[0] + di[4] * bl[1] + di[5] * bl[2]};
          const std::vector<int>& es = ptEdges[l];
          for (size_t a = 0; a < es.size(); a++) {
            const int i1 = poseIdx[ePose[es[a]]];
            if (i1 < 0) continue;
            const R* Bi = &Hpl[(size_t)18 * es[a]];
            R BD[6][3];
            for (int r = 0; r < 6; r++) {
              BD[r][0] = Bi[r * 3] * di[0] + Bi[r * 3 + 1] * di[1] + Bi[r * 3 + 2] * di[2];
              BD[r][1] = Bi[r * 3] * di[1] + Bi[r * 3 + 1] * di[3] + Bi[r * 3 + 2] * di[4];
              BD[r][2] = Bi[r * 3] * di[2] + Bi[r * 3 + 1] * di[4] + Bi[r * 3 + 2] * di[5];
              coef[6 * i1 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
            }
            for (size_t c = a; c < es.size(); c++) {
              const int i2 = poseIdx[ePose[es[c]]];
              if (i2 < 0) continue;
              const R* Bj = &Hpl[(size_t)18 * es[c]];
              for (int r = 0; r < 6; r++)
                for (int q = 0; q < 6; q++) {
                  const R v = BD[r][0] * Bj[q * 3] + BD[r][1] * Bj[q * 3 + 1] + BD[r][2] * Bj[q * 3 + 2];
                  Hs[(size_t)(6 * i1 + r) * sp + 6 * i2 + q] -= v;                      // upper block (i1 <= i2)
                  if (i1 != i2) Hs[(size_t)(6 * i2 + q) * sp + 6 * i1 + r] -= v;         // mirrored for the dense LL^T
                }
            }
          }
        }
        for (int i = 0; i < sp; i++) bs[i] = b[i] - coef[i];
        bool ok = sp == 0 ? true : cholesky(Hs, sp);
        if (ok && sp) cholSolve(Hs, sp, bs.data(), x.data());
        if (ok) {
          // landmark update, block_solver.hpp:420-444
          for (int l = 0; l < nL; l++) {
            R cl[3] = {b[sp + 3 * l], b[sp + 3 * l + 1], b[sp + 3 * l + 2]};
            for (int e : ptEdges[l]) {
              const int i1 = poseIdx[ePose[e]];
              if (i1 < 0) continue;
              const R* Bi = &Hpl[(size_t)18 * e];
              for (int c = 0; c < 3; c++)
                for (int r = 0; r < 6; r++) cl[c] -= Bi[r * 3 + c] * x[6 * i1 + r];
            }
            const R* di = &Dinv[6 * l];
            x[sp + 3 * l] = di[0] * cl[0] + di[1] * cl[1] + di[2] * cl[2];
            x[sp + 3 * l + 1] = di[1] * cl[0] + di[3] * cl[1] + di[4] * cl[2];
            x[sp + 3 * l + 2] = di[2] * cl[0] + di[4] * cl[1] + di[5] * cl[2];
          }
        }
        // update(), sparse_optimizer.cpp:433 (g2o applies _x even after a failed solve; x then holds the previous step)
        for (int i = 0; i < nP; i++) poses[poseOf[i]] = poseOplus(poses[poseOf[i]], &x[6 * i]);
        for (int l = 0; l < nL; l++) { V3& X = pts[ptOf[l]]; X.x += x[sp + 3 * l]; X.y += x[sp + 3 * l + 1]; X.z += x[sp + 3 * l + 2]; }
        computeErrors(act);
        R tempChi = robustChi2(act);
        if (!ok) tempChi = std::numeric_limits<R>::max();
        rho = currentChi - tempChi;
        R sc = 0;  // computeScale, levenberg.cpp:166-173
        for (size_t j = 0; j < x.size(); j++) sc += x[j] * (lambda * x[j] + b[j]);
        sc += 1e-3;
        rho /= sc;
        if (rho > 0 && std::isfinite(tempChi)) {
          R alpha = 1. - pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          poses = posesBak;  // pop()
          pts = ptsBak;
          if (!std::isfinite(lambda)) { qmax++; totalTrials++; break; }
        }
        qmax++;
        totalTrials++;
      } while (rho < 0 && qmax < opt.maxTrials && !stopped());
      log.push_back(IterLog{currentChi, lambda, qmax, stage});
      if (qmax == opt.maxTrials || rho == 0 || !std::isfinite(lambda)) break;  // SolverResult::Terminate
    }
  }
};

}  // namespace

extern "C" {

// poses: [K][7] = tx,ty,tz,qx,qy,qz,qw (in/out).  points: [P][3] (in/out).  edge_meas: [E][3] (u,v,ur; ur<0 => mono).
// outlier: [E] out (1 = the reference would erase the observation, optimizer.cpp:316-334).
// log: [max_log][4] doubles (chi2, lambda, trials, stage).  Returns the number of LM trials (linearise+solve+update each).
int yo_ba_solve(int K, int P, int E, double* poses, const uint8_t* poseFixed, double* points, const int32_t* ePose, const int32_t* ePoint,
                const double* eMeas, const double* eInfo, const double* camera5, const volatile uint8_t* stop, const void* options,
                uint8_t* outlier, double* log, int maxLog, int* nLog) {
  Solver S;
  S.K = K; S.P = P; S.E = E;
  S.poses.resize(K); S.fixed.assign(poseFixed, poseFixed + K); S.pts.resize(P);
  for (int k = 0; k < K; k++) {
    const double* p = poses + 7 * k;
    S.poses[k].t = {p[0], p[1], p[2]};
    S.poses[k].q = {p[3], p[4], p[5], p[6]};
    qnormalize(S.poses[k].q);
  }
  for (int p = 0; p < P; p++) S.pts[p] = {points[3 * p], points[3 * p + 1], points[3 * p + 2]};
  S.ePose = ePose; S.ePoint = ePoint; S.eMeas = eMeas; S.eInfo = eInfo;
  S.cam = {camera5[0], camera5[1], camera5[2], camera5[3], camera5[4]};
  S.stop = stop;
  S.opt = *(const Options*)options;
  S.level.assign(E, 0); S.robust.assign(E, (S.opt.flags & 2) ? 0 : 1); S.err.assign((size_t)3 * E, 0.0);
  if (nLog) *nLog = 0;
  if (S.stopped()) return 0;  // optimizer.cpp:284-286
  S.optimize(S.opt.iters1, 1);
  auto depthPositive = [&](int e) { return add(qrot(S.poses[ePose[e]].q, S.pts[ePoint[e]]), S.poses[ePose[e]].t).z > 0.0; };
  if (!(S.opt.flags & 1) && !S.stopped()) {  // optimizer.cpp:290-314 (bundleAdjust, :7-137, has no cull and no second stage)
    for (int e = 0; e < E; e++) {
      const double th = S.isStereo(e) ? S.opt.chi2Stereo : S.opt.chi2Mono;
      if (S.chi2Of(e) > th || !depthPositive(e)) S.level[e] = 1;
      S.robust[e] = 0;
    }
    S.optimize(S.opt.iters2, 2);
  }
  for (int e = 0; e < E; e++) {
    const double th = S.isStereo(e) ? S.opt.chi2Stereo : S.opt.chi2Mono;
    outlier[e] = (S.chi2Of(e) > th || !depthPositive(e)) ? 1 : 0;
  }
  for (int k = 0; k < K; k++) {
    double* p = poses + 7 * k;
    p[0] = S.poses[k].t.x; p[1] = S.poses[k].t.y; p[2] = S.poses[k].t.z;
    p[3] = S.poses[k].q.x; p[4] = S.poses[k].q.y; p[5] = S.poses[k].q.z; p[6] = S.poses[k].q.w;
  }
  for (int p = 0; p < P; p++) { points[3 * p] = S.pts[p].x; points[3 * p + 1] = S.pts[p].y; points[3 * p + 2] = S.pts[p].z; }
  int n = std::min((int)S.log.size(), maxLog);
  for (int i = 0; i < n; i++) { log[4 * i] = S.log[i].chi2; log[4 * i + 1] = S.log[i].lambda; log[4 * i + 2] = S.log[i].trials; log[4 * i + 3] = S.log[i].stage; }
  if (nLog) *nLog = n;
  return S.totalTrials;
}

// primitives for the known-answer / property tests
void yo_ba_residual(const double* pose7, const double* X, const double* z, int stereo, const double* camera5, double* e3) {
  Pose T{{pose7[0], pose7[1], pose7[2]}, {pose7[3], pose7[4], pose7[5], pose7[6]}};
  Cam c{camera5[0], camera5[1], camera5[2], camera5[3], camera5[4]};
  R d;
  residual(T, {X[0], X[1], X[2]}, z, stereo != 0, c, e3, &d);
}
void yo_ba_jacobians(const double* pose7, const double* X, int stereo, const double* camera5, double* A9, double* B18) {
  Pose T{{pose7[0], pose7[1], pose7[2]}, {pose7[3], pose7[4], pose7[5], pose7[6]}};
  Cam c{camera5[0], camera5[1], camera5[2], camera5[3], camera5[4]};
  R A[3][3], B[3][6];
  jacobians(T, {X[0], X[1], X[2]}, stereo != 0, c, A, B);
  memcpy(A9, A, sizeof(A));
  memcpy(B18, B, sizeof(B));
}
void yo_ba_pose_oplus(const double* pose7, const double* upd6, double* out7) {
  Pose T{{pose7[0], pose7[1], pose7[2]}, {pose7[3], pose7[4], pose7[5], pose7[6]}};
  Pose r = poseOplus(T, upd6);
  out7[0] = r.t.x; out7[1] = r.t.y; out7[2] = r.t.z; out7[3] = r.q.x; out7[4] = r.q.y; out7[5] = r.q.z; out7[6] = r.q.w;
}
int yo_ba_chol_solve(const double* A, int n, const double* b, double* x) {
  std::vector<R> a(A, A + (size_t)n * n);
  if (!cholesky(a, n)) return 0;
  cholSolve(a, n, b, x);
  return 1;
}
void yo_ba_huber(double e, double delta, double* rho2) { huber(e, delta, rho2, rho2 + 1); }

// ---- Optimizer::optimizePose (reference src/optimizer.cpp:358-501): pose-only LM on unary edges --------------------------------
// g2o pieces: EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose (types_six_dof_expmap.h:230-320, .cpp:415-494: the
// Jacobians use invz products, unlike the binary edges), BaseUnaryEdge::constructQuadraticForm, LinearSolverDense
// (solvers/dense/linear_solver_dense.h:66-109: dense 6x6 Cholesky, failure when not positive), the same Levenberg loop.
// pose7 in/out (tx,ty,tz,qx,qy,qz,qw); Xw [E][3]; meas [E][3] (ur < 0: monocular); info [E]; outlier [E] out.
// Returns initialCorrespondenceNum - badNum (:500), or 0 without touching anything when E < 3 (:443-445).
// chi2Log: [4] robust chi2 after each episode's optimize (NaN when the episode did not run).
int yo_pose_optimize(double* pose7, int E, const double* Xw, const double* meas, const double* info, const double* camera5,
                     uint8_t* outlier, double* chi2Log, int* trialsOut) {
  if (trialsOut) *trialsOut = 0;
  for (int k = 0; k < 4; k++) if (chi2Log) chi2Log[k] = std::numeric_limits<double>::quiet_NaN();
  if (E < 3) return 0;
  const Cam cam{camera5[0], camera5[1], camera5[2], camera5[3], camera5[4]};
  const R deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);   // :381-382
  Pose init{{pose7[0], pose7[1], pose7[2]}, {pose7[3], pose7[4], pose7[5], pose7[6]}};
  qnormalize(init.q);
  Pose T = init;
  std::vector<uint8_t> level(E, 0), robust(E, 1);
  std::vector<R> err((size_t)3 * E, 0.0);
  for (int e = 0; e < E; e++) outlier[e] = 0;
  auto stereoOf = [&](int e) { return meas[3 * e + 2] >= 0; };
  auto errOf = [&](int e) { R d; residual(T, {Xw[3 * e], Xw[3 * e + 1], Xw[3 * e + 2]}, &meas[3 * e], stereoOf(e), cam, &err[3 * e], &d); };
  auto chi2Of = [&](int e) { const R* r = &err[3 * e]; return info[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); };
  int bad = 0, trials = 0;
  for (int epi = 0; epi < 4; epi++) {
    T = init;                                   // setEstimate(frame pose) at the top of every episode (:455)
    std::vector<int> act;
    for (int e = 0; e < E; e++) if (!level[e]) act.push_back(e);
    auto robustChi2 = [&]() {
      R chi = 0;
      for (int e : act) {
        const R c = chi2Of(e);
        if (robust[e]) { R r0, r1; huber(c, stereoOf(e) ? deltaStereo : deltaMono, &r0, &r1); chi += r0; } else chi += c;
      }
      return chi;
    };
    if (!act.empty()) {                          // optimize(10), :457
      R lambda = 0, ni = 2, currentChi = 0;
      for (int it = 0; it < 10; it++) {
        for (int e : act) errOf(e);
        currentChi = robustChi2();
        R H[6][6] = {{0}}, b[6] = {0};
        for (int e : act) {
          const bool st = stereoOf(e);
          const int D = st ? 3 : 2;
          const V3 p = add(qrot(T.q, V3{Xw[3 * e], Xw[3 * e + 1], Xw[3 * e + 2]}), T.t);
          const R x = p.x, y = p.y, invz = 1.0 / p.z, invz2 = invz * invz;
          R J[3][6];
          J[0][0] = x * y * invz2 * cam.fx; J[0][1] = -(1 + (x * x * invz2)) * cam.fx; J[0][2] = y * invz * cam.fx;
          J[0][3] = -invz * cam.fx; J[0][4] = 0; J[0][5] = x * invz2 * cam.fx;
          J[1][0] = (1 + y * y * invz2) * cam.fy; J[1][1] = -x * y * invz2 * cam.fy; J[1][2] = -x * invz * cam.fy;
          J[1][3] = 0; J[1][4] = -invz * cam.fy; J[1][5] = y * invz2 * cam.fy;
          if (st) {
            J[2][0] = J[0][0] - cam.bf * y * invz2; J[2][1] = J[0][1] + cam.bf * x * invz2; J[2][2] = J[0][2];
            J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - cam.bf * invz2;
          }
          R rho1 = 1;
          if (robust[e]) { R r0; huber(chi2Of(e), st ? deltaStereo : deltaMono, &r0, &rho1); }
          const R w = info[e], W = rho1 * w;
          for (int r = 0; r < 6; r++) {
            for (int c = 0; c < 6; c++) { R s2 = 0; for (int d = 0; d < D; d++) s2 += J[d][r] * W * J[d][c]; H[r][c] += s2; }
            R s1 = 0;
            for (int d = 0; d < D; d++) s1 += J[d][r] * (-w * err[3 * e + d] * rho1);
            b[r] += s1;
          }
        }
        if (it == 0) { R mx = 0; for (int i = 0; i < 6; i++) mx = std::max(fabs(H[i][i]), mx); lambda = 1e-5 * mx; ni = 2; }
        R rho = 0, x6[6] = {0, 0, 0, 0, 0, 0};
        int qmax = 0;
        do {
          const Pose bak = T;
          std::vector<R> A(36);
          for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) A[r * 6 + c] = H[r][c] + (r == c ? lambda : 0.0);
          const bool ok = cholesky(A, 6);
          if (ok) cholSolve(A, 6, b, x6);
          T = poseOplus(T, x6);
          for (int e : act) errOf(e);
          R tempChi = robustChi2();
          if (!ok) tempChi = std::numeric_limits<R>::max();
          rho = currentChi - tempChi;
          R sc = 1e-3;
          for (int j = 0; j < 6; j++) sc += x6[j] * (lambda * x6[j] + b[j]);
          rho /= sc;
          if (rho > 0 && std::isfinite(tempChi)) {
            R alpha = 1. - pow((2 * rho - 1), 3);
            alpha = std::min(alpha, 2. / 3.);
            lambda *= std::max(1. / 3., alpha);
            ni = 2;
            currentChi = tempChi;
          } else {
            lambda *= ni; ni *= 2;
            T = bak;
            if (!std::isfinite(lambda)) { qmax++; trials++; break; }
          }
          qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;
      }
      if (chi2Log) chi2Log[epi] = currentChi;
    }
    bad = 0;                                     // classification, :458-493
    for (int e = 0; e < E; e++) {
      if (outlier[e]) errOf(e);                  // inactive edges carry a stale error: refreshed with the episode's final pose
      const float c2 = (float)chi2Of(e);
      const float th = stereoOf(e) ? 7.815f : 5.991f;
      if (c2 > th) { outlier[e] = 1; level[e] = 1; bad++; } else { outlier[e] = 0; level[e] = 0; }
      if (epi == 2) robust[e] = 0;
    }
    if (E < 10) break;                           // optimizer.edges().size() < 10, :494
  }
  pose7[0] = T.t.x; pose7[1] = T.t.y; pose7[2] = T.t.z; pose7[3] = T.q.x; pose7[4] = T.q.y; pose7[5] = T.q.z; pose7[6] = T.q.w;
  if (trialsOut) *trialsOut = trials;
  return E - bad;
}
}
