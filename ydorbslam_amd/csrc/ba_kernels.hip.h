// gfx950 HIP kernels of the local-BA back-end (double precision throughout, like g2o's number_t).
//
// Arithmetic restated from the g2o subset that Optimizer::localBundleAdjust executes (reference
// src/optimizer.cpp:138-352): residuals/Jacobians thirdParty/g2o/g2o/types/sba/types_six_dof_expmap.{h,cpp},
// SE3 algebra types/slam3d/se3quat.h, quadratic form core/base_binary_edge.hpp:64-136, Huber
// core/robust_kernel_impl.cpp:65-78, Schur complement + back-substitution core/block_solver.hpp:315-444.
// The structure is GPU-first instead of g2o's edge loop over a sparse block matrix:
//   * edges are sorted by (landmark, pose): one thread owns a landmark and accumulates H_ll, b_l and the
//     6x3 H_pl blocks of its observations in registers (no atomics);
//   * one workgroup owns a free pose and reduces its 6x6 H_pp and b_p over that pose's observations;
//   * the Schur complement  S = H_pp + lambda*I - sum_l W_l (H_ll + lambda*I)^-1 W_l^T  is formed directly in a
//     dense (6K)^2 buffer (K = 100 keyframes -> 2.9 MB, L2-resident): the (landmark, pose pair) items are bucketed by pose pair
//     once per solve and sorted by landmark, then one workgroup per 6x6 block sums its bucket on the FP64 matrix core - no atomics,
//     fixed summation order (a solve is bit-reproducible run to run);
//   * the reduced system is factorised by a blocked right-looking dense Cholesky (one launch per 32-column
//     panel, one workgroup per 32x32 trailing tile) and solved by a single-workgroup blocked substitution;
//   * LM control (lambda, accept/reject, chi2 cull) stays on the host exactly as
//     core/optimization_algorithm_levenberg.cpp:57-148 sequences it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace ydorb {
namespace ba {

typedef double R;
struct V3 { R x, y, z; };
struct Q4 { R x, y, z, w; };

__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 scale(V3 a, R s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {  // Eigen quaternion * vector
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = add(uv, uv);
  return add(add(v, scale(uv, q.w)), cross(qv, uv));
}
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ void qToR(Q4 q, R m[3][3]) {
  const R tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const R twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y,
          tyz = tz * q.y, tzz = tz * q.z;
  m[0][0] = 1 - (tyy + tzz); m[0][1] = txy - twz; m[0][2] = txz + twy;
  m[1][0] = txy + twz; m[1][1] = 1 - (txx + tzz); m[1][2] = tyz - twx;
  m[2][0] = txz - twy; m[2][1] = tyz + twx; m[2][2] = 1 - (txx + tyy);
}
__device__ __forceinline__ Q4 rToQ(const R a[3][3]) {  // Eigen Quaternion(Matrix3)
  Q4 q;
  R t = a[0][0] + a[1][1] + a[2][2];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (a[2][1] - a[1][2]) * t;
    q.y = (a[0][2] - a[2][0]) * t;
    q.z = (a[1][0] - a[0][1]) * t;
  } else {
    int i = 0;
    if (a[1][1] > a[0][0]) i = 1;
    if (a[2][2] > a[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(a[i][i] - a[j][j] - a[k][k] + 1.0);
    R c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (a[k][j] - a[j][k]) * t;
    c[j] = (a[j][i] + a[i][j]) * t;
    c[k] = (a[k][i] + a[i][k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}
__device__ __forceinline__ void qnormalize(Q4& q) {  // se3quat.h:280-285
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const R n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

struct Cam { R fx, fy, cx, cy, bf; };

struct EdgeSoA {          // active edges sorted by (landmark index, pose index)
  const int* pose;        // index into the pose array
  const int* pidx;        // free-pose (Hessian) index or -1 when the pose is fixed
  const int* pt;          // index into the point array
  const R* meas;          // [E][3] u, v, ur (ur < 0: monocular edge, optimizer.cpp:239)
  const R* info;          // [E] invSigma2 (information = I * invSigma2, optimizer.cpp:248,268)
  const uint8_t* robust;  // [E] Huber kernel attached
  int n;
};

__device__ __forceinline__ void load_pose(const R* poses, int k, V3& t, Q4& q) {
  const R* p = poses + 7 * k;
  t = {p[0], p[1], p[2]};
  q = {p[3], p[4], p[5], p[6]};
}

// computeError (types_six_dof_expmap.h:208-213, 269-274; cam_project .cpp:327-342)
__device__ __forceinline__ void residual(V3 t, Q4 q, V3 X, const R* z, bool stereo, const Cam& c, R e[3], R* depth) {
  const V3 p = add(qrot(q, X), t);
  *depth = p.z;
  if (stereo) {
    const R invz = 1.0f / p.z;
    const R u = p.x * invz * c.fx + c.cx, v = p.y * invz * c.fy + c.cy;
    const float bf = (float)c.bf;
    e[0] = z[0] - u; e[1] = z[1] - v; e[2] = z[2] - (u - bf * invz);
  } else {
    e[0] = z[0] - (p.x / p.z * c.fx + c.cx);
    e[1] = z[1] - (p.y / p.z * c.fy + c.cy);
    e[2] = 0;
  }
}
// linearizeOplus (types_six_dof_expmap.cpp:289-325, 357-403): A = de/dX (3x3, third row 0 for mono), B = de/dxi (3x6)
__device__ __forceinline__ void jacobians(V3 t, Q4 q, V3 X, bool stereo, const Cam& c, R A[3][3], R B[3][6]) {
  const V3 p = add(qrot(q, X), t);
  R Rm[3][3];
  qToR(q, Rm);
  const R x = p.x, y = p.y, z = p.z, z2 = z * z;
  if (stereo) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      A[0][j] = -c.fx * Rm[0][j] / z + c.fx * x * Rm[2][j] / z2;
      A[1][j] = -c.fy * Rm[1][j] / z + c.fy * y * Rm[2][j] / z2;
      A[2][j] = A[0][j] - c.bf * Rm[2][j] / z2;
    }
  } else {
    const R tmp[2][3] = {{c.fx, 0, -x / z * c.fx}, {0, c.fy, -y / z * c.fy}};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        R s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += (-1. / z * tmp[i][k]) * Rm[k][j];
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
#pragma unroll
    for (int j = 0; j < 6; j++) B[2][j] = 0;
  }
}
__device__ __forceinline__ void huber(R e, R delta, R* rho0, R* rho1) {  // robust_kernel_impl.cpp:65-78
  const R dsqr = delta * delta;
  if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
  else { const R s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

__device__ __forceinline__ R block_sum(R v, R* lds) {  // deterministic workgroup sum (blockDim.x multiple of 64, <= 1024)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) lds[wv] = v;
  __syncthreads();
  R s = 0;
  for (int i = 0; i < nw; i++) s += lds[i];
  return s;
}

// computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:63-116): err[e], per-workgroup partial chi2
__device__ __forceinline__ void b_errors(EdgeSoA Ed, const R* __restrict__ poses, const R* __restrict__ pts, Cam cam,
                                                R deltaMono, R deltaStereo, R* __restrict__ err, R* __restrict__ partial) {
  __shared__ R lds[4];
  const int e = blockIdx.x * 256 + threadIdx.x;
  R chi = 0;
  if (e < Ed.n) {
    V3 t; Q4 q;
    load_pose(poses, Ed.pose[e], t, q);
    const R* X = pts + 3 * Ed.pt[e];
    const R* z = Ed.meas + 3 * e;
    const bool st = z[2] >= 0;
    R r[3], d;
    residual(t, q, V3{X[0], X[1], X[2]}, z, st, cam, r, &d);
    if (Ed.info[e] != 0) {   // information 0 = culled after stage 1: not in the active set (no chi2, error not refreshed)
      err[3 * e] = r[0]; err[3 * e + 1] = r[1]; err[3 * e + 2] = r[2];
      const R c2 = Ed.info[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      if (Ed.robust[e]) { R r0, r1; huber(c2, st ? deltaStereo : deltaMono, &r0, &r1); chi = r0; }
      else chi = c2;
    }
  }
  const R s = block_sum(chi, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_errors(EdgeSoA Ed, const R* __restrict__ poses, const R* __restrict__ pts, Cam cam,
                                                R deltaMono, R deltaStereo, R* __restrict__ err, R* __restrict__ partial) { b_errors(Ed, poses, pts, cam, deltaMono, deltaStereo, err, partial); }

// final ordered sum of the partials: out[slot] = sum(partial[0..n))
__device__ __forceinline__ void b_sum_partials(const R* __restrict__ partial, int n, R* __restrict__ out, int slot,
                                                      const int* __restrict__ status) {
  __shared__ R lds[4];
  R v = 0;
  for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
  const R s = block_sum(v, lds);
  if (threadIdx.x == 0) {
    out[slot] = s;
    if (status) { out[6] = (R)status[0]; out[7] = (R)status[1]; }   // the factorisation status rides along in the same read-back
  }
}
__global__ __launch_bounds__(256) void k_sum_partials(const R* __restrict__ partial, int n, R* __restrict__ out, int slot,
                                                      const int* __restrict__ status) { b_sum_partials(partial, n, out, slot, status); }


// buildSystem, landmark side (constructQuadraticForm for `from` = point): one thread per active landmark.
// Hll[l] = sum A^T W A (6 unique), bl[l] = sum A^T (-rho' Omega e), Hpl[e] = B^T W A (6x3) for free poses.
__device__ __forceinline__ void b_build_points(EdgeSoA Ed, const int* __restrict__ ptStart, int nL, const R* __restrict__ poses,
                                                      const R* __restrict__ pts, Cam cam, R deltaMono, R deltaStereo,
                                                      const R* __restrict__ err, R* __restrict__ Hll, R* __restrict__ bl,
                                                      R* __restrict__ Hpl) {
  const int l = blockIdx.x * 128 + threadIdx.x;
  if (l >= nL) return;
  R h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  for (int e = ptStart[l]; e < ptStart[l + 1]; e++) {
    const R w = Ed.info[e];
    if (w == 0) {   // culled edge: no contribution; its Hpl block must read as zero for the Schur and back-substitution kernels
      if (Ed.pidx[e] >= 0)
        for (int k = 0; k < 18; k++) Hpl[(size_t)18 * e + k] = 0;
      continue;
    }
    V3 t; Q4 q;
    load_pose(poses, Ed.pose[e], t, q);
    const R* X = pts + 3 * Ed.pt[e];
    const bool st = Ed.meas[3 * e + 2] >= 0;
    R A[3][3], B[3][6];
    jacobians(t, q, V3{X[0], X[1], X[2]}, st, cam, A, B);
    const R r0 = err[3 * e], r1 = err[3 * e + 1], r2 = err[3 * e + 2];
    R rho1 = 1;
    if (Ed.robust[e]) { R rr; huber(w * (r0 * r0 + r1 * r1 + r2 * r2), st ? deltaStereo : deltaMono, &rr, &rho1); }
    const R W = rho1 * w;
    const R omr[3] = {-w * r0 * rho1, -w * r1 * rho1, -w * r2 * rho1};
    int k = 0;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = r; c < 3; c++, k++) h[k] += A[0][r] * W * A[0][c] + A[1][r] * W * A[1][c] + A[2][r] * W * A[2][c];
#pragma unroll
    for (int r = 0; r < 3; r++) b[r] += A[0][r] * omr[0] + A[1][r] * omr[1] + A[2][r] * omr[2];
    if (Ed.pidx[e] >= 0) {
      R* hp = Hpl + (size_t)18 * e;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) hp[r * 3 + c] = B[0][r] * W * A[0][c] + B[1][r] * W * A[1][c] + B[2][r] * W * A[2][c];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) Hll[(size_t)6 * l + k] = h[k];
  bl[3 * l] = b[0]; bl[3 * l + 1] = b[1]; bl[3 * l + 2] = b[2];
}
__global__ __launch_bounds__(128) void k_build_points(EdgeSoA Ed, const int* __restrict__ ptStart, int nL, const R* __restrict__ poses,
                                                      const R* __restrict__ pts, Cam cam, R deltaMono, R deltaStereo,
                                                      const R* __restrict__ err, R* __restrict__ Hll, R* __restrict__ bl,
                                                      R* __restrict__ Hpl) { b_build_points(Ed, ptStart, nL, poses, pts, cam, deltaMono, deltaStereo, err, Hll, bl, Hpl); }


// buildSystem, pose side: one workgroup per free pose, threads over that pose's edges; Hpp[i] (6x6) and bp[6i..].
__device__ __forceinline__ void b_build_poses(EdgeSoA Ed, const int* __restrict__ poseStart, const int* __restrict__ poseEdges,
                                                     const R* __restrict__ poses, const R* __restrict__ pts, Cam cam, R deltaMono,
                                                     R deltaStereo, const R* __restrict__ err, R* __restrict__ Hpp,
                                                     R* __restrict__ bp) {
  __shared__ R lds[4];
  const int i = blockIdx.x;
  R h[21], b[6];
#pragma unroll
  for (int k = 0; k < 21; k++) h[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) b[k] = 0;
  for (int j = poseStart[i] + threadIdx.x; j < poseStart[i + 1]; j += 256) {
    const int e = poseEdges[j];
    const R w = Ed.info[e];
    if (w == 0) continue;   // culled edge
    V3 t; Q4 q;
    load_pose(poses, Ed.pose[e], t, q);
    const R* X = pts + 3 * Ed.pt[e];
    const bool st = Ed.meas[3 * e + 2] >= 0;
    R A[3][3], B[3][6];
    jacobians(t, q, V3{X[0], X[1], X[2]}, st, cam, A, B);
    const R r0 = err[3 * e], r1 = err[3 * e + 1], r2 = err[3 * e + 2];
    R rho1 = 1;
    if (Ed.robust[e]) { R rr; huber(w * (r0 * r0 + r1 * r1 + r2 * r2), st ? deltaStereo : deltaMono, &rr, &rho1); }
    const R W = rho1 * w;
    const R omr[3] = {-w * r0 * rho1, -w * r1 * rho1, -w * r2 * rho1};
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = r; c < 6; c++, k++) h[k] += B[0][r] * W * B[0][c] + B[1][r] * W * B[1][c] + B[2][r] * W * B[2][c];
#pragma unroll
    for (int r = 0; r < 6; r++) b[r] += B[0][r] * omr[0] + B[1][r] * omr[1] + B[2][r] * omr[2];
  }
  // 27 sums: butterfly inside each wave, one LDS hand-off of the 4 wave partials, 27 threads finish (fixed order)
  __shared__ R part[4][27];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 21; k++) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) h[k] += __shfl_xor(h[k], o, 64);
    if (lane == 0) part[wv][k] = h[k];
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) b[k] += __shfl_xor(b[k], o, 64);
    if (lane == 0) part[wv][21 + k] = b[k];
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int k = threadIdx.x;
    const R sum = part[0][k] + part[1][k] + part[2][k] + part[3][k];
    if (k < 21) {
      int r = 0, rem = k;
      while (rem >= 6 - r) { rem -= 6 - r; r++; }
      const int c = r + rem;
      Hpp[(size_t)36 * i + r * 6 + c] = sum;
      Hpp[(size_t)36 * i + c * 6 + r] = sum;
    } else {
      bp[6 * i + (k - 21)] = sum;
    }
  }
  (void)lds;
}
__global__ __launch_bounds__(256) void k_build_poses(EdgeSoA Ed, const int* __restrict__ poseStart, const int* __restrict__ poseEdges,
                                                     const R* __restrict__ poses, const R* __restrict__ pts, Cam cam, R deltaMono,
                                                     R deltaStereo, const R* __restrict__ err, R* __restrict__ Hpp,
                                                     R* __restrict__ bp) { b_build_poses(Ed, poseStart, poseEdges, poses, pts, cam, deltaMono, deltaStereo, err, Hpp, bp); }


// max |diagonal| of the Hessian (computeLambdaInit, levenberg.cpp:150-164) -> out[slot]
__device__ __forceinline__ void b_max_diag(const R* __restrict__ Hpp, int nP, const R* __restrict__ Hll, int nL, R* __restrict__ out,
                                                  int slot) {
  __shared__ R lds[4];
  R m = 0;
  for (int i = threadIdx.x; i < 6 * nP; i += 256) m = fmax(m, fabs(Hpp[(size_t)36 * (i / 6) + (i % 6) * 7]));
  for (int i = threadIdx.x; i < 3 * nL; i += 256) {
    const int l = i / 3, d = i % 3;
    m = fmax(m, fabs(Hll[(size_t)6 * l + (d == 0 ? 0 : d == 1 ? 3 : 5)]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[slot] = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
}
__global__ __launch_bounds__(256) void k_max_diag(const R* __restrict__ Hpp, int nP, const R* __restrict__ Hll, int nL, R* __restrict__ out,
                                                  int slot) { b_max_diag(Hpp, nP, Hll, nL, out, slot); }


// D^-1 = (Hll + lambda I)^-1 (Eigen cofactor inverse, block_solver.hpp:350) and db = D^-1 b_l, per landmark
__device__ __forceinline__ void b_dinv(const R* __restrict__ Hll, const R* __restrict__ bl, int nL, R lambda, R* __restrict__ Dinv,
                                              R* __restrict__ db, int* __restrict__ status) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l == 0) { status[0] = 0; status[1] = 0; }   // first kernel of a trial: clears the factorisation status
  if (l >= nL) return;
  const R* d = Hll + (size_t)6 * l;
  const R a = d[0] + lambda, b = d[1], c = d[2], e = d[3] + lambda, f = d[4], i = d[5] + lambda;
  const R c00 = e * i - f * f, c01 = c * f - b * i, c02 = b * f - c * e;
  const R id = 1.0 / (a * c00 + b * c01 + c * c02);
  R o[6] = {c00 * id, c01 * id, c02 * id, (a * i - c * c) * id, (b * c - a * f) * id, (a * e - b * b) * id};
#pragma unroll
  for (int k = 0; k < 6; k++) Dinv[(size_t)6 * l + k] = o[k];
  const R b0 = bl[3 * l], b1 = bl[3 * l + 1], b2 = bl[3 * l + 2];
  db[3 * l] = o[0] * b0 + o[1] * b1 + o[2] * b2;
  db[3 * l + 1] = o[1] * b0 + o[3] * b1 + o[4] * b2;
  db[3 * l + 2] = o[2] * b0 + o[4] * b1 + o[5] * b2;
}
__global__ __launch_bounds__(256) void k_dinv(const R* __restrict__ Hll, const R* __restrict__ bl, int nL, R lambda, R* __restrict__ Dinv,
                                              R* __restrict__ db, int* __restrict__ status) { b_dinv(Hll, bl, nL, lambda, Dinv, db, status); }


// ---- Schur complement (block_solver.hpp:342-393) without atomics ---------------------------------------------------
// S(i2,i1) = [i1==i2](Hpp_i1 + lambda I) - sum over landmarks seen by both poses of  B_i2 D^-1 B_i1^T   (lower triangle, i2 >= i1).
// The contributions are bucketed ONCE per optimize() by pose pair (the graph structure does not change between LM trials):
// k_pair_count / k_pair_fill / k_pair_sort build, for every pose pair, the list of (edge of i1, edge of i2) of the landmarks
// they share, ordered by landmark.  Per trial one wave owns one 6x6 block and walks its list: 36 lanes = 36 entries, no
// reduction, no atomics, fixed summation order (the first version issued 13 M FP64 atomics per trial: 0.87 ms).
__device__ __forceinline__ int pair_bucket(int i1, int i2) { return i2 * (i2 + 1) / 2 + i1; }  // i2 >= i1

__global__ __launch_bounds__(256) void k_pair_count(EdgeSoA Ed, const int* __restrict__ ptStart, int nL, int* __restrict__ cnt) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nL) return;
  for (int a = ptStart[l]; a < ptStart[l + 1]; a++) {
    const int i1 = Ed.pidx[a];
    if (i1 < 0) continue;
    for (int b = a; b < ptStart[l + 1]; b++) {
      const int i2 = Ed.pidx[b];  // edges of a landmark are sorted by pose index: i2 >= i1
      if (i2 >= 0) atomicAdd(&cnt[pair_bucket(i1, i2)], 1);
    }
  }
}
// exclusive scan of n ints by one workgroup: out[0..n], out[n] = total
__global__ __launch_bounds__(256) void k_excl_scan(const int* __restrict__ in, int n, int* __restrict__ out) {
  __shared__ int wsum[4];
  __shared__ int carryS;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) carryS = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int i = c0 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int off = carryS;
    for (int j = 0; j < wv; j++) off += wsum[j];
    if (i < n) out[i] = off + x - v;
    __syncthreads();
    if (threadIdx.x == 0) carryS += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carryS;
}
__global__ __launch_bounds__(256) void k_pair_fill(EdgeSoA Ed, const int* __restrict__ ptStart, int nL, int* __restrict__ cursor,
                                                   int2* __restrict__ items) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nL) return;
  for (int a = ptStart[l]; a < ptStart[l + 1]; a++) {
    const int i1 = Ed.pidx[a];
    if (i1 < 0) continue;
    for (int b = a; b < ptStart[l + 1]; b++) {
      const int i2 = Ed.pidx[b];
      if (i2 >= 0) items[atomicAdd(&cursor[pair_bucket(i1, i2)], 1)] = make_int2(a, b);
    }
  }
}
// order every bucket by its first edge index (== by landmark): rank sort from LDS, one wave per bucket, out of place.
// Buckets with more than kSortCap entries are copied unsorted (their summation order is then not reproducible run to run).
constexpr int kSortCap = 2048;   // 32 KB of LDS keys per workgroup; C5 has 800 observations per pose (its diagonal buckets)
__global__ __launch_bounds__(256) void k_pair_sort(const int* __restrict__ start, int nBuckets, const int2* __restrict__ in,
                                                   int2* __restrict__ out) {
  __shared__ int keys[4][kSortCap];
  const int wv = threadIdx.x >> 6, bkt = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
  if (bkt >= nBuckets) return;
  const int s0 = start[bkt], m = start[bkt + 1] - s0;
  if (m > kSortCap) {
    for (int t = lane; t < m; t += 64) out[s0 + t] = in[s0 + t];
    return;
  }
  for (int t = lane; t < m; t += 64) keys[wv][t] = in[s0 + t].x;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int t = lane; t < m; t += 64) {
    const int2 it = in[s0 + t];
    int rank = 0;
    for (int u = 0; u < m; u++) rank += keys[wv][u] < it.x;
    out[s0 + rank] = it;
  }
}
// BD_e = Hpl_e D^-1 (6x3) for every observation of a free pose
__device__ __forceinline__ void b_bd(EdgeSoA Ed, const int* __restrict__ edgeLm, const R* __restrict__ Hpl, const R* __restrict__ Dinv,
                                            R* __restrict__ BD) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Ed.n || Ed.pidx[e] < 0) return;
  const R* di = Dinv + (size_t)6 * edgeLm[e];
  const R* B = Hpl + (size_t)18 * e;
  R* o = BD + (size_t)18 * e;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    o[r * 3] = B[r * 3] * di[0] + B[r * 3 + 1] * di[1] + B[r * 3 + 2] * di[2];
    o[r * 3 + 1] = B[r * 3] * di[1] + B[r * 3 + 1] * di[3] + B[r * 3 + 2] * di[4];
    o[r * 3 + 2] = B[r * 3] * di[2] + B[r * 3 + 1] * di[4] + B[r * 3 + 2] * di[5];
  }
}
__global__ __launch_bounds__(256) void k_bd(EdgeSoA Ed, const int* __restrict__ edgeLm, const R* __restrict__ Hpl, const R* __restrict__ Dinv,
                                            R* __restrict__ BD) { b_bd(Ed, edgeLm, Hpl, Dinv, BD); }

// bs_i = contrib * bp_i - sum over the observations e of pose i of Hpl_e (D^-1 b_l)   (one workgroup per free pose)
__device__ __forceinline__ void b_bs(EdgeSoA Ed, const int* __restrict__ poseStart, const int* __restrict__ poseEdges,
                                            const int* __restrict__ edgeLm, const R* __restrict__ Hpl, const R* __restrict__ db,
                                            const R* __restrict__ bp, R contrib, R* __restrict__ bs) {
  __shared__ R part[4][6];
  const int i = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  R acc[6] = {0, 0, 0, 0, 0, 0};
  for (int j = poseStart[i] + threadIdx.x; j < poseStart[i + 1]; j += 256) {
    const int e = poseEdges[j];
    const R* B = Hpl + (size_t)18 * e;
    const R* d = db + (size_t)3 * edgeLm[e];
#pragma unroll
    for (int r = 0; r < 6; r++) acc[r] += B[r * 3] * d[0] + B[r * 3 + 1] * d[1] + B[r * 3 + 2] * d[2];
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
    if (lane == 0) part[wv][r] = acc[r];
  }
  __syncthreads();
  if (threadIdx.x < 6) bs[6 * i + threadIdx.x] = contrib * bp[6 * i + threadIdx.x] - (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_bs(EdgeSoA Ed, const int* __restrict__ poseStart, const int* __restrict__ poseEdges,
                                            const int* __restrict__ edgeLm, const R* __restrict__ Hpl, const R* __restrict__ db,
                                            const R* __restrict__ bp, R contrib, R* __restrict__ bs) { b_bs(Ed, poseStart, poseEdges, edgeLm, Hpl, db, bp, contrib, bs); }

// One workgroup per lower 6x6 block (i2 >= i1); the last workgroup also writes the identity padding of S and bs.
// The block is the contraction  [Hpl_b rows q | 3m columns] x [BD_a rows r | 3m columns]^T  over the m landmarks the two poses
// share: it runs on the FP64 matrix core, v_mfma_f64_16x16x4_f64 (A[i = lane&15][k = lane>>4], B[k][j = lane&15],
// D col = lane&15, row = (lane>>4) + 4*reg; rows/cols >= 6 are fed zeros), four k-columns per instruction.
typedef double double4_t __attribute__((ext_vector_type(4)));
#ifndef SCHUR_WAVES
#define SCHUR_WAVES 4
#endif
constexpr int kSchurWaves = SCHUR_WAVES;   // waves per bucket
__device__ __forceinline__ void b_schur_pairs(const int* __restrict__ start, const int2* __restrict__ items, int nP, int nBuckets,
                                                     const R* __restrict__ BD, const R* __restrict__ Hpl, const R* __restrict__ Hpp, R lambda,
                                                     R contrib, int n, R* __restrict__ S, R* __restrict__ bs) {
  // One workgroup per bucket, its items split over the workgroup's waves (the buckets of poses that see the same landmarks hold hundreds of
  // items while most others are empty: with one wave per bucket ~700 waves carried the whole launch); the partial blocks
  // are added in wave order, so the result does not depend on timing.
  __shared__ R part[kSchurWaves][36];
  const int bkt = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (bkt >= nBuckets) {
    if (bkt == nBuckets) {  // the extra workgroup; padding rows/cols 6 nP .. n-1: identity (scaled like the rest for the all-reduce)
      for (int idx = threadIdx.x; idx < (n - 6 * nP) * n; idx += 64 * kSchurWaves) {
        const int r = 6 * nP + idx / n, c = idx % n;
        if (c <= r) S[(size_t)r * n + c] = r == c ? contrib : 0.0;
      }
      for (int r = 6 * nP + threadIdx.x; r < n; r += 64 * kSchurWaves) bs[r] = 0;
    }
    return;
  }
  int i2 = (int)((sqrt(8.0 * bkt + 1.0) - 1.0) * 0.5);
  while ((i2 + 1) * (i2 + 2) / 2 <= bkt) i2++;
  while (i2 * (i2 + 1) / 2 > bkt) i2--;
  const int i1 = bkt - i2 * (i2 + 1) / 2;
  const int i16 = lane & 15, kq = lane >> 4;
  const bool live = i16 < 6;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int b0 = start[bkt], b1 = start[bkt + 1];
  const int per = (((b1 - b0) + kSchurWaves - 1) / kSchurWaves + 3) & ~3;   // items per wave, a multiple of the 4 MFMA k-slots
  const int s0 = min(b0 + wv * per, b1), s1 = min(s0 + per, b1);
  // K ordering chosen for the loads, not for the landmarks: MFMA k-slot kq takes item 4g + kq and the three instructions of a
  // group take that item's columns c = 0, 1, 2 — so lane (row i16, slot kq) reads ITS row of ITS item as three consecutive
  // doubles from each operand (the 16 lanes of a slot cover one contiguous 144-byte block), no index shuffles, and 8 groups
  // (32 items) of loads are in flight before the first MFMA waits.
  for (int base = s0; base < s1; base += 32) {
    R av[8][3], bv[8][3];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const int idx = base + 4 * g + kq;
      const bool on = live && idx < s1;
      const int2 it = on ? items[idx] : make_int2(0, 0);
      const R* pa = Hpl + (size_t)18 * it.y + i16 * 3;
      const R* pb = BD + (size_t)18 * it.x + i16 * 3;
#pragma unroll
      for (int c = 0; c < 3; c++) { av[g][c] = on ? pa[c] : 0.0; bv[g][c] = on ? pb[c] : 0.0; }
    }
#pragma unroll
    for (int g = 0; g < 8; g++)
      if (base + 4 * g < s1) {
#pragma unroll
        for (int c = 0; c < 3; c++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][c], bv[g][c], acc, 0, 0, 0);
      }
  }
  if (live) {
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int q = kq + 4 * reg;
      if (q < 6) part[wv][q * 6 + i16] = acc[reg];
    }
  }
  __syncthreads();
  if (threadIdx.x < 36) {
    const int q = threadIdx.x / 6, r = threadIdx.x - 6 * q;
    R v = 0;
#pragma unroll
    for (int w = 0; w < kSchurWaves; w++) v -= part[w][threadIdx.x];
    if (i1 == i2) v += contrib * (Hpp[(size_t)36 * i1 + q * 6 + r] + (q == r ? lambda : 0.0));
    S[(size_t)(6 * i2 + q) * n + 6 * i1 + r] = v;
  }
}
__global__ __launch_bounds__(64 * kSchurWaves) void k_schur_pairs(const int* __restrict__ start, const int2* __restrict__ items, int nP, int nBuckets,
                                                     const R* __restrict__ BD, const R* __restrict__ Hpl, const R* __restrict__ Hpp, R lambda,
                                                     R contrib, int n, R* __restrict__ S, R* __restrict__ bs) { b_schur_pairs(start, items, nP, nBuckets, BD, Hpl, Hpp, lambda, contrib, n, S, bs); }


// ---- dense blocked right-looking Cholesky of S (n multiple of 32, lower triangle, row-major) ----------------------
// Launch kb (0 .. nb-1) has one workgroup per tile (i >= j >= kb) of the trailing lower triangle:
//   * every tile first applies the rank-32 update of the previous panel:  A(i,j) -= L(i,kb-1) L(j,kb-1)^T ;
//   * tiles of column kb then factor the panel: each recomputes the updated 32x32 diagonal tile (one extra rank-32 update),
//     factors it with a single wave in LDS (no workgroup barriers inside the 32 pivot steps) and solves its own tile against it;
//   * workgroup (kb,kb) stores the diagonal factor and its inverse in diagL / diagInv (S's own diagonal tile is left untouched
//     because the other workgroups of the column still read it).
// All (nb-kb)(nb-kb+1)/2 tiles of a launch run in parallel (the first, left-looking version kept <= 19 workgroups busy and took
// 19 x 68 us).  status[0] is set when a pivot is not positive (LinearSolverEigen reports failure -> the LM step is rejected,
// linear_solver_eigen.h:118-126).
constexpr int NB = 32;
constexpr int NBP = NB + 1;

// LL^T of the lower part of D (LDS, [NB][NBP]) by ONE wave; returns false on a non-positive pivot.
// Lane i (< 32) keeps row i in 32 registers (all loops fully unrolled, so every index is a compile-time constant).  Per pivot
// step the pivot is broadcast with v_readlane (constant lane -> scalar registers).  (History, CHOL_TIMING build: element-wise LDS
// updates ~30 us per tile; register rows with an LDS column vector and two wave barriers per pivot 10 us; all-readlane 9.2 us.)
__device__ __forceinline__ R bcast_lane(R v, int srcLane) {   // srcLane must be a compile-time constant after unrolling
  const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), srcLane), hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ R rsqrt_nr(R d) {   // v_rsq_f64 seed + two Newton steps (relative error ~1e-16 for normal d > 0)
  R y = __builtin_amdgcn_rsq(d);
  const R h = 0.5 * d;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
__device__ __forceinline__ bool potrf_wave(R (*D)[NBP], R (*Lp)[NB][4], R* rdiag, int lane) {
  // Blocked by 4 columns: inside a block the pivots and the (at most 3) in-block multipliers go through v_readlane; the block's
  // four finished columns then make ONE LDS round trip (row-major [row][4], double-buffered) and every later column takes its
  // rank-4 update from two broadcast ds_read_b128 + 4 FMAs — 8 round trips instead of 32, and the 992 v_readlane of the
  // all-readlane form (which made a single wave issue-bound: 9.2 us) shrink to 160.  Same operation order per element as the
  // rank-1 form (k ascending).
  const int i = lane & 31;
  R a[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) a[c] = D[i][c];
  bool ok = true;
#pragma unroll
  for (int b = 0; b < NB / 4; b++) {
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const int j = 4 * b + jj;
      const R d = bcast_lane(a[j], j);           // pivot: element (j, j) lives in lane j
      if (!(d > 0)) ok = false;
      const R inv = rsqrt_nr(d);                 // one reciprocal square root per pivot instead of a divide per row
      const R lij = a[j] * inv;                  // rows i < j hold stale values in a[j]; they are never used again
      a[j] = lij;
      if (lane == j) rdiag[j] = inv;
#pragma unroll
      for (int c = j + 1; c < 4 * b + 4; c++) a[c] -= lij * bcast_lane(lij, c);   // L(c, j) lives in lane c
    }
    if (b < NB / 4 - 1) {
      R (*buf)[4] = Lp[b & 1];
      if (lane < NB) {
        *reinterpret_cast<double2*>(&buf[i][0]) = make_double2(a[4 * b], a[4 * b + 1]);
        *reinterpret_cast<double2*>(&buf[i][2]) = make_double2(a[4 * b + 2], a[4 * b + 3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 4 * b + 4; c < NB; c++) {
        const double2 l01 = *reinterpret_cast<const double2*>(&buf[c][0]), l23 = *reinterpret_cast<const double2*>(&buf[c][2]);
        R v = a[c];
        v -= a[4 * b] * l01.x; v -= a[4 * b + 1] * l01.y; v -= a[4 * b + 2] * l23.x; v -= a[4 * b + 3] * l23.y;
        a[c] = v;
      }
    }
  }
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; c++) D[i][c] = c <= i ? a[c] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return ok;
}

// The same factorisation on the whole 256-thread workgroup: wave w keeps columns 8w .. 8w+7 of every row (row per lane, as above),
// the wave that owns a 4-column block factors it and publishes it through LDS, then ALL waves apply the block's rank-4 update to
// their own columns side by side.  One wave did 112 column updates in sequence between the 8 block factorisations; here at most 8
// follow each block before the next owner can start.  Same operations in the same order per element (k ascending), so the factor
// is bit-identical to potrf_wave's.  One workgroup barrier per block; *okFlag (pre-set to 1) is cleared on a non-positive pivot.
__device__ __forceinline__ void potrf_block(R (*D)[NBP], R (*Lp)[NB][4], R* rdiag, int* okFlag, int tid) {
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31;
  R a[8];
#pragma unroll
  for (int c = 0; c < 8; c++) a[c] = D[i][8 * wv + c];
  bool ok = true;
#pragma unroll
  for (int b = 0; b < NB / 4; b++) {
    const int owner = b >> 1, cb = 4 * (b & 1);
    R (*buf)[4] = Lp[b & 1];
    if (wv == owner) {
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const int j = 4 * b + jj;
        const R d = bcast_lane(a[cb + jj], j);     // pivot: element (j, j) lives in lane j
        if (!(d > 0)) ok = false;
        const R inv = rsqrt_nr(d);
        const R lij = a[cb + jj] * inv;            // rows i < j hold stale values; they are never used again
        a[cb + jj] = lij;
        if (lane == j) rdiag[j] = inv;
#pragma unroll
        for (int c = jj + 1; c < 4; c++) a[cb + c] -= lij * bcast_lane(lij, 4 * b + c);   // L(c, j) lives in lane c
      }
      if (b < NB / 4 - 1 && lane < NB) {
        *reinterpret_cast<double2*>(&buf[i][0]) = make_double2(a[cb], a[cb + 1]);
        *reinterpret_cast<double2*>(&buf[i][2]) = make_double2(a[cb + 2], a[cb + 3]);
      }
    }
    if (b < NB / 4 - 1) {
      __syncthreads();
      if (8 * wv + 7 >= 4 * b + 4) {   // this wave still has columns behind the block
        const double2 m01 = *reinterpret_cast<const double2*>(&buf[i][0]), m23 = *reinterpret_cast<const double2*>(&buf[i][2]);   // L(i, block)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int gc = 8 * wv + c;
          if (gc >= 4 * b + 4) {
            const double2 l01 = *reinterpret_cast<const double2*>(&buf[gc][0]), l23 = *reinterpret_cast<const double2*>(&buf[gc][2]);
            R v = a[c];
            v -= m01.x * l01.x; v -= m01.y * l01.y; v -= m23.x * l23.x; v -= m23.y * l23.y;
            a[c] = v;
          }
        }
      }
    }
  }
  if (!ok && lane == 0) *okFlag = 0;
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < 8; c++) D[i][8 * wv + c] = 8 * wv + c <= i ? a[c] : 0.0;
  }
}

__device__ __forceinline__ int tri_index(int t, int* row) {  // t -> (row, col) of a packed lower triangle, row-major
  int r = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= t) r++;
  while (r * (r + 1) / 2 > t) r--;
  *row = r;
  return t - r * (r + 1) / 2;
}

#ifdef CHOL_TIMING
__device__ long long g_cholClk[2][32][12];
#define CH_CLK(p) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 1)) g_cholClk[blockIdx.x][kb][p] = wall_clock64(); } while (0)
#else
#define CH_CLK(p) do { } while (0)
#endif
__device__ __forceinline__ void b_chol_step(R* __restrict__ S, R* __restrict__ diagL, R* __restrict__ diagInv, int n, int kb,
                                                   int* __restrict__ status, R* __restrict__ bvec, R* __restrict__ yv) {
  __shared__ R Ta[NB][NBP];   // this tile
  __shared__ R Dg[NB][NBP];   // diagonal tile of the panel (panel workgroups only)
  __shared__ R La[NB][NBP];   // L(i, kb-1)
  __shared__ R Lb[NB][NBP];   // L(j, kb-1)
  __shared__ R Lk[NB][NBP];   // L(kb, kb-1)
  __shared__ __align__(16) R Lp[2][NB][4];   // potrf: the current 4-column block of L, double-buffered
  __shared__ R rdiag[NB];   // 1 / L_jj
  __shared__ int sOk;
  CH_CLK(0);
  const int tid = threadIdx.x, tr = tid >> 3, tc4 = (tid & 7) * 4;
  if (blockIdx.x == (unsigned)((n / NB - kb) * (n / NB - kb + 1) / 2)) {
    // Extra workgroup (launches kb >= 1): the forward substitution L y = b rides along, off the factorisation's critical path.
    //   y_{kb-1} = Linv_{kb-1} b_{kb-1}   (block inverse stored by launch kb-1; b_{kb-1} already holds every earlier panel's update)
    //   b_r -= L(r, panel kb-1) y_{kb-1}  for all remaining rows (panel kb-1 is final; this launch only touches tiles >= kb).
    // The last block's y is computed by k_chol_solve.
    __shared__ R yprev[NB];
    {
      const int r = tid >> 3, part = tid & 7;   // 8 threads per row, 4 terms each
      const R* inv = diagInv + (size_t)(kb - 1) * NB * NB + r * NB;
      R p = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) p += inv[part * 4 + k] * bvec[(kb - 1) * NB + part * 4 + k];   // Linv is lower triangular (zeros above)
      p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64);
      if (part == 0) { yprev[r] = p; yv[(kb - 1) * NB + r] = p; }
    }
    __syncthreads();
    for (int r = kb * NB + tid; r < n; r += 256) {
      const R* row = S + (size_t)r * n + (kb - 1) * NB;
      R sres = 0;
#pragma unroll
      for (int k = 0; k < NB; k++) sres += row[k] * yprev[k];
      bvec[r] -= sres;
    }
    return;
  }
  int ri;
  const int cj = tri_index(blockIdx.x, &ri);
  const int i = kb + ri, j = kb + cj;          // tile (i, j), i >= j >= kb
  const bool panel = j == kb;
  CH_CLK(1);
  // The rank-32 update of the tile (and, on panel workgroups, of the diagonal tile) on the FP64 matrix core: wave w owns the 16 x 16
  // quadrant (w >> 1, w & 1), D = C - A B with A[i][k] = L(i, kb-1) rows, B[k][j] = L(j, kb-1)^T, eight v_mfma_f64_16x16x4_f64 per
  // quadrant.  Lane layout of the instruction: A and B element (i or j = lane % 16, k = lane / 16); D element (i = lane / 16 + 4 reg,
  // j = lane % 16).  The operand tiles go through LDS once (coalesced loads) instead of 5 LDS reads per 4 multiply-adds.
  const int mw = tid >> 6, ml = tid & 63, i16 = ml & 15, kq = ml >> 4, qr = 16 * (mw >> 1), qc = 16 * (mw & 1);
  double4_t acc, accD = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int reg = 0; reg < 4; reg++) {
    acc[reg] = S[(size_t)(i * NB + qr + kq + 4 * reg) * n + j * NB + qc + i16];
    if (panel) accD[reg] = S[(size_t)(kb * NB + qr + kq + 4 * reg) * n + kb * NB + qc + i16];
  }
  CH_CLK(2);
  if (kb > 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      La[tr][tc4 + c] = S[(size_t)(i * NB + tr) * n + (kb - 1) * NB + tc4 + c];
      Lb[tr][tc4 + c] = S[(size_t)(j * NB + tr) * n + (kb - 1) * NB + tc4 + c];
      if (panel) Lk[tr][tc4 + c] = S[(size_t)(kb * NB + tr) * n + (kb - 1) * NB + tc4 + c];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < NB; ks += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(0.0 - La[qr + i16][ks + kq], Lb[qc + i16][ks + kq], acc, 0, 0, 0);
    if (panel) {
#pragma unroll
      for (int ks = 0; ks < NB; ks += 4) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(0.0 - Lk[qr + i16][ks + kq], Lk[qc + i16][ks + kq], accD, 0, 0, 0);
    }
  }
  CH_CLK(3);
  if (!panel) {
#pragma unroll
    for (int reg = 0; reg < 4; reg++) S[(size_t)(i * NB + qr + kq + 4 * reg) * n + j * NB + qc + i16] = acc[reg];
    return;
  }
#pragma unroll
  for (int reg = 0; reg < 4; reg++) { Ta[qr + kq + 4 * reg][qc + i16] = acc[reg]; Dg[qr + kq + 4 * reg][qc + i16] = accD[reg]; }
  if (tid == 0) sOk = 1;
  __syncthreads();
  CH_CLK(4);
#ifndef CHOL_SKIP_POTRF
#ifdef CHOL_POTRF_ONE_WAVE
  if (tid < 64) {
    const bool ok = potrf_wave(Dg, Lp, rdiag, tid);
    if (tid == 0) sOk = ok ? 1 : 0;
  }
#else
  potrf_block(Dg, Lp, rdiag, &sOk, tid);
#endif
#else
  if (tid == 0) sOk = 1;
  if (tid < NB) rdiag[tid] = 1.0;
#endif
  __syncthreads();
  CH_CLK(5);
  if (!sOk) {
    if (tid == 0) atomicMax(status, 1);
    return;
  }
  // inverse of the diagonal factor, kept in LDS: the panel solve X L_kk^T = T then is a 32x32x32 product X = T * Linv^T on all 256
  // threads instead of 32 serial substitutions.  Linv is built bottom-up by 2x2 blocks, L = [A 0; B C] -> Linv = [A^-1 0; -C^-1 B A^-1  C^-1]:
  // the four 8x8 diagonal blocks by forward substitution (one column per lane on 32 lanes: a 28-term dependent chain; a direct
  // 32x32 inverse is a 528-term chain and measured 9 us, two 16x16 ones 2.0 us), then the two 16x16 blocks and the 32x32 block by
  // two small products each on the whole workgroup.
  if (tid < NB) {
    const int off = (tid >> 3) * 8, c = tid & 7;
    R z[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      R sres = r == c ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 8; k++) if (k < r) sres -= Dg[off + r][off + k] * z[k];
      z[r] = r < c ? 0.0 : sres * rdiag[off + r];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) La[off + r][off + c] = z[r];   // La is free after the update: La = Linv
  }
  CH_CLK(6);
  {
    const int r = tid >> 3, c0 = (tid & 7) * 4;   // zeros above the 8x8 diagonal blocks
#pragma unroll
    for (int c = c0; c < c0 + 4; c++) if ((c >> 3) > (r >> 3)) La[r][c] = 0.0;
  }
  __syncthreads();
  {  // 16x16 level, both diagonal blocks at once: p = which block, (r, c) in the 8x8 off-diagonal part
    const int p16 = (tid >> 6) & 1, r = (tid >> 3) & 7, c = tid & 7, o = 16 * p16;
    R t1 = 0;
    if (tid < 128) {
#pragma unroll
      for (int k = 0; k < 8; k++) t1 += Dg[o + 8 + r][o + k] * La[o + k][o + c];   // T1 = B A^-1
      Lb[o + r][c] = t1;
    }
    __syncthreads();
    if (tid < 128) {
      R m = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) m -= La[o + 8 + r][o + 8 + k] * Lb[o + k][c];     // M = -C^-1 T1
      La[o + 8 + r][o + c] = m;
    }
  }
  __syncthreads();
  {
    const int r = tid >> 4, c = tid & 15;   // T1 = B A^-1
    R t1 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) t1 += Dg[16 + r][k] * La[k][c];
    Lb[r][c] = t1;
  }
  __syncthreads();
  {
    const int r = tid >> 4, c = tid & 15;   // M = -C^-1 T1
    R m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) m -= La[16 + r][16 + k] * Lb[k][c];
    La[16 + r][c] = m;
  }
  __syncthreads();
  CH_CLK(7);
  if (i == kb) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      diagL[(size_t)kb * NB * NB + tr * NB + tc4 + c] = tc4 + c <= tr ? Dg[tr][tc4 + c] : 0;
      diagInv[(size_t)kb * NB * NB + tr * NB + tc4 + c] = La[tr][tc4 + c];
    }
    CH_CLK(8);
    return;
  }
  R xo[4] = {0, 0, 0, 0};
#pragma unroll 8
  for (int k = 0; k < NB; k++) {
    const R t = Ta[tr][k];
#pragma unroll
    for (int c = 0; c < 4; c++) xo[c] += t * La[tc4 + c][k];
  }
  CH_CLK(9);
#pragma unroll
  for (int c = 0; c < 4; c++) S[(size_t)(i * NB + tr) * n + kb * NB + tc4 + c] = xo[c];
  CH_CLK(10);
}
__global__ __launch_bounds__(256) void k_chol_step(R* __restrict__ S, R* __restrict__ diagL, R* __restrict__ diagInv, int n, int kb,
                                                   int* __restrict__ status, R* __restrict__ bvec, R* __restrict__ yv) { b_chol_step(S, diagL, diagInv, n, kb, status, bvec, yv); }


// Backward substitution L^T x = y (y comes out of the factorisation launches), single workgroup, blocked by 32 with the
// diagonal blocks applied through their inverses:  x_k = Linv_kk^T y_k ;  y_j -= L(k,j)^T x_k  (j < k).  It first finishes the
// forward substitution (last block of y).  The chain over the 19 blocks is serial, so each step is kept short: the block inverse (exactly 1024 numbers, one per
// thread) and the 32 L entries a thread needs for the update are fetched BEFORE the step's reduction (they do not depend on x),
// and x_k is a 32-way shuffle reduction on all 1024 threads instead of a 32-term loop on 32 of them.
__device__ __forceinline__ void b_chol_solve(const R* __restrict__ L, const R* __restrict__ diagInv, int n, const R* __restrict__ yin,
                                                     const R* __restrict__ bvec, R* __restrict__ x) {
  extern __shared__ R y[];  // [n]
  __shared__ R inv[NB][NB + 1];
  __shared__ R yk[NB];
  const int tid = threadIdx.x, nb = n / NB;
  for (int i = tid; i < n - NB; i += 1024) y[i] = yin[i];
  {  // forward substitution's last block: y_last = Linv_last b_last (thread (r, k) = (tid / 32, tid % 32); zeros above the diagonal)
    const int r = tid >> 5, k = tid & 31;
    R p = diagInv[(size_t)(nb - 1) * NB * NB + tid] * bvec[(nb - 1) * NB + k];
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
    if (k == 0) y[(nb - 1) * NB + r] = p;
  }
  R invNext = diagInv[(size_t)(nb - 1) * NB * NB + tid];
  for (int kb = nb - 1; kb >= 0; kb--) {
    const R invReg = invNext;                                       // element (r, c) = (tid / 32, tid % 32); fetched one step ahead:
    if (kb > 0) invNext = diagInv[(size_t)(kb - 1) * NB * NB + tid];   // the step needs it right after its first barrier
    R lrow[NB];
    const bool upd = tid < kb * NB;                                 // this thread's column of the update
#pragma unroll
    for (int r = 0; r < NB; r++) lrow[r] = upd ? L[(size_t)(kb * NB + r) * n + tid] : 0.0;
    __syncthreads();                                                // previous step's y updates are visible; inv/yk are free
    inv[tid >> 5][tid & 31] = invReg;
    __syncthreads();
    {  // x_k[c] = sum_{r >= c} Linv[r][c] y_k[r]: thread (c, r) = (tid / 32, tid % 32), reduced over the 32 lanes of a half wave
      const int c = tid >> 5, r = tid & 31;
      R p = r >= c ? inv[r][c] * y[kb * NB + r] : 0.0;
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
      if (r == 0) yk[c] = p;
    }
    __syncthreads();
    if (tid < NB) y[kb * NB + tid] = yk[tid];
    if (upd) {
      R sres = 0;
#pragma unroll
      for (int r = 0; r < NB; r++) sres += lrow[r] * yk[r];
      y[tid] -= sres;
    }
    // systems wider than the workgroup (n > 1024: global BA with > 170 free keyframes): the remaining columns of the update,
    // strided, L read after the reduction (no prefetch registers for them)
    for (int col = tid + 1024; col < kb * NB; col += 1024) {
      R sres = 0;
#pragma unroll 8
      for (int r = 0; r < NB; r++) sres += L[(size_t)(kb * NB + r) * n + col] * yk[r];
      y[col] -= sres;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) x[i] = y[i];
}
__global__ __launch_bounds__(1024) void k_chol_solve(const R* __restrict__ L, const R* __restrict__ diagInv, int n, const R* __restrict__ yin,
                                                     const R* __restrict__ bvec, R* __restrict__ x) { b_chol_solve(L, diagInv, n, yin, bvec, x); }


// Launch of k_chol_solve: y[n] sits in dynamic LDS, so n is bounded by the CU's 160 KiB (minus the kernel's 8.7 KiB of static LDS)
// and anything above the 64 KiB default needs the attribute.  Returns false when the system is too wide.
constexpr int kCholSolveMaxN = (160 * 1024 - 10 * 1024) / (int)sizeof(R) / NB * NB;   // 19 200 rows = 3 200 keyframes
inline bool launch_chol_solve(hipStream_t s, const R* L, const R* diagInv, int n, const R* yin, const R* bvec, R* x) {
  if (n > kCholSolveMaxN) return false;
  const size_t dyn = sizeof(R) * (size_t)n;
  if (dyn > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
    return false;
  hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(1024), dyn, s, L, diagInv, n, yin, bvec, x);
  return true;
}

// landmark step (block_solver.hpp:420-444): xl = D^-1 (b_l - sum_e Hpl_e^T xp[pose(e)])
__device__ __forceinline__ void b_backsub(EdgeSoA Ed, const int* __restrict__ ptStart, int nL, const R* __restrict__ Hpl,
                                                 const R* __restrict__ Dinv, const R* __restrict__ bl, const R* __restrict__ xp,
                                                 R* __restrict__ xl) {
  const int l = blockIdx.x * 128 + threadIdx.x;
  if (l >= nL) return;
  R c0 = bl[3 * l], c1 = bl[3 * l + 1], c2 = bl[3 * l + 2];
  for (int e = ptStart[l]; e < ptStart[l + 1]; e++) {
    const int i1 = Ed.pidx[e];
    if (i1 < 0) continue;
    const R* B = Hpl + (size_t)18 * e;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const R xv = xp[6 * i1 + r];
      c0 -= B[r * 3] * xv; c1 -= B[r * 3 + 1] * xv; c2 -= B[r * 3 + 2] * xv;
    }
  }
  const R* di = Dinv + (size_t)6 * l;
  xl[3 * l] = di[0] * c0 + di[1] * c1 + di[2] * c2;
  xl[3 * l + 1] = di[1] * c0 + di[3] * c1 + di[4] * c2;
  xl[3 * l + 2] = di[2] * c0 + di[4] * c1 + di[5] * c2;
}
__global__ __launch_bounds__(128) void k_backsub(EdgeSoA Ed, const int* __restrict__ ptStart, int nL, const R* __restrict__ Hpl,
                                                 const R* __restrict__ Dinv, const R* __restrict__ bl, const R* __restrict__ xp,
                                                 R* __restrict__ xl) { b_backsub(Ed, ptStart, nL, Hpl, Dinv, bl, xp, xl); }


// VertexSE3Expmap::oplusImpl: pose <- exp(u) * pose, u = [omega, upsilon] (se3quat.h:100-106, 218-257)
__device__ __forceinline__ void pose_oplus(V3& t, Q4& q, const R* u) {
  const V3 omega{u[0], u[1], u[2]}, ups{u[3], u[4], u[5]};
  const R theta = sqrt(omega.x * omega.x + omega.y * omega.y + omega.z * omega.z);
  const R Om[3][3] = {{0, -omega.z, omega.y}, {omega.z, 0, -omega.x}, {-omega.y, omega.x, 0}};
  R Om2[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) Om2[a][b] = Om[a][0] * Om[0][b] + Om[a][1] * Om[1][b] + Om[a][2] * Om[2][b];
  R ca, cb, cc;
  const bool small = theta < 0.00001;
  if (small) { ca = 1; cb = 0.5; cc = 1. / 6.; }
  else { ca = sin(theta) / theta; cb = (1 - cos(theta)) / (theta * theta); cc = (theta - sin(theta)) / pow(theta, 3.0); }
  R Rm[3][3], V[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      const R I = a == b ? 1.0 : 0.0;
      if (small) { Rm[a][b] = I + Om[a][b] + 0.5 * Om2[a][b]; V[a][b] = I + 0.5 * Om[a][b] + cc * Om2[a][b]; }
      else { Rm[a][b] = I + ca * Om[a][b] + cb * Om2[a][b]; V[a][b] = I + cb * Om[a][b] + cc * Om2[a][b]; }
    }
  Q4 eq = rToQ(Rm);
  qnormalize(eq);
  const V3 et{V[0][0] * ups.x + V[0][1] * ups.y + V[0][2] * ups.z, V[1][0] * ups.x + V[1][1] * ups.y + V[1][2] * ups.z,
              V[2][0] * ups.x + V[2][1] * ups.y + V[2][2] * ups.z};
  t = add(et, qrot(eq, t));
  q = qmul(eq, q);
  qnormalize(q);
}

// SparseOptimizer::update (sparse_optimizer.cpp:433): poses <- exp(dxi) * pose (se3quat.h:100-106, 218-257),
// points <- X + dX.  Reads `src`, writes `dst` (push/pop become a buffer swap on the host).
__device__ __forceinline__ void b_update(const R* __restrict__ srcPoses, const R* __restrict__ srcPts, R* __restrict__ dstPoses,
                                                R* __restrict__ dstPts, const int* __restrict__ poseOf, int nP,
                                                const int* __restrict__ ptOf, int nL, const R* __restrict__ xp, const R* __restrict__ xl) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nP) {
    const int k = poseOf[i];
    V3 t; Q4 q;
    load_pose(srcPoses, k, t, q);
    V3 nt = t; Q4 nq = q;
    pose_oplus(nt, nq, xp + 6 * i);
    R* o = dstPoses + 7 * k;
    o[0] = nt.x; o[1] = nt.y; o[2] = nt.z; o[3] = nq.x; o[4] = nq.y; o[5] = nq.z; o[6] = nq.w;
  }
  if (i < nL) {
    const int p = ptOf[i];
    dstPts[3 * p] = srcPts[3 * p] + xl[3 * i];
    dstPts[3 * p + 1] = srcPts[3 * p + 1] + xl[3 * i + 1];
    dstPts[3 * p + 2] = srcPts[3 * p + 2] + xl[3 * i + 2];
  }
}
__global__ __launch_bounds__(256) void k_update(const R* __restrict__ srcPoses, const R* __restrict__ srcPts, R* __restrict__ dstPoses,
                                                R* __restrict__ dstPts, const int* __restrict__ poseOf, int nP,
                                                const int* __restrict__ ptOf, int nL, const R* __restrict__ xp, const R* __restrict__ xl) { b_update(srcPoses, srcPts, dstPoses, dstPts, poseOf, nP, ptOf, nL, xp, xl); }


// computeScale (levenberg.cpp:166-173): sum_j x_j (lambda x_j + b_j) over pose and landmark unknowns -> per-workgroup partials
__device__ __forceinline__ void b_scale(const R* __restrict__ xp, const R* __restrict__ bp, int np6, const R* __restrict__ xl,
                                               const R* __restrict__ bl, int nl3, R lambda, R* __restrict__ partial) {
  __shared__ R lds[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  R v = 0;
  if (i < np6) v = xp[i] * (lambda * xp[i] + bp[i]);
  else if (i - np6 < nl3) { const int j = i - np6; v = xl[j] * (lambda * xl[j] + bl[j]); }
  const R s = block_sum(v, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_scale(const R* __restrict__ xp, const R* __restrict__ bp, int np6, const R* __restrict__ xl,
                                               const R* __restrict__ bl, int nl3, R lambda, R* __restrict__ partial) { b_scale(xp, bp, np6, xl, bl, nl3, lambda, partial); }


// The chi2 / depth test of optimizer.cpp:292-311 (after the first stage) and :318-334 (final outlier list) on the device, so that
// neither the errors (24 B per edge) nor the depths travel to the host between the stages.  chi2 = information * |error|^2 with the
// ORIGINAL information and the error of the edge's last evaluation (a culled edge keeps it: it left g2o's active set), depth of the
// point under the current estimate (isDepthPositive, types_six_dof_expmap.h:215-219,276-280).
// final == 0: a failing edge gets information 0 (= level 1: every kernel skips it) and every edge loses its robust kernel;
// final == 1: outlier[e] = fails.
__global__ __launch_bounds__(256) void k_cull(const int* __restrict__ ePose, const int* __restrict__ ePt, const R* __restrict__ meas, int E,
                                              const R* __restrict__ info0, const R* __restrict__ err, const R* __restrict__ poses,
                                              const R* __restrict__ pts, R chi2Mono, R chi2Stereo, int final, R* __restrict__ info,
                                              uint8_t* __restrict__ robust, uint8_t* __restrict__ outlier) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  V3 t; Q4 q;
  load_pose(poses, ePose[e], t, q);
  const R* X = pts + 3 * ePt[e];
  const R depth = add(qrot(q, V3{X[0], X[1], X[2]}), t).z;
  const R* r = err + 3 * e;
  const R chi2 = info0[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const R th = meas[3 * e + 2] >= 0 ? chi2Stereo : chi2Mono;
  const bool bad = chi2 > th || !(depth > 0.0);
  if (final) outlier[e] = bad ? 1 : 0;
  else { if (bad) info[e] = 0; robust[e] = 0; }
}


// ------------------------------------------------------------------------------------------------------------------------
// Optimizer::optimizePose (reference src/optimizer.cpp:358-501): pose-only Levenberg on unary reprojection edges,
// four episodes of <= 10 iterations from the same start pose with an inlier/outlier re-classification after each.
// GPU form: ONE workgroup = one frame's whole optimizePose call (all 4 episodes, every LM trial) in a single launch — the
// reference's per-iteration graph walk is ~40 dependent steps of a few microseconds each, far too fine for separate launches;
// a batch of frames is a grid of such workgroups.  Per pass every thread walks its edges (stride 256), the 28 sums
// (21 of H, 6 of b, chi2) are reduced in a fixed order (wave butterfly, then the 4 wave partials in order), and every thread
// then runs the scalar LM bookkeeping and the 6x6 Cholesky redundantly on the same numbers, so no broadcast is needed.
// Edge Jacobians follow EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus (invz products,
// types_six_dof_expmap.cpp:415-494); the error is the same computeError as the binary edges.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kPoseThreads = 256;

__device__ __forceinline__ void pose_block_sums(R (&v)[28], R (*part)[28], R (&out)[28]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 28; k++) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
  }
  __syncthreads();                       // the previous reduction's readers are done with `part`
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 28; k++) part[wv][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 28; k++) out[k] = ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k];
}
__device__ __forceinline__ R pose_block_sum1(R v, R (*part)[28]) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][0] = v;
  __syncthreads();
  return ((part[0][0] + part[1][0]) + part[2][0]) + part[3][0];
}
// LL^T of the 6x6 system + both substitutions (LinearSolverDense, linear_solver_dense.h:66-109); false when not positive definite
__device__ __forceinline__ bool pose_solve6(const R (&Hs)[21], R lambda, const R (&b)[6], R (&x)[6]) {
  R L[6][6];
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = r; c < 6; c++, k++) { L[c][r] = Hs[k] + (r == c ? lambda : 0.0); }   // lower triangle
#pragma unroll
  for (int j = 0; j < 6; j++) {
    R d = L[j][j];
#pragma unroll
    for (int q = 0; q < j; q++) d -= L[j][q] * L[j][q];
    if (!(d > 0)) return false;
    d = sqrt(d);
    L[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      R s2 = L[i][j];
#pragma unroll
      for (int q = 0; q < j; q++) s2 -= L[i][q] * L[j][q];
      L[i][j] = s2 / d;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    R s2 = b[i];
#pragma unroll
    for (int q = 0; q < i; q++) s2 -= L[i][q] * x[q];
    x[i] = s2 / L[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    R s2 = x[i];
#pragma unroll
    for (int q = i + 1; q < 6; q++) s2 -= L[q][i] * x[q];
    x[i] = s2 / L[i][i];
  }
  return true;
}

// state per edge in global scratch: err[3] (as last computed, stale for inactive edges like g2o's _error), flags (bit0 outlier/level,
// bit1 robust kernel off)
__global__ __launch_bounds__(kPoseThreads) void k_pose_optimize(int nFrames, const int* __restrict__ edgeStart, R* __restrict__ poses,
                                                                const R* __restrict__ Xw, const R* __restrict__ meas,
                                                                const R* __restrict__ info, Cam cam, R deltaMono, R deltaStereo,
                                                                R* __restrict__ err, uint8_t* __restrict__ flags,
                                                                uint8_t* __restrict__ outlier, int* __restrict__ nInliers,
                                                                R* __restrict__ chi2Log, int* __restrict__ trialsOut) {
  __shared__ R part[4][28];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= nFrames) return;
  const int e0 = edgeStart[f], E = edgeStart[f + 1] - e0;
  for (int k = tid; k < 4; k += kPoseThreads) chi2Log[4 * f + k] = __longlong_as_double(0x7ff8000000000000ll);
  if (E < 3) {                               // :443-445 (m_v_isOutliers of the few correspondences was already cleared at :392)
    for (int i = tid; i < E; i += kPoseThreads) outlier[e0 + i] = 0;
    if (tid == 0) { nInliers[f] = 0; trialsOut[f] = 0; }
    return;
  }
  V3 t0; Q4 q0;
  load_pose(poses, f, t0, q0);
  qnormalize(q0);
  for (int i = tid; i < E; i += kPoseThreads) { flags[e0 + i] = 0; outlier[e0 + i] = 0; }
  __syncthreads();
  V3 t = t0; Q4 q = q0;
  int bad = 0, trials = 0;
  for (int epi = 0; epi < 4; epi++) {
    t = t0; q = q0;                          // setEstimate(frame pose) at the top of every episode (:455)
    int nAct = 0;
    for (int i = tid; i < E; i += kPoseThreads) nAct += !(flags[e0 + i] & 1);
    nAct = (int)pose_block_sum1((R)nAct, part);
    if (nAct > 0) {
      R lambda = 0, ni = 2, currentChi = 0;
      for (int it = 0; it < 10; it++) {
        // computeActiveErrors + robust chi2 + buildSystem in one pass
        R acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0;
        for (int i = tid; i < E; i += kPoseThreads) {
          const int e = e0 + i;
          const uint8_t fl = flags[e];
          if (fl & 1) continue;
          const R* z = meas + 3 * e;
          const bool st = z[2] >= 0;
          const V3 X{Xw[3 * e], Xw[3 * e + 1], Xw[3 * e + 2]};
          R r[3], depth;
          residual(t, q, X, z, st, cam, r, &depth);
          err[3 * e] = r[0]; err[3 * e + 1] = r[1]; err[3 * e + 2] = r[2];
          const R w = info[e], c2 = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
          R rho0 = c2, rho1 = 1;
          if (!(fl & 2)) huber(c2, st ? deltaStereo : deltaMono, &rho0, &rho1);
          acc[27] += rho0;
          const V3 p = add(qrot(q, X), t);
          const R x = p.x, y = p.y, invz = 1.0 / p.z, invz2 = invz * invz;
          R J[3][6];
          J[0][0] = x * y * invz2 * cam.fx; J[0][1] = -(1 + (x * x * invz2)) * cam.fx; J[0][2] = y * invz * cam.fx;
          J[0][3] = -invz * cam.fx; J[0][4] = 0; J[0][5] = x * invz2 * cam.fx;
          J[1][0] = (1 + y * y * invz2) * cam.fy; J[1][1] = -x * y * invz2 * cam.fy; J[1][2] = -x * invz * cam.fy;
          J[1][3] = 0; J[1][4] = -invz * cam.fy; J[1][5] = y * invz2 * cam.fy;
          if (st) {
            J[2][0] = J[0][0] - cam.bf * y * invz2; J[2][1] = J[0][1] + cam.bf * x * invz2; J[2][2] = J[0][2];
            J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - cam.bf * invz2;
          } else {
#pragma unroll
            for (int c = 0; c < 6; c++) J[2][c] = 0;
          }
          const R W = rho1 * w;
          const R omr[3] = {-w * r[0] * rho1, -w * r[1] * rho1, st ? -w * r[2] * rho1 : 0.0};
          int k = 0;
#pragma unroll
          for (int rr = 0; rr < 6; rr++)
#pragma unroll
            for (int c = rr; c < 6; c++, k++) acc[k] += J[0][rr] * W * J[0][c] + J[1][rr] * W * J[1][c] + J[2][rr] * W * J[2][c];
#pragma unroll
          for (int rr = 0; rr < 6; rr++) acc[21 + rr] += J[0][rr] * omr[0] + J[1][rr] * omr[1] + J[2][rr] * omr[2];
        }
        R S[28];
        pose_block_sums(acc, part, S);
        R Hs[21], b[6];
#pragma unroll
        for (int k = 0; k < 21; k++) Hs[k] = S[k];
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = S[21 + k];
        currentChi = S[27];
        if (it == 0) {                       // computeLambdaInit
          R mx = 0;
          int k = 0;
#pragma unroll
          for (int rr = 0; rr < 6; rr++) { mx = fmax(fabs(Hs[k]), mx); k += 6 - rr; }
          lambda = 1e-5 * mx; ni = 2;
        }
        R rho = 0, x6[6] = {0, 0, 0, 0, 0, 0};
        int qmax = 0;
        do {
          const V3 tb = t; const Q4 qb = q;  // push()
          const bool ok = pose_solve6(Hs, lambda, b, x6);
          pose_oplus(t, q, x6);              // g2o applies _x even after a failed solve (it then holds the previous step)
          R chi = 0;
          for (int i = tid; i < E; i += kPoseThreads) {
            const int e = e0 + i;
            const uint8_t fl = flags[e];
            if (fl & 1) continue;
            const R* z = meas + 3 * e;
            const bool st = z[2] >= 0;
            R r[3], depth;
            residual(t, q, V3{Xw[3 * e], Xw[3 * e + 1], Xw[3 * e + 2]}, z, st, cam, r, &depth);
            err[3 * e] = r[0]; err[3 * e + 1] = r[1]; err[3 * e + 2] = r[2];
            const R c2 = info[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            R rho0 = c2, rho1;
            if (!(fl & 2)) huber(c2, st ? deltaStereo : deltaMono, &rho0, &rho1);
            chi += rho0;
          }
          R tempChi = pose_block_sum1(chi, part);
          if (!ok) tempChi = 1.7976931348623157e308;
          rho = currentChi - tempChi;
          R sc = 1e-3;
#pragma unroll
          for (int j = 0; j < 6; j++) sc += x6[j] * (lambda * x6[j] + b[j]);
          rho /= sc;
          if (rho > 0 && isfinite(tempChi)) {
            R alpha = 1. - pow((2 * rho - 1), 3.0);
            alpha = fmin(alpha, 2. / 3.);
            lambda *= fmax(1. / 3., alpha);
            ni = 2;
            currentChi = tempChi;
          } else {
            lambda *= ni; ni *= 2;
            t = tb; q = qb;                  // pop()
            if (!isfinite(lambda)) { qmax++; trials++; break; }
          }
          qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;
      }
      if (tid == 0) chi2Log[4 * f + epi] = currentChi;
    }
    // classification (:458-493): excluded edges get a fresh error with the episode's final pose, active ones keep the last evaluated
    int myBad = 0;
    for (int i = tid; i < E; i += kPoseThreads) {
      const int e = e0 + i;
      uint8_t fl = flags[e];
      const R* z = meas + 3 * e;
      const bool st = z[2] >= 0;
      if (fl & 1) {
        R r[3], depth;
        residual(t, q, V3{Xw[3 * e], Xw[3 * e + 1], Xw[3 * e + 2]}, z, st, cam, r, &depth);
        err[3 * e] = r[0]; err[3 * e + 1] = r[1]; err[3 * e + 2] = r[2];
      }
      const float c2 = (float)(info[e] * (err[3 * e] * err[3 * e] + err[3 * e + 1] * err[3 * e + 1] + err[3 * e + 2] * err[3 * e + 2]));
      const bool isBad = c2 > (st ? 7.815f : 5.991f);
      fl = (uint8_t)((fl & 2) | (isBad ? 1 : 0));
      if (epi == 2) fl |= 2;                 // setRobustKernel(0) before the last episode
      flags[e] = fl;
      outlier[e] = isBad ? 1 : 0;
      myBad += isBad;
    }
    bad = (int)pose_block_sum1((R)myBad, part);
    __syncthreads();                         // flags / err of this episode are visible to the next one
    if (E < 10) break;                       // optimizer.edges().size() < 10 (:494)
  }
  if (tid == 0) {
    R* o = poses + 7 * f;
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    nInliers[f] = E - bad;
    trialsOut[f] = trials;
  }
}

// ---------------------------------------------------------------------------------------------------
// Lock-step batch of independent problems (ydorb_ba_solve_batch): the same kernel bodies, one launch per phase for ALL problems
// of the batch, blockIdx.z = problem.  A latency chain of ~35 small launches per LM trial leaves the GPU nearly idle; B chains in
// one set of launches cost about the same wall time as one.  Every problem keeps its own buffers (BaDev = its pointers, sizes and
// the scalars of the trial at hand); grids are sized for the largest problem and workgroups beyond a problem's extent - or of a
// problem that sits this phase out (finished, or waiting while others retry a rejected step) - leave at once.  The arithmetic and
// its order are those of the single-problem launches, so a batched solve is bit-identical to its own single solve.
// ---------------------------------------------------------------------------------------------------
// Every buffer is named by its BYTE OFFSET from the BaDev array itself, not by a pointer: a pointer loaded from memory is a FLAT
// pointer to the compiler (flat_load / flat_store, which count on the LDS counter too and serialise with the LDS traffic of the
// Cholesky and Schur kernels), whereas `kernel argument + offset` keeps the global address space.
struct BaDev {
  long long ePose, ePidx, ePt, eMeas, eInfo, eRobust;   // the EdgeSoA arrays
  long long ptStart, poseStart, poseEdges, eLm, poseOf, ptOf, pairStart, pairItems;
  long long poses[2], pts[2];
  long long err, partial, Hll, bl, Hpl, BD, Hpp, S, diagL, diagInv, Dinv, db, xp, yv, xl, scal, status;
  Cam cam;
  R dM, dSt;
  int nL, nPf, Ea, n, nb, nBlkE, nBuckets;
  // the round at hand (rewritten by the host before every round)
  R lambda;
  int cur;        // which of poses[2] / pts[2] holds the current estimate
  int build;      // takes part in the chi2 / build-system launches of this round
  int chi2;       // ... and needs the errors of the current estimate recomputed first (first iteration, or after a rejected NaN step)
  int maxdiag;    // ... and its initial lambda (first iteration of an optimize() call)
  int trial;      // takes part in the trial launches of this round
};
#define YD_BA_AT(T, off) (reinterpret_cast<T*>(const_cast<char*>(reinterpret_cast<const char*>(all)) + (off)))
#define YD_BA_ED EdgeSoA{YD_BA_AT(const int, D.ePose), YD_BA_AT(const int, D.ePidx), YD_BA_AT(const int, D.ePt), YD_BA_AT(const R, D.eMeas), \
                         YD_BA_AT(const R, D.eInfo), YD_BA_AT(const uint8_t, D.eRobust), D.Ea}
#define YD_BA_PROB(flag)                         \
  const BaDev& D = all[blockIdx.z];              \
  if (!D.flag) return
__global__ __launch_bounds__(256) void kb_errors(const BaDev* __restrict__ all, int onTrial) {
  const BaDev& D = all[blockIdx.z];
  if (!(onTrial ? D.trial : (D.build && D.chi2)) || blockIdx.x >= (unsigned)D.nBlkE) return;
  const int buf = onTrial ? (D.cur ^ 1) : D.cur;
  b_errors(YD_BA_ED, YD_BA_AT(R, D.poses[buf]), YD_BA_AT(R, D.pts[buf]), D.cam, D.dM, D.dSt, YD_BA_AT(R, D.err), YD_BA_AT(R, D.partial));
}
// use 0: chi2 of the current estimate (no status), 1: chi2 of the trial estimate (+ factorisation status), 2: the scale sum
__global__ __launch_bounds__(256) void kb_sum_partials(const BaDev* __restrict__ all, int use) {
  const BaDev& D = all[blockIdx.z];
  if (!(use == 0 ? (D.build && D.chi2) : D.trial)) return;
  if (use == 2) b_sum_partials(YD_BA_AT(R, D.partial) + D.nBlkE, (6 * D.nPf + 3 * D.nL + 255) / 256, YD_BA_AT(R, D.scal), 2, nullptr);
  else b_sum_partials(YD_BA_AT(R, D.partial), D.nBlkE, YD_BA_AT(R, D.scal), 0, use == 1 ? YD_BA_AT(int, D.status) : nullptr);
}
__global__ __launch_bounds__(128) void kb_build_points(const BaDev* __restrict__ all) {
  YD_BA_PROB(build);
  if (blockIdx.x >= (unsigned)((D.nL + 127) / 128)) return;
  b_build_points(YD_BA_ED, YD_BA_AT(const int, D.ptStart), D.nL, YD_BA_AT(R, D.poses[D.cur]), YD_BA_AT(R, D.pts[D.cur]), D.cam, D.dM, D.dSt, YD_BA_AT(R, D.err), YD_BA_AT(R, D.Hll), YD_BA_AT(R, D.bl), YD_BA_AT(R, D.Hpl));
}
__global__ __launch_bounds__(256) void kb_build_poses(const BaDev* __restrict__ all) {
  YD_BA_PROB(build);
  if (blockIdx.x >= (unsigned)D.nPf) return;
  b_build_poses(YD_BA_ED, YD_BA_AT(const int, D.poseStart), YD_BA_AT(const int, D.poseEdges), YD_BA_AT(R, D.poses[D.cur]), YD_BA_AT(R, D.pts[D.cur]), D.cam, D.dM, D.dSt, YD_BA_AT(R, D.err), YD_BA_AT(R, D.Hpp), YD_BA_AT(R, D.Hpp) + (size_t)36 * D.nPf);
}
__global__ __launch_bounds__(256) void kb_max_diag(const BaDev* __restrict__ all) {
  const BaDev& D = all[blockIdx.z];
  if (!(D.build && D.maxdiag)) return;
  b_max_diag(YD_BA_AT(R, D.Hpp), D.nPf, YD_BA_AT(R, D.Hll), D.nL, YD_BA_AT(R, D.scal), 1);
}
__global__ __launch_bounds__(256) void kb_dinv(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)((D.nL + 255) / 256)) return;
  b_dinv(YD_BA_AT(R, D.Hll), YD_BA_AT(R, D.bl), D.nL, D.lambda, YD_BA_AT(R, D.Dinv), YD_BA_AT(R, D.db), YD_BA_AT(int, D.status));
}
__global__ __launch_bounds__(256) void kb_bd(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)D.nBlkE) return;
  b_bd(YD_BA_ED, YD_BA_AT(const int, D.eLm), YD_BA_AT(R, D.Hpl), YD_BA_AT(R, D.Dinv), YD_BA_AT(R, D.BD));
}
__global__ __launch_bounds__(256) void kb_bs(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)D.nPf) return;
  b_bs(YD_BA_ED, YD_BA_AT(const int, D.poseStart), YD_BA_AT(const int, D.poseEdges), YD_BA_AT(const int, D.eLm), YD_BA_AT(R, D.Hpl), YD_BA_AT(R, D.db), YD_BA_AT(R, D.Hpp) + (size_t)36 * D.nPf, 1.0, YD_BA_AT(R, D.S) + (size_t)D.n * D.n);
}
__global__ __launch_bounds__(64 * kSchurWaves) void kb_schur_pairs(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)(D.nBuckets + 1)) return;
  b_schur_pairs(YD_BA_AT(const int, D.pairStart), YD_BA_AT(const int2, D.pairItems), D.nPf, D.nBuckets, YD_BA_AT(R, D.BD), YD_BA_AT(R, D.Hpl), YD_BA_AT(R, D.Hpp), D.lambda, 1.0, D.n, YD_BA_AT(R, D.S), YD_BA_AT(R, D.S) + (size_t)D.n * D.n);
}
__global__ __launch_bounds__(256) void kb_chol_step(const BaDev* __restrict__ all, int kb) {
  YD_BA_PROB(trial);
  if (kb >= D.nb || blockIdx.x >= (unsigned)((D.nb - kb) * (D.nb - kb + 1) / 2 + (kb > 0))) return;
  b_chol_step(YD_BA_AT(R, D.S), YD_BA_AT(R, D.diagL), YD_BA_AT(R, D.diagInv), D.n, kb, YD_BA_AT(int, D.status), YD_BA_AT(R, D.S) + (size_t)D.n * D.n, YD_BA_AT(R, D.yv));
}
__global__ __launch_bounds__(1024) void kb_chol_solve(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  b_chol_solve(YD_BA_AT(R, D.S), YD_BA_AT(R, D.diagInv), D.n, YD_BA_AT(R, D.yv), YD_BA_AT(R, D.S) + (size_t)D.n * D.n, YD_BA_AT(R, D.xp));
}
__global__ __launch_bounds__(128) void kb_backsub(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)((D.nL + 127) / 128)) return;
  b_backsub(YD_BA_ED, YD_BA_AT(const int, D.ptStart), D.nL, YD_BA_AT(R, D.Hpl), YD_BA_AT(R, D.Dinv), YD_BA_AT(R, D.bl), YD_BA_AT(R, D.xp), YD_BA_AT(R, D.xl));
}
__global__ __launch_bounds__(256) void kb_update(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)((max(D.nPf, D.nL) + 255) / 256)) return;
  b_update(YD_BA_AT(R, D.poses[D.cur]), YD_BA_AT(R, D.pts[D.cur]), YD_BA_AT(R, D.poses[D.cur ^ 1]), YD_BA_AT(R, D.pts[D.cur ^ 1]), YD_BA_AT(const int, D.poseOf), D.nPf, YD_BA_AT(const int, D.ptOf), D.nL, YD_BA_AT(R, D.xp), YD_BA_AT(R, D.xl));
}
__global__ __launch_bounds__(256) void kb_scale(const BaDev* __restrict__ all) {
  YD_BA_PROB(trial);
  if (blockIdx.x >= (unsigned)((6 * D.nPf + 3 * D.nL + 255) / 256)) return;
  b_scale(YD_BA_AT(R, D.xp), YD_BA_AT(R, D.Hpp) + (size_t)36 * D.nPf, 6 * D.nPf, YD_BA_AT(R, D.xl), YD_BA_AT(R, D.bl), 3 * D.nL, D.lambda, YD_BA_AT(R, D.partial) + D.nBlkE);
}
#undef YD_BA_PROB
#undef YD_BA_ED
#undef YD_BA_AT

}  // namespace ba
}  // namespace ydorb
