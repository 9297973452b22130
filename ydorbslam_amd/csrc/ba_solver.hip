// Host driver of the MI355X local-BA back-end + its C ABI (include/ydorb/c_api.h, "Local bundle adjustment").
// Sequencing restated from the reference: two-stage schedule of Optimizer::localBundleAdjust
// (src/optimizer.cpp:284-334), SparseOptimizer::initializeOptimization/optimize
// (thirdParty/g2o/g2o/core/sparse_optimizer.cpp:208-280, 366-440) and the Levenberg-Marquardt trial loop
// (core/optimization_algorithm_levenberg.cpp:57-173).  All arithmetic on the graph runs in ba_kernels.hip.h;
// the host only sorts the edge list, keeps lambda / nu, and reads three scalars per trial.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <functional>
#include <string>
#include <thread>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>
#include <vector>

#include "../../include/ydorb/c_api.h"
#include "ba_kernels.hip.h"
#include "ydorb_host.h"

using namespace ydorb;
using namespace ydorb::ba;

namespace {

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ydorb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return YDORB_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool view = false;   // a slice of another DBuf (Ctx::upArena): not owned
  void setView(void* ptr, size_t bytes) { if (!view && p) (void)hipFree(p); p = ptr; cap = bytes; view = true; }
  int ensure(size_t bytes) {
    if (bytes <= cap && !view) return YDORB_OK;
    if (p && !view) (void)hipFree(p);
    view = false;
    p = nullptr; cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    if (hipMalloc(&p, want) != hipSuccess) { set_error("hipMalloc(%zu) failed", want); return YDORB_ERR_HIP; }
    cap = want;
    return YDORB_OK;
  }
  template <class T> T* as() { return reinterpret_cast<T*>(p); }
  void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
};

enum { PH_ERR = 0, PH_BUILD, PH_SCHUR, PH_SOLVE, PH_UPDATE, PH_COUNT };

struct Ctx {  // per-device scratch, reused across calls (localBundleAdjust runs on one thread, localMapping.cpp:29)
  int device = -1;
  hipStream_t stream = nullptr;
  DBuf poses[2], pts[2];
  DBuf upArena;   // the ordered edge arrays and index lists of a stage, laid out like the pinned staging area: ONE upload per stage; the twelve buffers
                  // below that the host fills (ePose .. ptOf except eInfo0 / eOutlier) are views into it
  DBuf ePose, ePidx, ePt, eMeas, eInfo, eInfo0, eRobust, eOutlier, eLm, ptStart, poseStart, poseEdges, poseOf, ptOf;
  DBuf err, partial, Hll, bl, Hpl, BD, Hpp, bp, S, diagL, diagInv, bs, Dinv, db, xp, xl, yv, scal, status, pairCnt, pairStart, pairCursor, pairA, pairB;
  hipEvent_t ev[2 * PH_COUNT + 2]{};
  bool evInit = false;
  double* hPin = nullptr;   // pinned read-back area: scal[8] + status[2] (one stream sync per LM trial)
  uint8_t* hStage = nullptr;   // pinned staging of the ordered edge arrays on their way up and of the results on their way down: an
  size_t hStageCap = 0;        // asynchronous copy out of / into pageable memory runs at ~8 GB/s and makes the host wait for the stream
  int stage(size_t bytes) {
    if (bytes <= hStageCap) return YDORB_OK;
    if (hStage) { (void)hipHostFree(hStage); hStage = nullptr; hStageCap = 0; }
    const size_t cap = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&hStage, cap) != hipSuccess) { hStage = nullptr; ydorb::set_error("hipHostMalloc(%zu) failed", cap); return YDORB_ERR_HIP; }
    hStageCap = cap;
    return YDORB_OK;
  }
  DBuf pStart, pPoses, pX, pMeas, pInfo, pErr, pFlags, pOutlier, pInl, pChi, pTrials;   // pose-only batches
  void releaseBuffers() {   // ydorb_ba_release: device scratch and pinned staging back to the system (stream and events stay)
    for (DBuf* b : {&poses[0], &poses[1], &pts[0], &pts[1], &upArena, &ePose, &ePidx, &ePt, &eMeas, &eInfo, &eInfo0, &eRobust, &eOutlier, &eLm, &ptStart,
                    &poseStart, &poseEdges, &poseOf, &ptOf, &err, &partial, &Hll, &bl, &Hpl, &BD, &Hpp, &bp, &S, &diagL, &diagInv, &bs, &Dinv, &db,
                    &xp, &xl, &yv, &scal, &status, &pairCnt, &pairStart, &pairCursor, &pairA, &pairB, &pStart, &pPoses, &pX, &pMeas, &pInfo, &pErr,
                    &pFlags, &pOutlier, &pInl, &pChi, &pTrials})
      b->release();
    if (hStage) { (void)hipHostFree(hStage); hStage = nullptr; hStageCap = 0; }
  }
};
// A small pool of contexts per device: one localBundleAdjust at a time is the reference's use (LocalMapping thread), but the solve
// is a latency chain that leaves most of the GPU idle, so several host threads (several maps / sessions) may solve concurrently,
// each on its own stream and scratch.
constexpr int kCtxPool = 8;
std::mutex g_mu[16][kCtxPool];
Ctx g_ctx[16][kCtxPool];
std::mutex g_pick;

struct Run {
  Ctx* c;
  const YdBaProblem* P;
  const YdBaOptions* O;
  YdBaResult* res;
  int cur = 0;  // which of poses[2]/pts[2] holds the current estimate
  bool noRobust = false;   // YDORB_BA_NO_ROBUST
  Cam cam;
  bool phaseTimes = false;   // YDORB_BA_PHASE_TIMES
  double phaseMs[PH_COUNT] = {0, 0, 0, 0, 0};
  bool pending[PH_COUNT] = {false, false, false, false, false};
  struct Sys { std::vector<int> act; int nL = 0, nPf = 0, Ea = 0, n = 0, nb = 0, nBlkE = 0, nBuckets = 0; } sys;   // stage 1's system, re-used by stage 2
  bool stopped() const { return P->stop && *P->stop; }
};

int allreduce(Run& R_, void* d_buf, int64_t count, int op) {
  const YdBaOptions* O = R_.O;
  if (!O->allreduce || O->world <= 1) return YDORB_OK;
  if (count > O->comm_doubles || !O->d_comm_buf) { set_error("BA comm buffer too small (%lld doubles needed)", (long long)count); return YDORB_ERR_CAPACITY; }
  HIPCHK(hipMemcpyAsync(O->d_comm_buf, d_buf, sizeof(double) * count, hipMemcpyDeviceToDevice, R_.c->stream));
  HIPCHK(hipStreamSynchronize(R_.c->stream));
  if (O->allreduce(O->allreduce_user, O->d_comm_buf, count, op) != 0) { set_error("BA all-reduce callback failed"); return YDORB_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(d_buf, O->d_comm_buf, sizeof(double) * count, hipMemcpyDeviceToDevice, R_.c->stream));
  return YDORB_OK;
}

// YDORB_BA_TRACE=1: host wall-clock marks of the solve on stderr (where the time between device phases goes)
static bool g_trace = getenv("YDORB_BA_TRACE") != nullptr && getenv("YDORB_BA_TRACE")[0] == '1';
static std::chrono::steady_clock::time_point g_t0;
static void trace(const char* what) {
  if (!g_trace) return;
  fprintf(stderr, "[ba %8.3f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0).count(), what);
}

struct PhaseTimer {
  Run& r; int ph; hipEvent_t a, b; bool on;
  PhaseTimer(Run& r_, int ph_) : r(r_), ph(ph_), a(r_.c->ev[2 * ph_]), b(r_.c->ev[2 * ph_ + 1]), on(r_.phaseTimes) {
    if (!on) return;   // YDORB_BA_PHASE_TIMES not asked for: no events on the stream
    collect(r_, ph_);  // an earlier recording of this phase's events must be read before they are re-recorded
    (void)hipEventRecord(a, r.c->stream);
  }
  void stop() { if (on) { (void)hipEventRecord(b, r.c->stream); on = false; r.pending[ph] = true; } }
  static void collect(Run& r_, int ph_) {
    if (!r_.pending[ph_]) return;
    float ms = 0;
    if (hipEventSynchronize(r_.c->ev[2 * ph_ + 1]) == hipSuccess && hipEventElapsedTime(&ms, r_.c->ev[2 * ph_], r_.c->ev[2 * ph_ + 1]) == hipSuccess)
      r_.phaseMs[ph_] += ms;
    r_.pending[ph_] = false;
  }
};

// initializeOptimization(0) for one optimize() call: host ordering of the level-0 edges, uploads, pose-pair buckets (stage 1), or the
// refreshed information / robust flags on stage 1's structures (stage 2); leaves S cleared and both estimate buffers in agreement.
// R_.sys.Ea == 0 afterwards means there is nothing to optimise.
int prepareStage(Run& R_, bool reuse) {
  Ctx& c = *R_.c;
  const YdBaProblem& P = *R_.P;
  const YdBaOptions& O = *R_.O;
  hipStream_t s = c.stream;
  const int K = P.n_poses, NP = P.n_points, E = P.n_edges;
  trace("optimize: begin");
  Run::Sys& Y = R_.sys;
  // Stage 2 (after the chi2 cull) re-uses stage 1's device structures: a culled edge keeps its slot with information 0, which
  // every kernel treats as "skip" — no second host ordering, upload or pair-bucket build.  Landmarks / poses that lose all their
  // edges stay in the system with H = lambda*I, b = 0, i.e. dx = 0: the same estimate g2o keeps by leaving them out.
  auto prepare = [&]() -> int {
    std::vector<int> act;
    for (int e = 0; e < E; e++) act.push_back(e);   // every edge starts at level 0 (optimizer.cpp:239-280)
    if (act.empty()) { Y.Ea = 0; return YDORB_OK; }
    // index mapping (buildIndexMapping, sparse_optimizer.cpp:168-192): free poses first, then landmarks, active ones only
    std::vector<int> poseIdx(K, -1), ptIdx(NP, -1), poseOf, ptOf;
    {
      std::vector<uint8_t> pu(K, 0), qu(NP, 0);
      for (int e : act) { pu[P.edge_pose[e]] = 1; qu[P.edge_point[e]] = 1; }
      for (int k = 0; k < K; k++) if (pu[k] && !P.pose_fixed[k]) { poseIdx[k] = (int)poseOf.size(); poseOf.push_back(k); }
      for (int p = 0; p < NP; p++) if (qu[p]) { ptIdx[p] = (int)ptOf.size(); ptOf.push_back(p); }
    }
    const int nP = (int)poseOf.size(), nL = (int)ptOf.size(), Ea = (int)act.size();
    if (O.world > 1) {
      // every rank must factorise the same reduced system: the free-pose set comes from the caller's fixed mask only
      poseOf.clear();
      for (int k = 0; k < K; k++) { poseIdx[k] = -1; if (!P.pose_fixed[k]) { poseIdx[k] = (int)poseOf.size(); poseOf.push_back(k); } }
    }
    const int nPf = (int)poseOf.size();
    (void)nP;
    {  // order by (landmark index, pose index, edge index): counting sort by landmark, then a tiny insertion sort per landmark
      std::vector<int> start(nL + 1, 0), sorted(Ea);
      for (int e : act) start[ptIdx[P.edge_point[e]] + 1]++;
      for (int l = 0; l < nL; l++) start[l + 1] += start[l];
      std::vector<int> fill(start.begin(), start.end() - 1);
      for (int e : act) sorted[fill[ptIdx[P.edge_point[e]]]++] = e;   // act is ascending: stable
      for (int l = 0; l < nL; l++)
        for (int a = start[l] + 1; a < start[l + 1]; a++) {
          const int e = sorted[a], pe = poseIdx[P.edge_pose[e]];
          int b = a - 1;
          while (b >= start[l] && (poseIdx[P.edge_pose[sorted[b]]] > pe || (poseIdx[P.edge_pose[sorted[b]]] == pe && sorted[b] > e))) { sorted[b + 1] = sorted[b]; b--; }
          sorted[b + 1] = e;
        }
      act.swap(sorted);
    }
    std::vector<int> hPose(Ea), hPidx(Ea), hPt(Ea), hLm(Ea), hPtStart(nL + 1, 0), hPoseStart(nPf + 1, 0), hPoseEdges;
    std::vector<double> hMeas((size_t)3 * Ea), hInfo(Ea);
    std::vector<uint8_t> hRobust(Ea);
    for (int i = 0; i < Ea; i++) {
      const int e = act[i];
      hPose[i] = P.edge_pose[e]; hPidx[i] = poseIdx[P.edge_pose[e]]; hPt[i] = P.edge_point[e]; hLm[i] = ptIdx[P.edge_point[e]];
      for (int d = 0; d < 3; d++) hMeas[3 * i + d] = P.edge_meas[3 * e + d];
      hInfo[i] = P.edge_inv_sigma2[e];
      hRobust[i] = R_.noRobust ? 0 : 1;
      hPtStart[hLm[i] + 1]++;
      if (hPidx[i] >= 0) hPoseStart[hPidx[i] + 1]++;
    }
    for (int l = 0; l < nL; l++) hPtStart[l + 1] += hPtStart[l];
    for (int i = 0; i < nPf; i++) hPoseStart[i + 1] += hPoseStart[i];
    hPoseEdges.resize(hPoseStart[nPf]);
    {
      std::vector<int> fill(hPoseStart.begin(), hPoseStart.end() - 1);
      for (int i = 0; i < Ea; i++) if (hPidx[i] >= 0) hPoseEdges[fill[hPidx[i]]++] = i;
    }
    const int n = std::max(NB, (6 * nPf + NB - 1) / NB * NB), nb = n / NB;
    const int nBlkE = (Ea + 255) / 256;
    int rc;
    if ((rc = c.eInfo0.ensure(sizeof(double) * Ea)) || (rc = c.eOutlier.ensure(Ea)) || (rc = c.err.ensure(sizeof(double) * 3 * Ea)) || (rc = c.partial.ensure(sizeof(double) * (nBlkE + (6 * nPf + 3 * nL + 255) / 256 + 1))) ||
        (rc = c.Hll.ensure(sizeof(double) * 6 * nL)) || (rc = c.bl.ensure(sizeof(double) * 3 * nL)) || (rc = c.Hpl.ensure(sizeof(double) * 18 * (size_t)Ea)) ||
        (rc = c.BD.ensure(sizeof(double) * 18 * (size_t)Ea)) ||
        (rc = c.Hpp.ensure(sizeof(double) * 42 * std::max(nPf, 1))) || (rc = c.S.ensure(sizeof(double) * ((size_t)n * n + n))) ||
        (rc = c.diagL.ensure(sizeof(double) * (size_t)nb * NB * NB)) || (rc = c.diagInv.ensure(sizeof(double) * (size_t)nb * NB * NB)) || (rc = c.Dinv.ensure(sizeof(double) * 6 * nL)) || (rc = c.db.ensure(sizeof(double) * 3 * nL)) ||
        (rc = c.xp.ensure(sizeof(double) * n)) || (rc = c.yv.ensure(sizeof(double) * n)) || (rc = c.xl.ensure(sizeof(double) * 3 * nL)) || (rc = c.scal.ensure(sizeof(double) * 8)) ||
        (rc = c.status.ensure(sizeof(int) * 2)))
      return rc;
    // Hpp and bp are contiguous ([36 nPf | 6 nPf]) so one all-reduce covers both; bs follows S for the same reason
    trace("optimize: host ordering done");
    {  // through the context's pinned staging area (true asynchronous copies at PCIe rate; the area is free again at the sync below)
      auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
      const size_t total = 4 * al(sizeof(int) * Ea) + al(sizeof(double) * 3 * Ea) + al(sizeof(double) * Ea) + al(Ea) + al(sizeof(int) * (nL + 1)) +
                           al(sizeof(int) * (nPf + 1)) + al(sizeof(int) * ptOf.size()) + al(sizeof(int) * hPoseEdges.size()) + al(sizeof(int) * poseOf.size());
      if ((rc = c.stage(total)) || (rc = c.upArena.ensure(total))) return rc;
      size_t off = 0;
      // every array is copied into the staging area and its device buffer becomes the slice of the arena at the same offset; ONE
      // hipMemcpyAsync then moves the stage's inputs (twelve calls before: with 16 set-up threads of a batch the runtime's lock was the cost)
      auto up = [&](DBuf& dst, const void* src, size_t bytes) {
        if (bytes) memcpy(c.hStage + off, src, bytes);
        dst.setView(reinterpret_cast<uint8_t*>(c.upArena.p) + off, bytes);
        off += al(bytes);
      };
  #define UP(buf, vec, T) up(c.buf, vec.data(), sizeof(T) * vec.size())
      UP(ePose, hPose, int); UP(ePidx, hPidx, int); UP(ePt, hPt, int); UP(eLm, hLm, int); UP(eMeas, hMeas, double); UP(eInfo, hInfo, double);
      UP(eRobust, hRobust, uint8_t); UP(ptStart, hPtStart, int); UP(poseStart, hPoseStart, int); UP(ptOf, ptOf, int);
      UP(poseEdges, hPoseEdges, int); UP(poseOf, poseOf, int);
  #undef UP
      if (off) HIPCHK(hipMemcpyAsync(c.upArena.p, c.hStage, off, hipMemcpyHostToDevice, s));
      // the original information: the chi2 tests between and after the stages use it (k_cull)
      HIPCHK(hipMemcpyAsync(c.eInfo0.p, c.eInfo.p, sizeof(double) * Ea, hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemsetAsync(c.err.p, 0, sizeof(double) * 3 * Ea, s));   // an edge that is never evaluated (stop flag) has error 0
    }
    trace("optimize: uploads enqueued");
    // pose-pair buckets of the Schur complement (structure is fixed for this optimize() call)
    const int nBuckets = nPf * (nPf + 1) / 2;
    size_t nItems = 0;
    for (int l = 0; l < nL; l++) {
      size_t m = 0;
      for (int i = hPtStart[l]; i < hPtStart[l + 1]; i++) m += hPidx[i] >= 0;
      nItems += m * (m + 1) / 2;
    }
    if ((rc = c.pairCnt.ensure(sizeof(int) * (nBuckets + 1))) || (rc = c.pairStart.ensure(sizeof(int) * (nBuckets + 1))) ||
        (rc = c.pairCursor.ensure(sizeof(int) * (nBuckets + 1))) || (rc = c.pairA.ensure(sizeof(int2) * std::max<size_t>(nItems, 1))) ||
        (rc = c.pairB.ensure(sizeof(int2) * std::max<size_t>(nItems, 1))))
      return rc;
    if (nBuckets > 0) {
      EdgeSoA Ed{c.ePose.as<int>(), c.ePidx.as<int>(), c.ePt.as<int>(), c.eMeas.as<double>(), c.eInfo.as<double>(), c.eRobust.as<uint8_t>(), Ea};
      HIPCHK(hipMemsetAsync(c.pairCnt.p, 0, sizeof(int) * (nBuckets + 1), s));
      hipLaunchKernelGGL(k_pair_count, dim3((nL + 255) / 256), dim3(256), 0, s, Ed, c.ptStart.as<int>(), nL, c.pairCnt.as<int>());
      hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(256), 0, s, c.pairCnt.as<int>(), nBuckets, c.pairStart.as<int>());
      HIPCHK(hipMemcpyAsync(c.pairCursor.p, c.pairStart.p, sizeof(int) * (nBuckets + 1), hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(k_pair_fill, dim3((nL + 255) / 256), dim3(256), 0, s, Ed, c.ptStart.as<int>(), nL, c.pairCursor.as<int>(), c.pairA.as<int2>());
      hipLaunchKernelGGL(k_pair_sort, dim3((nBuckets + 3) / 4), dim3(256), 0, s, c.pairStart.as<int>(), nBuckets, c.pairA.as<int2>(), c.pairB.as<int2>());
    }
    HIPCHK(hipStreamSynchronize(s));   // the host staging vectors above die with this scope
    Y.act.swap(act); Y.nL = nL; Y.nPf = nPf; Y.Ea = Ea; Y.n = n; Y.nb = nb; Y.nBlkE = nBlkE; Y.nBuckets = nBuckets;
    return YDORB_OK;
  };
  int rc;
  if (!reuse || Y.Ea == 0) {
    if ((rc = prepare())) return rc;
  }
  // (stage 2 re-uses stage 1's device structures as they are: k_cull already zeroed the information of the culled edges and
  // cleared the robust flags on the device)
  if (Y.Ea == 0) return YDORB_OK;
  // the two estimate buffers must agree on everything the update kernel does not write (fixed poses, points without edges)
  HIPCHK(hipMemcpyAsync(c.poses[R_.cur ^ 1].p, c.poses[R_.cur].p, sizeof(double) * 7 * K, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c.pts[R_.cur ^ 1].p, c.pts[R_.cur].p, sizeof(double) * 3 * NP, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemsetAsync(c.S.p, 0, sizeof(double) * ((size_t)Y.n * Y.n + Y.n), s));
  return YDORB_OK;
}

// one SparseOptimizer::optimize(iterations) on the level-0 edges
int optimize(Run& R_, int iterations, int stage, bool reuse) {
  int rc = prepareStage(R_, reuse);
  if (rc) return rc;
  Ctx& c = *R_.c;
  const YdBaProblem& P = *R_.P;
  const YdBaOptions& O = *R_.O;
  hipStream_t s = c.stream;
  Run::Sys& Y = R_.sys;
  if (Y.Ea == 0) return YDORB_OK;
  (void)P;
  const int nL = Y.nL, nPf = Y.nPf, Ea = Y.Ea, n = Y.n, nb = Y.nb, nBlkE = Y.nBlkE, nBuckets = Y.nBuckets;
  // Hpp and bp are contiguous ([36 nPf | 6 nPf]) so one all-reduce covers both; bs follows S for the same reason
  double* dHpp = c.Hpp.as<double>();
  double* dbp = dHpp + (size_t)36 * nPf;
  double* dS = c.S.as<double>();
  double* dbs = dS + (size_t)n * n;
  EdgeSoA Ed{c.ePose.as<int>(), c.ePidx.as<int>(), c.ePt.as<int>(), c.eMeas.as<double>(), c.eInfo.as<double>(), c.eRobust.as<uint8_t>(), Ea};
  const double dM = O.delta_mono, dSt = O.delta_stereo;
  double* hscal = c.hPin;
  int* hstatus = reinterpret_cast<int*>(c.hPin + 8);
  const bool multi = O.world > 1 && O.allreduce;

  auto computeChi2 = [&](int buf, double* out, bool withStatus) -> int {
    PhaseTimer t(R_, PH_ERR);
    hipLaunchKernelGGL(k_errors, dim3(nBlkE), dim3(256), 0, s, Ed, c.poses[buf].as<double>(), c.pts[buf].as<double>(), R_.cam, dM, dSt,
                       c.err.as<double>(), c.partial.as<double>());
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, c.partial.as<double>(), nBlkE, c.scal.as<double>(), 0, withStatus ? c.status.as<int>() : nullptr);
    t.stop();
    if (multi) { int r2 = allreduce(R_, c.scal.p, 1, 0); if (r2) return r2; }
    HIPCHK(hipMemcpyAsync(hscal, c.scal.p, sizeof(double) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (withStatus) { hstatus[0] = (int)hscal[6]; hstatus[1] = (int)hscal[7]; }
    *out = hscal[0];
    return YDORB_OK;
  };

  trace("optimize: pair buckets enqueued");
  double lambda = 0, ni = 2, currentChi = 0;
  bool lastAccepted = true;
  for (int it = 0; it < iterations && !R_.stopped(); it++) {
    // computeActiveErrors + activeRobustChi2 at the top of an iteration: after the first iteration the state is the trial
    // that was just accepted, whose errors (c.err) and chi2 are already there — same kernel, same inputs, same bits.
    // A trial can also end rejected without terminating the loop (rho = NaN: `rho < 0` and `rho == 0` are both false); then err
    // and chi2 belong to the rejected state and are recomputed on the kept one, as g2o does at the top of every iteration.
    if ((it == 0 || !lastAccepted) && (rc = computeChi2(R_.cur, &currentChi, false))) return rc;
    {  // buildSystem
      PhaseTimer t(R_, PH_BUILD);
      hipLaunchKernelGGL(k_build_points, dim3((nL + 127) / 128), dim3(128), 0, s, Ed, c.ptStart.as<int>(), nL, c.poses[R_.cur].as<double>(),
                         c.pts[R_.cur].as<double>(), R_.cam, dM, dSt, c.err.as<double>(), c.Hll.as<double>(), c.bl.as<double>(), c.Hpl.as<double>());
      if (nPf)
        hipLaunchKernelGGL(k_build_poses, dim3(nPf), dim3(256), 0, s, Ed, c.poseStart.as<int>(), c.poseEdges.as<int>(), c.poses[R_.cur].as<double>(),
                           c.pts[R_.cur].as<double>(), R_.cam, dM, dSt, c.err.as<double>(), dHpp, dbp);
      t.stop();
      if (multi && nPf) { if ((rc = allreduce(R_, dHpp, (int64_t)42 * nPf, 0))) return rc; }
    }
    if (it == 0) {  // computeLambdaInit
      hipLaunchKernelGGL(k_max_diag, dim3(1), dim3(256), 0, s, dHpp, nPf, c.Hll.as<double>(), nL, c.scal.as<double>(), 1);
      if (multi) { if ((rc = allreduce(R_, c.scal.as<double>() + 1, 1, 1))) return rc; }
      HIPCHK(hipMemcpyAsync(hscal, c.scal.p, sizeof(double) * 8, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      lambda = 1e-5 * hscal[1];
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const int nxt = R_.cur ^ 1;
      {
        PhaseTimer t(R_, PH_SCHUR);
        const double contrib = (!multi || O.rank == 0) ? 1.0 : 0.0;
        hipLaunchKernelGGL(k_dinv, dim3((nL + 255) / 256), dim3(256), 0, s, c.Hll.as<double>(), c.bl.as<double>(), nL, lambda, c.Dinv.as<double>(), c.db.as<double>(), c.status.as<int>());
        hipLaunchKernelGGL(k_bd, dim3(nBlkE), dim3(256), 0, s, Ed, c.eLm.as<int>(), c.Hpl.as<double>(), c.Dinv.as<double>(), c.BD.as<double>());
        if (nPf)
          hipLaunchKernelGGL(k_bs, dim3(nPf), dim3(256), 0, s, Ed, c.poseStart.as<int>(), c.poseEdges.as<int>(), c.eLm.as<int>(), c.Hpl.as<double>(),
                             c.db.as<double>(), dbp, contrib, dbs);
        hipLaunchKernelGGL(k_schur_pairs, dim3(nBuckets + 1), dim3(64 * kSchurWaves), 0, s, c.pairStart.as<int>(), c.pairB.as<int2>(), nPf, nBuckets,
                           c.BD.as<double>(), c.Hpl.as<double>(), dHpp, lambda, contrib, n, dS, dbs);
        t.stop();
      }
      if (multi) {  // sum of the per-rank landmark contributions (+ rank 0's Hpp, lambda, bp)
        if ((rc = allreduce(R_, dS, (int64_t)n * n + n, 0))) return rc;
      }
      {
        PhaseTimer t(R_, PH_SOLVE);
        for (int kb = 0; kb < nb; kb++)
          hipLaunchKernelGGL(k_chol_step, dim3((nb - kb) * (nb - kb + 1) / 2 + (kb > 0)), dim3(256), 0, s, dS, c.diagL.as<double>(), c.diagInv.as<double>(), n, kb,
                             c.status.as<int>(), dbs, c.yv.as<double>());
        if (!launch_chol_solve(s, dS, c.diagInv.as<double>(), n, c.yv.as<double>(), dbs, c.xp.as<double>())) {
          set_error("reduced camera system of %d rows is wider than the solve kernel's LDS (max %d)", n, kCholSolveMaxN);
          return YDORB_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(k_backsub, dim3((nL + 127) / 128), dim3(128), 0, s, Ed, c.ptStart.as<int>(), nL, c.Hpl.as<double>(), c.Dinv.as<double>(),
                           c.bl.as<double>(), c.xp.as<double>(), c.xl.as<double>());
        t.stop();
        HIPCHK(hipGetLastError());
      }
      {
        PhaseTimer t(R_, PH_UPDATE);
        hipLaunchKernelGGL(k_update, dim3((std::max(nPf, nL) + 255) / 256), dim3(256), 0, s, c.poses[R_.cur].as<double>(), c.pts[R_.cur].as<double>(),
                           c.poses[nxt].as<double>(), c.pts[nxt].as<double>(), c.poseOf.as<int>(), nPf, c.ptOf.as<int>(), nL, c.xp.as<double>(),
                           c.xl.as<double>());
        // computeScale: pose part once (rank 0), landmark part per rank
        {
          const int np6 = (!multi || O.rank == 0) ? 6 * nPf : 0;
          const int nb2 = (np6 + 3 * nL + 255) / 256;
          hipLaunchKernelGGL(k_scale, dim3(nb2), dim3(256), 0, s, c.xp.as<double>(), dbp, np6, c.xl.as<double>(), c.bl.as<double>(), 3 * nL, lambda,
                             c.partial.as<double>() + nBlkE);
          hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, c.partial.as<double>() + nBlkE, nb2, c.scal.as<double>(), 2, nullptr);
        }
        t.stop();
        if (multi) { if ((rc = allreduce(R_, c.scal.as<double>() + 2, 1, 0))) return rc; }
      }
      double tempChi;
      if ((rc = computeChi2(nxt, &tempChi, true))) return rc;  // also brings back scal[2], the factorisation status, and leaves err = errors of the trial state
      const double scaleSum = hscal[2];
      const bool ok2 = hstatus[0] == 0;
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = scaleSum + 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        R_.cur = nxt;  // discardTop(): keep the updated estimate
        lastAccepted = true;
      } else {
        lastAccepted = false;
        lambda *= ni;
        ni *= 2;  // pop(): the previous estimate is still in poses[cur]
        if (!std::isfinite(lambda)) { qmax++; R_.res->n_trials++; break; }
      }
      qmax++;
      R_.res->n_trials++;
    } while (rho < 0 && qmax < O.max_trials && !R_.stopped());
    YdBaResult* res = R_.res;
    if (res->n_log < 32) {
      res->log_chi2[res->n_log] = currentChi; res->log_lambda[res->n_log] = lambda; res->log_trials[res->n_log] = qmax; res->log_stage[res->n_log] = stage;
      res->n_log++;
    }
    res->n_iterations++;
    if (qmax == O.max_trials || rho == 0 || !std::isfinite(lambda)) break;  // SolverResult::Terminate
  }
  trace("optimize: LM loop done");
  for (int ph = 0; ph < PH_COUNT; ph++) PhaseTimer::collect(R_, ph);
  trace("optimize: errors read back");
  return YDORB_OK;
}



// validation shared by the single and the batched entry points; fills *Oout with the effective options
int checkProblem(const YdBaProblem* P, const YdBaOptions* optIn, YdBaResult* res, YdBaOptions* Oout) {
  if (!P || !res) { set_error("null argument"); return YDORB_ERR_INVALID_ARG; }
  YdBaOptions O;
  if (optIn) O = *optIn; else ydorb_ba_default_options(&O);
  uint8_t* outlier = res->edge_outlier;
  memset(res, 0, sizeof(*res));
  res->edge_outlier = outlier;
  const int K = P->n_poses, NP = P->n_points, E = P->n_edges;
  if (K < 0 || NP < 0 || E < 0 || (K && (!P->poses || !P->pose_fixed)) || (NP && !P->points) ||
      (E && (!P->edge_pose || !P->edge_point || !P->edge_meas || !P->edge_inv_sigma2)) || O.device < 0 || O.device >= 16 || O.max_trials < 1) {
    set_error("invalid BA problem");
    return YDORB_ERR_INVALID_ARG;
  }
  for (int e = 0; e < E; e++)
    if (P->edge_pose[e] < 0 || P->edge_pose[e] >= K || P->edge_point[e] < 0 || P->edge_point[e] >= NP) {
      set_error("edge %d references a vertex out of range", e);
      return YDORB_ERR_INVALID_ARG;
    }
  if (outlier) memset(outlier, 0, E);
  *Oout = O;
  return YDORB_OK;
}

// state upload at the start of a solve (the vertices g2o is handed at optimizer.cpp:185-230)
int beginSolve(Run& R_) {
  Ctx& c = *R_.c;
  const YdBaProblem* P = R_.P;
  const YdBaOptions& O = *R_.O;
  const int K = P->n_poses, NP = P->n_points;
  int rc;
  R_.noRobust = (O.flags & YDORB_BA_NO_ROBUST) != 0;
  R_.phaseTimes = (O.flags & YDORB_BA_PHASE_TIMES) != 0;
  R_.cam = Cam{P->fx, P->fy, P->cx, P->cy, P->bf};
  for (int i = 0; i < 2; i++)
    if ((rc = c.poses[i].ensure(sizeof(double) * 7 * K)) || (rc = c.pts[i].ensure(sizeof(double) * 3 * NP))) return rc;
  {  // SE3Quat's 7-vector constructor normalises the rotation (se3quat.h:80-86)
    std::vector<double> hp(P->poses, P->poses + (size_t)7 * K);
    for (int k = 0; k < K; k++) {
      double* q = &hp[7 * k + 3];
      if (q[3] < 0) for (int d = 0; d < 4; d++) q[d] = -q[d];
      const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int d = 0; d < 4; d++) q[d] /= nrm;
    }
    HIPCHK(hipMemcpyAsync(c.poses[0].p, hp.data(), sizeof(double) * 7 * K, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));   // hp dies here
  }
  HIPCHK(hipMemcpyAsync(c.pts[0].p, P->points, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, c.stream));
  return YDORB_OK;
}

int launchCull(Run& R_, int final) {
  Ctx& c = *R_.c;
  const YdBaOptions& O = *R_.O;
  const int Ea = R_.sys.Ea;
  if (Ea == 0) return YDORB_OK;
  hipLaunchKernelGGL(k_cull, dim3((Ea + 255) / 256), dim3(256), 0, c.stream, c.ePose.as<int>(), c.ePt.as<int>(), c.eMeas.as<double>(), Ea, c.eInfo0.as<double>(),
                     c.err.as<double>(), c.poses[R_.cur].as<double>(), c.pts[R_.cur].as<double>(), O.chi2_mono, O.chi2_stereo, final, c.eInfo.as<double>(),
                     c.eRobust.as<uint8_t>(), c.eOutlier.as<uint8_t>());
  HIPCHK(hipGetLastError());
  return YDORB_OK;
}

// optimizer.cpp:290-311: edges whose chi2 exceeds the threshold or whose depth is not positive leave the second stage; no kernels any more
int cullAfterFirstStage(Run& R_) { return launchCull(R_, 0); }

// optimizer.cpp:315-351 up to the write-back: the final outlier list and the estimates
int endSolve(Run& R_) {
  Ctx& c = *R_.c;
  const YdBaProblem* P = R_.P;
  const Run::Sys& Y = R_.sys;
  int rc = launchCull(R_, 1);
  if (rc) return rc;
  uint8_t* outlier = R_.res->edge_outlier;
  // down through the pinned staging area, then into the caller's (pageable) arrays
  const size_t bPoses = sizeof(double) * 7 * P->n_poses, bPts = sizeof(double) * 3 * P->n_points, oPts = (bPoses + 255) & ~(size_t)255,
               oOut = oPts + ((bPts + 255) & ~(size_t)255);
  if ((rc = c.stage(oOut + Y.Ea))) return rc;
  if (outlier && Y.Ea) HIPCHK(hipMemcpyAsync(c.hStage + oOut, c.eOutlier.p, Y.Ea, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.hStage, c.poses[R_.cur].p, bPoses, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipMemcpyAsync(c.hStage + oPts, c.pts[R_.cur].p, bPts, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));
  memcpy(P->poses, c.hStage, bPoses);
  memcpy(P->points, c.hStage + oPts, bPts);
  if (outlier) for (int i = 0; i < Y.Ea; i++) outlier[Y.act[i]] = c.hStage[oOut + i];   // device edges are in (landmark, pose) order
  if (R_.stopped()) R_.res->stopped = 1;
  return YDORB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ydorb_ba_solve_batch: independent problems advance in LOCK STEP through one set of launches per phase (blockIdx.z = problem,
// ba_kernels.hip.h "Lock-step batch").  The host keeps one LM state per problem - exactly the scalars and decisions of optimize()
// above - and every round (a) builds the system of the problems that start an iteration, (b) runs one LM trial of every unfinished
// problem, (c) reads all problems' three scalars back with ONE copy and decides accept / retry / terminate per problem.  A problem
// that is done with a stage goes through the same cull / second stage / read-back steps as a single solve while the others go on.
// ---------------------------------------------------------------------------------------------------------------------------
struct Job {
  Ctx ctx;                 // its own buffers; ctx.stream is the batch's stream (not owned)
  YdBaOptions O;
  Run* run = nullptr;
  int stage = 1, it = 0, iterations = 0, qmax = 0;
  double lambda = 0, ni = 2, currentChi = 0, rho = 0;
  bool lastAccepted = true, needBuild = false, done = false;
  bool pendingEnd = false;   // finished its LM schedule: final cull + read-back still to do (all of a group together, on the set-up threads)
  int rc = YDORB_OK;
  std::string errText;
  ~Job() { delete run; }
};
struct BatchPool {   // per device: contexts, stream and staging of the lock-step batches (one batch at a time per device)
  std::mutex mu;
  hipStream_t stream = nullptr;
  std::vector<Job*> jobs;          // grown on demand; buffers are kept between calls
  std::vector<hipStream_t> setupStreams;
  DBuf dDev, dScal;                // BaDev[B]; per problem 8 doubles (chi2, max diag, scale sum, ..., status copies) + 2 ints of status
  BaDev* hDev = nullptr;           // pinned
  double* hScal = nullptr;         // pinned
  int cap = 0;
};
BatchPool g_batch[16];
constexpr int kSetupThreads = 16;  // host threads of a batch's set-up phase
constexpr int kBatchGroup = 64;    // problems per lock-step group (C5-sized problems take ~40 MB each)

void fillDev(Job& J, BaDev& D, const BaDev* dDevBase, double* dScal, int* dStatus) {
  Ctx& c = J.ctx;
  const Run::Sys& Y = J.run->sys;
  memset(&D, 0, sizeof(D));
  auto off = [&](const void* p) { return (long long)(reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(dDevBase)); };   // see BaDev
  D.ePose = off(c.ePose.p); D.ePidx = off(c.ePidx.p); D.ePt = off(c.ePt.p); D.eMeas = off(c.eMeas.p); D.eInfo = off(c.eInfo.p); D.eRobust = off(c.eRobust.p);
  D.ptStart = off(c.ptStart.p); D.poseStart = off(c.poseStart.p); D.poseEdges = off(c.poseEdges.p); D.eLm = off(c.eLm.p);
  D.poseOf = off(c.poseOf.p); D.ptOf = off(c.ptOf.p); D.pairStart = off(c.pairStart.p); D.pairItems = off(c.pairB.p);
  for (int i = 0; i < 2; i++) { D.poses[i] = off(c.poses[i].p); D.pts[i] = off(c.pts[i].p); }
  D.err = off(c.err.p); D.partial = off(c.partial.p); D.Hll = off(c.Hll.p); D.bl = off(c.bl.p); D.Hpl = off(c.Hpl.p);
  D.BD = off(c.BD.p); D.Hpp = off(c.Hpp.p); D.S = off(c.S.p); D.diagL = off(c.diagL.p); D.diagInv = off(c.diagInv.p);
  D.Dinv = off(c.Dinv.p); D.db = off(c.db.p); D.xp = off(c.xp.p); D.yv = off(c.yv.p); D.xl = off(c.xl.p);
  D.scal = off(dScal); D.status = off(dStatus);
  D.cam = J.run->cam; D.dM = J.O.delta_mono; D.dSt = J.O.delta_stereo;
  D.nL = Y.nL; D.nPf = Y.nPf; D.Ea = Y.Ea; D.n = Y.n; D.nb = Y.nb; D.nBlkE = Y.nBlkE; D.nBuckets = Y.nBuckets;
}

int solveGroup(BatchPool& B, const YdBaProblem* probs, const YdBaOptions& Oin, YdBaResult* res, int n, int* rcEach) {
  hipStream_t s = B.stream;
  int rc;
  if ((rc = B.dDev.ensure(sizeof(BaDev) * n)) || (rc = B.dScal.ensure((sizeof(double) * 8 + sizeof(int) * 2) * n))) return rc;
  double* dScalAll = B.dScal.as<double>();
  int* dStatusAll = reinterpret_cast<int*>(dScalAll + (size_t)8 * n);
  std::vector<Job*> J(B.jobs.begin(), B.jobs.begin() + n);
  auto fail = [&](int j, int code) { J[j]->rc = code; J[j]->done = true; if (rcEach) rcEach[j] = code; };

  // stage transitions (per problem, on the shared stream) -----------------------------------------------------------------
  std::function<void(int)> finalize, endStage;
  auto startStage = [&](int j, int stage) {
    Job& X = *J[j];
    X.stage = stage; X.it = 0; X.qmax = 0; X.rho = 0; X.lastAccepted = true;
    X.iterations = stage == 1 ? X.O.iters1 : X.O.iters2;
    int r = prepareStage(*X.run, stage == 2);
    if (r) { fail(j, r); return; }
    if (X.run->sys.Ea == 0) { endStage(j); return; }
    if (X.run->sys.n > kCholSolveMaxN) { set_error("reduced camera system of %d rows is wider than the solve kernel's LDS (max %d)", X.run->sys.n, kCholSolveMaxN); fail(j, YDORB_ERR_UNSUPPORTED); return; }
    fillDev(X, B.hDev[j], B.dDev.as<BaDev>(), dScalAll + (size_t)8 * j, dStatusAll + (size_t)2 * j);
    if (!(X.it < X.iterations && !X.run->stopped())) { endStage(j); return; }   // `for (it = 0; it < iterations && !terminate(); ...)`
    X.needBuild = true;
  };
  finalize = [&](int j) {   // the final cull, read-back and outlier scatter of a problem (~0.1 ms each, mostly host): deferred, see the end
    J[j]->pendingEnd = true;
    J[j]->done = true;
  };
  endStage = [&](int j) {
    Job& X = *J[j];
    X.needBuild = false;
    int r;
    if (X.stage == 1 && !(X.O.flags & YDORB_BA_SINGLE_STAGE)) {
      if (!X.run->stopped()) {   // optimizer.cpp:290-314
        if ((r = cullAfterFirstStage(*X.run))) { fail(j, r); return; }
        startStage(j, 2);
        return;
      }
      X.run->res->stopped = 1;
    }
    finalize(j);
  };
  // end of an outer iteration: the log entry and SolverResult::Terminate, as in optimize()
  auto endIteration = [&](int j) {
    Job& X = *J[j];
    YdBaResult* r = X.run->res;
    if (r->n_log < 32) {
      r->log_chi2[r->n_log] = X.currentChi; r->log_lambda[r->n_log] = X.lambda; r->log_trials[r->n_log] = X.qmax; r->log_stage[r->n_log] = X.stage;
      r->n_log++;
    }
    r->n_iterations++;
    const bool terminate = X.qmax == X.O.max_trials || X.rho == 0 || !std::isfinite(X.lambda);
    X.it++;
    if (terminate || !(X.it < X.iterations && !X.run->stopped())) { endStage(j); return; }
    X.needBuild = true; X.qmax = 0; X.rho = 0;
  };

  // Set-up of every problem (validation, edge ordering, uploads, pose-pair buckets): ~2 ms of host work and pageable copies per C5-sized
  // problem, independent of the others, so it is spread over a few host threads, each on its own set-up stream; the lock-step
  // rounds below then run on the batch's one stream.
  {
    const int nt = std::max(1, std::min(n, kSetupThreads));
    while ((int)B.setupStreams.size() < nt) {
      hipStream_t st = nullptr;
      HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      B.setupStreams.push_back(st);
    }
    std::atomic<int> next{0};
    auto worker = [&](int t) {
      (void)hipSetDevice(Oin.device);
      hipStream_t st = B.setupStreams[t];
      for (int j = next.fetch_add(1); j < n; j = next.fetch_add(1)) {
        Job& X = *J[j];
        X.done = false; X.pendingEnd = false; X.rc = YDORB_OK; X.errText.clear();
        delete X.run; X.run = nullptr;
        if (rcEach) rcEach[j] = YDORB_OK;
        int r = checkProblem(&probs[j], &Oin, &res[j], &X.O);
        const YdBaProblem* P = &probs[j];
        if (!r) {
          if (P->stop && *P->stop) { res[j].stopped = 1; X.done = true; continue; }
          if (P->n_edges == 0 || P->n_poses == 0 || P->n_points == 0) { X.done = true; continue; }
          X.ctx.device = Oin.device; X.ctx.stream = st;
          X.run = new Run{&X.ctx, P, &X.O, &res[j]};
          r = beginSolve(*X.run);
        }
        if (!r) {
          startStage(j, 1);
          r = X.rc;
        }
        if (r) { X.errText = ydorb_last_error(); fail(j, r); }
        if (hipStreamSynchronize(st) != hipSuccess && !r) { X.errText = "hipStreamSynchronize failed"; fail(j, YDORB_ERR_HIP); }
        X.ctx.stream = s;
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(worker, t);
    worker(0);
    for (std::thread& th : pool) th.join();
    for (int j = 0; j < n; j++) if (J[j]->rc != YDORB_OK) set_error("%s", J[j]->errText.c_str());
  }
  trace("batch: set-up done");

  auto maxOver = [&](auto fn) { int m = 0; for (int j = 0; j < n; j++) if (!J[j]->done) m = std::max(m, fn(J[j]->run->sys)); return m; };
  auto rounds = [&]() -> int {
  while (true) {
    bool any = false;
    for (int j = 0; j < n; j++) any = any || !J[j]->done;
    if (!any) break;
    trace("batch: round begins");
    // (a) iteration starts: chi2 of the current estimate where needed, H and b, initial lambda -----------------------------------
    bool anyBuild = false, anyChi = false, anyDiag = false;
    for (int j = 0; j < n; j++) {
      Job& X = *J[j];
      BaDev& D = B.hDev[j];
      D.trial = 0;
      D.build = !X.done && X.needBuild;
      D.chi2 = D.build && (X.it == 0 || !X.lastAccepted);
      D.maxdiag = D.build && X.it == 0;
      D.cur = X.done ? 0 : X.run->cur;
      D.lambda = X.lambda;
      anyBuild = anyBuild || D.build; anyChi = anyChi || D.chi2; anyDiag = anyDiag || D.maxdiag;
    }
    const int gE = maxOver([](const Run::Sys& Y) { return Y.nBlkE; }), gL128 = maxOver([](const Run::Sys& Y) { return (Y.nL + 127) / 128; }),
              gL256 = maxOver([](const Run::Sys& Y) { return (Y.nL + 255) / 256; }), gP = maxOver([](const Run::Sys& Y) { return Y.nPf; }),
              gBk = maxOver([](const Run::Sys& Y) { return Y.nBuckets + 1; }), gNb = maxOver([](const Run::Sys& Y) { return Y.nb; }),
              gUpd = maxOver([](const Run::Sys& Y) { return (std::max(Y.nPf, Y.nL) + 255) / 256; }),
              gScale = maxOver([](const Run::Sys& Y) { return (6 * Y.nPf + 3 * Y.nL + 255) / 256; }), gN = maxOver([](const Run::Sys& Y) { return Y.n; });
    const BaDev* dDev = B.dDev.as<BaDev>();
    if (anyBuild) {
      HIPCHK(hipMemcpyAsync(B.dDev.p, B.hDev, sizeof(BaDev) * n, hipMemcpyHostToDevice, s));
      if (anyChi) {
        hipLaunchKernelGGL(kb_errors, dim3(gE, 1, n), dim3(256), 0, s, dDev, 0);
        hipLaunchKernelGGL(kb_sum_partials, dim3(1, 1, n), dim3(256), 0, s, dDev, 0);
      }
      hipLaunchKernelGGL(kb_build_points, dim3(gL128, 1, n), dim3(128), 0, s, dDev);
      if (gP) hipLaunchKernelGGL(kb_build_poses, dim3(gP, 1, n), dim3(256), 0, s, dDev);
      if (anyDiag) hipLaunchKernelGGL(kb_max_diag, dim3(1, 1, n), dim3(256), 0, s, dDev);
      HIPCHK(hipGetLastError());
      if (anyChi || anyDiag) {
        HIPCHK(hipMemcpyAsync(B.hScal, dScalAll, sizeof(double) * 8 * n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
      }
      for (int j = 0; j < n; j++) {
        Job& X = *J[j];
        const BaDev& D = B.hDev[j];
        if (!D.build) continue;
        if (D.chi2) X.currentChi = B.hScal[8 * j + 0];
        if (D.maxdiag) { X.lambda = 1e-5 * B.hScal[8 * j + 1]; X.ni = 2; }   // computeLambdaInit
        X.needBuild = false;
      }
    }
    // (b) one LM trial of every unfinished problem ----------------------------------------------------------------------------------
    for (int j = 0; j < n; j++) {
      Job& X = *J[j];
      BaDev& D = B.hDev[j];
      D.build = D.chi2 = D.maxdiag = 0;
      D.trial = !X.done;
      D.lambda = X.lambda;
      D.cur = X.done ? 0 : X.run->cur;
    }
    HIPCHK(hipMemcpyAsync(B.dDev.p, B.hDev, sizeof(BaDev) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kb_dinv, dim3(gL256, 1, n), dim3(256), 0, s, dDev);
    hipLaunchKernelGGL(kb_bd, dim3(gE, 1, n), dim3(256), 0, s, dDev);
    if (gP) hipLaunchKernelGGL(kb_bs, dim3(gP, 1, n), dim3(256), 0, s, dDev);
    hipLaunchKernelGGL(kb_schur_pairs, dim3(gBk, 1, n), dim3(64 * kSchurWaves), 0, s, dDev);
    for (int kb = 0; kb < gNb; kb++)
      hipLaunchKernelGGL(kb_chol_step, dim3((gNb - kb) * (gNb - kb + 1) / 2 + (kb > 0), 1, n), dim3(256), 0, s, dDev, kb);
    {
      const size_t dyn = sizeof(double) * (size_t)gN;
      if (dyn > 48 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kb_chol_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
      hipLaunchKernelGGL(kb_chol_solve, dim3(1, 1, n), dim3(1024), dyn, s, dDev);
    }
    hipLaunchKernelGGL(kb_backsub, dim3(gL128, 1, n), dim3(128), 0, s, dDev);
    hipLaunchKernelGGL(kb_update, dim3(gUpd, 1, n), dim3(256), 0, s, dDev);
    hipLaunchKernelGGL(kb_scale, dim3(gScale, 1, n), dim3(256), 0, s, dDev);
    hipLaunchKernelGGL(kb_sum_partials, dim3(1, 1, n), dim3(256), 0, s, dDev, 2);
    hipLaunchKernelGGL(kb_errors, dim3(gE, 1, n), dim3(256), 0, s, dDev, 1);
    hipLaunchKernelGGL(kb_sum_partials, dim3(1, 1, n), dim3(256), 0, s, dDev, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(B.hScal, dScalAll, sizeof(double) * 8 * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    trace("batch: round synchronised");
    // (c) the LM decision of every problem (optimization_algorithm_levenberg.cpp:95-146, as in optimize()) -----------------------------------
    for (int j = 0; j < n; j++) {
      Job& X = *J[j];
      if (X.done) continue;
      const double* hs = B.hScal + (size_t)8 * j;
      double tempChi = hs[0];
      const double scaleSum = hs[2];
      const bool ok2 = (int)hs[6] == 0;
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      X.rho = X.currentChi - tempChi;
      const double scale = scaleSum + 1e-3;
      X.rho /= scale;
      bool lambdaBroke = false;
      if (X.rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - pow((2 * X.rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        X.lambda *= std::max(1. / 3., alpha);
        X.ni = 2;
        X.currentChi = tempChi;
        X.run->cur ^= 1;  // discardTop(): keep the updated estimate
        X.lastAccepted = true;
      } else {
        X.lastAccepted = false;
        X.lambda *= X.ni;
        X.ni *= 2;  // pop(): the previous estimate is still in poses[cur]
        if (!std::isfinite(X.lambda)) lambdaBroke = true;
      }
      X.qmax++;
      X.run->res->n_trials++;
      if (lambdaBroke || !(X.rho < 0 && X.qmax < X.O.max_trials && !X.run->stopped())) endIteration(j);
    }
  }
  return YDORB_OK;
  };
  {
    // a HIP error inside the rounds ends the group: every member that has not been read back reports it (its poses / points / outlier
    // list were not written), members that failed earlier keep their own status
    int rr = rounds();
    if (rr == YDORB_OK && hipStreamSynchronize(s) != hipSuccess) { set_error("hipStreamSynchronize failed after the lock-step rounds"); rr = YDORB_ERR_HIP; }
    if (rr != YDORB_OK) {
      const std::string text = ydorb_last_error();
      for (int j = 0; j < n; j++)
        if (J[j]->rc == YDORB_OK && J[j]->run && (!J[j]->done || J[j]->pendingEnd)) { J[j]->pendingEnd = false; J[j]->errText = text; fail(j, rr); }
      set_error("%s", text.c_str());
      return rr;
    }
  }
  {  // endSolve of every finished problem, spread over the set-up threads and streams (the lock-step stream has drained)
    const int nt = std::max(1, std::min(n, (int)B.setupStreams.size()));
    std::atomic<int> next{0};
    auto worker = [&](int t) {
      (void)hipSetDevice(Oin.device);
      for (int j = next.fetch_add(1); j < n; j = next.fetch_add(1)) {
        Job& X = *J[j];
        if (!X.pendingEnd) continue;
        X.pendingEnd = false;
        X.ctx.stream = B.setupStreams[t];
        const int r = endSolve(*X.run);
        X.ctx.stream = s;
        if (r) { X.errText = ydorb_last_error(); fail(j, r); }
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(worker, t);
    worker(0);
    for (std::thread& th : pool) th.join();
    for (int j = 0; j < n; j++) if (J[j]->rc != YDORB_OK && !J[j]->errText.empty()) set_error("%s", J[j]->errText.c_str());
  }
  trace("batch: done");
  int first = YDORB_OK;
  for (int j = 0; j < n; j++) if (J[j]->rc != YDORB_OK && first == YDORB_OK) first = J[j]->rc;
  return first;
}

}  // namespace

extern "C" {

void ydorb_ba_default_options(YdBaOptions* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->iters1 = 5; o->iters2 = 10;
  o->chi2_mono = 5.991; o->chi2_stereo = 7.815;
  o->delta_mono = (double)(float)sqrt(5.991);   // `const float monoDelta = sqrt(5.991)`, optimizer.cpp:223
  o->delta_stereo = (double)(float)sqrt(7.815);
  o->max_trials = 10;
  o->device = 0;
  o->rank = 0; o->world = 1;
}

int ydorb_ba_solve(const YdBaProblem* P, const YdBaOptions* optIn, YdBaResult* res) {
  YdBaOptions O;
  int rc = checkProblem(P, optIn, res, &O);
  if (rc) return rc;
  const int K = P->n_poses, NP = P->n_points, E = P->n_edges;
  if (P->stop && *P->stop) { res->stopped = 1; return YDORB_OK; }  // optimizer.cpp:284-286
  if (E == 0 || K == 0 || NP == 0) return YDORB_OK;
  if ((rc = require_device(O.device))) return rc;
  g_t0 = std::chrono::steady_clock::now();
  trace("solve: begin");
  int slot = -1;
  {
    std::lock_guard<std::mutex> pick(g_pick);
    for (int i = 0; i < kCtxPool && slot < 0; i++)
      if (g_mu[O.device][i].try_lock()) slot = i;
  }
  if (slot < 0) { slot = 0; g_mu[O.device][0].lock(); }   // all busy: queue behind slot 0
  std::lock_guard<std::mutex> lock(g_mu[O.device][slot], std::adopt_lock);
  Ctx& c = g_ctx[O.device][slot];
  if (!c.stream) {
    c.device = O.device;
    HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    for (auto& e : c.ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipHostMalloc(&c.hPin, sizeof(double) * 16));
  }
  Run R_{&c, P, &O, res};
  if ((rc = beginSolve(R_))) return rc;
  trace("solve: state uploaded");
  hipEvent_t t0 = c.ev[2 * PH_COUNT], t1 = c.ev[2 * PH_COUNT + 1];
  HIPCHK(hipEventRecord(t0, c.stream));

  if ((rc = optimize(R_, O.iters1, 1, false))) return rc;
  if (O.flags & YDORB_BA_SINGLE_STAGE) {
    // bundleAdjust: one optimize() call, nothing culled
  } else if (!R_.stopped()) {  // optimizer.cpp:290-314
    if ((rc = cullAfterFirstStage(R_))) return rc;
    trace("solve: depths read");
    if ((rc = optimize(R_, O.iters2, 2, true))) return rc;
  } else {
    res->stopped = 1;
  }
  HIPCHK(hipEventRecord(t1, c.stream));
  if ((rc = endSolve(R_))) return rc;
  (void)hipEventElapsedTime(&res->ms_total, t0, t1);
  trace("solve: results read back");
  res->ms_errors = (float)R_.phaseMs[PH_ERR]; res->ms_build = (float)R_.phaseMs[PH_BUILD]; res->ms_schur = (float)R_.phaseMs[PH_SCHUR];
  res->ms_solve = (float)R_.phaseMs[PH_SOLVE]; res->ms_update = (float)R_.phaseMs[PH_UPDATE];
  return YDORB_OK;
}

int ydorb_pose_optimize(const YdPoseBatch* B, uint8_t* outlier, int32_t* n_inliers, double* chi2_log, int32_t* trials) {
  if (!B || B->n_frames < 0 || (B->n_frames && (!B->edge_start || !B->poses || !n_inliers)) || B->device < 0 || B->device >= 16) {
    set_error("invalid pose batch");
    return YDORB_ERR_INVALID_ARG;
  }
  const int n = B->n_frames;
  if (n == 0) return YDORB_OK;
  if (B->edge_start[0] != 0) { set_error("edge_start[0] must be 0"); return YDORB_ERR_INVALID_ARG; }
  for (int f = 0; f < n; f++)
    if (B->edge_start[f + 1] < B->edge_start[f]) { set_error("edge_start must be non-decreasing"); return YDORB_ERR_INVALID_ARG; }
  const int E = B->edge_start[n];
  if (E > 0 && (!B->points || !B->meas || !B->inv_sigma2 || !outlier)) { set_error("null edge arrays"); return YDORB_ERR_INVALID_ARG; }
  int rc = require_device(B->device);
  if (rc) return rc;
  int slot = -1;
  {
    std::lock_guard<std::mutex> pick(g_pick);
    for (int i = 0; i < kCtxPool && slot < 0; i++)
      if (g_mu[B->device][i].try_lock()) slot = i;
  }
  if (slot < 0) { slot = 0; g_mu[B->device][0].lock(); }
  std::lock_guard<std::mutex> lock(g_mu[B->device][slot], std::adopt_lock);
  Ctx& c = g_ctx[B->device][slot];
  if (!c.stream) {
    c.device = B->device;
    HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    for (auto& e : c.ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipHostMalloc(&c.hPin, sizeof(double) * 16));
  }
  hipStream_t s = c.stream;
  const size_t Ez = (size_t)std::max(E, 1);
  if ((rc = c.pStart.ensure(sizeof(int) * (n + 1))) || (rc = c.pPoses.ensure(sizeof(double) * 7 * n)) || (rc = c.pX.ensure(sizeof(double) * 3 * Ez)) ||
      (rc = c.pMeas.ensure(sizeof(double) * 3 * Ez)) || (rc = c.pInfo.ensure(sizeof(double) * Ez)) || (rc = c.pErr.ensure(sizeof(double) * 3 * Ez)) ||
      (rc = c.pFlags.ensure(Ez)) || (rc = c.pOutlier.ensure(Ez)) || (rc = c.pInl.ensure(sizeof(int) * n)) || (rc = c.pChi.ensure(sizeof(double) * 4 * n)) ||
      (rc = c.pTrials.ensure(sizeof(int) * n)))
    return rc;
  HIPCHK(hipMemcpyAsync(c.pStart.p, B->edge_start, sizeof(int) * (n + 1), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c.pPoses.p, B->poses, sizeof(double) * 7 * n, hipMemcpyHostToDevice, s));
  if (E) {
    HIPCHK(hipMemcpyAsync(c.pX.p, B->points, sizeof(double) * 3 * E, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c.pMeas.p, B->meas, sizeof(double) * 3 * E, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c.pInfo.p, B->inv_sigma2, sizeof(double) * E, hipMemcpyHostToDevice, s));
  }
  const Cam cam{B->fx, B->fy, B->cx, B->cy, B->bf};
  const double dM = (double)(float)sqrt(5.991), dS = (double)(float)sqrt(7.815);   // optimizer.cpp:381-382
  hipLaunchKernelGGL(k_pose_optimize, dim3(n), dim3(kPoseThreads), 0, s, n, c.pStart.as<int>(), c.pPoses.as<double>(), c.pX.as<double>(),
                     c.pMeas.as<double>(), c.pInfo.as<double>(), cam, dM, dS, c.pErr.as<double>(), c.pFlags.as<uint8_t>(),
                     c.pOutlier.as<uint8_t>(), c.pInl.as<int>(), c.pChi.as<double>(), c.pTrials.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(B->poses, c.pPoses.p, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(n_inliers, c.pInl.p, sizeof(int) * n, hipMemcpyDeviceToHost, s));
  if (E) HIPCHK(hipMemcpyAsync(outlier, c.pOutlier.p, E, hipMemcpyDeviceToHost, s));
  if (chi2_log) HIPCHK(hipMemcpyAsync(chi2_log, c.pChi.p, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, s));
  if (trials) HIPCHK(hipMemcpyAsync(trials, c.pTrials.p, sizeof(int) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return YDORB_OK;
}


int ydorb_ba_solve_batch(const YdBaProblem* probs, int32_t n, const YdBaOptions* opt, YdBaResult* res, int32_t threads, int32_t* rcEach) {
  if (n < 0 || (n > 0 && (!probs || !res))) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  if (opt && opt->world > 1) { set_error("the batched form solves whole problems: no landmark sharding"); return YDORB_ERR_INVALID_ARG; }
  if (n == 0) return YDORB_OK;
  YdBaOptions O;
  if (opt) O = *opt; else ydorb_ba_default_options(&O);
  if (O.device < 0 || O.device >= 16) { set_error("invalid device"); return YDORB_ERR_INVALID_ARG; }
  int rc = require_device(O.device);
  if (rc) return rc;
  BatchPool& B = g_batch[O.device];
  std::lock_guard<std::mutex> lock(B.mu);
  g_t0 = std::chrono::steady_clock::now();
  HIPCHK(hipSetDevice(O.device));
  if (!B.stream) HIPCHK(hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
  // `threads` is the number of problems advanced together (0 = as many as fit one group); the lock-step batch needs no host threads
  const int group = std::max(1, std::min<int>(threads > 0 ? threads : kBatchGroup, kBatchGroup));
  if (B.cap < group) {
    if (B.hDev) (void)hipHostFree(B.hDev);
    if (B.hScal) (void)hipHostFree(B.hScal);
    B.hDev = nullptr; B.hScal = nullptr; B.cap = 0;
    HIPCHK(hipHostMalloc(&B.hDev, sizeof(BaDev) * group));
    HIPCHK(hipHostMalloc(&B.hScal, sizeof(double) * 8 * group));
    B.cap = group;
  }
  while ((int)B.jobs.size() < group) B.jobs.push_back(new Job());
  int first = YDORB_OK;
  std::string firstText;
  for (int at = 0; at < n; at += group) {
    const int m = std::min(group, n - at);
    rc = solveGroup(B, probs + at, O, res + at, m, rcEach ? rcEach + at : nullptr);
    if (rc != YDORB_OK && first == YDORB_OK) { first = rc; firstText = ydorb_last_error(); }
  }
  if (first != YDORB_OK) set_error("%s", firstText.c_str());
  return first;
}

int ydorb_ba_release(int32_t device) {
  if (device < 0 || device >= 16) { set_error("invalid device"); return YDORB_ERR_INVALID_ARG; }
  int rc = require_device(device);
  if (rc) return rc;
  HIPCHK(hipSetDevice(device));
  for (int i = 0; i < kCtxPool; i++) {          // waits for a solve that holds the context
    std::lock_guard<std::mutex> lock(g_mu[device][i]);
    Ctx& c = g_ctx[device][i];
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    c.releaseBuffers();
  }
  BatchPool& B = g_batch[device];
  std::lock_guard<std::mutex> lock(B.mu);
  if (B.stream) (void)hipStreamSynchronize(B.stream);
  for (Job* j : B.jobs) { j->ctx.releaseBuffers(); delete j; }
  B.jobs.clear();
  B.dDev.release(); B.dScal.release();
  if (B.hDev) (void)hipHostFree(B.hDev);
  if (B.hScal) (void)hipHostFree(B.hScal);
  B.hDev = nullptr; B.hScal = nullptr; B.cap = 0;
  return YDORB_OK;
}

int ydorb_ba_dense_solve(int32_t device, const double* A, int32_t n0, const double* b, double* x, int32_t* ok) {
  if (!A || !b || !x || !ok || n0 < 1 || device < 0 || device >= 16) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  int rc = require_device(device);
  if (rc) return rc;
  const int n = (n0 + NB - 1) / NB * NB, nb = n / NB;
  std::vector<double> hA((size_t)n * n, 0.0), hb(n, 0.0);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) hA[(size_t)i * n + j] = (i < n0 && j < n0) ? A[(size_t)i * n0 + j] : (i == j ? 1.0 : 0.0);
    if (i < n0) hb[i] = b[i];
  }
  double *dA = nullptr, *dD = nullptr, *dI = nullptr, *db = nullptr, *dx = nullptr, *dy = nullptr;
  int* dst = nullptr;
  HIPCHK(hipMalloc(&dA, sizeof(double) * n * n));
  HIPCHK(hipMalloc(&dD, sizeof(double) * nb * NB * NB));
  HIPCHK(hipMalloc(&dI, sizeof(double) * nb * NB * NB));
  HIPCHK(hipMalloc(&db, sizeof(double) * n));
  HIPCHK(hipMalloc(&dx, sizeof(double) * n));
  HIPCHK(hipMalloc(&dy, sizeof(double) * n));
  HIPCHK(hipMalloc(&dst, sizeof(int) * 2));
  HIPCHK(hipMemcpy(dA, hA.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db, hb.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dst, 0, sizeof(int) * 2));
  for (int kb = 0; kb < nb; kb++) hipLaunchKernelGGL(k_chol_step, dim3((nb - kb) * (nb - kb + 1) / 2 + (kb > 0)), dim3(256), 0, 0, dA, dD, dI, n, kb, dst, db, dy);
  if (!launch_chol_solve(0, dA, dI, n, dy, db, dx)) {
    (void)hipFree(dA); (void)hipFree(dD); (void)hipFree(dI); (void)hipFree(db); (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dst);
    set_error("system of %d rows is wider than the solve kernel's LDS (max %d)", n, kCholSolveMaxN);
    return YDORB_ERR_UNSUPPORTED;
  }
  HIPCHK(hipGetLastError());
  std::vector<double> hx(n);
  int hst[2];
  HIPCHK(hipMemcpy(hx.data(), dx, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hst, dst, sizeof(hst), hipMemcpyDeviceToHost));
  *ok = hst[0] == 0;
  for (int i = 0; i < n0; i++) x[i] = hx[i];
#ifdef CHOL_TIMING
  {
    static long long clk[2][32][12];
    HIPCHK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_cholClk), sizeof(clk)));
    for (int w = 0; w < 2; w++)
      for (int kb = 0; kb < nb && kb < 32; kb += 6) {
        fprintf(stderr, "chol wg%d kb=%d:", w, kb);
        for (int p2 = 1; p2 < 11; p2++) fprintf(stderr, " %.2f", clk[w][kb][p2] ? (clk[w][kb][p2] - clk[w][kb][0]) / 100.0 : -1.0);
        fprintf(stderr, " us\n");
      }
    for (int kb = 1; kb < nb && kb < 32; kb++) fprintf(stderr, "%.1f ", (clk[0][kb][0] - clk[0][kb - 1][0]) / 100.0);
    fprintf(stderr, "us between step starts\n");
  }
#endif
  (void)hipFree(dA); (void)hipFree(dD); (void)hipFree(dI); (void)hipFree(db); (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dst);
  return YDORB_OK;
}

}  // extern "C"
