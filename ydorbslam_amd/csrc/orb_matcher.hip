// Host side of the MI355X descriptor matcher + its C ABI (include/ydorb/c_api.h, "Descriptor matcher").
// Mirrors YDORBSLAM::OrbMatcher's search-by-projection / search-by-BoW entry points
// (reference src/orbMatcher.cpp:24-239, 303-462) on POD views; all searching runs in the HIP kernels of
// match_kernels.hip.h.  No CPU fallback: the only host arithmetic is the one-pair popcount that the
// reference exposes as a static helper (orbMatcher.cpp:11-23) and the vocabulary-node merge-join that
// decides which buckets meet (a walk over two sorted id lists, orbMatcher.cpp:317-361).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/ydorb/c_api.h"
#include "match_kernels.hip.h"
#include "stereo_kernels.hip.h"
#include "ydorb_host.h"

using namespace ydorb;

namespace {

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ydorb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return YDORB_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

// k_grid_build keeps one 16-bit cell id per keypoint in dynamic LDS beside ~12 KiB of static tables: frames with more than
// ~26 k keypoints need the kernel's dynamic-LDS limit raised (the 16-bit index format allows up to 65 535).
int launchGridBuild(int nFrames, int cap, hipStream_t s, const FrameDev* frames) {
  const size_t dyn = sizeof(int16_t) * (size_t)std::max(cap, 0);
  if (dyn > 48 * 1024)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  hipLaunchKernelGGL(k_grid_build, dim3(nFrames), dim3(256), dyn, s, frames, cap);
  HIPCHK(hipGetLastError());
  return YDORB_OK;
}

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return YDORB_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 4096);
    if (hipMalloc(&p, want) != hipSuccess) { set_error("hipMalloc(%zu) failed", want); return YDORB_ERR_HIP; }
    cap = want;
    return YDORB_OK;
  }
  template <class T> T* as() { return reinterpret_cast<T*>(p); }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

__global__ void k_hamming_rows(const uint8_t* a, const uint8_t* b, int n, int* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = hamming256(a + (size_t)i * 32, b + (size_t)i * 32);
}

enum { MS_GRID = 0, MS_GATHER, MS_RESOLVE, MS_COUNT };
const char* kMatchStageNames[MS_COUNT] = {"grid_build", "gather_distances", "resolve"};

}  // namespace

struct ydorb_matcher {
  int device = 0;
  hipStream_t stream = nullptr;
  Buf kps, desc, rightX, queries, qdesc, taken, assigned, matchQ, qInfo, qPre, cellStart, cellIdx, pool, frames, calls, misc, kps2,
      desc2, feat, valid, qFeat, qRange, qAngle, sf, heads, sortedKp, sortedDesc, kps1, good1, good2, stereoPar, stereoCnt, stereoOut;
  size_t poolRecords = 1u << 20;
  // cached descriptors of the last batched launch (re-uploaded only when they change)
  std::vector<FrameDev> hFrames;
  std::vector<CallDev> hCalls;
  float hSf[8] = {0};
  bool hIdentAffine = false;
  std::vector<int32_t> consecPairs;
  int ovfPerKeypoint = 16;
  bool profiling = false;
  hipEvent_t ev[MS_COUNT + 1]{};
  double stageMs[MS_COUNT]{};
  int stageCalls = 0;
  bool evPending = false;
};

namespace {

void collect(ydorb_matcher* m) {
  if (!m->profiling || !m->evPending) return;
  m->evPending = false;
  if (hipEventQuery(m->ev[MS_COUNT]) != hipSuccess) return;   // a pipelined caller launched again before the events completed: sample dropped, never waited for
  for (int i = 0; i < MS_COUNT; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, m->ev[i], m->ev[i + 1]) == hipSuccess) m->stageMs[i] += ms;
  }
  m->stageCalls++;
}

// misc layout: [0] poolHead (unsigned), [1] status (int), [2] count (int)
int resetMisc(ydorb_matcher* m, hipStream_t s) {
  int rc = m->misc.ensure(64);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(m->misc.p, 0, 64, s));
  return YDORB_OK;
}

FrameDev makeFrame(const YdFrameView& v, const KeyPointDev* dk, const uint8_t* dd, const float* drx, int* cellStart, int* cellIdx, float4* skp,
                   uint8_t* sdesc) {
  FrameDev F{};
  F.kps = dk; F.desc = dd; F.rightX = drx; F.nPtr = nullptr; F.n = v.n;
  F.minX = v.min_x; F.minY = v.min_y;
  F.gridWInv = static_cast<float>(kGridCols) / (v.max_x - v.min_x);  // frame.cpp:99-100
  F.gridHInv = static_cast<float>(kGridRows) / (v.max_y - v.min_y);
  F.cellStart = cellStart; F.cellIdx = cellIdx; F.sortedKp = skp; F.sortedDesc = sdesc;
  return F;
}

int uploadFrame(ydorb_matcher* m, const YdFrameView* fv, FrameDev* out) {
  const int n = std::max(fv->n, 1);
  int rc;
  if ((rc = m->kps.ensure(sizeof(YdKeyPoint) * n)) || (rc = m->desc.ensure((size_t)32 * n)) || (rc = m->cellStart.ensure(sizeof(int) * (kGridCells + 1))) ||
      (rc = m->cellIdx.ensure(sizeof(int) * n)) || (rc = m->frames.ensure(sizeof(FrameDev))) || (rc = m->sortedKp.ensure(sizeof(float4) * n)) ||
      (rc = m->sortedDesc.ensure((size_t)32 * n)))
    return rc;
  if (fv->right_x && (rc = m->rightX.ensure(sizeof(float) * n))) return rc;
  if (fv->n > 0) {
    HIPCHK(hipMemcpyAsync(m->kps.p, fv->kps, sizeof(YdKeyPoint) * fv->n, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->desc.p, fv->desc, (size_t)32 * fv->n, hipMemcpyHostToDevice, m->stream));
    if (fv->right_x) HIPCHK(hipMemcpyAsync(m->rightX.p, fv->right_x, sizeof(float) * fv->n, hipMemcpyHostToDevice, m->stream));
  }
  *out = makeFrame(*fv, m->kps.as<KeyPointDev>(), m->desc.as<uint8_t>(), fv->right_x ? m->rightX.as<float>() : nullptr,
                   m->cellStart.as<int>(), m->cellIdx.as<int>(), m->sortedKp.as<float4>(), m->sortedDesc.as<uint8_t>());
  HIPCHK(hipMemcpyAsync(m->frames.p, out, sizeof(FrameDev), hipMemcpyHostToDevice, m->stream));
  return launchGridBuild(1, n, m->stream, m->frames.as<FrameDev>());
}

}  // namespace

extern "C" {

void ydorb_matcher_destroy(ydorb_matcher_t* m);

int ydorb_matcher_create(int32_t device, ydorb_matcher_t** out) {
  if (!out) { set_error("null argument"); return YDORB_ERR_INVALID_ARG; }
  *out = nullptr;
  int rc = require_device(device);
  if (rc) return rc;
  ydorb_matcher* m = new ydorb_matcher();
  m->device = device;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    delete m;
    return YDORB_ERR_HIP;
  }
  for (auto& e : m->ev) (void)hipEventCreate(&e);
  if (m->misc.ensure(64) != YDORB_OK || hipMemset(m->misc.p, 0, 64) != hipSuccess) {
    set_error("matcher scratch allocation failed");
    ydorb_matcher_destroy(m);
    return YDORB_ERR_HIP;
  }
  *out = m;
  return YDORB_OK;
}

void ydorb_matcher_destroy(ydorb_matcher_t* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipStreamSynchronize(m->stream);
  for (Buf* b : {&m->kps, &m->desc, &m->rightX, &m->queries, &m->qdesc, &m->taken, &m->assigned, &m->matchQ, &m->qInfo, &m->qPre, &m->cellStart,
                 &m->cellIdx, &m->pool, &m->frames, &m->calls, &m->misc, &m->kps2, &m->desc2, &m->feat, &m->valid, &m->qFeat, &m->qRange,
                 &m->qAngle, &m->sf, &m->heads, &m->sortedKp, &m->sortedDesc, &m->kps1, &m->good1, &m->good2, &m->stereoPar, &m->stereoCnt,
                 &m->stereoOut})
    b->release();
  for (auto& e : m->ev) if (e) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(m->stream);
  delete m;
}

int ydorb_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; i += 8) {
    uint64_t x, y;
    memcpy(&x, a + i, 8);
    memcpy(&y, b + i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

int ydorb_descriptor_distance_rows(ydorb_matcher_t* m, const uint8_t* a, const uint8_t* b, int32_t n, int32_t* out) {
  if (!m || !a || !b || !out || n < 0) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  if (n == 0) return YDORB_OK;
  HIPCHK(hipSetDevice(m->device));
  int rc;
  if ((rc = m->desc.ensure((size_t)32 * n)) || (rc = m->desc2.ensure((size_t)32 * n)) || (rc = m->assigned.ensure(sizeof(int) * n))) return rc;
  HIPCHK(hipMemcpyAsync(m->desc.p, a, (size_t)32 * n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->desc2.p, b, (size_t)32 * n, hipMemcpyHostToDevice, m->stream));
  hipLaunchKernelGGL(k_hamming_rows, dim3((n + 255) / 256), dim3(256), 0, m->stream, m->desc.as<uint8_t>(), m->desc2.as<uint8_t>(), n, m->assigned.as<int>());
  HIPCHK(hipMemcpyAsync(out, m->assigned.p, sizeof(int) * n, hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  return YDORB_OK;
}

int ydorb_distinctive_descriptors(ydorb_matcher_t* m, const uint8_t* desc, const int32_t* offsets, int32_t nPoints, int32_t* best) {
  if (!m || !offsets || !best || nPoints < 0) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  if (nPoints == 0) return YDORB_OK;
  const int total = offsets[nPoints];
  if (offsets[0] != 0 || total < 0 || (total > 0 && !desc)) { set_error("invalid offsets"); return YDORB_ERR_INVALID_ARG; }
  for (int p = 0; p < nPoints; p++) {
    const int cnt = offsets[p + 1] - offsets[p];
    if (cnt < 0 || cnt > 65535) { set_error("map point %d holds %d descriptors (0..65535 supported)", p, cnt); return YDORB_ERR_INVALID_ARG; }
  }
  HIPCHK(hipSetDevice(m->device));
  int rc;
  if ((rc = m->desc.ensure((size_t)32 * std::max(total, 1))) || (rc = m->qRange.ensure(sizeof(int) * (nPoints + 1))) ||
      (rc = m->assigned.ensure(sizeof(int) * nPoints)))
    return rc;
  if (total) HIPCHK(hipMemcpyAsync(m->desc.p, desc, (size_t)32 * total, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qRange.p, offsets, sizeof(int) * (nPoints + 1), hipMemcpyHostToDevice, m->stream));
  hipLaunchKernelGGL(k_distinctive, dim3((nPoints + 3) / 4), dim3(256), 0, m->stream, m->desc.as<uint8_t>(), m->qRange.as<int>(), nPoints, m->assigned.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(best, m->assigned.p, sizeof(int) * nPoints, hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  return YDORB_OK;
}

static int searchProjectionImpl(ydorb_matcher_t* m, int32_t mode, const YdFrameView* fv, const YdQuery* queries, const uint8_t* qdesc,
                                int32_t nq, float ratio, int32_t orbDist, int32_t checkOri, uint8_t* taken, int32_t* assigned,
                                int32_t* nMatches, std::vector<uint32_t>* recordsOut, const float* invSigma2 = nullptr, int nLevels = 0,
                                int32_t* bestOut = nullptr) {
  HIPCHK(hipSetDevice(m->device));
  const int n = fv->n;
  if (n > 65535) { set_error("frames with more than 65535 keypoints are not supported"); return YDORB_ERR_UNSUPPORTED; }
  if (nq == 0 || n == 0) { *nMatches = 0; return YDORB_OK; }
  for (int attempt = 0; attempt < 6; attempt++) {
    int rc;
    FrameDev F;
    if ((rc = uploadFrame(m, fv, &F))) return rc;
    if ((rc = m->queries.ensure(sizeof(YdQuery) * nq)) || (rc = m->qdesc.ensure((size_t)32 * nq)) || (rc = m->taken.ensure(n)) ||
        (rc = m->assigned.ensure(sizeof(int) * n)) || (rc = m->matchQ.ensure(sizeof(int) * nq)) || (rc = m->qInfo.ensure(sizeof(int2) * nq)) ||
        (rc = m->qPre.ensure(sizeof(uint2) * nq)) ||
        (rc = m->pool.ensure(sizeof(uint32_t) * ((size_t)nq * kSlot + m->poolRecords))) || (rc = m->calls.ensure(sizeof(CallDev))) || (rc = resetMisc(m, m->stream)))
      return rc;
    HIPCHK(hipMemcpyAsync(m->queries.p, queries, sizeof(YdQuery) * nq, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->qdesc.p, qdesc, (size_t)32 * nq, hipMemcpyHostToDevice, m->stream));
    if (taken) HIPCHK(hipMemcpyAsync(m->taken.p, taken, n, hipMemcpyHostToDevice, m->stream));
    else HIPCHK(hipMemsetAsync(m->taken.p, 0, n, m->stream));
    if (assigned) HIPCHK(hipMemcpyAsync(m->assigned.p, assigned, sizeof(int) * n, hipMemcpyHostToDevice, m->stream));
    CallDev C{};
    C.frame = 0; C.tkps = F.kps; C.qAngle = nullptr;
    C.queries = m->queries.as<QueryDev>(); C.qdesc = m->qdesc.as<uint8_t>(); C.nqPtr = nullptr; C.nq = nq;
    C.qInfo = m->qInfo.as<int2>(); C.taken = m->taken.as<uint8_t>(); C.assigned = m->assigned.as<int>(); C.matchQ = m->matchQ.as<int>();
    C.qPre = m->qPre.as<uint2>(); C.takenClear = taken ? 0 : 1;
    C.count = m->misc.as<int>() + 2; C.mode = mode; C.ratio = ratio; C.orbDist = orbDist; C.checkOri = checkOri;
    for (int i = 0; i < 8; i++) C.invSigma2[i] = (invSigma2 && i < nLevels) ? invSigma2[i] : 0.f;
    HIPCHK(hipMemcpyAsync(m->calls.p, &C, sizeof(CallDev), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_gather_projection, dim3((nq + 4 * kGatherQpw - 1) / (4 * kGatherQpw), 1), dim3(256), 0, m->stream, m->calls.as<CallDev>(), m->frames.as<FrameDev>(), nq,
                       m->pool.as<uint32_t>(), m->misc.as<unsigned>(), (unsigned)((size_t)nq * kSlot + m->poolRecords), m->misc.as<int>() + 1);
    int hmisc[3];
    if (!recordsOut) {
      const int takenWords = (n + 31) / 32;
      hipLaunchKernelGGL(k_resolve, dim3(1), dim3(64), 2 * sizeof(unsigned) * takenWords, m->stream, m->calls.as<CallDev>(), m->frames.as<FrameDev>(),
                         m->pool.as<uint32_t>(), takenWords);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hmisc, m->misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (hmisc[1] != 0) {  // record pool too small: grow and replay (the kernels wrote nothing past the pool)
      m->poolRecords = std::max<size_t>(m->poolRecords * 4, (size_t)(unsigned)hmisc[0] + 1024);
      continue;
    }
    if (recordsOut) {
      std::vector<int2> info(nq);
      HIPCHK(hipMemcpy(info.data(), m->qInfo.p, sizeof(int2) * nq, hipMemcpyDeviceToHost));
      recordsOut->resize(info[0].y);
      if (info[0].y) HIPCHK(hipMemcpy(recordsOut->data(), m->pool.as<uint32_t>() + info[0].x, sizeof(uint32_t) * info[0].y, hipMemcpyDeviceToHost));
      return YDORB_OK;
    }
    if (taken) HIPCHK(hipMemcpy(taken, m->taken.p, n, hipMemcpyDeviceToHost));
    if (assigned) HIPCHK(hipMemcpy(assigned, m->assigned.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (bestOut) HIPCHK(hipMemcpy(bestOut, m->matchQ.p, sizeof(int) * nq, hipMemcpyDeviceToHost));
    *nMatches = hmisc[2];
    return YDORB_OK;
  }
  set_error("candidate record pool kept overflowing");
  return YDORB_ERR_CAPACITY;
}

int ydorb_search_by_projection(ydorb_matcher_t* m, int32_t mode, const YdFrameView* fv, const YdQuery* queries, const uint8_t* qdesc,
                               int32_t nq, float ratio, int32_t orbDist, int32_t checkOri, uint8_t* taken, int32_t* assigned,
                               int32_t* nMatches) {
  if (!m || !fv || !nMatches || mode < 0 || (mode > 2 && mode != 7) || nq < 0 || fv->n < 0 || (nq > 0 && (!queries || !qdesc)) ||
      (fv->n > 0 && (!fv->kps || !fv->desc || !assigned)) || !(fv->max_x > fv->min_x) || !(fv->max_y > fv->min_y)) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  if (mode == 7) checkOri = 0;   // searchByProjectionInSim has no rotation-consistency step
  return searchProjectionImpl(m, mode, fv, queries, qdesc, nq, ratio, orbDist, checkOri, taken, assigned, nMatches, nullptr);
}

int ydorb_fuse_search(ydorb_matcher_t* m, const YdFrameView* fv, const YdQuery* queries, const uint8_t* qdesc, int32_t nq,
                      const float* invSigma2, int32_t nLevels, int32_t* best, int32_t* nFound) {
  return ydorb_window_search(m, fv, queries, qdesc, nq, invSigma2, nLevels, 50, best, nFound);
}

int ydorb_window_search(ydorb_matcher_t* m, const YdFrameView* fv, const YdQuery* queries, const uint8_t* qdesc, int32_t nq,
                        const float* invSigma2, int32_t nLevels, int32_t maxDist, int32_t* best, int32_t* nFound) {
  if (!m || !fv || !nFound || nq < 0 || maxDist < 0 || maxDist > 256 || fv->n < 0 || (nq > 0 && (!queries || !qdesc || !best)) || !invSigma2 || nLevels < 1 || nLevels > 8 ||
      (fv->n > 0 && (!fv->kps || !fv->desc)) || !(fv->max_x > fv->min_x) || !(fv->max_y > fv->min_y)) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  for (int q = 0; q < nq; q++) best[q] = -1;
  *nFound = 0;
  for (int i = 0; i < fv->n; i++)
    if (fv->kps[i].octave < 0 || fv->kps[i].octave >= nLevels) { set_error("keyframe feature %d: octave %d outside the %d-level table", i, fv->kps[i].octave, nLevels); return YDORB_ERR_INVALID_ARG; }
  return searchProjectionImpl(m, 6, fv, queries, qdesc, nq, 0.f, maxDist, 0, nullptr, nullptr, nFound, nullptr, invSigma2, nLevels, best);
}

int ydorb_frame_keypoints_in_area(ydorb_matcher_t* m, const YdFrameView* fv, float x, float y, float r, int32_t minLevel, int32_t maxLevel,
                                  int32_t* outIdx, int32_t cap, int32_t* nOut) {
  if (!m || !fv || !nOut || (cap > 0 && !outIdx) || fv->n < 0 || !(fv->max_x > fv->min_x) || !(fv->max_y > fv->min_y)) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  *nOut = 0;
  if (fv->n == 0) return YDORB_OK;
  YdQuery q{};
  q.u = x; q.v = y; q.r = r; q.min_level = minLevel; q.max_level = maxLevel; q.flags = 1;
  uint8_t zero[32] = {0};
  std::vector<uint32_t> rec;
  int dummy = 0;
  int rc = searchProjectionImpl(m, 2, fv, &q, zero, 1, 0.f, 0, 0, nullptr, nullptr, &dummy, &rec);
  if (rc) return rc;
  if ((int)rec.size() > cap) { set_error("%zu candidates, capacity %d", rec.size(), cap); return YDORB_ERR_CAPACITY; }
  for (size_t i = 0; i < rec.size(); i++) outIdx[i] = (int)(rec[i] & 0xFFFFu);
  *nOut = (int)rec.size();
  return YDORB_OK;
}

int ydorb_search_by_bow(ydorb_matcher_t* m, int32_t mode, const YdBowSide* A, const YdBowSide* B, float ratio, int32_t checkOri, int32_t* out,
                        int32_t* nMatches) {
  if (!m || !A || !B || !out || !nMatches || (mode != 3 && mode != 4) || A->n < 0 || B->n < 0 || (A->n > 0 && (!A->kps || !A->desc || !A->valid)) ||
      (B->n > 0 && (!B->kps || !B->desc)) || (mode == 4 && B->n > 0 && !B->valid)) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(m->device));
  const int nOutLen = mode == 3 ? B->n : A->n;
  for (int i = 0; i < nOutLen; i++) out[i] = -1;
  *nMatches = 0;
  if (A->n == 0 || B->n == 0) return YDORB_OK;
  if (B->n > 65535) { set_error("more than 65535 features per frame are not supported"); return YDORB_ERR_UNSUPPORTED; }
  // which vocabulary nodes meet: merge-join of the two ascending id lists (orbMatcher.cpp:317-361)
  std::vector<int> qFeat;
  std::vector<int2> qRange;
  std::vector<float> qAngle;
  size_t records = 0;
  {
    int a = 0, b = 0;
    const YdFeatureVector &fa = A->fv, &fb = B->fv;
    while (a < fa.n_nodes && b < fb.n_nodes) {
      if (fa.node_ids[a] == fb.node_ids[b]) {
        for (int ia = fa.node_start[a]; ia < fa.node_start[a + 1]; ia++) {
          const int idxA = fa.feat[ia];
          if (!A->valid[idxA]) continue;
          qFeat.push_back(idxA);
          qRange.push_back(make_int2(fb.node_start[b], fb.node_start[b + 1]));
          qAngle.push_back(A->kps[idxA].angle);
          records += (size_t)(fb.node_start[b + 1] - fb.node_start[b]);
        }
        a++; b++;
      } else if (fa.node_ids[a] < fb.node_ids[b]) {
        a = (int)(std::lower_bound(fa.node_ids, fa.node_ids + fa.n_nodes, fb.node_ids[b]) - fa.node_ids);
      } else {
        b = (int)(std::lower_bound(fb.node_ids, fb.node_ids + fb.n_nodes, fa.node_ids[a]) - fb.node_ids);
      }
    }
  }
  const int nq = (int)qFeat.size();
  if (nq == 0) return YDORB_OK;
  const int nFeatB = B->fv.node_start[B->fv.n_nodes];
  m->poolRecords = std::max<size_t>(m->poolRecords, records + 1024);
  int rc;
  if ((rc = m->desc.ensure((size_t)32 * A->n)) || (rc = m->desc2.ensure((size_t)32 * B->n)) || (rc = m->kps2.ensure(sizeof(YdKeyPoint) * B->n)) ||
      (rc = m->feat.ensure(sizeof(int) * std::max(nFeatB, 1))) || (rc = m->valid.ensure(B->n)) || (rc = m->qFeat.ensure(sizeof(int) * nq)) ||
      (rc = m->qRange.ensure(sizeof(int2) * nq)) || (rc = m->qAngle.ensure(sizeof(float) * nq)) || (rc = m->qInfo.ensure(sizeof(int2) * nq)) ||
      (rc = m->matchQ.ensure(sizeof(int) * nq)) || (rc = m->assigned.ensure(sizeof(int) * std::max(nq, B->n))) ||
      (rc = m->pool.ensure(sizeof(uint32_t) * m->poolRecords)) || (rc = m->calls.ensure(sizeof(CallDev))) || (rc = resetMisc(m, m->stream)))
    return rc;
  HIPCHK(hipMemcpyAsync(m->desc.p, A->desc, (size_t)32 * A->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->desc2.p, B->desc, (size_t)32 * B->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->kps2.p, B->kps, sizeof(YdKeyPoint) * B->n, hipMemcpyHostToDevice, m->stream));
  if (nFeatB) HIPCHK(hipMemcpyAsync(m->feat.p, B->fv.feat, sizeof(int) * nFeatB, hipMemcpyHostToDevice, m->stream));
  if (mode == 4) HIPCHK(hipMemcpyAsync(m->valid.p, B->valid, B->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qFeat.p, qFeat.data(), sizeof(int) * nq, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qRange.p, qRange.data(), sizeof(int2) * nq, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qAngle.p, qAngle.data(), sizeof(float) * nq, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemsetAsync(m->assigned.p, 0xFF, sizeof(int) * std::max(nq, B->n), m->stream));
  BowCallDev BC{};
  BC.descA = m->desc.as<uint8_t>(); BC.descB = m->desc2.as<uint8_t>(); BC.qFeat = m->qFeat.as<int>(); BC.qRange = m->qRange.as<int2>();
  BC.featB = m->feat.as<int>(); BC.validB = mode == 4 ? m->valid.as<uint8_t>() : nullptr; BC.nq = nq; BC.qInfo = m->qInfo.as<int2>();
  hipLaunchKernelGGL(k_gather_bow, dim3((nq + 3) / 4), dim3(256), 0, m->stream, BC, m->pool.as<uint32_t>(), m->misc.as<unsigned>(),
                     (unsigned)m->poolRecords, m->misc.as<int>() + 1);
  CallDev C{};
  C.frame = 0; C.tkps = m->kps2.as<KeyPointDev>(); C.qAngle = m->qAngle.as<float>(); C.queries = nullptr; C.qdesc = nullptr; C.nqPtr = nullptr;
  C.nq = nq; C.qInfo = m->qInfo.as<int2>(); C.taken = nullptr; C.assigned = m->assigned.as<int>(); C.matchQ = m->matchQ.as<int>();
  C.count = m->misc.as<int>() + 2; C.mode = mode; C.ratio = ratio; C.orbDist = 0; C.checkOri = checkOri;
  HIPCHK(hipMemcpyAsync(m->calls.p, &C, sizeof(CallDev), hipMemcpyHostToDevice, m->stream));
  const int takenWords = (B->n + 31) / 32;
  hipLaunchKernelGGL(k_resolve, dim3(1), dim3(64), 2 * sizeof(unsigned) * takenWords, m->stream, m->calls.as<CallDev>(), (const FrameDev*)nullptr,
                     m->pool.as<uint32_t>(), takenWords);
  HIPCHK(hipGetLastError());
  int hmisc[3];
  std::vector<int> res(std::max(nq, B->n));
  HIPCHK(hipMemcpyAsync(hmisc, m->misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipMemcpyAsync(res.data(), m->assigned.p, sizeof(int) * res.size(), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  if (hmisc[1] != 0) { set_error("bow record pool overflow"); return YDORB_ERR_CAPACITY; }
  if (mode == 3) {
    for (int i = 0; i < B->n; i++) out[i] = res[i] >= 0 ? qFeat[res[i]] : -1;
  } else {
    for (int q = 0; q < nq; q++) out[qFeat[q]] = res[q];
  }
  *nMatches = hmisc[2];
  return YDORB_OK;
}

int ydorb_search_for_triangulation(ydorb_matcher_t* m, const YdTriSide* A, const YdTriSide* B, const float* F, float ex, float ey,
                                   const float* sfB, const float* sf2B, int32_t nLevels, int32_t stereoOnly, int32_t checkOri, int32_t* out,
                                   int32_t* nMatches) {
  if (!m || !A || !B || !F || !sfB || !sf2B || !out || !nMatches || nLevels < 1 || nLevels > 8 || A->n < 0 || B->n < 0 ||
      (A->n > 0 && (!A->kps || !A->desc || !A->right_x || !A->has_map_point)) || (B->n > 0 && (!B->kps || !B->desc || !B->right_x || !B->has_map_point))) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(m->device));
  for (int i = 0; i < A->n; i++) out[i] = -1;
  *nMatches = 0;
  if (A->n == 0 || B->n == 0) return YDORB_OK;
  if (B->n > 65535) { set_error("more than 65535 features per frame are not supported"); return YDORB_ERR_UNSUPPORTED; }
  for (int i = 0; i < B->n; i++)
    if (B->kps[i].octave < 0 || B->kps[i].octave >= nLevels) { set_error("second keyframe: octave %d outside the %d-level tables", B->kps[i].octave, nLevels); return YDORB_ERR_INVALID_ARG; }
  // eligibility (:491-492, :497-500) is static: no map point yet, and a good stereo coordinate when only stereo points are wanted
  std::vector<uint8_t> goodA(A->n), goodB(B->n), validB(B->n);
  for (int i = 0; i < A->n; i++) goodA[i] = A->right_x[i] >= 0;
  for (int i = 0; i < B->n; i++) { goodB[i] = B->right_x[i] >= 0; validB[i] = !B->has_map_point[i] && (!stereoOnly || goodB[i]); }
  std::vector<int> qFeat;
  std::vector<int2> qRange;
  std::vector<float> qAngle;
  size_t records = 0;
  {
    int a = 0, b = 0;
    const YdFeatureVector &fa = A->fv, &fb = B->fv;
    while (a < fa.n_nodes && b < fb.n_nodes) {   // merge-join of the two ascending node id lists (:484-541)
      if (fa.node_ids[a] == fb.node_ids[b]) {
        for (int ia = fa.node_start[a]; ia < fa.node_start[a + 1]; ia++) {
          const int i1 = fa.feat[ia];
          if (A->has_map_point[i1] || (stereoOnly && !goodA[i1])) continue;
          qFeat.push_back(i1);
          qRange.push_back(make_int2(fb.node_start[b], fb.node_start[b + 1]));
          qAngle.push_back(A->kps[i1].angle);
          records += (size_t)(fb.node_start[b + 1] - fb.node_start[b]);
        }
        a++; b++;
      } else if (fa.node_ids[a] < fb.node_ids[b]) {
        a = (int)(std::lower_bound(fa.node_ids, fa.node_ids + fa.n_nodes, fb.node_ids[b]) - fa.node_ids);
      } else {
        b = (int)(std::lower_bound(fb.node_ids, fb.node_ids + fb.n_nodes, fa.node_ids[a]) - fb.node_ids);
      }
    }
  }
  const int nq = (int)qFeat.size();
  if (nq == 0) return YDORB_OK;
  const int nFeatB = B->fv.node_start[B->fv.n_nodes];
  m->poolRecords = std::max<size_t>(m->poolRecords, records + 1024);
  int rc;
  if ((rc = m->desc.ensure((size_t)32 * A->n)) || (rc = m->desc2.ensure((size_t)32 * B->n)) || (rc = m->kps2.ensure(sizeof(YdKeyPoint) * B->n)) ||
      (rc = m->kps1.ensure(sizeof(YdKeyPoint) * A->n)) || (rc = m->good1.ensure(A->n)) || (rc = m->good2.ensure(B->n)) ||
      (rc = m->feat.ensure(sizeof(int) * std::max(nFeatB, 1))) || (rc = m->valid.ensure(B->n)) || (rc = m->qFeat.ensure(sizeof(int) * nq)) ||
      (rc = m->qRange.ensure(sizeof(int2) * nq)) || (rc = m->qAngle.ensure(sizeof(float) * nq)) || (rc = m->qInfo.ensure(sizeof(int2) * nq)) ||
      (rc = m->matchQ.ensure(sizeof(int) * nq)) || (rc = m->assigned.ensure(sizeof(int) * std::max(nq, B->n))) ||
      (rc = m->pool.ensure(sizeof(uint32_t) * m->poolRecords)) || (rc = m->calls.ensure(sizeof(CallDev))) || (rc = resetMisc(m, m->stream)))
    return rc;
  HIPCHK(hipMemcpyAsync(m->desc.p, A->desc, (size_t)32 * A->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->desc2.p, B->desc, (size_t)32 * B->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->kps1.p, A->kps, sizeof(YdKeyPoint) * A->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->kps2.p, B->kps, sizeof(YdKeyPoint) * B->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->good1.p, goodA.data(), A->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->good2.p, goodB.data(), B->n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->valid.p, validB.data(), B->n, hipMemcpyHostToDevice, m->stream));
  if (nFeatB) HIPCHK(hipMemcpyAsync(m->feat.p, B->fv.feat, sizeof(int) * nFeatB, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qFeat.p, qFeat.data(), sizeof(int) * nq, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qRange.p, qRange.data(), sizeof(int2) * nq, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(m->qAngle.p, qAngle.data(), sizeof(float) * nq, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemsetAsync(m->assigned.p, 0xFF, sizeof(int) * std::max(nq, B->n), m->stream));
  BowCallDev BC{};
  BC.descA = m->desc.as<uint8_t>(); BC.descB = m->desc2.as<uint8_t>(); BC.qFeat = m->qFeat.as<int>(); BC.qRange = m->qRange.as<int2>();
  BC.featB = m->feat.as<int>(); BC.validB = m->valid.as<uint8_t>(); BC.nq = nq; BC.qInfo = m->qInfo.as<int2>();
  BC.tri = 1; BC.kpsA = m->kps1.as<KeyPointDev>(); BC.kpsB = m->kps2.as<KeyPointDev>(); BC.goodA = m->good1.as<uint8_t>(); BC.goodB = m->good2.as<uint8_t>();
  for (int i = 0; i < 9; i++) BC.F[i] = F[i];
  BC.ex = ex; BC.ey = ey;
  for (int i = 0; i < 8; i++) { BC.sfB[i] = i < nLevels ? sfB[i] : 0.f; BC.sf2B[i] = i < nLevels ? sf2B[i] : 0.f; }
  hipLaunchKernelGGL(k_gather_bow, dim3((nq + 3) / 4), dim3(256), 0, m->stream, BC, m->pool.as<uint32_t>(), m->misc.as<unsigned>(),
                     (unsigned)m->poolRecords, m->misc.as<int>() + 1);
  CallDev C{};
  C.frame = 0; C.tkps = m->kps2.as<KeyPointDev>(); C.qAngle = m->qAngle.as<float>(); C.queries = nullptr; C.qdesc = nullptr; C.nqPtr = nullptr;
  C.nq = nq; C.qInfo = m->qInfo.as<int2>(); C.taken = nullptr; C.assigned = m->assigned.as<int>(); C.matchQ = m->matchQ.as<int>();
  C.count = m->misc.as<int>() + 2; C.mode = 5; C.ratio = 0.f; C.orbDist = 0; C.checkOri = checkOri;
  HIPCHK(hipMemcpyAsync(m->calls.p, &C, sizeof(CallDev), hipMemcpyHostToDevice, m->stream));
  const int takenWords = (B->n + 31) / 32;
  hipLaunchKernelGGL(k_resolve, dim3(1), dim3(64), 2 * sizeof(unsigned) * takenWords, m->stream, m->calls.as<CallDev>(), (const FrameDev*)nullptr,
                     m->pool.as<uint32_t>(), takenWords);
  HIPCHK(hipGetLastError());
  int hmisc[3];
  std::vector<int> res(std::max(nq, B->n));
  HIPCHK(hipMemcpyAsync(hmisc, m->misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipMemcpyAsync(res.data(), m->assigned.p, sizeof(int) * res.size(), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  if (hmisc[1] != 0) { set_error("bow record pool overflow"); return YDORB_ERR_CAPACITY; }
  for (int q = 0; q < nq; q++) out[qFeat[q]] = res[q];
  *nMatches = hmisc[2];
  return YDORB_OK;
}

int ydorb_stereo_matches(ydorb_matcher_t* m, const YdStereoSide* L, const YdStereoSide* R, int32_t nPairs, float bf, float b, int32_t flags,
                         float* rightX, float* depth, int32_t* nKept, int32_t* status, void* stream) {
  if (!m || !L || !R || !L->extractor || !R->extractor || !L->kps || !L->desc || !L->n || !R->kps || !R->desc || !R->n || !rightX || !depth ||
      nPairs < 1 || L->cap < 1 || R->cap < 1 || !(bf > 0.f) || !(b > 0.f)) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  if (R->cap > kStereoMaxRight) { set_error("right cap %d > %d", R->cap, kStereoMaxRight); return YDORB_ERR_CAPACITY; }
  PyramidView vl, vr;
  int rc;
  if ((rc = extractor_pyramid_view(L->extractor, &vl)) || (rc = extractor_pyramid_view(R->extractor, &vr))) return rc;
  if (vl.device != m->device || vr.device != m->device) { set_error("extractors and matcher live on different devices"); return YDORB_ERR_INVALID_ARG; }
  if (vl.nLevels != vr.nLevels) { set_error("left and right pyramids differ in depth"); return YDORB_ERR_INVALID_ARG; }
  for (int l = 0; l < vl.nLevels; l++)
    if (vl.w[l] != vr.w[l] || vl.h[l] != vr.h[l] || vl.scale[l] != vr.scale[l]) { set_error("left and right pyramids differ at level %d", l); return YDORB_ERR_INVALID_ARG; }
  const int lastL = L->first_frame + (nPairs - 1) * L->frame_step, lastR = R->first_frame + (nPairs - 1) * R->frame_step;
  if (L->first_frame < 0 || lastL < 0 || L->first_frame >= vl.frames || lastL >= vl.frames || R->first_frame < 0 || lastR < 0 ||
      R->first_frame >= vr.frames || lastR >= vr.frames) {
    set_error("pair frames outside the extractors' last call (%d / %d frames)", vl.frames, vr.frames);
    return YDORB_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(m->device));
  const bool dev = flags & YDORB_STEREO_DEVICE_POINTERS;
  hipStream_t s = stream ? (hipStream_t)stream : m->stream;
  const size_t nl = (size_t)nPairs * L->cap, nr = (size_t)nPairs * R->cap;
  if ((rc = m->stereoPar.ensure(sizeof(StereoDev))) || (rc = m->stereoCnt.ensure(sizeof(int) * 4 * nPairs)) ||
      (rc = m->stereoOut.ensure(sizeof(int) * 2 * nPairs)))
    return rc;
  StereoDev P{};
  if (dev) {
    P.kpsL = reinterpret_cast<const KeyPointDev*>(L->kps); P.descL = L->desc; P.nL = L->n;
    P.kpsR = reinterpret_cast<const KeyPointDev*>(R->kps); P.descR = R->desc; P.nR = R->n;
    P.rightX = rightX; P.depth = depth;
  } else {
    // the pyramids must be complete before the kernels read them from another stream
    HIPCHK(hipStreamSynchronize((hipStream_t)vl.stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)vr.stream));
    if ((rc = m->kps1.ensure(sizeof(YdKeyPoint) * nl)) || (rc = m->desc.ensure(32 * nl)) || (rc = m->kps2.ensure(sizeof(YdKeyPoint) * nr)) ||
        (rc = m->desc2.ensure(32 * nr)) || (rc = m->qRange.ensure(sizeof(int) * 2 * nPairs)) || (rc = m->rightX.ensure(sizeof(float) * 2 * nl)))
      return rc;
    HIPCHK(hipMemcpyAsync(m->kps1.p, L->kps, sizeof(YdKeyPoint) * nl, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->desc.p, L->desc, 32 * nl, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->kps2.p, R->kps, sizeof(YdKeyPoint) * nr, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->desc2.p, R->desc, 32 * nr, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->qRange.p, L->n, sizeof(int) * nPairs, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->qRange.as<int>() + nPairs, R->n, sizeof(int) * nPairs, hipMemcpyHostToDevice, s));
    P.kpsL = m->kps1.as<KeyPointDev>(); P.descL = m->desc.as<uint8_t>(); P.nL = m->qRange.as<int>();
    P.kpsR = m->kps2.as<KeyPointDev>(); P.descR = m->desc2.as<uint8_t>(); P.nR = m->qRange.as<int>() + nPairs;
    P.rightX = m->rightX.as<float>(); P.depth = m->rightX.as<float>() + nl;
  }
  for (int l = 0; l < vl.nLevels; l++) {
    P.pyrL[l] = vl.roi[l]; P.pyrR[l] = vr.roi[l];
    P.w[l] = vl.w[l]; P.h[l] = vl.h[l]; P.pitchL[l] = vl.pitch[l]; P.pitchR[l] = vr.pitch[l];
    P.scale[l] = vl.scale[l]; P.invScale[l] = vl.invScale[l];
  }
  P.frameStrideL = vl.frameStride; P.frameStrideR = vr.frameStride;
  P.capL = L->cap; P.capR = R->cap; P.frameL0 = L->first_frame; P.frameLStep = L->frame_step; P.frameR0 = R->first_frame; P.frameRStep = R->frame_step;
  P.nLevels = vl.nLevels; P.flags = flags; P.bf = bf; P.maxD = bf / b;   // :382
  P.counters = m->stereoCnt.as<int>(); P.keptOut = m->stereoOut.as<int>(); P.statusOut = m->stereoOut.as<int>() + nPairs;
  // replay form: the right-keypoint table (8 bytes each) + the index sorted by first band row (row starts, fill cursors, 2 bytes per keypoint)
  size_t ldsReplay = (size_t)R->cap * 8;
  {
    const size_t bytes = (size_t)R->cap * 8 + sizeof(int) * (2 * (size_t)vl.h[0] + 1) + 2 * (size_t)R->cap + 16;
    const bool noLists = getenv("YDORB_STEREO_NO_ROW_LISTS") != nullptr;   // diagnostic: force the scan form (tests compare the two)
    if (!noLists && !(flags & YDORB_STEREO_INDEX_BY_KEYPOINT) && bytes <= 120 * 1024 && vl.h[0] < 4095) { P.rowLists = 1; ldsReplay = bytes; }
  }
  // P travels as a kernel argument: an asynchronous copy out of pageable host memory makes the host wait for everything queued on
  // the stream before it - here the extraction the association waits for - and a caller that pipelines steps would run in lock step
  HIPCHK(hipMemsetAsync(m->stereoCnt.p, 0, sizeof(int) * 4 * nPairs, s));
  HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.rightX), 0xBF800000u, nl, s));   // -1.0f, :363-364
  HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(P.depth), 0xBF800000u, nl, s));
  const size_t lds = (size_t)R->cap * 8;
  const StereoDev& dP = P;
  if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stereo<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (ldsReplay > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stereo<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsReplay));
  if (flags & YDORB_STEREO_INDEX_BY_KEYPOINT)
    hipLaunchKernelGGL(k_stereo<false>, dim3((L->cap + kStereoChunk - 1) / kStereoChunk, nPairs), dim3(256), lds, s, dP, 0, 0);
  else {
    // slices of the serial walk (see k_stereo): a batch keeps every launch below ~2 ms; a single pair (the adapter's call) is one launch
    static const int sliceEnv = getenv("YDORB_STEREO_SLICE") ? atoi(getenv("YDORB_STEREO_SLICE")) : 0;
    const int slice = sliceEnv > 0 ? sliceEnv : (nPairs >= 8 ? 1024 : L->cap);
    for (int k0 = 0; k0 < L->cap; k0 += slice)
      hipLaunchKernelGGL(k_stereo<true>, dim3(1, nPairs), dim3(256), ldsReplay, s, dP, k0, std::min(k0 + slice, L->cap));
  }
  hipLaunchKernelGGL(k_stereo_outliers, dim3(nPairs), dim3(256), 0, s, P);
  HIPCHK(hipGetLastError());
  if (dev) {
    if (nKept) HIPCHK(hipMemcpyAsync(nKept, P.keptOut, sizeof(int) * nPairs, hipMemcpyDeviceToDevice, s));
    if (status) HIPCHK(hipMemcpyAsync(status, P.statusOut, sizeof(int) * nPairs, hipMemcpyDeviceToDevice, s));
    return YDORB_OK;
  }
  HIPCHK(hipMemcpyAsync(rightX, P.rightX, sizeof(float) * nl, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(depth, P.depth, sizeof(float) * nl, hipMemcpyDeviceToHost, s));
  if (nKept) HIPCHK(hipMemcpyAsync(nKept, P.keptOut, sizeof(int) * nPairs, hipMemcpyDeviceToHost, s));
  if (status) HIPCHK(hipMemcpyAsync(status, P.statusOut, sizeof(int) * nPairs, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return YDORB_OK;
}

int ydorb_match_pairs_device(ydorb_matcher_t* m, const YdFrameSetDev* Q, const YdFrameSetDev* T, const int32_t* pairs, int32_t nCalls,
                             int32_t width, int32_t height, float th, const float* scaleFactors, int32_t nLevels, const float* d_affine,
                             int32_t checkOri, int32_t* d_assigned, int32_t* d_counts, void* stream) {
  if (!m || !Q || !T || !pairs || !Q->d_kps || !Q->d_desc || !Q->d_n || !T->d_kps || !T->d_desc || !T->d_n || !d_assigned || !d_counts || !scaleFactors ||
      Q->cap < 1 || Q->cap > 65535 || T->cap != Q->cap || Q->n_frames < 1 || T->n_frames < 1 || nCalls < 1 || nLevels < 1 || nLevels > 8 || width < 1 ||
      height < 1) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  for (int c = 0; c < nCalls; c++)
    if (pairs[2 * c] < 0 || pairs[2 * c] >= Q->n_frames || pairs[2 * c + 1] < 0 || pairs[2 * c + 1] >= T->n_frames) {
      set_error("pair %d references a frame out of range", c);
      return YDORB_ERR_INVALID_ARG;
    }
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = stream ? (hipStream_t)stream : m->stream;
  // grids are built only for the frames that appear as a pair's TARGET: on the multi-GPU path T is the all-gathered set (world x F
  // frames) of which this rank searches its own F
  std::vector<int> tmap(T->n_frames, -1);
  std::vector<int> tused;
  for (int c = 0; c < nCalls; c++) {
    const int tf = pairs[2 * c + 1];
    if (tmap[tf] < 0) { tmap[tf] = (int)tused.size(); tused.push_back(tf); }
  }
  const int cap = Q->cap, nFrames = (int)tused.size();
  const size_t poolPerCall = (size_t)cap * kSlot + (size_t)cap * m->ovfPerKeypoint;  // fixed slots + overflow region (records of queries with > kSlot candidates)
  int rc;
  if ((rc = m->queries.ensure(sizeof(QueryDev) * (size_t)cap * nCalls)) || (rc = m->taken.ensure((size_t)cap * nCalls)) ||
      (rc = m->matchQ.ensure(sizeof(int) * (size_t)cap * nCalls)) || (rc = m->qInfo.ensure(sizeof(int2) * (size_t)cap * nCalls)) ||
      (rc = m->qPre.ensure(sizeof(uint2) * (size_t)cap * nCalls)) ||
      (rc = m->cellStart.ensure(sizeof(int) * (size_t)(kGridCells + 1) * nFrames)) || (rc = m->cellIdx.ensure(sizeof(int) * (size_t)cap * nFrames)) ||
      (rc = m->pool.ensure(sizeof(uint32_t) * poolPerCall * nCalls)) || (rc = m->frames.ensure(sizeof(FrameDev) * nFrames)) ||
      (rc = m->calls.ensure(sizeof(CallDev) * nCalls)) || (rc = m->sf.ensure(sizeof(float) * 8 + sizeof(float) * 6 * nCalls)) || (rc = m->misc.ensure(64)) ||
      (rc = m->heads.ensure(sizeof(unsigned) * nCalls)) || (rc = m->sortedKp.ensure(sizeof(float4) * (size_t)cap * nFrames)) ||
      (rc = m->sortedDesc.ensure((size_t)32 * cap * nFrames)))
    return rc;
  std::vector<FrameDev> hf(nFrames);
  std::vector<CallDev> hc(nCalls);
  const float minX = 0.f, minY = 0.f, maxX = (float)width, maxY = (float)height;  // Frame::computeImageBounds without distortion
  for (int f = 0; f < nFrames; f++) {
    const int tf = tused[f];
    FrameDev F{};
    F.kps = reinterpret_cast<const KeyPointDev*>(T->d_kps) + (size_t)tf * cap; F.desc = T->d_desc + (size_t)tf * cap * 32; F.rightX = nullptr;
    F.nPtr = T->d_n + tf; F.n = 0; F.minX = minX; F.minY = minY;
    F.gridWInv = static_cast<float>(kGridCols) / (maxX - minX); F.gridHInv = static_cast<float>(kGridRows) / (maxY - minY);
    F.cellStart = m->cellStart.as<int>() + (size_t)f * (kGridCells + 1); F.cellIdx = m->cellIdx.as<int>() + (size_t)f * cap;
    F.sortedKp = m->sortedKp.as<float4>() + (size_t)f * cap; F.sortedDesc = m->sortedDesc.as<uint8_t>() + (size_t)f * cap * 32;
    hf[f] = F;
  }
  for (int c = 0; c < nCalls; c++) {
    const int qf = pairs[2 * c], tf = tmap[pairs[2 * c + 1]];
    CallDev C{};
    C.frame = tf; C.tkps = hf[tf].kps; C.qAngle = nullptr; C.queries = m->queries.as<QueryDev>() + (size_t)c * cap;
    C.qkps = reinterpret_cast<const KeyPointDev*>(Q->d_kps) + (size_t)qf * cap;
    C.qdesc = Q->d_desc + (size_t)qf * cap * 32; C.nqPtr = Q->d_n + qf; C.nq = 0; C.qInfo = m->qInfo.as<int2>() + (size_t)c * cap;
    C.qPre = m->qPre.as<uint2>() + (size_t)c * cap; C.takenClear = 1;
    C.taken = m->taken.as<uint8_t>() + (size_t)c * cap; C.assigned = d_assigned + (size_t)c * cap; C.matchQ = m->matchQ.as<int>() + (size_t)c * cap;
    C.count = d_counts + c; C.mode = 1; C.ratio = 0.9f; C.orbDist = 0; C.checkOri = checkOri;
    hc[c] = C;
  }
  float hsf[8] = {0};
  for (int l = 0; l < nLevels; l++) hsf[l] = scaleFactors[l];
  const bool same = m->hFrames.size() == hf.size() && m->hCalls.size() == hc.size() && !memcmp(m->hFrames.data(), hf.data(), sizeof(FrameDev) * hf.size()) &&
                    !memcmp(m->hCalls.data(), hc.data(), sizeof(CallDev) * hc.size()) && !memcmp(m->hSf, hsf, sizeof(hsf)) && m->hIdentAffine == !d_affine;
  if (!same) {
    HIPCHK(hipStreamSynchronize(s));
    m->hFrames = hf;
    m->hCalls = hc;
    memcpy(m->hSf, hsf, sizeof(hsf));
    m->hIdentAffine = !d_affine;
    HIPCHK(hipMemcpy(m->frames.p, m->hFrames.data(), sizeof(FrameDev) * nFrames, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->calls.p, m->hCalls.data(), sizeof(CallDev) * nCalls, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->sf.p, hsf, sizeof(hsf), hipMemcpyHostToDevice));
    if (!d_affine) {
      std::vector<float> ident((size_t)6 * nCalls, 0.f);
      for (int c = 0; c < nCalls; c++) { ident[6 * c] = 1.f; ident[6 * c + 4] = 1.f; }
      HIPCHK(hipMemcpy(m->sf.as<float>() + 8, ident.data(), sizeof(float) * ident.size(), hipMemcpyHostToDevice));
    }
  }
  const float* aff = d_affine ? d_affine : m->sf.as<float>() + 8;
  const bool prof = m->profiling;
  collect(m);
  // per-call pool heads, taken flags and the all -1 assignment are written by k_queries_from_keypoints (the overflow status, misc[1], is
  // sticky until synchronize reads it)
  if (prof) HIPCHK(hipEventRecord(m->ev[0], s));
  hipLaunchKernelGGL(k_queries_from_keypoints, dim3((cap + 255) / 256, nCalls), dim3(256), 0, s, m->calls.as<CallDev>(), cap, aff, th, m->sf.as<float>(),
                     nLevels, minX, maxX, minY, maxY, m->heads.as<unsigned>());
  { const int rcg = launchGridBuild(nFrames, cap, s, m->frames.as<FrameDev>()); if (rcg) return rcg; }
  if (prof) HIPCHK(hipEventRecord(m->ev[1], s));
  hipLaunchKernelGGL(k_gather_projection, dim3((cap + 4 * kGatherQpw - 1) / (4 * kGatherQpw), nCalls), dim3(256), 0, s, m->calls.as<CallDev>(), m->frames.as<FrameDev>(), cap,
                     m->pool.as<uint32_t>(), m->heads.as<unsigned>(), (unsigned)poolPerCall, m->misc.as<int>() + 1);
  if (prof) HIPCHK(hipEventRecord(m->ev[2], s));
  const int takenWords = (cap + 31) / 32;
  hipLaunchKernelGGL(k_resolve, dim3(nCalls), dim3(64), 2 * sizeof(unsigned) * takenWords, s, m->calls.as<CallDev>(), m->frames.as<FrameDev>(),
                     m->pool.as<uint32_t>(), takenWords);
  if (prof) { HIPCHK(hipEventRecord(m->ev[3], s)); m->evPending = true; }
  HIPCHK(hipGetLastError());
  return YDORB_OK;
}

int ydorb_match_consecutive_device(ydorb_matcher_t* m, const YdKeyPoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int32_t cap,
                                   int32_t nFrames, int32_t width, int32_t height, float th, const float* scaleFactors, int32_t nLevels,
                                   const float* d_affine, int32_t checkOri, int32_t* d_assigned, int32_t* d_counts, void* stream) {
  if (!m || nFrames < 2) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  const YdFrameSetDev S{d_kps, d_desc, d_n, nFrames, cap};
  if ((int)m->consecPairs.size() != 2 * (nFrames - 1)) {
    m->consecPairs.resize((size_t)2 * (nFrames - 1));
    for (int c = 0; c < nFrames - 1; c++) { m->consecPairs[2 * c] = c; m->consecPairs[2 * c + 1] = c + 1; }
  }
  return ydorb_match_pairs_device(m, &S, &S, m->consecPairs.data(), nFrames - 1, width, height, th, scaleFactors, nLevels, d_affine, checkOri, d_assigned,
                                  d_counts, stream);
}

int ydorb_hamming_topk(ydorb_matcher_t* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, const int32_t* candOffsets,
                       const int32_t* candIdx, YdMatch2* out) {
  static_assert(sizeof(YdMatch2) == sizeof(TopkOut), "YdMatch2 layout");
  if (!m || nq < 0 || nt < 0 || nt > 65535 || (nq && (!q || !out)) || (nt && !t) || ((candOffsets == nullptr) != (candIdx == nullptr))) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  if (nq == 0) return YDORB_OK;
  int rc = require_device(m->device);
  if (rc) return rc;
  HIPCHK(hipSetDevice(m->device));
  size_t nCand = 0;
  if (candOffsets) {
    if (candOffsets[0] != 0) { set_error("cand_offsets[0] must be 0"); return YDORB_ERR_INVALID_ARG; }
    for (int i = 0; i < nq; i++) {
      const int n = candOffsets[i + 1] - candOffsets[i];
      if (n < 0 || n > 65535) { set_error("candidate list %d has %d entries (0..65535 supported)", i, n); return YDORB_ERR_INVALID_ARG; }
    }
    nCand = (size_t)candOffsets[nq];
  }
  hipStream_t s = m->stream;
  if ((rc = m->qdesc.ensure((size_t)32 * nq)) || (rc = m->desc.ensure((size_t)32 * std::max(nt, 1))) || (rc = m->pool.ensure(sizeof(TopkOut) * (size_t)nq)) ||
      (rc = m->cellStart.ensure(sizeof(int) * ((size_t)nq + 1))) || (rc = m->cellIdx.ensure(sizeof(int) * std::max<size_t>(nCand, 1))))
    return rc;
  HIPCHK(hipMemcpyAsync(m->qdesc.p, q, (size_t)32 * nq, hipMemcpyHostToDevice, s));
  if (nt) HIPCHK(hipMemcpyAsync(m->desc.p, t, (size_t)32 * nt, hipMemcpyHostToDevice, s));
  if (candOffsets) {
    HIPCHK(hipMemcpyAsync(m->cellStart.p, candOffsets, sizeof(int) * ((size_t)nq + 1), hipMemcpyHostToDevice, s));
    if (nCand) HIPCHK(hipMemcpyAsync(m->cellIdx.p, candIdx, sizeof(int) * nCand, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(k_topk_csr, dim3((nq + 3) / 4), dim3(256), 0, s, m->qdesc.as<uint8_t>(), nq, m->desc.as<uint8_t>(), nt,
                     candOffsets ? m->cellStart.as<int>() : nullptr, candOffsets ? m->cellIdx.as<int>() : nullptr, m->pool.as<TopkOut>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, m->pool.p, sizeof(TopkOut) * (size_t)nq, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return YDORB_OK;
}

int ydorb_hamming_topk_device(ydorb_matcher_t* m, const uint8_t* d_qdesc, const int32_t* d_nq, const uint8_t* d_tdesc, const int32_t* d_nt, int32_t cap,
                              int32_t nPairs, YdMatch2* d_out, void* stream) {
  if (!m || !d_qdesc || !d_nq || !d_tdesc || !d_nt || !d_out || cap < 1 || cap > 65535 || nPairs < 1) { set_error("invalid argument"); return YDORB_ERR_INVALID_ARG; }
  int rc = require_device(m->device);
  if (rc) return rc;
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = stream ? (hipStream_t)stream : m->stream;
  hipLaunchKernelGGL(k_topk_allpairs, dim3((cap + 255) / 256, nPairs), dim3(256), 0, s, d_qdesc, d_nq, (size_t)cap * 32, d_tdesc, d_nt, (size_t)cap * 32, cap,
                     reinterpret_cast<TopkOut*>(d_out));
  HIPCHK(hipGetLastError());
  return YDORB_OK;
}

int ydorb_matcher_synchronize(ydorb_matcher_t* m) {
  if (!m) { set_error("null handle"); return YDORB_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipDeviceSynchronize());
  collect(m);
  if (m->misc.p) {
    int hmisc[2];
    HIPCHK(hipMemcpy(hmisc, m->misc.p, sizeof(hmisc), hipMemcpyDeviceToHost));
    if (hmisc[1] != 0) {
      (void)hipMemset(m->misc.p, 0, 8);
      m->ovfPerKeypoint *= 4;   // the next batched call gets a larger overflow region
      set_error("candidate record pool overflow in the batched search: results of that call are incomplete; the overflow region is now %d records per keypoint, call again",
                m->ovfPerKeypoint);
      return YDORB_ERR_CAPACITY;
    }
  }
  return YDORB_OK;
}

int ydorb_matcher_set_profiling(ydorb_matcher_t* m, int32_t on) {
  if (!m) return YDORB_ERR_INVALID_ARG;
  m->profiling = on != 0;
  for (double& v : m->stageMs) v = 0;
  m->stageCalls = 0;
  m->evPending = false;
  return YDORB_OK;
}

int ydorb_matcher_stage_times(ydorb_matcher_t* m, int32_t maxStages, const char** names, float* ms, int32_t* nStages) {
  if (!m || !nStages) return YDORB_ERR_INVALID_ARG;
  const int n = std::min<int>(maxStages, MS_COUNT);
  for (int i = 0; i < n; i++) {
    if (names) names[i] = kMatchStageNames[i];
    if (ms) ms[i] = m->stageCalls ? (float)(m->stageMs[i] / m->stageCalls) : 0.f;
  }
  *nStages = n;
  return YDORB_OK;
}

}  // extern "C"
