// Bag-of-words transform on the MI355X + its C ABI (include/ydorb/c_api.h, "Vocabulary").
// Replaces DBoW3::Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (reference
// thirdParty/DBow3/src/Vocabulary.cpp:752-824; per-feature descent :836-874; BowVector.cpp:31-88; FeatureVector.cpp:31-45) as called
// by Frame::computeBoW (src/frame.cpp:265-272), for a batch of frames per call.
//
//   k_bow_descend : one thread per descriptor walks the tree: at every level the Hamming distance to each child in Node::children
//                   order, first minimum wins (:858-865); records word, weight and the node at level L - levelsup.  Neighbouring
//                   threads diverge in what they read, but the upper levels of the tree live in L2 and a descriptor needs only
//                   k * L (= 60 for the ORB vocabulary) distances.
//   k_bow_assemble: one workgroup per frame turns the per-feature (word, node) records into the two std::map-ordered containers:
//                   bitonic sort of (id << 13 | feature) keys in LDS, run boundaries by a block scan, then per word the value the
//                   reference's sequence of `+=` produces (the same weight added count times, in order) and the normalisation
//                   with its sequential sum in word order (one thread; the order of a floating-point sum is part of the result).
// No CPU fallback; the host only uploads the tree and sizes buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/ydorb/c_api.h"
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

constexpr int kBowMaxFeatures = 8192;   // per frame: 64-bit sort keys of one frame stay in LDS (64 KB)
constexpr int kFeatBits = 13;

struct BowTreeDev {
  const int* childBegin; const int* childIds; const uint8_t* nodeDesc; const double* nodeWeight; const int* nodeWord;
  int nNodes, L;
};

__global__ __launch_bounds__(256) void k_bow_descend(BowTreeDev T, const uint8_t* __restrict__ desc, const int* __restrict__ nFeat, int cap,
                                                     int levelsup, int* __restrict__ word, int* __restrict__ node, double* __restrict__ weight,
                                                     int* __restrict__ status) {
  const int frame = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  if (f >= min(nFeat[frame], cap)) return;
  const size_t at = (size_t)frame * cap + f;
  const uint4 a0 = *reinterpret_cast<const uint4*>(desc + at * 32), a1 = *reinterpret_cast<const uint4*>(desc + at * 32 + 16);
  const int nidLevel = T.L - levelsup;   // :845
  int fin = 0, level = 0, nid = nidLevel <= 0 ? 0 : -1;
  int cb = T.childBegin[0], ce = T.childBegin[1];
  do {
    ++level;
    unsigned best = 0xFFFFFFFFu;
    for (int c = cb; c < ce; c++) {
      const int id = T.childIds[c];
      const uint4 b0 = *reinterpret_cast<const uint4*>(T.nodeDesc + (size_t)id * 32), b1 = *reinterpret_cast<const uint4*>(T.nodeDesc + (size_t)id * 32 + 16);
      const unsigned d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      if (d < best) { best = d; fin = id; }   // first minimum in children order
    }
    if (level == nidLevel) nid = fin;
    cb = T.childBegin[fin]; ce = T.childBegin[fin + 1];
  } while (ce > cb);
  if (nid < 0) { nid = fin; atomicOr(&status[frame], 1); }   // leaf above level L - levelsup: `nid` is read uninitialised in the reference (:777)
  word[at] = T.nodeWord[fin];
  node[at] = nid;
  weight[at] = T.nodeWeight[fin];
}

struct BowOut {
  int* bowWord; double* bowValue; int* nWords;     // [frames][cap], [frames]
  int* fvNode; int* fvStart; int* fvFeat; int* nFvNodes;   // [frames][cap], [frames][cap + 1], [frames][cap], [frames]
};

// in-LDS bitonic sort of n2 (power of two) 64-bit keys, ascending
__device__ void bitonic_sort(unsigned long long* key, int n2) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n2 >> 1); t += 256) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// exclusive scan of one int per thread over the 256 threads; returns the thread's offset, *total = sum
__device__ int block_scan_excl(int v, int* wsum, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wv; w++) off += wsum[w];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return off + x - v;
}

// Runs of equal ids in the sorted keys -> (id, start) lists.  ids[e], starts[e] for run e; starts[nRuns] = m.  Returns nRuns.
__device__ int run_starts(const unsigned long long* key, int m, int n2, int* ids, int* starts, int* wsum) {
  const int per = n2 / 256 > 0 ? n2 / 256 : 1;           // contiguous keys per thread
  const int i0 = threadIdx.x * per, i1 = min(i0 + per, m);
  int cnt = 0;
  for (int i = i0; i < i1; i++) cnt += i == 0 || (key[i] >> kFeatBits) != (key[i - 1] >> kFeatBits);
  int total;
  int e = block_scan_excl(cnt, wsum, &total);
  for (int i = i0; i < i1; i++)
    if (i == 0 || (key[i] >> kFeatBits) != (key[i - 1] >> kFeatBits)) { ids[e] = (int)(key[i] >> kFeatBits); starts[e] = i; e++; }
  if (threadIdx.x == 0) starts[total] = m;
  __syncthreads();
  return total;
}

// grid (frames), 256 threads, dynamic LDS = n2 * 8 bytes (n2 = cap rounded up to a power of two)
__global__ __launch_bounds__(256) void k_bow_assemble(const int* __restrict__ word, const int* __restrict__ node, const double* __restrict__ weight,
                                                      const int* __restrict__ nFeat, int cap, int n2, int weighting, int norm, BowOut O) {
  extern __shared__ unsigned long long key[];
  __shared__ int wsum[4];
  __shared__ double sNorm;
  const int frame = blockIdx.x, tid = threadIdx.x;
  const int n = min(nFeat[frame], cap);
  const size_t base = (size_t)frame * cap;
  int* ids = O.bowWord + base;
  int* starts = O.fvStart + (size_t)frame * (cap + 1);   // scratch for the word runs first, the feature-vector offsets afterwards
  // ---- BowVector: words in ascending id (std::map order), features of a word in ascending index (insertion order) ----
  int mLocal = 0;
  for (int f = tid; f < n2; f += 256) {
    const bool live = f < n && weight[base + f] > 0;   // `if (w > 0)`: stopped words add nothing (:781, :809)
    key[f] = live ? ((unsigned long long)(unsigned)word[base + f] << kFeatBits) | (unsigned)f : ~0ull;
    mLocal += live;
  }
  __syncthreads();
  int m;
  (void)block_scan_excl(mLocal, wsum, &m);
  bitonic_sort(key, n2);
  const int nWords = run_starts(key, m, n2, ids, starts, wsum);
  for (int e = tid; e < nWords; e += 256) {
    const int cnt = starts[e + 1] - starts[e];
    const double w = weight[base + (int)(key[starts[e]] & ((1u << kFeatBits) - 1))];
    double v = w;                                   // BowVector::addWeight: insert, then += per further feature (BowVector.cpp:31-43)
    if (weighting <= 1) for (int r = 1; r < cnt; r++) v += w;   // TF_IDF / TF; IDF / BINARY keep the first (addIfNotExist, :47-56)
    if (weighting <= 1 && norm == 0) v /= (double)nWords;       // Vocabulary.cpp:789-795
    O.bowValue[base + e] = v;
  }
  __syncthreads();
  if (norm != 0) {   // BowVector::normalize (BowVector.cpp:60-88): sequential sum in word order
    if (tid == 0) {
      double s = 0.0;
      if (norm == 1) for (int e = 0; e < nWords; e++) s += fabs(O.bowValue[base + e]);
      else {
        for (int e = 0; e < nWords; e++) s += O.bowValue[base + e] * O.bowValue[base + e];
        s = sqrt(s);
      }
      sNorm = s;
    }
    __syncthreads();
    if (sNorm > 0.0) for (int e = tid; e < nWords; e += 256) O.bowValue[base + e] /= sNorm;
  }
  if (tid == 0) O.nWords[frame] = nWords;
  __syncthreads();
  // ---- FeatureVector: nodes in ascending id, each with its features in ascending index (FeatureVector.cpp:31-45) ----
  for (int f = tid; f < n2; f += 256) {
    const bool live = f < n && weight[base + f] > 0;
    key[f] = live ? ((unsigned long long)(unsigned)node[base + f] << kFeatBits) | (unsigned)f : ~0ull;
  }
  __syncthreads();
  bitonic_sort(key, n2);
  const int nNodes = run_starts(key, m, n2, O.fvNode + base, starts, wsum);
  for (int i = tid; i < m; i += 256) O.fvFeat[base + i] = (int)(key[i] & ((1u << kFeatBits) - 1));
  if (tid == 0) O.nFvNodes[frame] = nNodes;
}

struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return YDORB_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    if (hipMalloc(&p, want) != hipSuccess) { set_error("hipMalloc(%zu) failed", want); return YDORB_ERR_HIP; }
    cap = want;
    return YDORB_OK;
  }
  template <class T> T* as() { return reinterpret_cast<T*>(p); }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct ydorb_vocabulary {
  int device = 0, nNodes = 0, L = 0, weighting = 0, norm = 1;
  hipStream_t stream = nullptr;
  DBuf childBegin, childIds, nodeDesc, nodeWeight, nodeWord;   // the tree
  DBuf desc, nFeat, word, node, weight, status, out;           // per-call scratch
  std::mutex mu;   // Frame::computeBoW and KeyFrame::computeBoW reach one vocabulary from the tracking, mapping and loop-closing threads
};

extern "C" {

int ydorb_vocabulary_create(const YdVocabularyTree* t, int32_t device, ydorb_vocabulary_t** out) {
  if (!t || !out || t->n_nodes < 1 || !t->child_begin || !t->child_ids || !t->node_desc || !t->node_weight || !t->node_word || t->levels < 1 ||
      t->weighting < 0 || t->weighting > 3 || t->norm < 0 || t->norm > 2) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  const int nn = t->n_nodes, nc = t->child_begin[nn];
  if (t->child_begin[0] != 0 || nc < 0) { set_error("child_begin must start at 0"); return YDORB_ERR_INVALID_ARG; }
  for (int i = 0; i < nn; i++)
    if (t->child_begin[i + 1] < t->child_begin[i]) { set_error("child_begin is not ascending at node %d", i); return YDORB_ERR_INVALID_ARG; }
  if (t->child_begin[1] == 0) { set_error("the root has no children"); return YDORB_ERR_INVALID_ARG; }
  for (int c = 0; c < nc; c++)
    if (t->child_ids[c] <= 0 || t->child_ids[c] >= nn) { set_error("child id %d out of range", t->child_ids[c]); return YDORB_ERR_INVALID_ARG; }
  // every descent must end: children come after... not required by DBoW3, so walk depth instead (a cycle would hang a kernel)
  {
    std::vector<int> depth(nn, -1);
    depth[0] = 0;
    std::vector<int> stack{0};
    while (!stack.empty()) {
      const int u = stack.back();
      stack.pop_back();
      for (int c = t->child_begin[u]; c < t->child_begin[u + 1]; c++) {
        const int v = t->child_ids[c];
        if (depth[v] >= 0) { set_error("node %d has two parents (not a tree)", v); return YDORB_ERR_INVALID_ARG; }
        depth[v] = depth[u] + 1;
        stack.push_back(v);
      }
    }
  }
  int rc = require_device(device);
  if (rc) return rc;
  ydorb_vocabulary* v = new ydorb_vocabulary;
  v->device = device; v->nNodes = nn; v->L = t->levels; v->weighting = t->weighting; v->norm = t->norm;
  auto fail = [&](int code) { ydorb_vocabulary_destroy(v); return code; };
  if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(YDORB_ERR_HIP); }
  if ((rc = v->childBegin.ensure(sizeof(int) * (nn + 1))) || (rc = v->childIds.ensure(sizeof(int) * std::max(nc, 1))) ||
      (rc = v->nodeDesc.ensure((size_t)32 * nn)) || (rc = v->nodeWeight.ensure(sizeof(double) * nn)) || (rc = v->nodeWord.ensure(sizeof(int) * nn)))
    return fail(rc);
  if (hipMemcpy(v->childBegin.p, t->child_begin, sizeof(int) * (nn + 1), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->childIds.p, t->child_ids, sizeof(int) * nc, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->nodeDesc.p, t->node_desc, (size_t)32 * nn, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->nodeWeight.p, t->node_weight, sizeof(double) * nn, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->nodeWord.p, t->node_word, sizeof(int) * nn, hipMemcpyHostToDevice) != hipSuccess) {
    set_error("uploading the vocabulary failed");
    return fail(YDORB_ERR_HIP);
  }
  *out = v;
  return YDORB_OK;
}

void ydorb_vocabulary_destroy(ydorb_vocabulary_t* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
  for (DBuf* b : {&v->childBegin, &v->childIds, &v->nodeDesc, &v->nodeWeight, &v->nodeWord, &v->desc, &v->nFeat, &v->word, &v->node, &v->weight,
                  &v->status, &v->out})
    b->release();
  delete v;
}

int ydorb_vocabulary_transform(ydorb_vocabulary_t* v, const uint8_t* desc, const int32_t* n, int32_t nFrames, int32_t cap, int32_t levelsup,
                               int32_t* bowWord, double* bowValue, int32_t* nWords, int32_t* fvNode, int32_t* fvStart, int32_t* fvFeat,
                               int32_t* nFvNodes, int32_t* status) {
  if (!v || !desc || !n || nFrames < 1 || cap < 1 || !bowWord || !bowValue || !nWords || !fvNode || !fvStart || !fvFeat || !nFvNodes) {
    set_error("invalid argument");
    return YDORB_ERR_INVALID_ARG;
  }
  if (cap > kBowMaxFeatures) { set_error("more than %d features per frame are not supported", kBowMaxFeatures); return YDORB_ERR_UNSUPPORTED; }
  for (int f = 0; f < nFrames; f++)
    if (n[f] < 0 || n[f] > cap) { set_error("frame %d: %d features, capacity %d", f, n[f], cap); return YDORB_ERR_INVALID_ARG; }
  std::lock_guard<std::mutex> lock(v->mu);
  HIPCHK(hipSetDevice(v->device));
  const size_t tot = (size_t)nFrames * cap;
  // out: bowWord | fvNode | fvFeat | fvStart (cap + 1 per frame) | nWords | nFvNodes (ints), then bowValue (doubles)
  const size_t outInts = 3 * tot + (size_t)nFrames * (cap + 1) + 2 * (size_t)nFrames, outIntsPad = (outInts + 1) & ~(size_t)1;
  int rc;
  if ((rc = v->desc.ensure(32 * tot)) || (rc = v->nFeat.ensure(sizeof(int) * nFrames)) || (rc = v->word.ensure(sizeof(int) * tot)) ||
      (rc = v->node.ensure(sizeof(int) * tot)) || (rc = v->weight.ensure(sizeof(double) * tot)) || (rc = v->status.ensure(sizeof(int) * nFrames)) ||
      (rc = v->out.ensure(sizeof(int) * outIntsPad + sizeof(double) * tot)))
    return rc;
  hipStream_t s = v->stream;
  HIPCHK(hipMemcpyAsync(v->desc.p, desc, 32 * tot, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(v->nFeat.p, n, sizeof(int) * nFrames, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(v->status.p, 0, sizeof(int) * nFrames, s));
  BowTreeDev T{v->childBegin.as<int>(), v->childIds.as<int>(), v->nodeDesc.as<uint8_t>(), v->nodeWeight.as<double>(), v->nodeWord.as<int>(), v->nNodes, v->L};
  hipLaunchKernelGGL(k_bow_descend, dim3((cap + 255) / 256, nFrames), dim3(256), 0, s, T, v->desc.as<uint8_t>(), v->nFeat.as<int>(), cap, levelsup,
                     v->word.as<int>(), v->node.as<int>(), v->weight.as<double>(), v->status.as<int>());
  int* oi = v->out.as<int>();
  BowOut O;
  O.bowWord = oi; O.fvNode = oi + tot; O.fvFeat = oi + 2 * tot; O.fvStart = oi + 3 * tot;
  O.nWords = O.fvStart + (size_t)nFrames * (cap + 1); O.nFvNodes = O.nWords + nFrames;
  O.bowValue = reinterpret_cast<double*>(oi + outIntsPad);
  int n2 = 256;
  while (n2 < cap) n2 <<= 1;
  const size_t lds = (size_t)n2 * 8;
  if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bow_assemble), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_bow_assemble, dim3(nFrames), dim3(256), lds, s, v->word.as<int>(), v->node.as<int>(), v->weight.as<double>(), v->nFeat.as<int>(), cap,
                     n2, v->weighting, v->norm, O);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(bowWord, O.bowWord, sizeof(int) * tot, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(bowValue, O.bowValue, sizeof(double) * tot, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(nWords, O.nWords, sizeof(int) * nFrames, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(fvNode, O.fvNode, sizeof(int) * tot, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(fvStart, O.fvStart, sizeof(int) * (size_t)nFrames * (cap + 1), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(fvFeat, O.fvFeat, sizeof(int) * tot, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(nFvNodes, O.nFvNodes, sizeof(int) * nFrames, hipMemcpyDeviceToHost, s));
  if (status) HIPCHK(hipMemcpyAsync(status, v->status.p, sizeof(int) * nFrames, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return YDORB_OK;
}

}  // extern "C"
