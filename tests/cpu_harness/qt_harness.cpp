// CPU harness for the product's host/device quad-tree core (ydorbslam_amd/csrc/quadtree_core.h).
// Runs the workgroup algorithm with a 1-thread context so its index logic can be checked against
// the oracle without a GPU.  Test-only; the product path always runs the HIP kernel.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../ydorbslam_amd/csrc/quadtree_core.h"

namespace {
struct CpuCtx {
  int tid() const { return 0; }
  int nthreads() const { return 1; }
  void sync() {}
  unsigned scan_incl_u32(unsigned v, unsigned* total) { *total = v; return v; }
  unsigned long long scan_incl_u64(unsigned long long v, unsigned long long* total) { *total = v; return v; }
  void count_child(unsigned long long* cc, int k, int q) { if (k >= 0) cc[k] += 1ull << (16 * q); }
};
}  // namespace

extern "C" int qt_cpu_distribute(const uint32_t* cands, int n, int rootX1, int rootY1, int quota, uint32_t* out) {
  using namespace ydorb;
  const int nodeCap = std::max(4 * quota, 4);
  std::vector<QtGeom> g0(nodeCap), g1(nodeCap);
  std::vector<uint32_t> c0(nodeCap), c1(nodeCap), b0(nodeCap), b1(nodeCap);
  std::vector<unsigned long long> cc(nodeCap);
  std::vector<uint16_t> ci(nodeCap * 4);
  QtShared S{{g0.data(), g1.data()}, {c0.data(), c1.data()}, {b0.data(), b1.data()}, cc.data(), ci.data()};
  std::vector<uint32_t> ca(cands, cands + n), cb(n), keys(n);
  std::vector<uint16_t> na(n), nb(n);
  QtGlobal G{{ca.data(), cb.data()}, {na.data(), nb.data()}};
  CpuCtx cx;
  return qt_distribute(cx, S, G, n, rootX1, rootY1, quota, nodeCap, out);
}

// std::sort front() emulation vs the real thing
extern "C" int qt_cpu_sort_front(const uint32_t* resp, int m) {
  std::vector<uint32_t> keys(m);
  for (int i = 0; i < m; i++) keys[i] = (resp[i] << 16) | (uint32_t)i;
  return ydorb::qt_sort_front(keys.data(), m);
}
extern "C" int qt_std_sort_front(const uint32_t* resp, int m) {
  struct E { float response; int idx; };
  std::vector<E> v(m);
  for (int i = 0; i < m; i++) v[i] = E{(float)resp[i], i};
  std::sort(v.begin(), v.end(), [](E& a, E& b) { return a.response > b.response; });
  return v.front().idx;
}
extern "C" void qt_cpu_heap_sort(uint32_t* keys, int m) { ydorb::qt_heap_sort(keys, m); }
extern "C" void qt_std_heap_sort(uint32_t* keys, int m) {
  auto cmp = [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); };
  std::make_heap(keys, keys + m, cmp);
  std::sort_heap(keys, keys + m, cmp);
}

// ---- flat (histogram + stable sort) formulation, sequential restatement of what k_quadtree_flat does in parallel ----
#include "../../ydorbslam_amd/csrc/quadtree_flat.h"
extern "C" int qt_flat_distribute(const uint32_t* cands, int n, int rootX1, int rootY1, int quota, uint32_t* out, int* passes) {
  using namespace ydorb;
  if (n <= 0 || quota <= 0) return 0;
  const int D = kQtFlatDepth;
  std::vector<uint32_t> key(n);
  std::vector<uint16_t> H(kQtFlatBins, 0);
  for (int i = 0; i < n; i++) {
    key[i] = qt_path_key(cands[i], rootX1, rootY1);
    for (int d = 0; d <= D; d++) H[qt_flat_level_off(d) + (key[i] >> (2 * (D - d)))]++;
  }
  int nodes[kQtFlatDepth + 1] = {0}, leaves[kQtFlatDepth + 1] = {0};
  nodes[0] = 1; leaves[0] = n == 1;
  for (int d = 1; d <= D; d++)
    for (uint32_t b = 0; b < (1u << (2 * d)); b++) {
      const int c = H[qt_flat_level_off(d) + b], pc = H[qt_flat_level_off(d - 1) + (b >> 2)];
      if (pc >= 2 && c >= 1) nodes[d]++;
      if (pc >= 2 && c == 1) leaves[d]++;
    }
  int K;
  const int P = qt_flat_passes(nodes, leaves, quota, &K);
  if (passes) *passes = P;
  if (P < 0) return -2;
  std::vector<uint32_t> R(n), order(n);
  for (int i = 0; i < n; i++) {
    int d = 0;
    while (d < P && H[qt_flat_level_off(d) + (key[i] >> (2 * (D - d)))] != 1) d++;
    R[i] = ((uint32_t)(P - d) << (2 * P)) | ((key[i] >> (2 * (D - d))) ^ qt_flat_mask(d));
    order[i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return R[a] < R[b]; });
  int k = 0;
  std::vector<uint32_t> keys;
  for (int s = 0; s < n && k < quota;) {
    int e = s;
    while (e < n && R[order[e]] == R[order[s]]) e++;
    keys.resize(e - s);
    for (int j = s; j < e; j++) keys[j - s] = ((uint32_t)qt_r(cands[order[j]]) << 16) | (uint32_t)(j - s);
    out[k++] = cands[order[s + qt_sort_front(keys.data(), e - s)]];
    s = e;
  }
  return k;
}

// all-pairs variant (small n): group-first depth e and alone-depth s straight from pairwise common prefixes of 30-bit keys
extern "C" int qt_pair_distribute(const uint32_t* cands, int n, int rootX1, int rootY1, int quota, uint32_t* out, int* passes) {
  using namespace ydorb;
  if (n <= 0 || quota <= 0) return 0;
  const int D = kQtPairDepth;
  std::vector<uint32_t> key(n);
  std::vector<int> e(n, 0), s(n, 0);
  for (int i = 0; i < n; i++) key[i] = qt_path_key<kQtPairDepth>(cands[i], rootX1, rootY1);
  int diff[kQtPairDepth + 3] = {0}, leaves[kQtPairDepth + 2] = {0}, nodes[kQtPairDepth + 2] = {0};
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      if (j == i) continue;
      const uint32_t x = key[i] ^ key[j];
      const int dpt = x ? ((__builtin_clz(x) - 2) >> 1) + 1 : D + 1;
      s[i] = std::max(s[i], dpt);
      if (j < i) e[i] = std::max(e[i], dpt);
    }
    if (e[i] <= D) { diff[e[i]]++; diff[std::min(s[i], D) + 1]--; }
    if (s[i] <= D) leaves[s[i]]++;
  }
  int run = 0;
  for (int d = 0; d <= D; d++) { run += diff[d]; nodes[d] = run; }
  int K;
  const int P = qt_flat_passes(nodes, leaves, quota, &K, D);
  if (passes) *passes = P;
  if (P < 0) return -2;
  std::vector<unsigned long long> R(n);
  std::vector<uint32_t> order(n);
  for (int i = 0; i < n; i++) {
    const int d = std::min(P, s[i]);
    R[i] = ((unsigned long long)(P - d) << (2 * P)) | ((key[i] >> (2 * (D - d))) ^ qt_flat_mask(d));
    order[i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return R[a] < R[b]; });
  int k = 0;
  std::vector<uint32_t> keys;
  for (int st = 0; st < n && k < quota;) {
    int en = st;
    while (en < n && R[order[en]] == R[order[st]]) en++;
    keys.resize(en - st);
    for (int j = st; j < en; j++) keys[j - st] = ((uint32_t)qt_r(cands[order[j]]) << 16) | (uint32_t)(j - st);
    out[k++] = cands[order[st + qt_sort_front(keys.data(), en - st)]];
    st = en;
  }
  return k;
}
