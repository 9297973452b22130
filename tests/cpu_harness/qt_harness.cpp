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
