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

// ---- rank form, sequential restatement of k_qt_fast (extract_kernels.hip.h): table keys, depth-6 pyramid with aligned levels, list
// positions from ONE pass over the quads of the groups d = P .. 1 in list order, per-node maxima with the smallest / largest original
// index, introsort replay only for tied nodes of more than 16 members (their members gathered out of order and re-ranked) ----
extern "C" int qt_rank_distribute(const uint32_t* cands, int n, int rootX1, int rootY1, int quota, uint32_t* out, int* passes) {
  using namespace ydorb;
  if (n <= 0 || quota <= 0) return 0;
  if (n == 1) { out[0] = cands[0]; if (passes) *passes = 1; return 1; }
  if (quota <= 1) return -2;
  const int D = kQfDepth;
  std::vector<uint32_t> xs(rootX1 + 1), ys(rootY1 + 1);
  for (int x = 0; x <= rootX1; x++) xs[x] = qf_axis_digits(x, rootX1);
  for (int y = 0; y <= rootY1; y++) ys[y] = qf_axis_digits(y, rootY1) << 1;
  std::vector<uint16_t> H(kQfPyrU16, 0);
  std::vector<uint32_t> st(n);
  for (int i = 0; i < n; i++) {
    const uint32_t key = qf_key(xs[qt_x(cands[i])], ys[qt_y(cands[i])]);
    if (key != qt_path_key<kQfDepth>(cands[i], rootX1, rootY1)) return -9;
    st[i] = key | ((uint32_t)qt_r(cands[i]) << 16);
    H[qf_off(D) + key]++;
  }
  int nodes[kQfDepth + 1] = {0}, leaves[kQfDepth + 1] = {0};
  nodes[0] = 1;
  for (int d = D - 1; d >= 0; d--)
    for (int b = 0; b < (1 << (2 * d)); b++) {
      const uint16_t* c = &H[qf_off(d + 1) + 4 * b];
      const unsigned sum = c[0] + c[1] + c[2] + c[3];
      H[qf_off(d) + b] = (uint16_t)sum;
      if (sum >= 2)
        for (int q = 0; q < 4; q++) { nodes[d + 1] += c[q] != 0; leaves[d + 1] += c[q] == 1; }
    }
  int K = 0;
  int P = qt_flat_passes(nodes, leaves, quota, &K, D);
  if (passes) *passes = P;
  const bool pairs = P < 1;
  if (pairs && n > 1024) return -2;
  int nOut = K < quota ? K : quota;
  std::vector<uint8_t> big;
  if (pairs) {   // all-pairs form of the kernel: alone depth / group-first depth from common prefixes, node position = distinct smaller ranks
    const int DP = kQtPairDepth;
    std::vector<uint32_t> kk(n);
    std::vector<int> sA(n, 0), eA(n, 0);
    for (int i = 0; i < n; i++) kk[i] = qt_path_key<kQtPairDepth>(cands[i], rootX1, rootY1);
    int diff[kQtPairDepth + 3] = {0}, lf[kQtPairDepth + 2] = {0}, nd[kQtPairDepth + 2] = {0};
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) {
        if (j == i) continue;
        const uint32_t x = kk[i] ^ kk[j];
        const int dpt = x ? ((__builtin_clz(x) - 2) >> 1) + 1 : DP + 1;
        sA[i] = std::max(sA[i], dpt);
        if (j < i) eA[i] = std::max(eA[i], dpt);
      }
      if (eA[i] <= DP) { diff[eA[i]]++; diff[std::min(sA[i], DP) + 1]--; }
      if (sA[i] <= DP) lf[sA[i]]++;
    }
    int run = 0;
    for (int d = 0; d <= DP; d++) { run += diff[d]; nd[d] = run; }
    P = qt_flat_passes(nd, lf, quota, &K, DP);
    if (passes) *passes = P;
    if (P < 1) return -2;
    nOut = K < quota ? K : quota;
    big.assign(nOut + 2, 0);
    std::vector<unsigned long long> RR(n);
    for (int i = 0; i < n; i++) {
      const int dA = std::min(P, sA[i]);
      RR[i] = ((unsigned long long)(P - dA) << (2 * P)) | (unsigned long long)((kk[i] >> (2 * (DP - dA))) ^ qt_flat_mask(dA));
    }
    std::vector<uint8_t> first(n, 1);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++) if (RR[j] == RR[i]) first[i] = 0;
    for (int i = 0; i < n; i++) {
      int kp = 0, mm = 0;
      for (int j = 0; j < n; j++) { kp += (RR[j] < RR[i]) && first[j]; mm += RR[j] == RR[i]; }
      if (kp < nOut && mm > 16) big[kp] = 1;
      st[i] = (st[i] & 0xFFFF0000u) | (uint32_t)(kp < nOut ? kp : 0x7FFF);
    }
  }
  for (int i = 0; i < n && !pairs; i++) {
    const int key = (int)(st[i] & 0xFFFu);
    int slot = qf_off(P) + (key >> (2 * (D - P)));
    for (int d = P - 1; d >= 1; d--) {
      const int sl = qf_off(d) + (key >> (2 * (D - d)));
      if (H[sl] == 1) slot = sl;
    }
    st[i] = (st[i] & 0xFFFF0000u) | (uint32_t)slot;
  }
  if (!pairs) big.assign(nOut + 2, 0);
  if (!pairs) {
    const int TQ = ((1 << (2 * P)) - 1) / 3;
    unsigned pos = 0;
    for (int q = 0; q < TQ; q++) {
      int d = P, start = 0;
      while (q >= start + (1 << (2 * (d - 1)))) { start += 1 << (2 * (d - 1)); d--; }
      const int pq = q - start, bq = pq ^ (int)(qt_flat_mask(d) >> 2), qslot = qf_off(d) + 4 * bq;
      const unsigned c[4] = {H[qslot], H[qslot + 1], H[qslot + 2], H[qslot + 3]};
      const unsigned sum = c[0] + c[1] + c[2] + c[3];
      unsigned o[4];
      for (int e = 0; e < 4; e++) {
        const unsigned cn = c[3 - e];
        const bool fl = sum >= 2 && (d == P ? cn >= 1 : cn == 1);
        unsigned v = 0x7FFFu;
        if (fl) {
          if ((int)pos < nOut) { v = pos | (cn > 16 ? 0x8000u : 0u); if (cn > 16) big[pos] = 1; }
          pos++;
        }
        o[3 - e] = v;
      }
      for (int e = 0; e < 4; e++) H[qslot + e] = (uint16_t)o[e];
    }
    if ((int)pos != K) return -3;
  }
  std::vector<uint32_t> lo(nOut + 2, 0), hi(nOut + 2, 0);
  for (int i = 0; i < n; i++) {
    const unsigned pv = pairs ? st[i] & 0x7FFFu : H[st[i] & 0xFFFFu] & 0x7FFFu;
    const bool ok = (int)pv < nOut;
    const uint32_t rs = st[i] & 0xFFFF0000u;
    if (ok) { lo[pv] = std::max(lo[pv], rs | (0xFFFFu - (uint32_t)i)); hi[pv] = std::max(hi[pv], rs | (uint32_t)i); }
    st[i] = rs | (ok ? pv : 0xFFFFu);
  }
  for (int k = 0; k < nOut; k++) {
    const unsigned iLo = 0xFFFFu - (lo[k] & 0xFFFFu), iHi = hi[k] & 0xFFFFu;
    if (iLo == iHi || !big[k]) { out[k] = cands[iLo]; continue; }
    std::vector<uint32_t> idx, rr;                       // members gathered in REVERSE order, as an arbitrary append order would
    for (int i = n - 1; i >= 0; i--)
      if ((st[i] & 0xFFFFu) == (unsigned)k) { idx.push_back((uint32_t)i); rr.push_back(st[i] >> 16); }
    const int m = (int)idx.size();
    std::vector<uint32_t> keys(m), vals(m);
    for (int t = 0; t < m; t++) {
      int r = 0;
      for (int u = 0; u < m; u++) r += idx[u] < idx[t];
      keys[r] = (rr[t] << 16) | (uint32_t)r; vals[r] = idx[t];
    }
    out[k] = cands[vals[qt_sort_front(keys.data(), m)]];
  }
  return nOut;
}
