// Flat (pass-free) formulation of the quad-tree keypoint thinning — same result as qt_distribute() in quadtree_core.h
// (reference src/orbExtractor.cpp:455-544 + :4-54), derived from three observations about the reference's list:
//   1. A candidate's path through the (quirky) geometry never depends on the other candidates: its child digit at every
//      depth can be computed up front (qt_path_key).
//   2. A depth-d prefix is a list node after pass d iff its parent holds >= 2 candidates; it is a leaf (moved behind the
//      new nodes, never split again) iff it holds exactly 1.  So with the histogram pyramid H_d[prefix] the list length
//      after pass p is  K_p = #{b : H_p[b] >= 1, H_{p-1}[b>>2] >= 2} + sum_{d<p} #{b : H_d[b] == 1, H_{d-1}[b>>2] >= 2},
//      and the reference's loop `while (K grew && K < quota)` stops after P = min{p >= 1 : !(K_p > K_{p-1} && K_p < quota)}.
//   3. push_front of (n1..n4) while walking the list front-to-back makes the depth-P nodes appear ordered by their digits
//      with alternating direction (last digit descending, the one before ascending, ...), i.e. ascending in
//      prefix XOR 0b..110011; leaves follow grouped by depth (deepest first), each group in its own depth's order.
// Final list rank of a candidate's node:  R = (P - d) << 2P | (prefix_d ^ mask_d),  d = min(P, first depth where it is alone).
// A stable sort by R puts every node's members together in their original order, which is what the std::sort tie
// emulation (qt_sort_front) needs.
#pragma once
#include "quadtree_core.h"

namespace ydorb {

constexpr int kQtPairDepth = 15;                      // all-pairs variant for small n: 30-bit path keys
constexpr int kQtFlatDepth = 7;                       // histogram pyramid depth: 4^7 = 16384 bins (deeper trees: pass algorithm)
constexpr int kQtFlatBins = (4 * 16384 - 1) / 3;      // 1 + 4 + ... + 4^7 = 21845
YD_HD inline int qt_flat_level_off(int d) { return ((1 << (2 * d)) - 1) / 3; }  // offset of depth d inside the pyramid

// digits MSD-first: depth-d prefix = key >> 2*(kQtFlatDepth-d)
template <int DEPTH = kQtFlatDepth>
YD_HD inline uint32_t qt_path_key(uint32_t c, int rootX1, int rootY1) {
  // qt_child()/qt_quadrant() unrolled on plain ints (same arithmetic; the harness checks it against the pass algorithm)
  const int x = qt_x(c), y = qt_y(c);
  int x0 = 0, x1 = rootX1, y0 = 0, y1 = rootY1;
  uint32_t key = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int d = 0; d < DEPTH; d++) {
    const int cx = qt_center(x0, x1), cy = qt_center(y0, y1);
    const bool l = x < cx, t = y < cy;
    key = (key << 2) | (uint32_t)(l ? (t ? 0 : 2) : (t ? 1 : 3));
    if (l) x1 = cx; else x0 = cx;
    if (t) y1 = cy;
    else { y1 = l ? y1 : y0; y0 = cy; }   // node4 inherits the parent's TOP edge as its "bottom" (orbExtractor.cpp:27)
  }
  return key;
}
// complement digits j = d, d-2, d-4, ... (1-based from the most significant) of a right-aligned depth-d prefix
YD_HD inline uint32_t qt_flat_mask(int d) { return 0x33333333u & ((1u << (2 * d)) - 1u); }

// K_p bookkeeping from per-depth (#nodes, #leaves) counts; returns P (number of passes the reference runs), or -1 when the
// loop would still be running after kQtFlatDepth passes.  nodes[d], leaves[d] for d = 0..kQtFlatDepth.
YD_HD inline int qt_flat_passes(const int* nodes, const int* leaves, int quota, int* Kfinal, int maxDepth = kQtFlatDepth) {
  int K = 1, last = 0, p = 0, leafSum = 0;
  while (K > last && K < quota) {
    if (p == maxDepth) return -1;
    last = K;
    leafSum += leaves[p];
    p++;
    K = nodes[p] + leafSum;
  }
  *Kfinal = K;
  return p;
}

// ---- rank form (k_qt_fast, extract_kernels.hip.h): no sort, list positions from one scan over the permuted bin spaces --------------
constexpr int kQfDepth = 6;                                    // histogram pyramid depth of the rank form: 4^6 = 4096 bins at the bottom
YD_HD constexpr int qf_off(int d) { return d == 0 ? 0 : 4 + ((1 << (2 * d)) - 4) / 3; }   // 8-byte aligned levels: 0, 4, 8, 24, 88, 344, 1368
constexpr int kQfPyrU16 = 1368 + 4096;
// The split digits of one axis: the reference's x tests depend on x only, its y tests on y only until the first child 3 (the
// line-27 slip hands node 4 the parent's TOP edge as its bottom: from there on every y test fails).  Digits most significant first,
// complemented (bit set = "not left" / "not top"), spread to every second bit: x digits on the even bits of the key, y digits << 1.
YD_HD inline uint32_t qf_axis_digits(int c, int hi, int depth = kQfDepth) {
  int a = 0, b = hi;
  uint32_t bits = 0;
  for (int d = 0; d < depth; d++) {
    const int m = qt_center(a, b);
    const bool lt = c < m;
    bits = (bits << 2) | (lt ? 0u : 1u);
    if (lt) b = m; else a = m;
  }
  return bits;
}
YD_HD inline uint32_t qf_key(uint32_t xDigits, uint32_t yDigitsShifted) {   // == qt_path_key<kQfDepth> (the harness checks every pixel)
  uint32_t key = xDigits | yDigitsShifted;
  const uint32_t both = key & (key >> 1) & 0x555u;
#if defined(__HIP_DEVICE_COMPILE__)
  key |= both ? ((1u << (31 - __clz((int)both))) - 1u) & 0xAAAu : 0u;
#else
  key |= both ? ((1u << (31 - __builtin_clz(both))) - 1u) & 0xAAAu : 0u;
#endif
  return key;
}

}  // namespace ydorb
