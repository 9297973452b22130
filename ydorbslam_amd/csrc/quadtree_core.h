// Quad-tree keypoint thinning for one (frame, pyramid level), written for ONE workgroup.
//
// Behaviour restated from reference src/orbExtractor.cpp:455-544 (distributeQuadTree) and :4-54
// (QuadTreeNode::divideNode) — but re-designed as flat arrays + prefix sums so a 256..1024-thread
// workgroup can run it with LDS-resident nodes:
//   * the reference's std::list with push_front becomes an index formula: after one pass the list is
//       [children of the LAST divisible node (n4,n3,n2,n1) ... children of the FIRST] ++ [old leaves],
//     so a child's new index is a suffix sum over nodes of their non-empty child counts;
//   * the per-node std::vector<KeyPoint> becomes a contiguous segment of a candidate array that is
//     stably 4-way partitioned each pass with one block-wide scan of four packed 16-bit counters;
//   * "best keypoint per node" = front() of libstdc++ std::sort by response (:536-539): ties are
//     resolved exactly as introsort would (first max for <=16 elements, else the left-most partition
//     chain of __introsort_loop is replayed), see qt_sort_front().
// The same template runs single-threaded on the host (tests/cpu harness) and as a HIP workgroup.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define YD_HD __host__ __device__
#else
#define YD_HD
#endif

namespace ydorb {

// candidate packing: x:12 | y:12 | response:8   (border-relative coords, orbExtractor.cpp:587-589)
YD_HD inline uint32_t qt_pack(int x, int y, int r) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)r << 24); }
YD_HD inline int qt_x(uint32_t c) { return (int)(c & 0xFFFu); }
YD_HD inline int qt_y(uint32_t c) { return (int)((c >> 12) & 0xFFFu); }
YD_HD inline int qt_r(uint32_t c) { return (int)(c >> 24); }

// The four numbers divideNode actually reads: tl.x, tr.x, tl.y, bl.y (:8-9).
struct QtGeom { int16_t x0, x1, y0, y1; };

YD_HD inline int qt_center(int a, int b) {  // ceil(float(a+b)/2.0) for small non-negative ints
  return (a + b + 1) >> 1;
}
// child q = 0..3 <-> _node1.._node4 of divideNode, including the line-27 slip that gives node4 the
// PARENT's top edge as its "bottom" (n3.br = (cx, tl.y); n4.bl = n3.br).
YD_HD inline QtGeom qt_child(const QtGeom& g, int cx, int cy, int q) {
  QtGeom c;
  switch (q) {
    case 0: c.x0 = g.x0; c.x1 = (int16_t)cx; c.y0 = g.y0; c.y1 = (int16_t)cy; break;
    case 1: c.x0 = (int16_t)cx; c.x1 = g.x1; c.y0 = g.y0; c.y1 = (int16_t)cy; break;
    case 2: c.x0 = g.x0; c.x1 = (int16_t)cx; c.y0 = (int16_t)cy; c.y1 = g.y1; break;
    default: c.x0 = (int16_t)cx; c.x1 = g.x1; c.y0 = (int16_t)cy; c.y1 = g.y0; break;
  }
  return c;
}
YD_HD inline int qt_quadrant(uint32_t c, int cx, int cy) {  // :37-47
  const bool l = qt_x(c) < cx, t = qt_y(c) < cy;
  return l ? (t ? 0 : 2) : (t ? 1 : 3);
}

// ---- libstdc++ std::sort(first,last,[](a,b){return a.response>b.response;}) -> index of front() ----
// keys[i] = (response << 16) | i ; comparison looks at the response only.
YD_HD inline bool qt_gt(uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); }
template <class P> YD_HD inline void qt_swap(P a, int i, int j) { uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }

template <class P> YD_HD inline void qt_push_heap(P a, int hole, int top, uint32_t v) {
  int parent = (hole - 1) / 2;
  while (hole > top && qt_gt(a[parent], v)) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = v;
}
template <class P> YD_HD inline void qt_adjust_heap(P a, int hole, int len, uint32_t v) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (qt_gt(a[child], a[child - 1])) child--;
    a[hole] = a[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[hole] = a[child - 1];
    hole = child - 1;
  }
  qt_push_heap(a, hole, top, v);
}
template <class P> YD_HD inline void qt_heap_sort(P a, int len) {  // __partial_sort(first,last,last)
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      uint32_t v = a[parent];
      qt_adjust_heap(a, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    uint32_t v = a[last];
    a[last] = a[0];
    qt_adjust_heap(a, 0, last, v);
  }
}
// a[0..m): scratch holding the keys in member order; returns the member index that std::sort leaves at front().
template <class P> YD_HD inline int qt_sort_front(P a, int m) {
  int hi = m;
  if (m > 16) {
    int depth = 0;
    for (int t = m; t > 1; t >>= 1) depth++;
    depth *= 2;
    while (hi > 16) {
      if (depth == 0) {
        qt_heap_sort(a, hi);
        return (int)(a[0] & 0xFFFFu);
      }
      --depth;
      const int mid = hi / 2;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        const int x = 1, y = mid, z = hi - 1;
        if (qt_gt(a[x], a[y])) {
          if (qt_gt(a[y], a[z])) qt_swap(a, 0, y);
          else if (qt_gt(a[x], a[z])) qt_swap(a, 0, z);
          else qt_swap(a, 0, x);
        } else if (qt_gt(a[x], a[z])) qt_swap(a, 0, x);
        else if (qt_gt(a[y], a[z])) qt_swap(a, 0, z);
        else qt_swap(a, 0, y);
      }
      int f = 1, l = hi;  // __unguarded_partition(first+1, last, pivot=first)
      const uint32_t piv = a[0];   // slot 0 is never touched by the swaps below (f >= 1)
      for (;;) {
        while (qt_gt(a[f], piv)) ++f;
        --l;
        while (qt_gt(piv, a[l])) --l;
        if (!(f < l)) break;
        qt_swap(a, f, l);
        ++f;
      }
      hi = f;  // recursion on [cut,last) never touches [first,cut)
    }
  }
  // stable insertion sort of a block that dominates everything to its right: front = first maximum
  int best = 0;
  for (int i = 1; i < hi; i++)
    if (qt_gt(a[i], a[best])) best = i;
  return (int)(a[best] & 0xFFFFu);
}

// ---- workgroup algorithm -------------------------------------------------------------------------
#ifndef QT_ITEMS
#define QT_ITEMS 8
#endif
// On the device the node table always lives in LDS; typing the pointers with the LDS address space makes every access a
// ds_* instruction (generic pointers compiled to flat_* loads, which cost global-memory latency even for LDS addresses).
#if defined(__HIP_DEVICE_COMPILE__)
#define QT_LDS __attribute__((address_space(3)))
#else
#define QT_LDS
#endif
// Node tables (arrays of nodeCap entries): LDS on the device in the normal case, heap on the host; `InLds = false` on the device puts
// them in HBM scratch — only for quotas so large (few levels x thousands of features) that 48 B x 4 x quota exceeds the LDS.
template <bool InLds> struct QtNodePtr {
  typedef QtGeom* Geom; typedef uint32_t* U32; typedef unsigned long long* U64; typedef uint16_t* U16;
};
template <> struct QtNodePtr<true> {
  typedef QT_LDS QtGeom* Geom; typedef QT_LDS uint32_t* U32; typedef QT_LDS unsigned long long* U64; typedef QT_LDS uint16_t* U16;
};
template <bool InLds>
struct QtSharedT {
  typename QtNodePtr<InLds>::Geom geom[2];
  typename QtNodePtr<InLds>::U32 cnt[2];
  typename QtNodePtr<InLds>::U32 base[2];
  typename QtNodePtr<InLds>::U64 cc;        // packed 4x16 child counts; re-used as the segment-start prefix in sweep 2
  typename QtNodePtr<InLds>::U16 childIdx;  // [nodeCap*4] new list index of child q (slot 0: own new index for a leaf)
};
typedef QtSharedT<true> QtShared;
template <bool InLds> struct QtCandPtr { typedef uint32_t* U32; typedef uint16_t* U16; };
template <> struct QtCandPtr<true> { typedef QT_LDS uint32_t* U32; typedef QT_LDS uint16_t* U16; };
template <bool InLds>
struct QtCandT {            // per (frame, level) candidate state, n entries each: LDS when the level fits, else HBM scratch
  typename QtCandPtr<InLds>::U32 cand[2];  // ping-pong; after the last pass the idle half holds the keys of the std::sort replay
  typename QtCandPtr<InLds>::U16 node[2];
};
typedef QtCandT<false> QtGlobal;

// Ctx: tid(), nthreads(), sync(), scan_incl_u64(v,&total), scan_incl_u32(v,&total), count_child(cc,k,q) (k < 0: lane idle)
template <class Ctx, class Shared, class Cand>
YD_HD int qt_distribute(Ctx& cx, const Shared& S, const Cand& G, int n, int rootX1, int rootY1, int quota,
                        int nodeCap, uint32_t* out) {
  const int tid = cx.tid(), nt = cx.nthreads();
  if (n <= 0 || quota <= 0) return 0;  // resize(desired) of an empty or zero-quota list
  int cur = 0;
  if (tid == 0) {
    S.geom[0][0] = QtGeom{0, (int16_t)rootX1, 0, (int16_t)rootY1};
    S.cnt[0][0] = (uint32_t)n;
    S.base[0][0] = 0;
  }
  for (int p = tid; p < n; p += nt) G.node[0][p] = 0;
  cx.sync();
  int K = 1, last = 0;
  while (K > last && K < quota) {
    last = K;
    const int nxt = cur ^ 1;
    const auto geom = S.geom[cur];
    const auto cnt = S.cnt[cur];
    const auto base = S.base[cur];
    for (int k = tid; k < K; k += nt) S.cc[k] = 0;
    cx.sync();
    // sweep 1: child counts.  Candidates of one node are contiguous, so the lanes of a wave mostly hit the same counter:
    // the context aggregates equal-node lanes with ballots and issues one LDS atomic per (wave, node) instead of 64
    // same-address atomics (which the LDS serialises).
    for (int p0 = 0; p0 < n; p0 += nt) {
      const int p = p0 + tid;
      int k = -1, q = 0;
      if (p < n) {
        const int kk = G.node[cur][p];
        if (cnt[kk] > 1) {
          const QtGeom g = geom[kk];
          q = qt_quadrant(G.cand[cur][p], qt_center(g.x0, g.x1), qt_center(g.y0, g.y1));
          k = kk;
        }
      }
      cx.count_child(S.cc, k, q);
    }
    cx.sync();
    // node phases A/A': each thread owns a contiguous run of `per` nodes, so every phase needs ONE workgroup scan.
    // A: new front nodes.  Children of LATER list nodes come first -> walk the list in reverse (suffix sum as a prefix).
    const int per = (K + nt - 1) / nt;
    int M = 0;
    {
      int mine = 0;
      for (int j = 0; j < per; j++) {
        const int kr = tid * per + j;
        if (kr < K) {
          const int k = K - 1 - kr;
          if (cnt[k] > 1) {
            const unsigned long long ccv = S.cc[k];
            for (int q = 0; q < 4; q++) mine += ((ccv >> (16 * q)) & 0xFFFF) != 0;
          }
        }
      }
      unsigned total;
      const unsigned incl = cx.scan_incl_u32((unsigned)mine, &total);
      int idx = (int)incl - mine;
      for (int j = 0; j < per; j++) {
        const int kr = tid * per + j;
        if (kr < K) {
          const int k = K - 1 - kr;
          if (cnt[k] > 1) {
            const unsigned long long ccv = S.cc[k];
            for (int q = 3; q >= 0; q--)  // n4 is pushed last -> sits first
              if ((ccv >> (16 * q)) & 0xFFFF) S.childIdx[k * 4 + q] = (uint16_t)idx++;
          }
        }
      }
      M = (int)total;
    }
    int Knew = M;
    {  // A': leaves keep their relative order behind all new nodes
      int mine = 0;
      for (int j = 0; j < per; j++) {
        const int k = tid * per + j;
        if (k < K && cnt[k] <= 1) mine++;
      }
      unsigned total;
      const unsigned incl = cx.scan_incl_u32((unsigned)mine, &total);
      int idx = M + (int)incl - mine;
      for (int j = 0; j < per; j++) {
        const int k = tid * per + j;
        if (k < K && cnt[k] <= 1) S.childIdx[k * 4] = (uint16_t)idx++;
      }
      Knew += (int)total;
    }
    if (Knew > nodeCap) return -1;  // cannot happen for nodeCap >= 4*quota
    cx.sync();
    // node phase B: write the new node table
    for (int k = tid; k < K; k += nt) {
      const QtGeom g = geom[k];
      if (cnt[k] > 1) {
        const unsigned long long ccv = S.cc[k];
        const int cxm = qt_center(g.x0, g.x1), cym = qt_center(g.y0, g.y1);
        for (int q = 0; q < 4; q++) {
          const uint32_t c = (uint32_t)((ccv >> (16 * q)) & 0xFFFF);
          if (c) {
            const int idx = S.childIdx[k * 4 + q];
            S.geom[nxt][idx] = qt_child(g, cxm, cym, q);
            S.cnt[nxt][idx] = c;
          }
        }
      } else {
        const int idx = S.childIdx[k * 4];
        S.geom[nxt][idx] = g;
        S.cnt[nxt][idx] = cnt[k];
      }
    }
    cx.sync();
    // node phase C: segment bases of the new list (exclusive scan of counts in new list order)
    {
      const int perN = (Knew + nt - 1) / nt;
      unsigned mine = 0;
      for (int j = 0; j < perN; j++) {
        const int k = tid * perN + j;
        if (k < Knew) mine += S.cnt[nxt][k];
      }
      unsigned total;
      const unsigned incl = cx.scan_incl_u32(mine, &total);
      unsigned run = incl - mine;
      for (int j = 0; j < perN; j++) {
        const int k = tid * perN + j;
        if (k < Knew) { S.base[nxt][k] = run; run += S.cnt[nxt][k]; }
      }
    }
    cx.sync();
    // sweep 2: stable 4-way partition (ordered scan of packed quadrant counters).  Each thread owns QT_ITEMS consecutive
    // candidates, so one workgroup scan covers nthreads*QT_ITEMS positions (4096 at 512 threads): two barriers per pass
    // instead of four per 512 candidates.
    {
      unsigned long long carry = 0;
      for (int c0 = 0; c0 < n; c0 += nt * QT_ITEMS) {
        int kk[QT_ITEMS], qq[QT_ITEMS];
        uint32_t cc[QT_ITEMS];
        unsigned long long loc[QT_ITEMS], tot = 0;
#pragma unroll
        for (int i = 0; i < QT_ITEMS; i++) {
          const int p = c0 + tid * QT_ITEMS + i;
          kk[i] = 0; qq[i] = -1; cc[i] = 0;
          unsigned long long v = 0;
          if (p < n) {
            kk[i] = G.node[cur][p];
            cc[i] = G.cand[cur][p];
            if (cnt[kk[i]] > 1) {
              const QtGeom g = geom[kk[i]];
              qq[i] = qt_quadrant(cc[i], qt_center(g.x0, g.x1), qt_center(g.y0, g.y1));
              v = 1ull << (16 * qq[i]);
            }
          }
          loc[i] = tot;
          tot += v;
        }
        unsigned long long total;
        const unsigned long long incl = cx.scan_incl_u64(tot, &total);
        const unsigned long long texcl = carry + incl - tot;
#pragma unroll
        for (int i = 0; i < QT_ITEMS; i++) {
          const int p = c0 + tid * QT_ITEMS + i;
          loc[i] += texcl;
          if (p < n && (uint32_t)p == base[kk[i]]) S.cc[kk[i]] = loc[i];  // prefix at segment start
        }
        cx.sync();
#pragma unroll
        for (int i = 0; i < QT_ITEMS; i++) {
          const int p = c0 + tid * QT_ITEMS + i;
          if (p < n) {
            int idx, rank;
            if (qq[i] >= 0) {
              idx = S.childIdx[kk[i] * 4 + qq[i]];
              rank = (int)(((loc[i] - S.cc[kk[i]]) >> (16 * qq[i])) & 0xFFFF);
            } else {
              idx = S.childIdx[kk[i] * 4];
              rank = 0;
            }
            const uint32_t np = S.base[nxt][idx] + (uint32_t)rank;
            G.cand[nxt][np] = cc[i];
            G.node[nxt][np] = (uint16_t)idx;
          }
        }
        carry += total;
        cx.sync();
      }
    }
    K = Knew;
    cur = nxt;
    cx.sync();
  }
  // best keypoint per node, list order, truncated to the quota (:534-543)
  const int nOut = K < quota ? K : quota;
#if defined(QT_DBG_STOP) && QT_DBG_STOP == 2
  return 0;
#endif
  const auto keysAll = G.cand[cur ^ 1];
  {  // keys of every candidate, segment-relative index in the low half: filled by all threads, coalesced
    const auto nodeOf = G.node[cur];
    const auto cd = G.cand[cur];
    for (int p = tid; p < n; p += nt) keysAll[p] = ((uint32_t)qt_r(cd[p]) << 16) | ((uint32_t)p - S.base[cur][nodeOf[p]]);
  }
  cx.sync();
  for (int k = tid; k < nOut; k += nt) {
    const uint32_t b = S.base[cur][k], m = S.cnt[cur][k];
    const auto keys = keysAll + b;
    int best = 0;
    if (m > 16) {
      best = qt_sort_front(keys, (int)m);
    } else {
      for (uint32_t i = 1; i < m; i++)
        if (qt_gt(keys[i], keys[best])) best = (int)i;
    }
    out[k] = G.cand[cur][b + best];
  }
  cx.sync();
  return nOut;
}

}  // namespace ydorb
