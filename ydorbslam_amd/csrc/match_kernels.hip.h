// gfx950 HIP kernels of the descriptor matcher (reference src/orbMatcher.cpp + the Frame grid of
// src/frame.cpp:249-264,327-361).
//
// The reference's searches are sequential and greedy: a keypoint that receives a map point is skipped by
// every later query (orbMatcher.cpp:40,106,194,325), so the result depends on query order.  The GPU design
// splits each search into
//   (1) k_grid_build      one workgroup per frame: the 64x48 candidate grid as a CSR (counting sort that keeps
//                         ascending keypoint index inside a cell, i.e. push_back order);
//   (2) k_gather_*        one 64-lane wave per query, all queries of all calls in parallel: candidate list in
//                         the reference's scan order + 256-bit Hamming distance (4 x 64-bit popcount per pair),
//                         appended to a record pool as (dist << 16 | idx);
//   (3) k_resolve         one wave per search call: replays the queries in order; per query a wavefront
//                         min-reduction over (dist, scan position) of the not-yet-taken candidates gives exactly
//                         the best / second-best the sequential loop would keep, then the acceptance rule,
//                         the "taken" update and the rotation histogram are applied.
// Everything is integer except the window arithmetic, which uses the same single float operations as the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#pragma clang fp contract(off)

namespace ydorb {

constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;  // frame.hpp:137-138
constexpr int kHistLen = 30, kThHigh = 100, kThLow = 50;                             // orbMatcher.cpp:7-9

// A pointer that a kernel reads out of a record in memory (CallDev, FrameDev, ...) is a FLAT pointer to the compiler: its loads and
// stores become flat_* instructions, which count on the LDS counter as well, so every LDS read behind one waits for the memory
// round trip.  G(p) re-types it as a global-address-space pointer (all these records only ever point into HBM).
template <class T> using gptr = __attribute__((address_space(1))) T*;
template <class T> __device__ __forceinline__ gptr<T> G(T* p) { return (gptr<T>)p; }

struct KeyPointDev { float x, y, size, angle, response; int octave, class_id; };
struct QueryDev {  // == YdQuery
  float u, v, r;
  int minLevel, maxLevel;
  float ur, rs, angle;
  int level, flags;
};
struct FrameDev {  // one target frame of a batched call
  const KeyPointDev* kps;
  const uint8_t* desc;
  const float* rightX;  // may be null (monocular: every entry <= 0)
  const int* nPtr;      // device count (batched pipeline) or null
  int n;                // used when nPtr is null
  float minX, minY, gridWInv, gridHInv;
  int* cellStart;       // [kGridCells + 1]
  int* cellIdx;         // [cap]
  float4* sortedKp;     // [cap] (x, y, octave bits, index bits) in grid (CSR) order: a window column is one contiguous run
  uint8_t* sortedDesc;  // [cap][32] descriptors in the same order
};
__device__ __forceinline__ int frame_n(const FrameDev& F) { return F.nPtr ? *F.nPtr : F.n; }

// ---------------------------------------------------------------------------------------------------
// Frame::assignKeyPointsToGrid + computeLocationInGrid (frame.cpp:249-264,327-336): rounds, and uses minX for y.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_build(const FrameDev* __restrict__ frames, int cap) {
  __shared__ int hist[kGridCells];
  __shared__ int wsum[4];
  extern __shared__ int16_t cellOf[];  // [cap]
  const FrameDev F = frames[blockIdx.x];
  const int n = min(frame_n(F), cap);
  for (int c = threadIdx.x; c < kGridCells; c += 256) hist[c] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const KeyPointDev kp = F.kps[i];
    const int lx = (int)roundf(__fmul_rn(__fsub_rn(kp.x, F.minX), F.gridWInv));
    const int ly = (int)roundf(__fmul_rn(__fsub_rn(kp.y, F.minX), F.gridHInv));
    int cell = -1;
    if (!(lx < 0 || lx >= kGridCols || ly < 0 || ly >= kGridRows)) {
      cell = lx * kGridRows + ly;
      atomicAdd(&hist[cell], 1);
    }
    cellOf[i] = (int16_t)cell;
  }
  __syncthreads();
  // exclusive scan of 3072 counts: 12 per thread
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int local[12], s = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) { local[k] = s; s += hist[threadIdx.x * 12 + k]; }
  int x = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  int off = x - s;
  for (int j = 0; j < wv; j++) off += wsum[j];
#pragma unroll
  for (int k = 0; k < 12; k++) F.cellStart[threadIdx.x * 12 + k] = off + local[k];
  if (threadIdx.x == 255) F.cellStart[kGridCells] = off + s;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 12; k++) hist[threadIdx.x * 12 + k] = off + local[k];  // now: start offsets
  __syncthreads();
  // stable fill: rank inside the cell = number of earlier keypoints with the same cell
  for (int i = threadIdx.x; i < n; i += 256) {
    const int cell = cellOf[i];
    if (cell < 0) continue;
    int rank = 0;
    for (int j = 0; j < i; j++) rank += cellOf[j] == cell;
    const int pos = hist[cell] + rank;
    F.cellIdx[pos] = i;
    const KeyPointDev kp = F.kps[i];
    F.sortedKp[pos] = make_float4(kp.x, kp.y, __int_as_float(kp.octave), __int_as_float(i));
    const uint4* d = reinterpret_cast<const uint4*>(F.desc + (size_t)i * 32);
    uint4* o = reinterpret_cast<uint4*>(F.sortedDesc + (size_t)pos * 32);
    o[0] = d[0]; o[1] = d[1];
  }
}

__device__ __forceinline__ int hamming256(const uint8_t* a, const uint8_t* b) {
  const uint4 a0 = *reinterpret_cast<const uint4*>(a), a1 = *reinterpret_cast<const uint4*>(a + 16);
  const uint4 b0 = *reinterpret_cast<const uint4*>(b), b1 = *reinterpret_cast<const uint4*>(b + 16);
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

struct CallDev {          // one search call (one target frame, one ordered query list)
  int frame;              // index into frames[] (projection family)
  const KeyPointDev* tkps; // target keypoints (octave / angle of the winners)
  const float* qAngle;    // per-query source angle (BoW family); null -> queries[q].angle
  const QueryDev* queries;
  const uint8_t* qdesc;   // [nq][32]
  const int* nqPtr;       // device query count or null
  int nq;
  int2* qInfo;            // [nq] (pool base, record count)
  uint8_t* taken;         // [n] in/out
  int* assigned;          // [n] in/out (query index or -1)
  int* matchQ;            // [nq] scratch: accepted target index per query (or -1), used by the histogram cull
  int* count;             // out: matchNum
  int mode;               // 0 M1, 1 M3, 2 M4 (projection family); 3 M5, 4 M6 (BoW family); 5 searchForTriangulation; 6 fuse search; 7 searchByProjectionInSim
  float ratio;
  int orbDist, checkOri;
  float invSigma2[8];     // mode 6: the keyframe's m_v_invScaleFactorSquares
  const KeyPointDev* qkps; // device-resident pair form: the query frame's keypoints (k_queries_from_keypoints builds `queries` from them)
  uint2* qPre;            // [nq] or null: per query the best / second-best record among the candidates that are free BEFORE the call (gather
                          // kernel): the ordered resolve takes them as they are unless one of the two was taken by an earlier query
  int takenClear;         // the taken flags are all zero at the start of the call (device pipeline): the gather need not read them
};
__device__ __forceinline__ int call_nq(const CallDev& C) { return C.nqPtr ? *C.nqPtr : C.nq; }

// ---------------------------------------------------------------------------------------------------
// Projection family, phase 1: Frame::getKeyPointsInArea (frame.cpp:337-361) for one query per wave, with the
// static part of the candidate test (level window, per-axis distance test, stereo consistency) and the
// descriptor distance.  Records keep the reference's scan order (ix, iy, insertion).
// ---------------------------------------------------------------------------------------------------
// Pool layout per call: [maxQ fixed slots of kSlot records][overflow region with one atomic head per call].  A query whose
// window holds <= kSlot keypoints (the common case) writes into its own slot without any atomic; only larger windows
// allocate from the overflow region.  (A single global head serialised at ~90 returning atomics/us — MI355X_MICROARCH
// "dequeue" — and was the whole cost of the first version of this kernel.)
// The grid cell of (ix, iy) is ix*48+iy, so the cells of one window COLUMN are contiguous in the CSR: the window is
// nx <= 64 contiguous runs, visited in exactly the reference's (ix, iy, insertion) order.
// wave-wide unsigned min with DPP (quad_perm xor 1/2, row_half_mirror, row_mirror, row_bcast15/31 -> lane 63): ~8 VALU
// ops instead of six ds_bpermute round trips; this reduction runs twice per query on the serial resolve path.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define YD_DPP_MIN(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false))
  YD_DPP_MIN(0xB1, 0xF);   // quad_perm [1,0,3,2]
  YD_DPP_MIN(0x4E, 0xF);   // quad_perm [2,3,0,1]
  YD_DPP_MIN(0x141, 0xF);  // row_half_mirror
  YD_DPP_MIN(0x140, 0xF);  // row_mirror
  YD_DPP_MIN(0x142, 0xA);  // row_bcast15 into rows 1,3
  YD_DPP_MIN(0x143, 0xC);  // row_bcast31 into rows 2,3
#undef YD_DPP_MIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

constexpr int kSlot = 64;
#ifndef GATHER_QPW
#define GATHER_QPW 1
#endif
constexpr int kGatherQpw = GATHER_QPW;   // queries per wave (measured per 511 pairs x 1000 queries: 1 -> 0.268 ms, 2 -> 0.276, 4 -> 0.327: the registers of
                                          // several queries in flight cost more occupancy than the prefetched window look-up saves)
__global__ __launch_bounds__(256) void k_gather_projection(const CallDev* __restrict__ calls, const FrameDev* __restrict__ frames,
                                                           int maxQ, uint32_t* __restrict__ pool, unsigned* __restrict__ poolHeads,
                                                           unsigned poolPerCall, int* __restrict__ status) {
  __shared__ int pref[4][64];
  __shared__ int sStart[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * 4 + wv) * kGatherQpw;
  const CallDev C = calls[blockIdx.y];
  const int nqAll = min(maxQ, call_nq(C));
  if (q0 >= nqAll) return;
  const FrameDev F = frames[C.frame];
  // A wave walks kGatherQpw consecutive queries; the cell bounds of the next query are requested before the current one is processed.
  QueryDev Qs[kGatherQpw];
#pragma unroll
  for (int i = 0; i < kGatherQpw; i++) Qs[i] = C.queries[min(q0 + i, nqAll - 1)];
  struct Window { bool ok; int nx, start, cnt; };
  auto window = [&](const QueryDev& Q) -> Window {
    Window W{false, 0, 0, 0};
    if (!(Q.flags & 1)) return W;
    const int minCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, F.minX), Q.r), F.gridWInv)));
    const int maxCellX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, F.minX), Q.r), F.gridWInv)));
    const int minCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.v, F.minY), Q.r), F.gridHInv)));
    const int maxCellY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.v, F.minY), Q.r), F.gridHInv)));
    if (!(minCellX < kGridCols && maxCellX >= 0 && minCellY < kGridRows && maxCellY >= 0 && minCellX <= maxCellX && minCellY <= maxCellY)) return W;
    W.ok = true;
    W.nx = maxCellX - minCellX + 1;  // <= 64
    if (lane < W.nx) {
      const int c0 = (minCellX + lane) * kGridRows;
      W.start = F.cellStart[c0 + minCellY];
      W.cnt = F.cellStart[c0 + maxCellY + 1] - W.start;
    }
    return W;
  };
  Window Wn = window(Qs[0]);
#pragma unroll
  for (int qi = 0; qi < kGatherQpw; qi++) {
  const int q = q0 + qi;
  if (q >= nqAll) break;
  const QueryDev Q = Qs[qi];
  const Window Wc = Wn;
  if (qi + 1 < kGatherQpw && q + 1 < nqAll) Wn = window(Qs[qi + 1]);
  int2 info = make_int2(0, 0);
  if (Wc.ok) {
    {
      const int nx = Wc.nx;
      const int start = Wc.start, cnt = Wc.cnt;
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
      const int total = __shfl(incl, 63, 64);
      if (total > 0) {
        pref[wv][lane] = incl;
        sStart[wv][lane] = start;
        __builtin_amdgcn_wave_barrier();
        // static part of the candidate test for window position t (everything but the descriptor distance)
        auto staticTest = [&](int t, int& pos, int& idx) -> bool {
          int lo = 0, hi = nx - 1;  // first column whose inclusive prefix exceeds t
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (pref[wv][mid] > t) hi = mid; else lo = mid + 1; }
          pos = sStart[wv][lo] + (t - (lo ? pref[wv][lo - 1] : 0));
          const float4 kp = F.sortedKp[pos];
          const int octave = __float_as_int(kp.z);
          idx = __float_as_int(kp.w);
          bool pass = true;
          if (Q.minLevel > 0 || Q.maxLevel >= 0)
            if (octave < Q.minLevel || (Q.maxLevel >= 0 && octave < Q.maxLevel)) pass = false;
          if (!(fabsf(__fsub_rn(kp.x, Q.u)) > Q.r && fabsf(__fsub_rn(kp.y, Q.v)) < Q.r)) pass = false;
          if (pass && C.mode == 6) {
            // fuseByProjection's candidate test (orbMatcher.cpp:711-718): level predicted-1 .. predicted and the chi-square test
            // on the squared reprojection error — pow(float, 2.0) sums in double, stored to float, times the float inverse sigma^2
            const float rx = F.rightX ? F.rightX[idx] : -1.0f;
            const double ax = (double)__fsub_rn(kp.x, Q.u), ay = (double)__fsub_rn(kp.y, Q.v), ar = (double)__fsub_rn(rx, Q.ur);
            const float monoErr = (float)(ax * ax + ay * ay);
            const float stereoErr = (float)((double)monoErr + ar * ar);
            const bool lv = octave >= Q.level - 1 && octave <= Q.level;
            const float is2 = calls[blockIdx.y].invSigma2[octave & 7];   // indexed from global memory: a dynamic index into the local copy of the call record would move the whole record to scratch
            pass = lv && ((rx >= 0 && (double)__fmul_rn(stereoErr, is2) <= 7.81) || (rx < 0 && (double)__fmul_rn(monoErr, is2) <= 5.99));
          } else if (pass && C.mode == 7) {
            pass = octave >= Q.level - 1 && octave <= Q.level;   // searchByProjectionInSim's explicit level window (orbMatcher.cpp:283-285)
          } else if (pass && C.mode != 2 && F.rightX) {
            const float rx = F.rightX[idx];
            if (!(rx <= 0 || fabsf(__fsub_rn(Q.ur, rx)) <= Q.rs)) pass = false;
          }
          return pass;
        };
        // Records needed = candidates that pass the static test.  A window with <= kSlot keypoints fits its fixed slot whatever
        // passes; a larger one (wide windows of the coarse levels, dense frames: most queries of a 1241x376 / 2000-feature frame)
        // is counted first - a second walk over keypoint records that are in L1/L2 by then - so that the slot still serves
        // it in the common case and the overflow region only ever holds what is really written.
        int need = total;
        if (total > kSlot) {
          need = 0;
          for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            int pos, idx;
            const bool pass = t < total && staticTest(t, pos, idx);
            need += __popcll(__ballot(pass));
          }
        }
        unsigned base;
        bool ok = true;
        if (need <= kSlot) {
          base = blockIdx.y * poolPerCall + (unsigned)q * kSlot;
        } else {
          unsigned off = 0;
          if (lane == 0) off = atomicAdd(poolHeads + blockIdx.y, (unsigned)need);
          off = __shfl(off, 0, 64);
          const unsigned ovfBase = (unsigned)maxQ * kSlot;
          ok = ovfBase + off + (unsigned)need <= poolPerCall;
          base = blockIdx.y * poolPerCall + ovfBase + off;
          if (!ok && lane == 0) atomicMax(status, 1);
        }
        if (ok && need > 0) {
          int written = 0;
          unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, r1 = 0, r2 = 0;   // this lane's two smallest keys (dist << 23 | record position) and their records
          const bool pre = C.qPre != nullptr, readTaken = pre && C.taken && !C.takenClear;
          const uint4* qd = reinterpret_cast<const uint4*>(C.qdesc + (size_t)q * 32);
          const uint4 qa = qd[0], qb = qd[1];
          for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            int pos = 0, idx = 0, dist = 0;
            const bool pass = t < total && staticTest(t, pos, idx);
            if (pass) {
              const uint4* td = reinterpret_cast<const uint4*>(F.sortedDesc + (size_t)pos * 32);
              const uint4 ta = td[0], tb = td[1];
              dist = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) + __popc(qb.x ^ tb.x) +
                     __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
            }
            const unsigned long long m = __ballot(pass);
            if (pass) {
              const int wpos = written + __popcll(m & ((1ull << lane) - 1ull));
              const uint32_t rec = ((uint32_t)dist << 16) | (uint32_t)idx;
              pool[base + wpos] = rec;
              if (pre && !(readTaken && C.taken[idx])) {
                const unsigned key = ((unsigned)dist << 23) | (unsigned)wpos;
                if (key < k1) { k2 = k1; r2 = r1; k1 = key; r1 = rec; } else if (key < k2) { k2 = key; r2 = rec; }
              }
            }
            written += __popcll(m);
          }
          if (pre) {   // first minimum and first minimum of the rest over the wave (keys are distinct: the position is part of them)
            const unsigned b1 = wave_min_u32(k1);
            const bool own = k1 == b1 && b1 != 0xFFFFFFFFu;
            const unsigned long long o1 = __ballot(own);
            const unsigned br = o1 ? (unsigned)__builtin_amdgcn_readlane((int)r1, (int)__builtin_ctzll(o1)) : 0xFFFFFFFFu;
            if (own) { k1 = k2; r1 = r2; }
            const unsigned b2 = wave_min_u32(k1);
            const unsigned long long o2 = __ballot(k1 == b2 && b2 != 0xFFFFFFFFu);
            const unsigned sr = o2 ? (unsigned)__builtin_amdgcn_readlane((int)r1, (int)__builtin_ctzll(o2)) : 0xFFFFFFFFu;
            if (lane == 0) C.qPre[q] = make_uint2(o1 ? br : 0xFFFFFFFFu, sr);
          }
          info = make_int2((int)base, written);
        }
      }
    }
  }
  if (lane == 0) C.qInfo[q] = info;
  __builtin_amdgcn_wave_barrier();   // pref / sStart are reused by the wave's next query
  }
}

// BoW family, phase 1: the candidate list of a query is a vocabulary-node bucket of the other frame
// (orbMatcher.cpp:323-335, 404-417): qRange[q] = [begin,end) into featB, qFeat[q] = feature index in A.
struct BowCallDev {
  const uint8_t* descA; const uint8_t* descB;
  const int* qFeat; const int2* qRange; const int* featB; const uint8_t* validB;  // validB null for mode 3
  int nq;
  int2* qInfo;
  // searchForTriangulation only (tri != 0): first/second keyframe keypoints, stereo flags, F (row-major), epipole, level tables
  int tri;
  const KeyPointDev* kpsA; const KeyPointDev* kpsB;
  const uint8_t* goodA; const uint8_t* goodB;
  float F[9], ex, ey, sfB[8], sf2B[8];
};
// isEpipolarLineDistCorrect (orbMatcher.cpp:808-819) and the epipole-distance exemption (:503-506), every operation as written
__device__ __forceinline__ bool tri_pair_ok(const BowCallDev& B, const KeyPointDev& k1, bool good1, const KeyPointDev& k2, bool good2) {
  if (!(good1 || good2)) {
    const float dx = __fsub_rn(B.ex, k2.x), dy = __fsub_rn(B.ey, k2.y);
    const double far = (double)dx * (double)dx + (double)dy * (double)dy;      // pow(float, 2.0) + pow(float, 2.0): exact squares in double
    if (!(far >= (double)__fmul_rn(100.0f, B.sfB[k2.octave]))) return false;
  }
  const float la = __fadd_rn(__fadd_rn(__fmul_rn(B.F[0], k1.x), __fmul_rn(B.F[3], k1.y)), B.F[6]);
  const float lb = __fadd_rn(__fadd_rn(__fmul_rn(B.F[1], k1.x), __fmul_rn(B.F[4], k1.y)), B.F[7]);
  const float lc = __fadd_rn(__fadd_rn(__fmul_rn(B.F[2], k1.x), __fmul_rn(B.F[5], k1.y)), B.F[8]);
  const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
  if (!(den > 0.0f)) return false;
  const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2.x), __fmul_rn(lb, k2.y)), lc);
  return (double)__fdiv_rn(__fmul_rn(num, num), __fmul_rn(den, den)) < 3.841 * (double)B.sf2B[k2.octave];
}
__global__ __launch_bounds__(256) void k_gather_bow(BowCallDev B, uint32_t* __restrict__ pool, unsigned* __restrict__ poolHead,
                                                    unsigned poolCap, int* __restrict__ status) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wv;
  if (q >= B.nq) return;
  const int2 rg = B.qRange[q];
  const int total = rg.y - rg.x;
  int2 info = make_int2(0, 0);
  if (total > 0) {
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(poolHead, (unsigned)total);
    base = __shfl(base, 0, 64);
    if (base + (unsigned)total > poolCap) {
      if (lane == 0) atomicMax(status, 1);
    } else {
      const uint8_t* qd = B.descA + (size_t)B.qFeat[q] * 32;
      int written = 0;
      for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        bool pass = false;
        int idx = 0, dist = 0;
        if (t < total) {
          idx = B.featB[rg.x + t];
          pass = !B.validB || B.validB[idx];
          if (pass) dist = hamming256(qd, B.descB + (size_t)idx * 32);
          if (pass && B.tri) {   // static part of searchForTriangulation's acceptance test (the dynamic part, `<= best` and `matched`, is the replay's)
            const int i1 = B.qFeat[q];
            pass = dist <= kThLow && tri_pair_ok(B, B.kpsA[i1], B.goodA[i1] != 0, B.kpsB[idx], B.goodB[idx] != 0);
          }
        }
        const unsigned long long m = __ballot(pass);
        if (pass) pool[base + written + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)dist << 16) | (uint32_t)idx;
        written += __popcll(m);
      }
      info = make_int2((int)base, written);
    }
  }
  if (lane == 0) B.qInfo[q] = info;
}

// ---------------------------------------------------------------------------------------------------
// Phase 2: ordered replay.  best = first minimum over the not-taken candidates in scan order; second = first
// minimum of the rest — which is exactly what the if / else-if chain at orbMatcher.cpp:44-53 leaves behind.
// ---------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------
// MapPoint::computeDistinctiveDescriptors (reference src/mapPoint.cpp:191-213), a batch of map points: one wave per point.
// Lane = descriptor row i (rows beyond 64 in further rounds).  The median the reference takes after sorting a row of the
// distance matrix, v[(int)(0.5 m)], is the smallest t with #{j : d(i,j) <= t} > (int)(0.5 m): nine bisection steps over
// t in [0, 256], each a pass over the other rows — whose addresses are wave-uniform, so they arrive as scalar loads.  The
// winner is the first row with the least median: wave-min of (median << 16 | row).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t* __restrict__ desc, const int* __restrict__ offsets, int nPoints,
                                                     int* __restrict__ best) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= nPoints) return;
  const int o0 = __builtin_amdgcn_readfirstlane(offsets[p]), m = __builtin_amdgcn_readfirstlane(offsets[p + 1]) - o0;
  if (m <= 0) { if (lane == 0) best[p] = -1; return; }
  const uint8_t* D = desc + (size_t)o0 * 32;
  const int kth = (int)(0.5 * m);
  unsigned win = 0xFFFFFFFFu;
  for (int i0 = 0; i0 < m; i0 += 64) {
    const int i = i0 + lane;
    if (i < m) {
      const uint4 a0 = *reinterpret_cast<const uint4*>(D + (size_t)i * 32), a1 = *reinterpret_cast<const uint4*>(D + (size_t)i * 32 + 16);
      int lo = 0, hi = 256;   // smallest t with count(d <= t) > kth
      while (lo < hi) {
        const int t = (lo + hi) >> 1;
        int cnt = 0;
        for (int j = 0; j < m; j++) {
          const uint4 b0 = *reinterpret_cast<const uint4*>(D + (size_t)j * 32), b1 = *reinterpret_cast<const uint4*>(D + (size_t)j * 32 + 16);
          const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                        __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
          cnt += d <= t;
        }
        if (cnt > kth) hi = t; else lo = t + 1;
      }
      win = min(win, ((unsigned)lo << 16) | (unsigned)min(i, 65535));
    }
  }
  win = wave_min_u32(win);
  if (lane == 0) best[p] = (int)(win & 0xFFFFu);
}

constexpr int kResolveWindow = 512;   // queries whose (base,count) are staged in LDS at a time (10 KB: this one-wave workgroup's LDS is what the other
                                       // lanes' kernels cannot use meanwhile; 2048 = 41 KB x 4 workgroups per CU left them none)

__global__ __launch_bounds__(64) void k_resolve(const CallDev* __restrict__ calls, const FrameDev* __restrict__ frames,
                                                const uint32_t* __restrict__ pool, int takenWords) {
  extern __shared__ unsigned takenBits[];  // [takenWords], then claimBits [takenWords] (the group-parallel replay's scratch)
  __shared__ int hist[kHistLen];
  __shared__ int3 live[kResolveWindow];    // (query, pool base, record count) of the window's non-empty queries, in order
  __shared__ uint2 livePre[kResolveWindow]; // their pre-computed (best, second) records (projection family; see CallDev::qPre)
  const int lane = threadIdx.x;
#ifdef RESOLVE_TIMING
  long long tclk[6]; tclk[0] = wall_clock64();
#define RS_CLK(i) tclk[i] = wall_clock64()
#else
#define RS_CLK(i) do { } while (0)
#endif
#ifndef YDORB_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);   // one wave per call, serial by definition: its latency is the call's latency (see k_stereo)
#endif
  const CallDev C = calls[blockIdx.x];
  const int nq = call_nq(C);
  const bool bow = C.mode >= 3 && C.mode <= 5;
  const bool lastWins = C.mode == 5;
  const auto kps = G(C.tkps);
  const auto gTaken = G(C.taken);
  const auto gMatchQ = G(C.matchQ), gAssigned = G(C.assigned), gCount = G(C.count);
  const auto gQInfo = G(reinterpret_cast<const int*>(C.qInfo));          // (HIP's vector classes cannot be copied out of a typed address space)
  const auto gQPre = G(reinterpret_cast<const unsigned*>(C.qPre));
  const auto gQueries = G(C.queries);
  const auto gQAngle = G(C.qAngle);
  const int n = takenWords * 32;
  unsigned* claimBits = takenBits + takenWords;
  for (int w = lane; w < takenWords; w += 64) {
    unsigned bits = 0;
    for (int b = 0; b < 32; b++) {
      const int i = w * 32 + b;
      if (C.taken && i < n && gTaken[i]) bits |= 1u << b;
    }
    takenBits[w] = bits;
    claimBits[w] = 0;
  }
  if (lane < kHistLen) hist[lane] = 0;
  for (int q = lane; q < nq; q += 64) gMatchQ[q] = -1;
  __syncthreads();
  RS_CLK(1);
  int matchNum = 0;
  for (int w0 = 0; w0 < nq; w0 += kResolveWindow) {
    // stage the window's non-empty queries (ordered ballot compaction): the serial loop then touches LDS only
    int nLive = 0;
    for (int q0 = w0; q0 < min(nq, w0 + kResolveWindow); q0 += 64) {
      const int q = q0 + lane;
      int2 info = make_int2(0, 0);
      if (q < nq) info = make_int2(gQInfo[2 * q], gQInfo[2 * q + 1]);
      const unsigned long long m = __ballot(info.y > 0);
      if (info.y > 0) {
        const int slot = nLive + __popcll(m & ((1ull << lane) - 1ull));
        // bit 30 of the query index: an accepted match marks its keypoint as taken (frame.m_v_sptrMapPoints[idx] with observations, :40 /
        // :103): staged here so that the replay below never waits for a dependent global load
        const int qflags = bow ? 3 : gQueries[q].flags;
        live[slot] = make_int3(q | ((qflags & 2) ? (1 << 30) : 0), info.x, info.y);
        if (C.qPre) livePre[slot] = make_uint2(gQPre[2 * q], gQPre[2 * q + 1]);
      }
      nLive += __popcll(m);
    }
    __syncthreads();
#ifdef RESOLVE_TIMING
    tclk[5] = wall_clock64();
#endif
    // The replay proper, in two instantiations: with the gather kernel's pre-computed (best, second) - the projection searches - the
    // common path runs no wave reduction and waits for no record load.  What is left is ~100 dependent instructions of ONE wave per
    // query (a lone wave issues one every 4-8 cycles) and three LDS round trips: 0.4 us per query, 0.28 ms per 1000-query frame pair,
    // measured with in-kernel stamps (-DRESOLVE_TIMING); the pairs of a call run side by side.
    auto replay = [&](auto preTag) {
    constexpr bool havePre = decltype(preTag)::value;
    const bool needSecond = C.mode == 0;
    unsigned recNext = 0;
    if (!havePre && nLive > 0 && lane < live[0].z) recNext = pool[live[0].y + lane];
    auto serial_one = [&](int e) {
      const int3 L3 = live[e];
      const int q = L3.x & 0x3FFFFFFF;
      const bool qObs = (L3.x >> 30) & 1;
      unsigned best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;  // key = dist(9) << 23 | scan position(23)
      unsigned bestRec = 0, secondRec = 0;
      bool reduce = true;
      if (havePre) {
        // The gather kernel left the best / second-best record among the candidates that were free before the call.  The free set
        // only shrinks while the queries are replayed, so they still are the first minimum and the first minimum of the rest unless
        // an earlier query took one of them: only then the records are reduced again (rare: ~50 of 1000 queries assign anything).
        const uint2 pr = livePre[e];
        const bool t1 = pr.x != 0xFFFFFFFFu && ((takenBits[(pr.x & 0xFFFFu) >> 5] >> (pr.x & 31u)) & 1u);
        const bool t2 = needSecond && pr.y != 0xFFFFFFFFu && ((takenBits[(pr.y & 0xFFFFu) >> 5] >> (pr.y & 31u)) & 1u);
        if (!t1 && !t2) {
          reduce = false;
          if (pr.x != 0xFFFFFFFFu) { best = (pr.x >> 16) << 23; bestRec = pr.x; }
          if (pr.y != 0xFFFFFFFFu) { second = (pr.y >> 16) << 23; secondRec = pr.y; }
        }
      }
      unsigned recFirst = recNext;
      if (!havePre && e + 1 < nLive) {  // prefetch the next query's first 64 records while this one is reduced
        const int3 N3 = live[e + 1];
        recNext = lane < N3.z ? pool[N3.y + lane] : 0u;
      }
      if (havePre && reduce && lane < L3.z) recFirst = pool[L3.y + lane];
      for (int t0 = 0; reduce && t0 < L3.z; t0 += 64) {
        const int t = t0 + lane;
        unsigned key = 0xFFFFFFFFu, rec = 0;
        if (t < L3.z) {
          rec = t0 == 0 ? recFirst : pool[L3.y + t];
          const unsigned idx = rec & 0xFFFFu;
          // searchForTriangulation keeps a candidate when `dist <= best` (:503): of equal distances the LAST one wins, so its scan
          // position is stored complemented and the same first-minimum reduction picks it
          if (!((takenBits[idx >> 5] >> (idx & 31)) & 1u)) key = ((rec >> 16) << 23) | (lastWins ? 0x7FFFFFu - (unsigned)t : (unsigned)t);
        }
        const unsigned b1 = wave_min_u32(key);
        if (b1 == 0xFFFFFFFFu) continue;
        const unsigned b2 = wave_min_u32(key == b1 ? 0xFFFFFFFFu : key);
        // keys are unique, so the low 6 bits of the scan position name the owning lane of this chunk
        const unsigned l1 = lastWins ? (0x7FFFFFu - (b1 & 0x7FFFFFu)) & 63u : b1 & 63u, l2 = lastWins ? (0x7FFFFFu - (b2 & 0x7FFFFFu)) & 63u : b2 & 63u;
        const unsigned r1 = (unsigned)__builtin_amdgcn_readlane((int)rec, (int)l1);
        const unsigned r2 = b2 != 0xFFFFFFFFu ? (unsigned)__builtin_amdgcn_readlane((int)rec, (int)l2) : 0u;
        if (b1 < best) {
          if (best < b2) { second = best; secondRec = bestRec; } else { second = b2; secondRec = r2; }
          best = b1; bestRec = r1;
        } else if (b1 < second) { second = b1; secondRec = r1; }
      }
      if (best != 0xFFFFFFFFu) {
        const int bestDist = (int)(best >> 23), secondDist = second == 0xFFFFFFFFu ? 256 : (int)(second >> 23);
        const int bestIdx = (int)(bestRec & 0xFFFFu);
        bool accept;
        if (C.mode == 0) {
          const int bestLevel = kps[bestIdx].octave, secondLevel = second == 0xFFFFFFFFu ? -1 : kps[secondRec & 0xFFFFu].octave;
          accept = bestDist <= kThHigh && (bestLevel != secondLevel || (float)bestDist <= __fmul_rn(C.ratio, (float)secondDist));
        } else if (C.mode == 1) accept = bestDist < kThHigh;
        else if (C.mode == 2) accept = bestDist <= C.orbDist;
        else if (C.mode == 5) accept = true;   // every record already passed dist <= 50 and the geometric tests
        else if (C.mode == 6) accept = bestDist <= C.orbDist;   // fuseByProjection :725 / fuseBySim3 :789 (50), searchBySim3 :625 (100); nothing is taken
        else if (C.mode == 7) accept = bestDist <= kThLow;   // searchByProjectionInSim :293
        else accept = bestDist <= kThLow && (float)bestDist < __fmul_rn(C.ratio, (float)secondDist);
        if (accept) {
          matchNum++;
          if (lane == 0) {
            if (C.mode == 4 || C.mode == 5) gAssigned[q] = bestIdx;   // out[firstIdx] = second-keyframe index; q is remapped by the host
            else gAssigned[bestIdx] = q;
            const bool nowTaken = C.mode == 6 ? false : C.mode >= 2 ? true : qObs;
            if (nowTaken) takenBits[bestIdx >> 5] |= 1u << (bestIdx & 31);
            gMatchQ[q] = bestIdx;
          }
          // (one wave per call: LDS accesses of a wave are ordered, so the next query sees the bit; no barrier and no wait for the stores)
          __builtin_amdgcn_wave_barrier();
        }
      }
    };
    if (!havePre) {
      for (int e = 0; e < nLive; e++) serial_one(e);
    } else {
      // With the gather kernel's (best, second) records a query's outcome is known without looking at any other query - unless an
      // earlier query of the replay takes (or is assigned) the keypoint one of its records names.  So the queries go 64 at a time, one
      // per lane, and every lane decides from its own records.
      //  * Modes whose acceptance is `best distance within a bound` alone (1, 2, 6, 7): losing candidates can only RAISE a query's best
      //    distance, so a query whose pre-computed best fails stays rejected whatever the others take - it is inert.  Of the accepting
      //    lanes, the first one whose keypoint is already taken or is also the keypoint of a LOWER accepting lane is the first whose
      //    outcome depends on the order: the lanes below it are committed in one step, that one query is replayed by the serial form
      //    (which defines the result), and the rest of the group is looked at again.
      //  * The other modes (ratio / level tests on the second-best record): the group is committed in one step when no record names a
      //    keypoint taken before the group or claimed by another lane, else replayed in order.
      // ~120 of 1000 queries accept anything: a dozen steps per group of 64 at worst instead of a 1000-step chain per frame pair.
      const bool monotone = C.mode == 1 || C.mode == 2 || C.mode == 6 || C.mode == 7;
      for (int g0 = 0; g0 < nLive; g0 += 64) {
        const int e = g0 + lane;
        const bool in = e < nLive;
        const int3 L3 = in ? live[e] : make_int3(0, 0, 0);
        const uint2 pr = in ? livePre[e] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        const bool has1 = pr.x != 0xFFFFFFFFu, has2 = pr.y != 0xFFFFFFFFu;
        const unsigned i1 = pr.x & 0xFFFFu, i2 = pr.y & 0xFFFFu;
        bool would = false;              // the outcome if no other query interfered
        if (has1) {
          const int bestDist = (int)(pr.x >> 16), secondDist = has2 ? (int)(pr.y >> 16) : 256;
          if (C.mode == 0) {
            const int bestLevel = kps[i1].octave, secondLevel = has2 ? kps[i2].octave : -1;
            would = bestDist <= kThHigh && (bestLevel != secondLevel || (float)bestDist <= __fmul_rn(C.ratio, (float)secondDist));
          } else if (C.mode == 1) would = bestDist < kThHigh;
          else if (C.mode == 2) would = bestDist <= C.orbDist;
          else if (C.mode == 5) would = true;
          else if (C.mode == 6) would = bestDist <= C.orbDist;
          else if (C.mode == 7) would = bestDist <= kThLow;
          else would = bestDist <= kThLow && (float)bestDist < __fmul_rn(C.ratio, (float)secondDist);
        }
        auto commit = [&](bool mine) {     // the accepted queries of `mine` lanes: assignment, taken bit, match list
          matchNum += __popcll(__ballot(mine));
          if (mine) {
            const int q = L3.x & 0x3FFFFFFF;
            const bool qObs = (L3.x >> 30) & 1;
            if (C.mode == 4 || C.mode == 5) gAssigned[q] = (int)i1;
            else gAssigned[i1] = q;
            const bool nowTaken = C.mode == 6 ? false : C.mode >= 2 ? true : qObs;
            if (nowTaken) atomicOr(&takenBits[i1 >> 5], 1u << (i1 & 31u));
            gMatchQ[q] = (int)i1;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
        };
        if (monotone) {
          int first = 0;                   // lanes below `first` are settled
          while (true) {
            const bool acc = would && lane >= first;
            bool conflict = acc && ((takenBits[i1 >> 5] >> (i1 & 31u)) & 1u);
            for (unsigned long long rem = __ballot(acc); rem; rem &= rem - 1ull) {   // the same keypoint on a lower accepting lane
              const int a = __ffsll((long long)rem) - 1;
              const unsigned ia = (unsigned)__builtin_amdgcn_readlane((int)i1, a);
              conflict = conflict || (acc && lane > a && i1 == ia);
            }
            const unsigned long long cm = __ballot(conflict);
            const int c = cm ? __ffsll((long long)cm) - 1 : 64;
            commit(acc && lane < c);
            if (c == 64) break;
            serial_one(g0 + c);
            first = c + 1;
          }
        } else {
          const bool stale = (has1 && ((takenBits[i1 >> 5] >> (i1 & 31u)) & 1u)) || (needSecond && has2 && ((takenBits[i2 >> 5] >> (i2 & 31u)) & 1u));
          const bool accept = would && !stale;
          bool conflict = stale;
          if (accept) conflict = conflict || ((atomicOr(&claimBits[i1 >> 5], 1u << (i1 & 31u)) >> (i1 & 31u)) & 1u);   // two accepted queries, one keypoint
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
          if (has1 && !accept) conflict = conflict || ((claimBits[i1 >> 5] >> (i1 & 31u)) & 1u);                        // its best keypoint goes to another query
          if (needSecond && has2) conflict = conflict || ((claimBits[i2 >> 5] >> (i2 & 31u)) & 1u);                      // its second keypoint goes to another query
          const bool redo = __ballot(conflict) != 0ull;
          __builtin_amdgcn_wave_barrier();
          if (accept) atomicAnd(&claimBits[i1 >> 5], ~(1u << (i1 & 31u)));
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
          if (redo) for (int e2 = g0; e2 < min(g0 + 64, nLive); e2++) serial_one(e2);
          else commit(accept);
        }
      }
    }
    };
    if (C.qPre) replay(std::true_type{}); else replay(std::false_type{});
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  RS_CLK(2);
  // rotation histogram (orbMatcher.cpp:119-153): bin = round((a1 - a2 [+360]) / 30), keep the three largest bins
  if (C.mode != 0 && C.checkOri) {
    const float factor = (float)(1.0 / kHistLen);
    for (int q = lane; q < nq; q += 64) {
      const int t = gMatchQ[q];
      if (t < 0) continue;
      const float a1 = C.qAngle ? gQAngle[q] : gQueries[q].angle;
      const float a2 = kps[t].angle;
      float rot = __fsub_rn(a1, a2);
      if (rot < 0.0f) rot = (float)((double)rot + 360.0);
      int bin = (int)roundf(__fmul_rn(rot, factor));
      if (bin == kHistLen) bin = 0;
      atomicAdd(&hist[bin], 1);
      gMatchQ[q] = t | (bin << 24);
    }
    __syncthreads();
    int i1 = -1, i2 = -1, i3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < kHistLen; i++) {  // computeThreeMaxima, orbMatcher.cpp:827-854 (redundantly per lane)
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
      else if (s > max3) { max3 = s; i3 = i; }
    }
    if (max2 < max1 / 10) { i3 = -1; i2 = -1; }
    else if (max3 < max1 / 10) { i3 = -1; }
    int culled = 0;
    for (int q = lane; q < nq; q += 64) {
      const int v = gMatchQ[q];
      if (v < 0) continue;
      const int bin = v >> 24, t = v & 0xFFFFFF;
      if (bin != i1 && bin != i2 && bin != i3) {
        if (C.mode == 4 || C.mode == 5) gAssigned[q] = -1; else gAssigned[t] = -1;
        culled++;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) culled += __shfl_xor(culled, o, 64);
    matchNum -= culled;
  }
  __syncthreads();
  RS_CLK(3);
  if (C.taken)
    for (int i = lane; i < n; i += 64) {
      const int w = i >> 5;
      if (w < takenWords) gTaken[i] = (takenBits[w] >> (i & 31)) & 1u;
    }
  if (lane == 0) *gCount = matchNum;
#ifdef RESOLVE_TIMING
  RS_CLK(4);
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) printf("resolve blk %d: init %.1f us, staging %.1f us, loop %.1f us, histogram %.1f us, tail %.1f us (nq %d, matches %d)\n", (int)blockIdx.x,
      (tclk[1] - tclk[0]) / 100.0, (tclk[5] - tclk[1]) / 100.0 + 1000 * 0 + 0.0 * (tclk[2] - tclk[1]) + (tclk[2] - tclk[5]) / 100.0 * 0 + 0, (tclk[2] - tclk[5]) / 100.0, (tclk[3] - tclk[2]) / 100.0, (tclk[4] - tclk[3]) / 100.0, nq, matchNum);
#endif
}

// queries for the device-resident consecutive-frame search (bench / streaming pipeline): the "projection" of a
// last-frame keypoint is its own position moved by a per-pair 2x3 affine map (identity = constant-position
// motion model); window th * scaleFactor[octave], level window octave-1..octave+1 — the non-forward /
// non-backward branch of orbMatcher.cpp:95-101.
__global__ __launch_bounds__(256) void k_queries_from_keypoints(const CallDev* __restrict__ calls, int cap, const float* __restrict__ affine, float th,
                                                                const float* __restrict__ scaleFactors, int nLevels,
                                                                float minX, float maxX, float minY, float maxY, unsigned* __restrict__ poolHeads) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  // the call's clean slate rides along (three memset launches on the stream otherwise): nothing assigned, nothing taken, empty overflow pool
  calls[f].assigned[i] = -1;
  calls[f].taken[i] = 0;
  if (i == 0 && poolHeads) poolHeads[f] = 0;
  const KeyPointDev* __restrict__ kps = calls[f].qkps;
  QueryDev* __restrict__ out = const_cast<QueryDev*>(calls[f].queries);
  QueryDev Q{};
  if (i < *calls[f].nqPtr) {
    const KeyPointDev kp = kps[i];
    const float* A = affine + (size_t)f * 6;
    Q.u = __fadd_rn(__fadd_rn(__fmul_rn(A[0], kp.x), __fmul_rn(A[1], kp.y)), A[2]);
    Q.v = __fadd_rn(__fadd_rn(__fmul_rn(A[3], kp.x), __fmul_rn(A[4], kp.y)), A[5]);
    const int oct = min(max(kp.octave, 0), nLevels - 1);
    Q.r = __fmul_rn(th, scaleFactors[oct]);
    Q.minLevel = oct - 1;
    Q.maxLevel = oct + 1;
    Q.ur = 0; Q.rs = 0;
    Q.angle = kp.angle;
    Q.level = oct;
    const bool inImage = Q.u >= minX && Q.u < maxX && Q.v >= minY && Q.v < maxY;  // Frame::isInImage, frame.cpp:291-294
    Q.flags = inImage ? 3 : 0;
  }
  out[i] = Q;
}

// ---------------------------------------------------------------------------------------------------
// Brute-force 256-bit Hamming top-2 (north_star; SURVEY 8(b) ydorb_hamming_topk).  The answer per query is what the reference's
// if / else-if chain leaves after walking the candidates in order (orbMatcher.cpp:39-52, 327-334, ...): best = FIRST candidate with
// the least distance, second = first candidate with the least distance among the others; both start at 256 and only a strictly
// smaller distance replaces them.  With key = distance << 16 | rank (rank = position in the candidate list, so keys are distinct and
// ordered by (distance, rank)) that is: the two smallest keys below 256 << 16.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
struct TopkOut { int bestDist, bestIdx, secondDist, secondIdx, bestRank, secondRank; };   // == YdMatch2
constexpr unsigned kTopkInit = 256u << 16;

// all pairs, one lane per query: the lane keeps its query in 8 registers, the workgroup stages 256 targets at a time in LDS and every
// lane walks them with wave-uniform (broadcast) ds_read_b128: per pair 8 v_xor + 8 v_bcnt (accumulating) + key + min + med3.
// grid = (ceil(cap / 256), pairs).  q / t: [pairs][cap][32] with per-pair counts nq / nt (device).
__global__ __launch_bounds__(256) void k_topk_allpairs(const uint8_t* __restrict__ qdesc, const int* __restrict__ nqPtr, size_t qStride,
                                                       const uint8_t* __restrict__ tdesc, const int* __restrict__ ntPtr, size_t tStride,
                                                       int cap, TopkOut* __restrict__ out) {
  __shared__ __align__(16) uint4 tl[256][2];
  const int pr = blockIdx.y;
  const int nq = min(nqPtr[pr], cap), nt = min(ntPtr[pr], cap);
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= nq) return;   // the whole workgroup is past the queries
  const uint8_t* Q = qdesc + (size_t)pr * qStride;
  const uint8_t* T = tdesc + (size_t)pr * tStride;
  uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
  if (q < nq) { qa = reinterpret_cast<const uint4*>(Q + (size_t)q * 32)[0]; qb = reinterpret_cast<const uint4*>(Q + (size_t)q * 32)[1]; }
  unsigned k1 = kTopkInit, k2 = kTopkInit;
  for (int t0 = 0; t0 < nt; t0 += 256) {
    __syncthreads();
    if (t0 + (int)threadIdx.x < nt) {
      tl[threadIdx.x][0] = reinterpret_cast<const uint4*>(T + (size_t)(t0 + threadIdx.x) * 32)[0];
      tl[threadIdx.x][1] = reinterpret_cast<const uint4*>(T + (size_t)(t0 + threadIdx.x) * 32)[1];
    }
    __syncthreads();
    auto one = [&](int j) {
      const uint4 ta = tl[j][0], tb = tl[j][1];
      const unsigned d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) + __popc(qb.x ^ tb.x) +
                         __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
      const unsigned key = (d << 16) | (unsigned)(t0 + j);
      const unsigned lo = min(k1, key);
      k2 = umed3(k1, k2, key);   // second smallest of (k1 <= k2, key)
      k1 = lo;
    };
    if (nt - t0 >= 256) {
#pragma unroll 4
      for (int j = 0; j < 256; j++) one(j);
    } else {
      for (int j = 0; j < nt - t0; j++) one(j);
    }
  }
  if (q < nq) {
    TopkOut o;
    o.bestDist = (int)(k1 >> 16); o.bestIdx = o.bestRank = k1 < kTopkInit ? (int)(k1 & 0xFFFFu) : -1;
    o.secondDist = (int)(k2 >> 16); o.secondIdx = o.secondRank = k2 < kTopkInit ? (int)(k2 & 0xFFFFu) : -1;
    out[(size_t)pr * cap + q] = o;
  }
}

// candidate lists (CSR), one wave per query: lanes stride over the list, each keeps its two smallest keys; the wave's best is a
// DPP wave-min (wave_min_u32 above), the lane that held it promotes its own runner-up, and a second wave-min gives the second.
__global__ __launch_bounds__(256) void k_topk_csr(const uint8_t* __restrict__ qdesc, int nq, const uint8_t* __restrict__ tdesc, int nt,
                                                  const int* __restrict__ candOff, const int* __restrict__ candIdx, TopkOut* __restrict__ out) {
  const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const int c0 = candOff ? candOff[q] : 0, c1 = candOff ? candOff[q + 1] : nt;
  const uint4 qa = reinterpret_cast<const uint4*>(qdesc + (size_t)q * 32)[0], qb = reinterpret_cast<const uint4*>(qdesc + (size_t)q * 32)[1];
  unsigned k1 = kTopkInit, k2 = kTopkInit;
  for (int r = lane; r < c1 - c0; r += 64) {
    const int idx = candIdx ? candIdx[c0 + r] : r;
    if (idx < 0 || idx >= nt) continue;
    const uint4 ta = reinterpret_cast<const uint4*>(tdesc + (size_t)idx * 32)[0], tb = reinterpret_cast<const uint4*>(tdesc + (size_t)idx * 32)[1];
    const unsigned d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) + __popc(qb.x ^ tb.x) +
                       __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
    const unsigned key = (d << 16) | (unsigned)r;
    const unsigned lo = min(k1, key);
    k2 = umed3(k1, k2, key);
    k1 = lo;
  }
  const unsigned b = wave_min_u32(k1);
  if (k1 == b && b < kTopkInit) k1 = k2;          // keys are distinct: exactly one lane held the best
  const unsigned s2 = wave_min_u32(k1);
  if (lane == 0) {
    TopkOut o;
    o.bestDist = (int)(b >> 16); o.bestRank = b < kTopkInit ? (int)(b & 0xFFFFu) : -1;
    o.secondDist = (int)(s2 >> 16); o.secondRank = s2 < kTopkInit ? (int)(s2 & 0xFFFFu) : -1;
    o.bestIdx = o.bestRank < 0 ? -1 : (candIdx ? candIdx[c0 + o.bestRank] : o.bestRank);
    o.secondIdx = o.secondRank < 0 ? -1 : (candIdx ? candIdx[c0 + o.secondRank] : o.secondRank);
    out[q] = o;
  }
}

}  // namespace ydorb
