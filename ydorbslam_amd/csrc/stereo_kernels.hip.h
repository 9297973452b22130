// Stereo association of one rectified pair: Frame::computeStereoMatches (reference src/frame.cpp:362-477).
//
// Per left keypoint: best right descriptor among the right keypoints whose row band covers the left row (:368-402), an
// 11x11 L1 block match over 11 column shifts at the keypoint's pyramid level (:406-436), a parabola through the three
// distances round the best shift (:441-446) and disparity -> depth (:448-459); per pair: the outlier rule on the sorted
// block-match minima (:464-472).
//
// The reference indexes the left descriptor and the output slot with a counter that only advances when a keypoint reaches
// the end of the loop body (:462), which makes the result a serial chain over the keypoints: s(k+1) = s(k) + c(k, s(k)), where
// c says whether keypoint k, matched with descriptor s(k), reaches `leftIdx++`.  k_stereo<true> replays that chain exactly: the
// four waves of a pair's workgroup take the keypoints in turn, each evaluating its keypoint with the index it expects (c = 1 for the
// keypoints still in flight) and committing in keypoint order once the exact index is published (see the kernel); the 64 lanes of a
// wave share the keypoint's candidate scan and block match, and the loads a step depends on (the wave's next keypoint, the two descriptor
// rows its next guess can be) are issued one step ahead.  (Solving the chain as a fixed
// point of parallel evaluation rounds was measured and dropped: with a lagging index ~3 % of the evaluations still end in a
// `continue`, each depending on s, so the settled prefix grows by only 10-50 keypoints per round - DESIGN.md.)
// k_stereo<false> is the per-keypoint form (descriptor and slot = the keypoint's own index, YDORB_STEREO_INDEX_BY_KEYPOINT)
// with one wave per keypoint.  Everything else is the same device function, restated in oracle/stereo_oracle.cpp.
//
// Row bands: instead of the reference's per-row index lists the kernel keeps one (lo, hi, octave, x) record per right
// keypoint in LDS and scans them in index order — the same candidates in the same order (the lists are filled in keypoint
// order, :373-379), so "first minimum" is the minimum of (distance << 16 | index).  The replay form adds an index of the right
// keypoints sorted by the FIRST row of their band (2 bytes each): the candidates of a row are then the entries whose first row lies in
// [row - widest band, row], filtered by the band test.  (The reference's lists themselves - every keypoint in each of its ~9 rows -
// took 44 KB of LDS per pair at 2000 features; the workgroup's LDS is what the other lanes' kernels cannot use while the serial walk
// runs: +32 KB cost the pipeline 12 %, tools/bench_stereo_lds.sh.)
#pragma once
#include "match_kernels.hip.h"

#pragma clang fp contract(off)

namespace ydorb {

constexpr int kStereoChunk = 32;       // left keypoints per workgroup in the per-keypoint form
constexpr int kStereoMaxRight = 8192;  // right keypoints per pair (LDS table: 8 bytes each)
constexpr int kStereoOrbDist = (100 + 50) / 2;   // (m_int_highThd + m_int_lowThd) / 2, frame.cpp:365
#ifndef STEREO_REPLAY_WAVES
#define STEREO_REPLAY_WAVES 4
#endif
constexpr int kStereoReplayWaves = STEREO_REPLAY_WAVES;   // waves that share the serial walk of a pair (1: the plain walk; n: n - 1 steps of speculation;
                                                            // one pair of 2000 keypoints: 2.40 / 1.71 / 1.37 / 1.21 ms for 1 / 2 / 3 / 4 waves, tools/stereo_chain_time.py)

struct StereoDev {
  const KeyPointDev* kpsL; const uint8_t* descL; const int* nL;
  const KeyPointDev* kpsR; const uint8_t* descR; const int* nR;
  const uint8_t* pyrL[8]; const uint8_t* pyrR[8];   // ROI origin of each level in frame 0 of the extractor's last call
  long long frameStrideL, frameStrideR;
  int capL, capR, frameL0, frameLStep, frameR0, frameRStep;
  int w[8], h[8], pitchL[8], pitchR[8];
  float scale[8], invScale[8];
  int nLevels, flags;
  float bf, maxD;
  float* rightX; float* depth;
  int* counters;   // per pair: [0] measurements kept, [1] of those with block-match minimum 0, [2] status bits, [3] unused
  int* keptOut; int* statusOut;
  int rowLists;   // replay kernel: 1 = LDS holds the first-row-sorted index of the right keypoints (0 = scan all right keypoints per step)
};

struct StereoAcc { int kept, zeros, status; };
struct StereoRes { bool complete, kept, zero; int status; float rx, depth; };

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One left keypoint (kx, ky, octave o) matched with the left descriptor row (a0, a1).  Wave-uniform control flow; `complete`
// says whether the reference's loop body reaches `leftIdx++` (:462).
__device__ StereoRes stereo_one(const StereoDev& P, int pair, float kx, float ky, int o, uint4 a0, uint4 a1, int nR,
                                const float* __restrict__ rx, const unsigned* __restrict__ rinfo, unsigned short* cand, int lane,
                                const int* __restrict__ rowStart = nullptr, const unsigned short* __restrict__ rowEnt = nullptr, int bandMax = 0) {
  StereoRes Rz{false, false, false, 0, -1.0f, -1.0f};
  const int row = (int)ky;
  if (!(ky >= 0.0f) || row >= P.h[0]) { Rz.status = 1; return Rz; }   // out-of-range row index at :389 (undefined in the reference)
  const uint8_t* dr = P.descR + (size_t)pair * P.capR * 32;
  const float xlo = kx - P.maxD, xhi = kx;   // :395 (minD = 0)
  unsigned best = 0xFFFFFFFFu;
  bool any = false;
  // Two phases so that the descriptor rows of all candidates are in flight together: the scan (LDS only) compacts the indices
  // that pass the static tests into the wave's list, then lane i takes candidate i.  A full list (64) is drained in between.
  const unsigned long long below = (1ull << lane) - 1ull;
  int cnt = 0;
  auto drain = [&](int n) {
    if (lane < n) {
      const int j = cand[lane];
      const uint4 b0 = *reinterpret_cast<const uint4*>(dr + (size_t)j * 32), b1 = *reinterpret_cast<const uint4*>(dr + (size_t)j * 32 + 16);
      const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                    __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      best = min(best, ((unsigned)d << 16) | (unsigned)j);
    }
  };
  if (rowStart) {
    // rowEnt = the right keypoints sorted by the first row of their band, rowStart[r] = first entry whose band starts at row >= r: the
    // candidates of this row start in [row - bandMax, row] (bandMax = widest band - 1 of this pair).  Two entries per lane and trip, so
    // that the usual ~90 entries are one trip with all LDS reads - and then all descriptor rows - in flight together; entry order does not
    // matter because the winner is the minimum of (distance << 16 | index).
    const int e0 = rowStart[max(row - bandMax, 0)], e1 = rowStart[row + 1];
    for (int e = e0 + lane; e < e1; e += 128) {
      const bool h1 = e + 64 < e1;
      const int ja = rowEnt[e], jb = h1 ? rowEnt[e + 64] : ja;
      const unsigned ia = rinfo[ja], ib = rinfo[jb];
      const float xa = rx[ja], xb = rx[jb];
      const int oa = (int)(ia >> 24), ob = (int)(ib >> 24);
      const bool ra = row <= (int)((ia >> 12) & 0xFFF), rb = h1 && row <= (int)((ib >> 12) & 0xFFF);   // first row <= row holds for every entry
      any |= ra | rb;
      const bool pa = ra && oa >= o - 1 && oa <= o + 1 && xa >= xlo && xa <= xhi, pb = rb && ob >= o - 1 && ob <= o + 1 && xb >= xlo && xb <= xhi;
      uint4 b0 = {}, b1 = {}, c0 = {}, c1 = {};
      if (pa) { b0 = *reinterpret_cast<const uint4*>(dr + (size_t)ja * 32); b1 = *reinterpret_cast<const uint4*>(dr + (size_t)ja * 32 + 16); }
      if (pb) { c0 = *reinterpret_cast<const uint4*>(dr + (size_t)jb * 32); c1 = *reinterpret_cast<const uint4*>(dr + (size_t)jb * 32 + 16); }
      if (pa) {
        const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                      __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
        best = min(best, ((unsigned)d << 16) | (unsigned)ja);
      }
      if (pb) {
        const int d = __popc(a0.x ^ c0.x) + __popc(a0.y ^ c0.y) + __popc(a0.z ^ c0.z) + __popc(a0.w ^ c0.w) + __popc(a1.x ^ c1.x) +
                      __popc(a1.y ^ c1.y) + __popc(a1.z ^ c1.z) + __popc(a1.w ^ c1.w);
        best = min(best, ((unsigned)d << 16) | (unsigned)jb);
      }
    }
  } else {
    for (int j0 = 0; j0 < nR; j0 += 256) {   // four table rows per lane and trip: their LDS reads are in flight together
      unsigned inf[4];
      float x[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int j = min(j0 + 64 * u + lane, nR - 1);
        inf[u] = rinfo[j];
        x[u] = rx[j];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int j = j0 + 64 * u + lane;
        const int lo = inf[u] & 0xFFF, hi = (inf[u] >> 12) & 0xFFF, oc = (int)(inf[u] >> 24);
        const bool inRow = j < nR && row >= lo && row <= hi;
        any |= inRow;
        const bool pass = inRow && oc >= o - 1 && oc <= o + 1 && x[u] >= xlo && x[u] <= xhi;
        const unsigned long long m = __ballot(pass);
        if (m) {
          if (pass) cand[cnt + __popcll(m & below)] = (unsigned short)j;
          cnt += __popcll(m);
          __builtin_amdgcn_wave_barrier();
          if (cnt >= 64) {
            drain(64);
            const unsigned short keep = cand[64 + lane];
            __builtin_amdgcn_wave_barrier();
            cand[lane] = keep;
            __builtin_amdgcn_wave_barrier();
            cnt -= 64;
          }
        }
      }
    }
    drain(cnt);
  }
  if (!__any(any) || !(kx >= 0.0f)) return Rz;   // :389
  best = wave_min_u32(best);
  const int bestDist = best == 0xFFFFFFFFu ? 256 : (int)(best >> 16);
  if (bestDist < kStereoOrbDist) {   // :406
    const int bestRight = (int)(best & 0xFFFFu);
    const float inv = P.invScale[o];
    const int lx = (int)roundf(kx * inv), ly = (int)roundf(ky * inv), rsx = (int)roundf(rx[bestRight] * inv);   // :408-410
    const int W = P.w[o], H = P.h[o];
    if (ly - 5 < 0 || ly + 6 >= H || lx - 5 < 0 || lx + 6 >= W) return Rz;   // :414-416
    if (rsx < 0 || rsx + 11 >= W) return Rz;                                  // :424-426
    if (rsx - 10 < 0) { Rz.status = 2; return Rz; }                           // negative colRange at :427 (cv::Exception in the reference)
    const int pL = P.pitchL[o], pR = P.pitchR[o];
    const uint8_t* Lp = P.pyrL[o] + (long long)(P.frameL0 + pair * P.frameLStep) * P.frameStrideL + (long long)ly * pL + lx;
    const uint8_t* Rp = P.pyrR[o] + (long long)(P.frameR0 + pair * P.frameRStep) * P.frameStrideR + (long long)ly * pR + rsx;
    int acc[11];
#pragma unroll
    for (int i = 0; i < 11; i++) acc[i] = 0;
    const int lc = Lp[0];
    for (int t = lane; t < 121; t += 64) {   // 11x11 window, centre-subtracted L1 distance for the 11 shifts (:416-434)
      const int dy = t / 11 - 5, dx = t % 11 - 5;
      const int a = (int)Lp[dy * pL + dx] - lc;
      const uint8_t* rr = Rp + dy * pR + dx;
#pragma unroll
      for (int i = 0; i < 11; i++) {
        const int b = (int)rr[i - 5] - (int)Rp[i - 5];
        acc[i] += abs(a - b);
      }
    }
    int sadBest = 256, bestCol = 0;   // :419-420: an int minimum that starts at 256
#pragma unroll
    for (int i = 0; i < 11; i++) {
      acc[i] = wave_sum_i32(acc[i]);
      if (acc[i] < sadBest) { sadBest = acc[i]; bestCol = i - 5; }
    }
    if (bestCol == -5 || bestCol == 5) return Rz;   // :437-439
    float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
    for (int i = 0; i < 11; i++) {
      if (i == bestCol + 4) d1 = (float)acc[i];
      if (i == bestCol + 5) d2 = (float)acc[i];
      if (i == bestCol + 6) d3 = (float)acc[i];
    }
    const float delta = (float)((double)(d1 - d3) / (2.0 * ((double)(d1 + d3) - 2.0 * (double)d2)));   // :441-443
    if (delta < -1 || delta > 1) return Rz;                                                           // :444-446
    float bestRightX = __fmul_rn(P.scale[o], __fadd_rn(__fadd_rn((float)rsx, delta), (float)bestCol));  // :448
    float disparity = __fsub_rn(kx, bestRightX);
    if (disparity >= 0.0f && disparity < P.maxD) {
      if (disparity <= 0) {
        disparity = 0.01f;
        bestRightX = (float)((double)kx - 0.01);
      }
      Rz.kept = true;
      Rz.zero = sadBest == 0;
      Rz.depth = __fdiv_rn(P.bf, disparity);   // :455-457
      Rz.rx = bestRightX;
    }
  }
  Rz.complete = true;
  return Rz;
}

// row band of each right keypoint (:373-379) as an LDS record; all threads of the workgroup, ends with a barrier
__device__ __forceinline__ void stereo_right_table(const StereoDev& P, int pair, int nR, float* rx, unsigned* rinfo) {
  const float lastRow = (float)P.h[0] - 1.0f;
  for (int j = threadIdx.x; j < nR; j += blockDim.x) {
    const KeyPointDev kp = P.kpsR[(size_t)pair * P.capR + j];
    const float r = 2.0f * P.scale[kp.octave & 7];
    int lo = (int)fmaxf(floorf(kp.y - r), 0.0f);
    int hi = (int)fminf(ceilf(kp.y + r), lastRow);
    if (lo > hi || hi < 0) { lo = 4095; hi = 0; }
    rx[j] = kp.x;
    rinfo[j] = (unsigned)lo | ((unsigned)hi << 12) | ((unsigned)kp.octave << 24);
  }
  __syncthreads();
}

// grid (REPLAY ? 1 : ceil(capL / kStereoChunk), pairs), 256 threads, dynamic LDS = capR * 8 bytes.  Outputs were set to -1
// (:363-364) by the host's fill.
// Replay form in slices: launch [kBegin, kEnd) walks that range of left keypoints and hands the lagging index to the next launch
// through counters[pair * 4 + 3].  One launch over a whole 2000-keypoint frame is a 3.8 ms single-wave kernel, and everything that
// shares its hardware queue (the device has 4; streams are mapped onto them round-robin) waits behind it; slices of a few hundred
// keypoints let the other streams' launches in between.  (The right-keypoint table and the row lists are rebuilt per slice: ~20 us.)
template <bool REPLAY>
__global__ __launch_bounds__(256) void k_stereo(const StereoDev P, int kBegin, int kEnd) {   // P by value: a kernel argument, no parameter copy for the stream to wait on
  extern __shared__ unsigned char stereoLds[];
  __shared__ unsigned short candList[4][128];   // per wave: indices of the right keypoints that passed the static tests
  const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nL = min(P.nL[pair], P.capL), nR = min(P.nR[pair], P.capR);
  float* rx = reinterpret_cast<float*>(stereoLds);
  unsigned* rinfo = reinterpret_cast<unsigned*>(rx + P.capR);
  const int k0 = REPLAY ? 0 : blockIdx.x * kStereoChunk, k1 = REPLAY ? P.capL : min(k0 + kStereoChunk, P.capL);
  if (!REPLAY && k0 >= nL) return;
  __shared__ int replayTag[8], replayS[8];   // replay form, several waves: the exact index after each step (see below)
  if (REPLAY && tid < 8) { replayTag[tid] = 0; replayS[tid] = 0; }   // (the table's barrier orders this before the walk)
  stereo_right_table(P, pair, nR, rx, rinfo);
  // Replay form: the right keypoints sorted by the first row of their band (counting sort over the rows, order inside a row arbitrary) -
  // a step of the serial walk then reads the ~90 entries whose band can cover its row instead of scanning every right keypoint.
  int* rowStart = nullptr;
  unsigned short* rowEnt = nullptr;
  int bandMax = 0;
  if (REPLAY && P.rowLists > 0) {
    const int rows = P.h[0];
    int* rs = reinterpret_cast<int*>(rinfo + P.capR);          // [rows + 1]
    int* cur = rs + rows + 1;                                   // [rows] counts, then fill cursors
    unsigned short* ent = reinterpret_cast<unsigned short*>(cur + rows);   // [nR]
    __shared__ int rlWave[4];
    __shared__ int rlBand;
    for (int r = tid; r < rows; r += 256) cur[r] = 0;
    if (tid == 0) rlBand = 0;
    __syncthreads();
    int wmax = 0;
    for (int j = tid; j < nR; j += 256) {
      const unsigned inf = rinfo[j];
      const int lo = inf & 0xFFF, hi = (inf >> 12) & 0xFFF;
      if (lo <= hi) { atomicAdd(&cur[lo], 1); wmax = max(wmax, hi - lo); }   // (an empty band, lo 4095 > hi 0, is in no row's list)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
    if (lane == 0) atomicMax(&rlBand, wmax);
    __syncthreads();
    {  // exclusive scan of the row counts: contiguous chunk per thread + block scan of the chunk sums
      const int per = (rows + 255) / 256, r0 = tid * per, r1 = min(r0 + per, rows);
      int sum = 0;
      for (int r = r0; r < r1; r++) sum += cur[r];
      int x = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
      if (lane == 63) rlWave[wave] = x;
      __syncthreads();
      int off = x - sum;
      for (int w = 0; w < wave; w++) off += rlWave[w];
      for (int r = r0; r < r1; r++) { rs[r] = off; off += cur[r]; }
      if (tid == 255) rs[rows] = off;
      __syncthreads();
    }
    for (int r = tid; r < rows; r += 256) cur[r] = 0;
    __syncthreads();
    for (int j = tid; j < nR; j += 256) {
      const unsigned inf = rinfo[j];
      const int lo = inf & 0xFFF, hi = (inf >> 12) & 0xFFF;
      if (lo <= hi) ent[rs[lo] + atomicAdd(&cur[lo], 1)] = (unsigned short)j;
    }
    __syncthreads();
    rowStart = rs;
    rowEnt = ent;
    bandMax = rlBand;
  }
  StereoAcc A{0, 0, 0};
  const KeyPointDev* kl = P.kpsL + (size_t)pair * P.capL;
  const uint4* dl = reinterpret_cast<const uint4*>(P.descL + (size_t)pair * P.capL * 32);
  float* outRx = P.rightX + (size_t)pair * P.capL;
  float* outDepth = P.depth + (size_t)pair * P.capL;
  if (REPLAY) {
    if (wave >= kStereoReplayWaves || nL == 0 || kBegin >= nL) return;
    // waves whose latency is the call's latency: highest issue priority among the waves of their SIMD (the other streams'
    // throughput kernels fill the CU and would otherwise get 7 of 8 issue slots)
#ifndef YDORB_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const int kStop = min(kEnd, nL);
    const int sStart = kBegin > 0 ? P.counters[pair * 4 + 3] : 0;
    if (kStereoReplayWaves == 1) {
      int s = sStart;
      // one step ahead: keypoint k + 1 and descriptor row s + 1 (the row the next step needs is s or s + 1)
      float kx = kl[kBegin].x, ky = kl[kBegin].y;
      int o = kl[kBegin].octave;
      uint4 a0 = dl[2 * s], a1 = dl[2 * s + 1];
      for (int k = kBegin; k < kStop; k++) {
        const int kn = min(k + 1, nL - 1), sn = min(s + 1, nL - 1);
        const float nkx = kl[kn].x, nky = kl[kn].y;
        const int no = kl[kn].octave;
        const uint4 n0 = dl[2 * sn], n1 = dl[2 * sn + 1];
        const StereoRes r = stereo_one(P, pair, kx, ky, o, a0, a1, nR, rx, rinfo, candList[wave], lane, rowStart, rowEnt, bandMax);
        if (r.kept && lane == 0) { outRx[s] = r.rx; outDepth[s] = r.depth; }
        A.kept += r.kept; A.zeros += r.zero; A.status |= r.status;
        if (r.complete) { s++; a0 = n0; a1 = n1; }
        kx = nkx; ky = nky; o = no;
      }
      if (lane == 0 && kStop < nL) P.counters[pair * 4 + 3] = s;
    } else {
      // NW waves take the keypoints in turn (wave w: k = kBegin + w, + NW, ...).  The index s_k a step needs is s_{k-1} + c of the previous
      // step - which is still running on another wave - so the step is evaluated with the guess "every keypoint since the last index this
      // wave knows exactly reaches leftIdx++" (true for ~97 % of the keypoints), checked against the exact s_k the previous step's wave
      // publishes through LDS when it commits, and evaluated again in the rare other case.  Commits stay in keypoint order (a step commits
      // only after its predecessor published), so outputs, counters and the hand-over to the next slice are those of the serial walk; NW
      // steps are in flight instead of one.  replayTag[k & 7] == k + 1 says replayS[k & 7] holds s_{k+1}; commits are in order, so the slot's
      // next writer (step k + 8) cannot get there before its reader (step k + 1) has committed.  Every step of [kBegin, kStop) is executed
      // and published, and a wave without steps just leaves: no wave waits for something that never comes.
      int k = kBegin + wave;
      int sGuess = min(sStart + wave, nL - 1);
      float kx = 0.f, ky = 0.f;
      int o = 0;
      uint4 a0{}, a1{};
      if (k < kStop) { kx = kl[k].x; ky = kl[k].y; o = kl[k].octave; a0 = dl[2 * sGuess]; a1 = dl[2 * sGuess + 1]; }
      int sLast = sStart;
      constexpr int NW = kStereoReplayWaves;
      for (; k < kStop; k += NW) {
        // ahead of their use: this wave's next keypoint and the two descriptor rows its guess can be (s_k + NW - 1 or s_k + NW when the guess holds)
        const int kn = min(k + NW, nL - 1), sn = min(sGuess + NW - 1, nL - 1), sm = min(sGuess + NW, nL - 1);
        const float nkx = kl[kn].x, nky = kl[kn].y;
        const int no = kl[kn].octave;
        const uint4 n0 = dl[2 * sn], n1 = dl[2 * sn + 1], m0 = dl[2 * sm], m1 = dl[2 * sm + 1];
        StereoRes r = stereo_one(P, pair, kx, ky, o, a0, a1, nR, rx, rinfo, candList[wave], lane, rowStart, rowEnt, bandMax);
        int sk = sStart;
        if (k > kBegin) {
          const int slot = (k - 1) & 7;
          while (__hip_atomic_load(&replayTag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != k) __builtin_amdgcn_s_sleep(1);
          sk = __hip_atomic_load(&replayS[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool redo = sk != sGuess;
        if (redo) {
          a0 = dl[2 * sk]; a1 = dl[2 * sk + 1];
          r = stereo_one(P, pair, kx, ky, o, a0, a1, nR, rx, rinfo, candList[wave], lane, rowStart, rowEnt, bandMax);
        }
        if (r.kept && lane == 0) { outRx[sk] = r.rx; outDepth[sk] = r.depth; }
        A.kept += r.kept; A.zeros += r.zero; A.status |= r.status;
        sLast = sk + (r.complete ? 1 : 0);                                   // s_{k+1}
        if (lane == 0) {
          __hip_atomic_store(&replayS[k & 7], sLast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&replayTag[k & 7], k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const int gNext = min(sLast + NW - 1, nL - 1);                         // this wave's next step guesses that the steps in between complete
        if (redo) { a0 = dl[2 * gNext]; a1 = dl[2 * gNext + 1]; }
        else if (r.complete) { a0 = m0; a1 = m1; }
        else { a0 = n0; a1 = n1; }
        sGuess = gNext;
        kx = nkx; ky = nky; o = no;
      }
      // the wave that ran the slice's last step hands s to the next slice
      if (lane == 0 && kStop < nL && ((kStop - 1 - kBegin) % NW) == wave) P.counters[pair * 4 + 3] = sLast;
    }
  } else {
    for (int k = k0 + wave; k < min(k1, nL); k += 4) {
      const StereoRes r = stereo_one(P, pair, kl[k].x, kl[k].y, kl[k].octave, dl[2 * k], dl[2 * k + 1], nR, rx, rinfo, candList[wave], lane);
      if (r.kept && lane == 0) { outRx[k] = r.rx; outDepth[k] = r.depth; }
      A.kept += r.kept; A.zeros += r.zero; A.status |= r.status;
    }
  }
  if (lane == 0) {
    if (A.kept) atomicAdd(&P.counters[pair * 4 + 0], A.kept);
    if (A.zeros) atomicAdd(&P.counters[pair * 4 + 1], A.zeros);
    if (A.status) atomicOr(&P.counters[pair * 4 + 2], A.status);
  }
}

// :464-472.  The sorted list is walked from its smallest entry and left at the first one below 2.1 x the median, so the
// loop removes every measurement when the median block-match minimum is 0 and none otherwise.
__global__ __launch_bounds__(256) void k_stereo_outliers(const StereoDev P) {
  const int pair = blockIdx.x;
  const int kept = P.counters[pair * 4 + 0], zeros = P.counters[pair * 4 + 1];
  if (threadIdx.x == 0) {
    P.keptOut[pair] = kept;
    P.statusOut[pair] = P.counters[pair * 4 + 2];
  }
  if (kept > 0 && zeros >= kept / 2 + 1) {
    for (int i = threadIdx.x; i < P.capL; i += 256) {
      const size_t at = (size_t)pair * P.capL + i;
      if (P.depth[at] > 0.0f) { P.depth[at] = -2.0f; P.rightX[at] = -2.0f; }
    }
  }
}

}  // namespace ydorb
