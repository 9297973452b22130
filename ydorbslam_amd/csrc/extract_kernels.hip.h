// gfx950 HIP kernels of the ORB front-end (pyramid, per-cell FAST-9/16 + NMS + ordered compaction,
// quad-tree thinning, 7x7 Gaussian, intensity-centroid orientation, steered rBRIEF).
//
// Behaviour follows reference src/orbExtractor.cpp (cited per kernel); the structure does not: frames are
// batched (grid.y/z = frame), every stage is one launch over all frames x levels, all intermediate
// products stay in HBM/L2, and nothing returns to the host between the image upload and the
// keypoint/descriptor download.  All arithmetic that decides a result bit is integer or single IEEE
// operations (this TU is compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "quadtree_core.h"
#include "quadtree_flat.h"

#pragma clang fp contract(off)

namespace ydorb {

constexpr int kPad = 19;          // m_int_maxPadSize, orbExtractor.hpp:71
constexpr int kBorder = 16;       // m_int_maxPadSize - 3, orbExtractor.cpp:549
constexpr int kMaxLevels = 8;
constexpr int kTileMax = 72;      // FAST cell sub-image is at most (59+1+6) px wide
#ifndef QT_THREADS
#define QT_THREADS 512
#endif
constexpr int kQtThreads = QT_THREADS;     // threads per (frame, level) unit

struct LevelDev {
  int w, h, pitch;          // interior size, padded-row pitch (bytes)
  int padOff;               // byte offset of the padded buffer inside a frame's pyramid block
  int blurOff, blurPitch;   // unpadded blurred level inside a frame's blur block
  int cellBegin, nCells;    // this level's cells in the cell table
  int quota;                // m_v_keyPointsNumsPerLevel[level]
  int candOff;              // entry offset of the level's quad-tree scratch inside a frame's scratch
  int kpOff;                // entry offset of the level's keypoints inside a frame's level-keypoint array
  int tabOff;               // offset of the resize tables (level >= 1)
  int nodeTabOff;           // >= 0: byte offset of this level's node tables inside a frame's HBM node scratch (quota too large for LDS), else -1
  float scale;              // m_v_scaleFactors[level]
  float size;               // (float)(int)(31 * scale), orbExtractor.cpp:595
};
struct PlanDev {
  int nLevels, nCellsTotal, cellCap, sumQuota;
  int maxX[16];             // m_v_maxXcords[0..15]
  unsigned nodeTabFrameStride;         // bytes of HBM node scratch per frame (0 when every level's node table fits the LDS)
  int blurTileBegin[kMaxLevels + 1];   // first 64x32 blur tile of each level in the flat per-frame tile list
  int borderBegin[kMaxLevels + 1];     // first border thread of each level in k_pyr_borders' flat per-frame list (multiples of 256)
  LevelDev lv[kMaxLevels];
};
#ifndef BLUR_TH
#define BLUR_TH 58
#endif
// Output tile of k_blur: 64 x BLUR_TH (any height with BLUR_TH + 6 even).  58 + 6 = 64 source rows = 32 row pairs x 16 dword groups =
// exactly two full passes of the workgroup in the horizontal stage, and a 10 % halo instead of 19 % with 32-row tiles: 0.516 -> 0.457 ms
// per 512 frames alone, resident-frames pipeline 212.8 -> 216.2 Mkeypoints/s (26 rows: 0.577 ms, 210.6).  (Round 2 kept 32: the blur then
// had to outlast the eight quad-tree launches it was hiding; k_qt_fast ended that.)
constexpr int kBlurTW = 64, kBlurTH = BLUR_TH;
struct CellDev {            // FAST sub-image [x0,x1) x [y0,y1) in level coordinates (orbExtractor.cpp:562-581)
  short level, x0, y0, x1, y1, pad0;
  int srcOff;               // byte offset of the sub-image's first pixel inside a frame's pyramid block (host-computed)
};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// ------------------------------------------------------------------------------------------------
// Pyramid level 0: copyMakeBorder(image, 19 px, BORDER_REFLECT_101)  (orbExtractor.cpp:618).
// One thread writes 4 consecutive bytes of the padded buffer.
// ------------------------------------------------------------------------------------------------
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef unsigned long long __attribute__((aligned(1))) u64_unaligned;

// Workgroup = 64 x 4 threads; a thread writes one dword (4 px) in each of kPyrRows consecutive rows, so a workgroup covers
// 256 B x 16 rows and the per-column tables are loaded once per thread (one-row workgroups were dispatch-bound).
#ifndef PYR_ROWS
#define PYR_ROWS 8
#endif
constexpr int kPyrRows = PYR_ROWS;
// The level kernels write INTERIOR rows only, one dword per thread and row, every lane on the same (fast) path: a dword that
// straddles the left/right edge is computed from clamped columns and its pad bytes are rewritten by k_pyr_borders afterwards.
// (Computing the reflect-101 pad inside these kernels sent two thirds of the waves through a byte-gather slow path for the sake
// of a few lanes.)  All loads of a thread's kPyrRows rows are issued before the first use.
__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ img, int stride, size_t frameStride,
                                                    uint8_t* __restrict__ pyr, size_t pyrFrameStride, LevelDev L) {
  const int wx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int yb = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kPyrRows;   // interior row
  const int x0 = wx * 4 - kPad;                              // level column of this dword's first byte
  if (x0 + 3 < 0 || x0 >= L.w) return;                       // pure pad dword: k_pyr_borders
  const uint8_t* src = img + (size_t)blockIdx.z * frameStride;
  uint8_t* dst = pyr + (size_t)blockIdx.z * pyrFrameStride + L.padOff + (size_t)kPad * L.pitch + wx * 4;
  // 4 source bytes from a clamped address; an edge dword is shifted so that its interior bytes land where they belong
  const int xa = L.w >= 4 ? min(max(x0, 0), L.w - 4) : 0, sh = 8 * (x0 - xa);
  uint32_t v[kPyrRows];
#pragma unroll
  for (int r = 0; r < kPyrRows; r++) {
    const uint8_t* row = src + (size_t)min(yb + r, L.h - 1) * stride;
    if (L.w >= 4) {
      v[r] = *reinterpret_cast<const u32_unaligned*>(row + xa);
    } else {
      v[r] = 0;
      for (int b = 0; b < L.w; b++) v[r] |= (uint32_t)row[b] << (8 * b);
    }
    v[r] = sh < 0 ? v[r] << (-sh) : v[r] >> sh;
  }
#pragma unroll
  for (int r = 0; r < kPyrRows; r++)
    if (yb + r < L.h) *reinterpret_cast<uint32_t*>(dst + (size_t)(yb + r) * L.pitch) = v[r];
}

// ------------------------------------------------------------------------------------------------
// Pyramid level l>=1: cv::resize(level l-1, INTER_LINEAR) (orbExtractor.cpp:614-615), interior only.
// Coefficient tables (11-bit fixed point) are built on the host.  The 4 outputs of a thread read source columns
// sx[0] .. sx[3]+1, at most 8 consecutive bytes for scale factors <= 2, fetched as one unaligned 8-byte load per source row.
// The row tables are indexed by a wave-uniform row number (scalar loads); the source column is computed in the kernel, only
// the weights come from the column table.  A tap that falls outside the source interior has weight 0, so the (not yet
// written) pad bytes of the source level never reach a result.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_resize(uint8_t* __restrict__ pyr, size_t pyrFrameStride, LevelDev Lp, LevelDev L,
                                                    double scaleX, const short* __restrict__ alpha,
                                                    const int* __restrict__ yofs, const short* __restrict__ beta) {
  const int wx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int yb = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kPyrRows;   // interior row
  const int x0 = wx * 4 - kPad;
  if (x0 + 3 < 0 || x0 >= L.w) return;                       // pure pad dword: k_pyr_borders
  uint8_t* frame = pyr + (size_t)blockIdx.z * pyrFrameStride;
  const uint8_t* S = frame + Lp.padOff + (size_t)kPad * Lp.pitch + kPad;  // ROI origin of the source level
  uint8_t* dst = frame + L.padOff + (size_t)kPad * L.pitch + wx * 4;
  int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int dx = min(max(x0 + b, 0), L.w - 1);             // edge dwords: pad bytes repeat the edge column (rewritten later)
    // source column computed here (the same IEEE sequence as the host's coefficient table): the pixel loads then wait for
    // no table load, only the weights do
    const float fx = (float)__dsub_rn(__dmul_rn((double)dx + 0.5, scaleX), 0.5);
    sx[b] = min(max((int)floorf(fx), 0), Lp.w - 1);
    a0[b] = alpha[2 * dx]; a1[b] = alpha[2 * dx + 1];
    sx1[b] = min(sx[b] + 1, Lp.w - 1);   // a1 == 0 whenever sx+1 is outside
  }
  const bool fast = sx[3] - sx[0] <= 6;   // always, for scale factors <= 2
  const uint8_t *r0p[kPyrRows], *r1p[kPyrRows];
  int b0[kPyrRows], b1[kPyrRows];
#pragma unroll
  for (int r = 0; r < kPyrRows; r++) {
    const int dy = min(yb + r, L.h - 1);
    const int sy = yofs[dy];
    b0[r] = beta[2 * dy]; b1[r] = beta[2 * dy + 1];
    r0p[r] = S + (size_t)min(max(sy, 0), Lp.h - 1) * Lp.pitch;
    r1p[r] = S + (size_t)min(max(sy + 1, 0), Lp.h - 1) * Lp.pitch;
  }
  uint32_t v[kPyrRows];
  if (fast) {
    // Both taps of an output as one u16 pair picked out of the 8 loaded bytes (v_perm_b32, selector fixed per column) and
    // weighted by (alpha0, alpha1) in one v_dot2_u32_u16.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    uint32_t sel[4];
    u16x2 ab[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t off = (uint32_t)(sx[b] - sx[0]);
      sel[b] = off | 0x0c000c00u | ((off + 1) << 16);        // byte off -> bits 0..7, byte off+1 -> bits 16..23, zeros between
      ab[b] = u16x2{(unsigned short)a0[b], (unsigned short)a1[b]};
    }
    unsigned long long w0[kPyrRows], w1[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
      w0[r] = *reinterpret_cast<const u64_unaligned*>(r0p[r] + sx[0]);
      w1[r] = *reinterpret_cast<const u64_unaligned*>(r1p[r] + sx[0]);
    }
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
      v[r] = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)(w0[r] >> 32), (uint32_t)w0[r], sel[b]);
        const uint32_t p1 = __builtin_amdgcn_perm((uint32_t)(w1[r] >> 32), (uint32_t)w1[r], sel[b]);
        const int h0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p0), ab[b], 0u, false);
        const int h1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p1), ab[b], 0u, false);
        const int o = (((b0[r] * (h0 >> 4)) >> 16) + ((b1[r] * (h1 >> 4)) >> 16) + 2) >> 2;
        v[r] |= (uint32_t)(o & 0xFF) << (8 * b);
      }
    }
  } else {   // scale factor > 2: byte taps
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
      v[r] = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int h0 = r0p[r][sx[b]] * a0[b] + r0p[r][sx1[b]] * a1[b];
        const int h1 = r1p[r][sx[b]] * a0[b] + r1p[r][sx1[b]] * a1[b];
        const int o = (((b0[r] * (h0 >> 4)) >> 16) + ((b1[r] * (h1 >> 4)) >> 16) + 2) >> 2;
        v[r] |= (uint32_t)(o & 0xFF) << (8 * b);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kPyrRows; r++)
    if (yb + r < L.h) *reinterpret_cast<uint32_t*>(dst + (size_t)(yb + r) * L.pitch) = v[r];
}

// ------------------------------------------------------------------------------------------------
// Level kernels that also write the level's reflect-101 pad (orbExtractor.cpp:612-621 in the same pass as :614-618): no border
// launch for such a level.  A thread owns dword `wx` of the padded row in kPyrRowsF consecutive interior rows.
//   left / right pad : a pad byte p of a row is the row's own byte 38 - p (left) or 2w + 36 - p (right); the at most two dwords that
//                      hold the mirrored bytes of a pad dword live in the same wave (the host shifts the lane -> dword mapping so that
//                      neither end group straddles a 64-dword boundary), so they arrive by ds_bpermute and two v_perm_b32 whose
//                      selectors come from a per-column table (PyrPadEntry) - executed only by waves that hold pad dwords;
//   top / bottom pad : the thread that writes interior row y in 1..19 stores the same finished dword to padded row 19 - y, and the
//                      one that writes row y in h-20..h-2 to row 2(h-1) - y + 19.
// Every byte is still written exactly once, and the source level's own pads are complete when the next level reads it.
// Per-column constants (selectors, weights) and per-row constants (source row, weights) are host tables, padded so that no index
// needs a clamp: the row entries of a wave are one scalar load, a column's entries two or three 16-byte loads.
// Levels the scheme does not cover (w or h < 20, scale factor > 2) keep k_pyr_level0 / k_pyr_resize + k_pyr_borders.
// ------------------------------------------------------------------------------------------------
struct PyrPadEntry { uint32_t laneA4, laneB4, selM, selF; };      // 4 * source lane of the two mirrored dwords, perm(A, B, selM), perm(M, own, selF)
struct PyrColEntry { uint32_t sel[4]; uint32_t ab[4]; };          // v_perm selector of the two taps (relative to the column's first tap), alpha0 | alpha1 << 16
constexpr uint32_t kPermIdentity = 0x03020100u;
#ifndef PYR_ROWS_F
#define PYR_ROWS_F 8
#endif
constexpr int kPyrRowsF = PYR_ROWS_F;

__device__ __forceinline__ uint32_t pyr_finish_dword(uint32_t v, const PyrPadEntry& pe) {
  const uint32_t A = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pe.laneA4, (int)v);
  const uint32_t B = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pe.laneB4, (int)v);
  const uint32_t M = __builtin_amdgcn_perm(A, B, pe.selM);
  return __builtin_amdgcn_perm(M, v, pe.selF);
}
// stores of one finished dword: its own row and, near the top / bottom edge, the mirrored pad row (row offsets relative to interior row 0)
__device__ __forceinline__ void pyr_store_rows(uint8_t* dst, int pitch, int y, int h, uint32_t v) {
  if (y >= h) return;
  *reinterpret_cast<uint32_t*>(dst + (ptrdiff_t)y * pitch) = v;
  if (y >= 1 && y <= kPad) *reinterpret_cast<uint32_t*>(dst - (ptrdiff_t)y * pitch) = v;
  if (y >= h - 1 - kPad && y <= h - 2) *reinterpret_cast<uint32_t*>(dst + (ptrdiff_t)(2 * (h - 1) - y) * pitch) = v;
}

__global__ __launch_bounds__(256) void k_pyr_level0_f(const uint8_t* __restrict__ img, int stride, size_t frameStride,
                                                      uint8_t* __restrict__ pyr, size_t pyrFrameStride, LevelDev L,
                                                      const PyrPadEntry* __restrict__ padTab, int shift) {
  const int wx = blockIdx.x * 64 + (threadIdx.x & 63) - shift;
  const int yb = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kPyrRowsF;
  const int nd = (L.w + 2 * kPad + 3) >> 2;                  // dwords with at least one byte below w + 38
  if (wx < 0 || wx >= nd) return;
  const int x0 = wx * 4 - kPad;
  const uint8_t* src = img + (size_t)blockIdx.z * frameStride;
  uint8_t* dst = pyr + (size_t)blockIdx.z * pyrFrameStride + L.padOff + (size_t)kPad * L.pitch + wx * 4;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 peRaw = reinterpret_cast<const u32x4*>(padTab)[wx];   // one 16-byte load, issued with the pixel loads
  const PyrPadEntry pe{peRaw.x, peRaw.y, peRaw.z, peRaw.w};
  const int xa = min(max(x0, 0), L.w - 4), sh = 8 * (x0 - xa);   // w >= 20 here
  uint32_t v[kPyrRowsF];
#pragma unroll
  for (int r = 0; r < kPyrRowsF; r++) {
    const uint8_t* row = src + (size_t)min(yb + r, L.h - 1) * stride;
    v[r] = *reinterpret_cast<const u32_unaligned*>(row + xa);
  }
  const int shl = min(max(-sh, 0), 31), shr = min(max(sh, 0), 31);   // |sh| <= 24 for dwords that keep an interior byte; others are all pad
#pragma unroll
  for (int r = 0; r < kPyrRowsF; r++) v[r] = sh < 0 ? v[r] << shl : v[r] >> shr;
  if (__builtin_amdgcn_ballot_w64(pe.selF != kPermIdentity)) {
#pragma unroll
    for (int r = 0; r < kPyrRowsF; r++) v[r] = pyr_finish_dword(v[r], pe);
  }
#pragma unroll
  for (int r = 0; r < kPyrRowsF; r++) pyr_store_rows(dst, L.pitch, yb + r, L.h, v[r]);
}

__global__ __launch_bounds__(256) void k_pyr_resize_f(uint8_t* __restrict__ pyr, size_t pyrFrameStride, LevelDev Lp, LevelDev L,
                                                      double scaleX, const PyrColEntry* __restrict__ colTab,
                                                      const PyrPadEntry* __restrict__ padTab, const int2* __restrict__ rowTab, int shift) {
  const int wx = blockIdx.x * 64 + (threadIdx.x & 63) - shift;
  const int yb = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kPyrRowsF;
  const int nd = (L.w + 2 * kPad + 3) >> 2;
  if (wx < 0 || wx >= nd) return;
  uint8_t* frame = pyr + (size_t)blockIdx.z * pyrFrameStride;
  const uint8_t* S = frame + Lp.padOff + (size_t)kPad * Lp.pitch + kPad;  // ROI origin of the source level
  uint8_t* dst = frame + L.padOff + (size_t)kPad * L.pitch + wx * 4;
  // first tap of the dword's first (clamped) column: the same IEEE sequence as the host table, so the pixel loads wait for no table
  const int dx0 = min(max(wx * 4 - kPad, 0), L.w - 1);
  const float fx = (float)__dsub_rn(__dmul_rn((double)dx0 + 0.5, scaleX), 0.5);
  const int base = min(max((int)floorf(fx), 0), Lp.w - 1);
  const uint8_t* Sb = S + base;
  unsigned long long w0[kPyrRowsF], w1[kPyrRowsF];
  int bw[kPyrRowsF];
#pragma unroll
  for (int r = 0; r < kPyrRowsF; r++) {
    const int2 rt = rowTab[yb + r];                          // wave-uniform index, table padded to the grid's rows
    bw[r] = rt.y;
    w0[r] = *reinterpret_cast<const u64_unaligned*>(Sb + (size_t)min(max(rt.x, 0), Lp.h - 1) * Lp.pitch);
    w1[r] = *reinterpret_cast<const u64_unaligned*>(Sb + (size_t)min(max(rt.x + 1, 0), Lp.h - 1) * Lp.pitch);
  }
  const PyrColEntry ce = colTab[wx];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 peRaw = reinterpret_cast<const u32x4*>(padTab)[wx];   // one 16-byte load, issued with the pixel loads
  const PyrPadEntry pe{peRaw.x, peRaw.y, peRaw.z, peRaw.w};
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  uint32_t v[kPyrRowsF];
#pragma unroll
  for (int r = 0; r < kPyrRowsF; r++) {
    const int b0 = bw[r] & 0xffff, b1 = (int)((uint32_t)bw[r] >> 16);
    v[r] = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)(w0[r] >> 32), (uint32_t)w0[r], ce.sel[b]);
      const uint32_t p1 = __builtin_amdgcn_perm((uint32_t)(w1[r] >> 32), (uint32_t)w1[r], ce.sel[b]);
      const int h0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p0), __builtin_bit_cast(u16x2, ce.ab[b]), 0u, false);
      const int h1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p1), __builtin_bit_cast(u16x2, ce.ab[b]), 0u, false);
      const int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      v[r] |= (uint32_t)(o & 0xFF) << (8 * b);
    }
  }
  if (__builtin_amdgcn_ballot_w64(pe.selF != kPermIdentity)) {
#pragma unroll
    for (int r = 0; r < kPyrRowsF; r++) v[r] = pyr_finish_dword(v[r], pe);
  }
#pragma unroll
  for (int r = 0; r < kPyrRowsF; r++) pyr_store_rows(dst, L.pitch, yb + r, L.h, v[r]);
}

// ------------------------------------------------------------------------------------------------
// copyMakeBorder(BORDER_REFLECT_101) of every level (orbExtractor.cpp:612-621), one launch for all levels and frames after the
// level kernels: a thread owns one dword of the padded level that holds at least one pad byte — the 19 full rows above and
// below, and up to 6 dwords at either end of an interior row — keeps the interior bytes it finds there and fills each pad byte
// from the reflected interior pixel.  Bytes beyond w + 38 (pitch slack) are zeroed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_borders(uint8_t* __restrict__ pyr, size_t pyrFrameStride, PlanDev P) {
  const int f = blockIdx.y;
  int t = blockIdx.x * 256 + threadIdx.x, level = 0;
#pragma unroll
  for (int l = 1; l < kMaxLevels; l++)
    if (l < P.nLevels && (int)(blockIdx.x * 256) >= P.borderBegin[l]) level = l;   // level starts are workgroup-aligned
  t -= P.borderBegin[level];
  const LevelDev L = P.lv[level];
  const int wd = L.pitch >> 2, rightStart = (kPad + L.w) >> 2;
  // The list of a level is ordered so that waves are (nearly) pure: first the dwords of the 38 pad rows that lie entirely over
  // interior columns - a straight dword copy of the reflected row - then the pad rows' end dwords, then the interior rows' ends.
  constexpr int xdLo = (kPad + 3) / 4;                       // first dword whose 4 columns are all >= 0
  const int xdEnd = min(rightStart, wd);                     // first dword with a column >= w
  const int nFast = max(xdEnd - xdLo, 0), nEdge = wd - nFast;
  uint8_t* base = pyr + (size_t)f * pyrFrameStride + L.padOff;
  if (t < 2 * kPad * nFast) {
    const int r = t / nFast, xf = xdLo + t - r * nFast;
    const int yy = r < kPad ? r : L.h + r;
    const int rr = reflect101(yy - kPad, L.h);
    reinterpret_cast<uint32_t*>(base + (size_t)yy * L.pitch)[xf] = reinterpret_cast<const uint32_t*>(base + (size_t)(rr + kPad) * L.pitch)[xf];
    return;
  }
  t -= 2 * kPad * nFast;
  int y, xd;
  if (t < 2 * kPad * nEdge) {
    const int r = t / nEdge, k = t - r * nEdge;
    xd = nFast == 0 ? k : (k < xdLo ? k : xdEnd + k - xdLo);
    y = r < kPad ? r : L.h + r;                  // rows 0..18 and h+19 .. h+37
  } else {
    const int u = t - 2 * kPad * nEdge, row = u / 12, j = u - row * 12;
    if (row >= L.h) return;
    y = kPad + row;
    xd = j < 6 ? j : rightStart + j - 6;
    if (xd >= wd || (j < 6 && (4 * xd >= kPad || xd >= rightStart))) return;   // left: dwords 0..4; tiny levels: the right end wins
  }
  uint32_t* dw = reinterpret_cast<uint32_t*>(base + (size_t)y * L.pitch) + xd;
  const int Y = y - kPad, ry = reflect101(Y, L.h);
  const bool rowInside = Y >= 0 && Y < L.h;
  const uint32_t old = rowInside ? *dw : 0u;
  const uint8_t* srow = base + (size_t)(ry + kPad) * L.pitch + kPad;
  uint32_t v = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int X = 4 * xd + b - kPad;
    uint32_t px = 0;
    if (X < L.w + kPad) px = (rowInside && X >= 0 && X < L.w) ? (old >> (8 * b)) & 255u : srow[reflect101(X, L.w)];
    v |= px << (8 * b);
  }
  *dw = v;
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 (cv::FAST, call site orbExtractor.cpp:581).  The kernel is VALU-issue bound, and on gfx950 the vector instructions
// come in two speed classes (tools/ubench/valu_rate*.hip, 8 waves per SIMD: ~1.05 ns per wave-instruction for 32-bit add / sub /
// and / or / xor / shifts, f32 add / mul / fma and the 16-bit VOP2 forms v_min/max/add/sub/mul_lo_u16; ~1.8 ns for everything that
// needs a VOP3 encoding - packed i16, perm, bfe, mad, min3/med3, lshl_or - and also for 32-bit and f32 min / max, compares, cndmask,
// mbcnt, cvt).  So the per-pixel work below is written on uint16_t values (the compiler selects the 16-bit VOP2 forms) with as
// few compares / conversions as possible:
//   compass test : a 9-arc of the 16-ring always contains >= 2 of the 4 compass pixels, so "second largest compass pixel
//                  > v + t or second smallest < v - t" is necessary: 8 x v_min/max_u16, two subtractions whose sign bits are
//                  the two answers, ONE compare for the ballot;
//   arc test + score : d = (v - ring) * polarity as 8 packed i16 pairs (d[j], d[j+8]) built by one v_pk_mad_i16 each; the
//                  maximum over the 16 circular 9-arcs of min(d) from prefix / suffix minima of the two 8-blocks (29 packed
//                  operations; the doubling network took 39).  That maximum IS OpenCV's cornerScore<16> + 1, and "> t" IS the
//                  strict 9-contiguous segment test, so one network answers both;
//   NMS          : strictly greater than the maximum of the 8 neighbours in the cell's score map.
// ONE WAVE PER CELL: the 4 waves of a workgroup take 4 consecutive cells and never meet (no workgroup barrier, no cross-wave
// offsets); inside a wave LDS accesses are ordered.  A wave walks its band two rows per step (32 lanes per row; cells wider than
// 32 px - only where a level is < 62 px across - one row per step), keeps the pixels that pass a stage in its own LDS list,
// compacted in place with ballots (row-major = the order cv::FAST returns them), and feeds the list to the next stage, so the
// expensive arc network runs on full waves of survivors of the whole cell.
// ------------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_swap(s16x2 x) { return __builtin_shufflevector(x, x, 1, 0); }
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
// max over the 16 circular 9-arcs of the minimum of d[0..15], d as pairs P[j] = (d[j], d[j + 8]).
// The arc that starts at k < 8 is d[k..7] + d[8..8+k] = min(suffix_lo[k], prefix_hi[k]); the one that starts at 8 + k is
// d[8+k..15] + d[0..k] = min(suffix_hi[k], prefix_lo[k]): both at once as pk_min(suffix[k], swap(prefix[k])).
__device__ __forceinline__ int fast_arc_best(const s16x2 (&P)[8]) {
  s16x2 pre[8], suf[8];
  pre[0] = P[0];
#pragma unroll
  for (int j = 1; j < 8; j++) pre[j] = pk_min(pre[j - 1], P[j]);
  suf[7] = P[7];
#pragma unroll
  for (int j = 6; j >= 0; j--) suf[j] = pk_min(suf[j + 1], P[j]);
  s16x2 m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) m[j] = pk_min(suf[j], pk_swap(pre[j]));
  const s16x2 r = pk_max(pk_max(pk_max(m[0], m[1]), pk_max(m[2], m[3])), pk_max(pk_max(m[4], m[5]), pk_max(m[6], m[7])));
  return max((int)r.x, (int)r.y);
}

// number of set bits of a ballot below this lane (v_mbcnt_lo + v_mbcnt_hi)
__device__ __forceinline__ int wave_rank(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// LDS of one wave: [tile rows x PITCH][score map][candidate list]; sizes come from the plan's largest cell (host: fastLdsLayout)
struct FastLds { int tileBytes, scoreBytes, listBytes; };

// ------------------------------------------------------------------------------------------------
// One wave per (cell, frame): cv::FAST(cell sub-image, thr, nms=true) — orbExtractor.cpp:562-590.
// FAST never looks outside the sub-image, so each cell has a private 3-px dead border and NMS sees
// zeros outside the cell's detection band.  Survivors are written in row-major order (the order
// cv::FAST returns them) into the cell's fixed slot; the quad-tree kernel concatenates cells in
// (row, col) order, which reproduces keyPointsToDistr.  The retry at :583 uses the same threshold
// (m_int_minFastThd is initialised from _initFastThd, :318), so it is a no-op and is not launched.
// PITCH = LDS row pitch of the tile (48 when every cell tile is <= 44 px wide, else 80): a compile-time constant, so the ring
// offsets are immediates of the ds_read instructions.
// ------------------------------------------------------------------------------------------------
template <int PITCH>
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, size_t pyrFrameStride, PlanDev P,
                                                    const CellDev* __restrict__ cells, int cellFirst, int cellEnd, int thr, FastLds lds,
                                                    uint32_t* __restrict__ cellCount, uint32_t* __restrict__ cellCand) {
  extern __shared__ __align__(16) uint8_t fastSmem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* tile = fastSmem + (size_t)wv * (lds.tileBytes + lds.scoreBytes + lds.listBytes);
  uint8_t* score = tile + lds.tileBytes;
  uint16_t* list = reinterpret_cast<uint16_t*>(score + lds.scoreBytes);   // (band row << 6) | band column | darker-arcs flag << 14 | brighter << 15
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own L2); gridDim.x is a multiple of 8, so
  // block (bx, f) runs on XCD bx % 8.  Inside every group of 8 frames the (bx, f) pairs are permuted so that XCD x processes ALL
  // cells of frame 8*(f/8) + x: the 6-px halo a cell shares with its neighbours (and the level a frame's cells share) is then
  // fetched into one L2 instead of eight, while different XCDs still stream different frames (no channel hot-spot).
  int grp = blockIdx.x, f = blockIdx.y;
  if ((f | 7) < (int)gridDim.y) {   // a full group of 8 frames
    grp = (f & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    f = (f & ~7) | (blockIdx.x & 7);
  }
  const int cellId = cellFirst + grp * 4 + wv;   // a launch takes the cells [cellFirst, cellEnd) of every frame (whole levels)
  if (cellId >= cellEnd) return;   // (waves never synchronise with each other)
  const CellDev c = cells[cellId];
  const int pitch = P.lv[c.level].pitch;
  const int tw = c.x1 - c.x0, th = c.y1 - c.y0;
  const int bw = tw - 6, bh = th - 6;  // detection band
  const size_t slot = (size_t)f * P.nCellsTotal + cellId;
  if (bw <= 0 || bh <= 0) {
    if (lane == 0) cellCount[slot] = 0;
    return;
  }
  {  // tile load: unaligned 8-byte loads (reads past tw stay inside the level's 19-px padding), ALL of a lane's rows in flight before
     // the first LDS write: the kernel is otherwise paced by the round trips of these loads (one wave = one cell, nothing to overlap).
     // PITCH 48: lane = (row mod 8, 8-byte column 0..5), <= 6 rows per lane; PITCH 80: (row mod 4, column 0..9), groups of 6 rows.
    constexpr int COLS = PITCH == 48 ? 8 : 16, ROWS = 64 / COLS, NR = 6;
    const uint8_t* sub = pyr + (size_t)f * pyrFrameStride + c.srcOff;
    const int wpr = (tw + 7) >> 3, ty0 = lane / COLS, wd = lane % COLS;
    const bool colOk = wd < wpr;
    for (int tyb = 0; tyb < th; tyb += ROWS * NR) {
      unsigned long long r[NR];
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int ty = tyb + ty0 + ROWS * k;
        r[k] = (colOk && ty < th) ? *reinterpret_cast<const u64_unaligned*>(sub + (unsigned)(ty * pitch + 8 * wd)) : 0ull;   // scalar base + 32-bit lane offset
      }
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int ty = tyb + ty0 + ROWS * k;
        if (colOk && ty < th) *reinterpret_cast<unsigned long long*>(tile + ty * PITCH + 8 * wd) = r[k];
      }
    }
  }
  const int sw = bw + 2, sh = bh + 2;  // score map with a zero ring
  for (int i = lane; i < (sw * sh + 3) >> 2; i += 64) reinterpret_cast<uint32_t*>(score)[i] = 0;
  __builtin_amdgcn_wave_barrier();
  // stage A: compass test over the band.  Two rows of <= 32 columns per step, or one row of <= 64.
  int n1 = 0;
  {
    const bool wide = bw > 32;
    const int bxl = wide ? lane : (lane & 31), byl = wide ? 0 : (lane >> 5), rowsPerIt = wide ? 1 : 2;
    const uint8_t* p = tile + (byl + 3) * PITCH + bxl + 3;
    unsigned ent = (unsigned)(byl << 6) | (unsigned)bxl;
    const uint16_t t16 = (uint16_t)thr;
    auto step = [&](uint16_t vmask) {      // vmask: 0xFFFF on lanes inside the band, 0x7FFF outside (their sign bit never survives)
      const uint16_t v = p[0], a = p[3 * PITCH], b = p[-3 * PITCH], cc = p[3], d = p[-3];
      const uint16_t lo1 = min(a, b), hi1 = max(a, b), lo2 = min(cc, d), hi2 = max(cc, d);
      const uint16_t mx = max(lo1, lo2), mn = min(hi1, hi2);
      const uint16_t second = max(mx, mn), third = min(mn, mx);
      const uint16_t s1 = (uint16_t)((uint16_t)(v + t16) - second);   // sign set: >= 2 brighter compass pixels
      const uint16_t s2 = (uint16_t)(third - (uint16_t)(v - t16));    // sign set: >= 2 darker compass pixels
      const bool keep = (int16_t)((uint16_t)(s1 | s2) & vmask) < 0;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
      if (keep) list[n1 + wave_rank(m)] = (uint16_t)(ent | (s1 & 0x8000u) | ((s2 & 0x8000u) >> 1));
      n1 += __popcll(m);
      p += rowsPerIt * PITCH;
      ent += rowsPerIt * 64;
    };
    const bool colOk = bxl < bw;
    // (opaque to the optimiser: it otherwise turns the mask back into `compare AND lane-valid` and rebuilds the ballot from a 0 / 1 select)
    unsigned vmA = colOk ? 0xFFFFu : 0x7FFFu, vmF = colOk && byl == 0 ? 0xFFFFu : 0x7FFFu;
    asm volatile("" : "+v"(vmA), "+v"(vmF));
    const uint16_t vmAll = (uint16_t)vmA, vmFirst = (uint16_t)vmF;
    int by0 = 0;
    for (; by0 + rowsPerIt <= bh; by0 += rowsPerIt) step(vmAll);
    if (by0 < bh) step(vmFirst);   // odd band height: only the first row of the last step is inside
  }
  __builtin_amdgcn_wave_barrier();
  // stage B: arc test and score in one network; corners stay in the list (compacted in place: writes never pass the reads)
  int n2 = 0;
  for (int base = 0; base < n1; base += 64) {
    const int j = base + lane;
    bool keep = false;
    unsigned ent = 0;
    if (j < n1) {
      ent = list[j];
      const int by = (ent >> 6) & 127, bx = ent & 63;
      const uint8_t* p = tile + (by + 3) * PITCH + bx + 3;
      const short v = (short)p[0];
      // One polarity per pixel: the compass test already says whether the darker or the brighter arcs can pass (a pixel cannot
      // be a corner both ways: 9 + 9 > 16), so d is sign-adjusted and ONE min-network runs.  The few pixels whose compass test
      // passed both ways get the other polarity in a wave-uniform extra step.
      const bool dark = (ent & 0x4000u) != 0;
      const s16x2 sg = dark ? s16x2{1, 1} : s16x2{-1, -1};
      const s16x2 vs = s16x2{v, v} * sg, ng = -sg;
      s16x2 Pd[8];   // (v - ring) * sg = vs + ring * (-sg): one packed multiply-add per opposite pair
      Pd[0] = s16x2{(short)p[3 * PITCH], (short)p[-3 * PITCH]} * ng + vs;
      Pd[1] = s16x2{(short)p[3 * PITCH + 1], (short)p[-3 * PITCH - 1]} * ng + vs;
      Pd[2] = s16x2{(short)p[2 * PITCH + 2], (short)p[-2 * PITCH - 2]} * ng + vs;
      Pd[3] = s16x2{(short)p[PITCH + 3], (short)p[-PITCH - 3]} * ng + vs;
      Pd[4] = s16x2{(short)p[3], (short)p[-3]} * ng + vs;
      Pd[5] = s16x2{(short)p[-PITCH + 3], (short)p[PITCH - 3]} * ng + vs;
      Pd[6] = s16x2{(short)p[-2 * PITCH + 2], (short)p[2 * PITCH - 2]} * ng + vs;
      Pd[7] = s16x2{(short)p[-3 * PITCH + 1], (short)p[3 * PITCH - 1]} * ng + vs;
      int best = fast_arc_best(Pd);
      if (__builtin_amdgcn_ballot_w64((ent & 0xC000u) == 0xC000u)) {   // some pixel of this chunk needs the brighter arcs too
        if ((ent & 0xC000u) == 0xC000u) {
          s16x2 Nd[8];
#pragma unroll
          for (int k = 0; k < 8; k++) Nd[k] = -Pd[k];
          best = max(best, fast_arc_best(Nd));
        }
      }
      keep = best > thr;
      if (keep) score[(by + 1) * sw + bx + 1] = (uint8_t)(best - 1);   // cornerScore<16>: max(t, arcs) - 1
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (keep) list[n2 + wave_rank(m)] = (uint16_t)ent;
    n2 += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  // stage C: NMS (strictly greater than the 8 neighbours; zeros outside the band) and the ordered output
  int n3 = 0;
  uint32_t* dst = cellCand + slot * P.cellCap;
  for (int base = 0; base < n2; base += 64) {
    const int j = base + lane;
    bool keep = false;
    uint32_t rec = 0;
    if (j < n2) {
      const unsigned ent = list[j];
      const int by = (ent >> 6) & 127, bx = ent & 63;
      const uint8_t* q = &score[(by + 1) * sw + bx + 1];
      const uint16_t s = q[0];
      const uint16_t nb = max(max(max((uint16_t)q[-1], (uint16_t)q[1]), max((uint16_t)q[-sw - 1], (uint16_t)q[-sw])),
                              max(max((uint16_t)q[-sw + 1], (uint16_t)q[sw - 1]), max((uint16_t)q[sw], (uint16_t)q[sw + 1])));
      keep = s > nb;
      rec = qt_pack(c.x0 + 3 + bx - kBorder, c.y0 + 3 + by - kBorder, s);
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    const int pos = n3 + wave_rank(m);
    if (keep && pos < P.cellCap) dst[pos] = rec;
    n3 += __popcll(m);
  }
  if (lane == 0) cellCount[slot] = (uint32_t)min(n3, P.cellCap);
}


struct QtBlockCtx {   // workgroup of kQtThreads threads: barriers + LDS hand-off of the per-wave scan totals
  unsigned* w32;
  unsigned long long* w64;
  __device__ int tid() const { return threadIdx.x; }
  __device__ int nthreads() const { return blockDim.x; }
  __device__ void sync() { __syncthreads(); }
  __device__ unsigned scan_incl_u32(unsigned v, unsigned* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) w32[wv] = x;
    __syncthreads();
    unsigned off = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
      const unsigned s = w32[i];
      if (i < wv) off += s;
      tot += s;
    }
    __syncthreads();
    *total = tot;
    return x + off;
  }
  __device__ unsigned long long scan_incl_u64(unsigned long long v, unsigned long long* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned lo = __shfl_up((unsigned)x, d, 64), hi = __shfl_up((unsigned)(x >> 32), d, 64);
      if (lane >= d) x += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) w64[wv] = x;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
      const unsigned long long s = w64[i];
      if (i < wv) off += s;
      tot += s;
    }
    __syncthreads();
    *total = tot;
    return x + off;
  }
  // all lanes of the wave call this together; lanes with k < 0 only take part in the ballots
  template <class P64> __device__ void count_child(P64 cc, int k, int q) {
    unsigned long long remaining = __ballot(k >= 0);
    const int lane = threadIdx.x & 63;
    while (remaining) {
      const int leader = __ffsll((long long)remaining) - 1;
      const int k0 = __shfl(k, leader, 64);
      const bool mine = k == k0;
      const unsigned long long m = __ballot(mine);
      const unsigned long long c0 = __popcll(__ballot(mine && q == 0)), c1 = __popcll(__ballot(mine && q == 1));
      const unsigned long long c2 = __popcll(__ballot(mine && q == 2)), c3 = __popcll(__ballot(mine && q == 3));
      if (lane == leader) __hip_atomic_fetch_add(cc + k0, c0 | (c1 << 16) | (c2 << 32) | (c3 << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      remaining &= ~m;
    }
  }

};

// Wave-synchronous context: ONE 64-lane wave runs a whole (frame, level).  The first version used a 512-thread workgroup and
// spent its time in ~25 __syncthreads per split pass (measured: 0.44 of 0.75 ms in the passes, 1 workgroup per CU); a wave
// needs no workgroup barrier at all, LDS/L1 keep program order inside a wave, and ~1000 (frame, level) units run at once.
struct QtWaveCtx {
  __device__ int tid() const { return threadIdx.x; }
  __device__ int nthreads() const { return 64; }
  __device__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ unsigned scan_incl_u32(unsigned v, unsigned* total) {
    const int lane = threadIdx.x;
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x;
  }
  __device__ unsigned long long scan_incl_u64(unsigned long long v, unsigned long long* total) {
    const int lane = threadIdx.x;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned lo = __shfl_up((unsigned)x, d, 64), hi = __shfl_up((unsigned)(x >> 32), d, 64);
      if (lane >= d) x += ((unsigned long long)hi << 32) | lo;
    }
    *total = ((unsigned long long)__shfl((unsigned)(x >> 32), 63, 64) << 32) | __shfl((unsigned)x, 63, 64);
    return x;
  }
  // all lanes of the wave call this together; lanes with k < 0 only take part in the ballots
  template <class P64> __device__ void count_child(P64 cc, int k, int q) {
    unsigned long long remaining = __ballot(k >= 0);
    const int lane = threadIdx.x & 63;
    while (remaining) {
      const int leader = __ffsll((long long)remaining) - 1;
      const int k0 = __shfl(k, leader, 64);
      const bool mine = k == k0;
      const unsigned long long m = __ballot(mine);
      const unsigned long long c0 = __popcll(__ballot(mine && q == 0)), c1 = __popcll(__ballot(mine && q == 1));
      const unsigned long long c2 = __popcll(__ballot(mine && q == 2)), c3 = __popcll(__ballot(mine && q == 3));
      if (lane == leader) __hip_atomic_fetch_add(cc + k0, c0 | (c1 << 16) | (c2 << 32) | (c3 << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      remaining &= ~m;
    }
  }

};

// dynamic LDS layout: [cc|geom0|geom1|childIdx|cnt0|cnt1|base0|base1] for nodeCap nodes, keys/cand[ldsCandCap] u32 x2,
// node ids [ldsCandCap] u16 x2, cellBase[maxCells+1]
__device__ __forceinline__ void qt_pass_unit(const PlanDev& P, const uint32_t* __restrict__ cellCount,
                                             const uint32_t* __restrict__ cellCand, uint32_t* __restrict__ qtCand,
                                             uint16_t* __restrict__ qtNode, size_t qtFrameStride, int nodeCap, int ldsCandCap, int level, int f,
                                             uint32_t* __restrict__ lvlKp, int* __restrict__ lvlCount, int* __restrict__ status,
                                             uint8_t* __restrict__ nodeScratch) {
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ unsigned w32[16];
  __shared__ unsigned long long w64[16];
  const int lane = threadIdx.x;
  const LevelDev L = P.lv[level];
  bool nodesInHbm = false;
  if (nodeCap <= 0) {   // one launch over all levels (-nodeCap = nodes the launch's LDS holds): each level derives its own table size
    const int have = -nodeCap;
    nodeCap = 4 * max(L.quota, 1);
    nodesInHbm = nodeCap > have;
  }
  if (L.nodeTabOff >= 0) nodesInHbm = true;   // the plan gave this level HBM node tables (its quota does not fit the LDS)
  if (nodesInHbm && (L.nodeTabOff < 0 || !nodeScratch)) {
    if (lane == 0) { atomicMax(status, 3); lvlCount[f * kMaxLevels + level] = 0; }
    return;
  }
  uint8_t* sp = smem;
  QtShared S;
  QtSharedT<false> SG;
  if (!nodesInHbm) {
    S.cc = (QT_LDS unsigned long long*)(sp); sp += sizeof(unsigned long long) * nodeCap;
    S.geom[0] = (QT_LDS QtGeom*)(sp); sp += sizeof(QtGeom) * nodeCap;
    S.geom[1] = (QT_LDS QtGeom*)(sp); sp += sizeof(QtGeom) * nodeCap;
    S.childIdx = (QT_LDS uint16_t*)(sp); sp += sizeof(uint16_t) * 4 * nodeCap;
    S.cnt[0] = (QT_LDS uint32_t*)(sp); sp += sizeof(uint32_t) * nodeCap;
    S.cnt[1] = (QT_LDS uint32_t*)(sp); sp += sizeof(uint32_t) * nodeCap;
    S.base[0] = (QT_LDS uint32_t*)(sp); sp += sizeof(uint32_t) * nodeCap;
    S.base[1] = (QT_LDS uint32_t*)(sp); sp += sizeof(uint32_t) * nodeCap;
  } else {   // same layout in this (frame, level)'s slice of the HBM node scratch
    uint8_t* gp = nodeScratch + (size_t)f * P.nodeTabFrameStride + L.nodeTabOff;
    SG.cc = (unsigned long long*)(gp); gp += sizeof(unsigned long long) * nodeCap;
    SG.geom[0] = (QtGeom*)(gp); gp += sizeof(QtGeom) * nodeCap;
    SG.geom[1] = (QtGeom*)(gp); gp += sizeof(QtGeom) * nodeCap;
    SG.childIdx = (uint16_t*)(gp); gp += sizeof(uint16_t) * 4 * nodeCap;
    SG.cnt[0] = (uint32_t*)(gp); gp += sizeof(uint32_t) * nodeCap;
    SG.cnt[1] = (uint32_t*)(gp); gp += sizeof(uint32_t) * nodeCap;
    SG.base[0] = (uint32_t*)(gp); gp += sizeof(uint32_t) * nodeCap;
    SG.base[1] = (uint32_t*)(gp);
    ldsCandCap = 0;                          // candidates from HBM scratch as well: the launch's LDS only holds the cell bases
  }
  QT_LDS uint32_t* ldsCand = (QT_LDS uint32_t*)(sp); sp += sizeof(uint32_t) * 2 * ldsCandCap;
  QT_LDS uint16_t* ldsNode = (QT_LDS uint16_t*)(sp); sp += sizeof(uint16_t) * 2 * ldsCandCap;
  QT_LDS uint32_t* cellBase = (QT_LDS uint32_t*)(sp);
  QtBlockCtx cx{w32, w64};

  // concatenate the level's cells in (row, col) order == keyPointsToDistr (orbExtractor.cpp:584-590)
  const uint32_t* cnts = cellCount + (size_t)f * P.nCellsTotal + L.cellBegin;
  unsigned n = 0;
  const int NTQ = cx.nthreads();   // k_quadtree: kQtThreads; k_quadtree_list: kQtListThreads
  for (int c0 = 0; c0 < L.nCells; c0 += NTQ) {
    const int c = c0 + lane;
    const unsigned v = c < L.nCells ? cnts[c] : 0u;
    unsigned tot;
    const unsigned incl = cx.scan_incl_u32(v, &tot);
    if (c < L.nCells) cellBase[c] = n + incl - v;
    n += tot;
  }
  cx.sync();
  const size_t so = (size_t)f * qtFrameStride + L.candOff;
  const size_t cap = (size_t)L.nCells * P.cellCap;
  // Two copies of the algorithm, one over LDS-resident candidates and one over HBM scratch, selected here (not through a
  // runtime pointer select): the compiler then proves the address space and emits ds_* instead of flat_* for the LDS copy.
  const bool inLds = (int)n <= ldsCandCap;
  QtCandT<true> GL{{ldsCand, ldsCand + ldsCandCap}, {ldsNode, ldsNode + ldsCandCap}};
  QtGlobal GG{{qtCand + 2 * so, qtCand + 2 * so + cap}, {qtNode + 2 * so, qtNode + 2 * so + cap}};
  int nOut = 0;
  if (n > 65535u) {
    if (lane == 0) atomicMax(status, 1);  // more candidates than the packed 16-bit counters allow
  } else {
    // one thread per cell: its (few) entries are independent loads
    for (int c = lane; c < L.nCells; c += NTQ) {
      const unsigned m = cnts[c], b = cellBase[c];
      const uint32_t* src = cellCand + ((size_t)f * P.nCellsTotal + L.cellBegin + c) * P.cellCap;
      if (inLds) { for (unsigned i = 0; i < m; i++) ldsCand[b + i] = src[i]; }
      else { for (unsigned i = 0; i < m; i++) GG.cand[0][b + i] = src[i]; }
    }
    cx.sync();
#if defined(QT_DBG_STOP) && QT_DBG_STOP == 1
    if (lane == 0) lvlCount[f * kMaxLevels + level] = 0;
    return;
#endif
    uint32_t* outKp = lvlKp + (size_t)f * P.sumQuota + L.kpOff;
    if (nodesInHbm) nOut = qt_distribute(cx, SG, GG, (int)n, L.w - 2 * kBorder, L.h - 2 * kBorder, L.quota, nodeCap, outKp);
    else if (inLds) nOut = qt_distribute(cx, S, GL, (int)n, L.w - 2 * kBorder, L.h - 2 * kBorder, L.quota, nodeCap, outKp);
    else nOut = qt_distribute(cx, S, GG, (int)n, L.w - 2 * kBorder, L.h - 2 * kBorder, L.quota, nodeCap, outKp);
    if (nOut < 0) {
      if (lane == 0) atomicMax(status, 2);
      nOut = 0;
    }
  }
  if (lane == 0) lvlCount[f * kMaxLevels + level] = nOut;
}
// every unit of a level: grid (1, frames) per level (YDORB_QT_PASS=1, or levels whose quota is too large for k_qt_fast's LDS)
__global__ __launch_bounds__(kQtThreads) void k_quadtree(PlanDev P, const uint32_t* __restrict__ cellCount,
                                                 const uint32_t* __restrict__ cellCand, uint32_t* __restrict__ qtCand,
                                                 uint16_t* __restrict__ qtNode, size_t qtFrameStride, int nodeCap, int ldsCandCap, int levelBase,
                                                 uint32_t* __restrict__ lvlKp, int* __restrict__ lvlCount, int* __restrict__ status,
                                                 uint8_t* __restrict__ nodeScratch) {
  qt_pass_unit(P, cellCount, cellCand, qtCand, qtNode, qtFrameStride, nodeCap, ldsCandCap, levelBase + blockIdx.x, blockIdx.y, lvlKp, lvlCount, status, nodeScratch);
}
// the units k_qt_fast handed over (rare): a FEW workgroups walk the hand-over list.  (One workgroup per (frame, level) unit that leaves at
// once when its unit was not handed over looked free - 4.5 us alone - but 8192 workgroups of 512 threads and ~40 KB of LDS queue behind the
// other lanes' kernels: 2.7 ms on average in the four-lane pipeline, in front of the lane's descriptor kernel; even 32 of them waited
// 0.25 - 0.9 ms for their slots with an EMPTY list.)  The host sizes the grid from the list length of an earlier call (*lastCount, copied
// back with the level maxima): one workgroup while nothing is handed over, up to kQtPassWorkgroups when a stream's frames need it - any grid
// >= 1 walks the whole list.
constexpr int kQtPassWorkgroups = 256;
constexpr int kQtListThreads = 256;   // the footprint of the other kernels' workgroups (4 waves): placed as soon as one of those retires
__global__ __launch_bounds__(kQtThreads) void k_quadtree_list(PlanDev P, const uint32_t* __restrict__ cellCount,
                                                 const uint32_t* __restrict__ cellCand, uint32_t* __restrict__ qtCand,
                                                 uint16_t* __restrict__ qtNode, size_t qtFrameStride, int nodeCap, uint32_t* __restrict__ lvlKp,
                                                 int* __restrict__ lvlCount, int* __restrict__ status, const int* __restrict__ passCount,
                                                 const int* __restrict__ passList, uint8_t* __restrict__ nodeScratch, int* __restrict__ lastCount) {
  const int n = *passCount;
  if (blockIdx.x == 0 && threadIdx.x == 0) *lastCount = n;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int unit = passList[i];
    qt_pass_unit(P, cellCount, cellCand, qtCand, qtNode, qtFrameStride, nodeCap, 0, unit % kMaxLevels, unit / kMaxLevels, lvlKp, lvlCount, status, nodeScratch);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Quad-tree thinning without passes and without a sort (closed form: quadtree_flat.h).  One 256-thread workgroup per (frame, level)
// unit, ONE launch for a range of levels, ~20 KB of LDS, one register of state per candidate (ITEMS per thread, flat index
// i = j*256 + tid):
//   1. scan of the level's cell counts -> cell bases (u16); a thread finds candidate i by a 10-step binary search over them and
//      loads it straight from the FAST kernel's cell slots (no staging copy);
//   2. path key to depth 6 from two LDS tables: the x digits of the reference's split sequence depend on x only, the y digits on
//      y only until the first child 3 (divideNode's line-27 slip gives node 4 the parent's TOP edge as its bottom: from there on
//      every y test fails), so key = xs[x] | ys[y], then every y bit below the first digit 3 is set (checked against qt_path_key for
//      every pixel of several level sizes).  Keys feed a histogram pyramid H_d (4^6 bins at the bottom, u16, packed LDS atomics),
//      parents from children; per-depth node / leaf tallies give the number of passes P and the list length K the reference's
//      loop ends with (qt_flat_passes);
//   3. the final list is [depth-P nodes, ascending in prefix ^ mask_P] ++ [leaves of depth P-1, ascending in prefix ^ mask_{P-1}] ++ ...
//      (quadtree_flat.h, observation 3), so a node's list position is its rank in that order: ONE block scan over the flags of the
//      concatenated (permuted) bin spaces; the pyramid is overwritten IN PLACE with the positions (bit 15: more than 16 members);
//   4. every candidate looks its node's position up (first depth where it is alone, then the position) and joins two LDS atomic
//      maxima per node: (response, smallest original index) and (response, largest original index);
//   5. a node whose two maxima name the same candidate, or with <= 16 members (insertion sort: first maximum), is done.  Only a
//      tied maximum in a node of more than 16 members needs libstdc++'s introsort replayed (qt_sort_front): its members are
//      collected (LDS append, then ranked by original index) into the dead pyramid and one thread per such node runs the replay.
// A tree deeper than the pyramid - sparse scenes, where nearly every candidate ends up alone: the usual case on real images - with at
// most 1024 candidates takes the all-pairs form instead of steps 2-3: 30-bit keys (15 levels), a candidate's "alone" depth and its
// group's first depth from pairwise common prefixes, the same tallies -> P, K, then its node's list position = number of distinct
// smaller ranks R (quadtree_flat.h), again by comparing all pairs; steps 4-5 are shared.
// Units the kernel does not take - more than 256*ITEMS candidates, more than 1024 cells, a deep tree with more than 1024 candidates,
// quota <= 1, tie buffers too small - raise needPass and k_quadtree (the pass algorithm) processes exactly those afterwards.
// ------------------------------------------------------------------------------------------------
constexpr int kQfThreads = 256;
constexpr int kQfMaxCells = 1024;
constexpr int kQfTieCap = kQfPyrU16 * 2 / 8;           // (key, value) pairs the dead pyramid holds: 1366
constexpr int kQfMaxTied = 32;
constexpr int kQfSmall = 1024;                         // all-pairs form (trees deeper than the pyramid) up to this many candidates
constexpr int kQfRegionA = kQfSmall * (4 + 8 + 1) > kQfPyrU16 * 2 ? kQfSmall * (4 + 8 + 1) : kQfPyrU16 * 2;   // pyramid | all-pairs arrays | tie buffers
// tabLen = (w - 32 + 1) + (h - 32 + 1) of the launch's largest level
inline size_t qt_fast_lds_bytes(int quotaMax, int tabLen) {
  return (size_t)kQfRegionA + (size_t)(kQfMaxCells + 8) * 2 + (size_t)((tabLen + 7) & ~7) * 2 + (size_t)(quotaMax + 2) * 8 + (size_t)((quotaMax + 2 + 15) & ~15) + 64;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, d, 64), hi = __shfl_xor((unsigned)(v >> 32), d, 64);
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

template <int ITEMS>
__global__ __launch_bounds__(kQfThreads) void k_qt_fast(PlanDev P, const uint32_t* __restrict__ cellCount, const uint32_t* __restrict__ cellCand,
                                                        int levelFirst, int quotaMax, int tabLen, uint32_t* __restrict__ lvlKp,
                                                        int* __restrict__ lvlCount, uint8_t* __restrict__ needPass, int* __restrict__ lvlMaxN,
                                                        int* __restrict__ passCount, int* __restrict__ passList) {
  constexpr int NT = kQfThreads, CAP = NT * ITEMS, D = kQfDepth;
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ unsigned w32[4];
  __shared__ unsigned long long sRed[4];
  __shared__ int sTieN, sFlag, sTieCnt[kQfMaxTied], sTieNode[kQfMaxTied];
  const int f = blockIdx.x, level = levelFirst + blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const LevelDev L = P.lv[level];
  const int unit = f * kMaxLevels + level, quota = L.quota;
  QT_LDS uint16_t* hist = (QT_LDS uint16_t*)smem;
  QT_LDS uint32_t* hist32 = (QT_LDS uint32_t*)smem;
  QT_LDS uint16_t* cellBase = (QT_LDS uint16_t*)(smem + kQfRegionA);                          // [kQfMaxCells + 1]
  QT_LDS uint16_t* xs = cellBase + kQfMaxCells + 8;                                           // [w - 32 + 1], then ys [h - 32 + 1]
  QT_LDS uint32_t* nodeLo = (QT_LDS uint32_t*)(xs + ((tabLen + 7) & ~7));                     // [quotaMax + 2] each
  QT_LDS uint32_t* nodeHi = nodeLo + quotaMax + 2;
  QT_LDS uint8_t* nodeBig = (QT_LDS uint8_t*)(nodeHi + quotaMax + 2);

  auto block_scan_excl = [&](unsigned v, unsigned* total) -> unsigned {   // exclusive prefix over the 256 threads (two barriers)
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) w32[wv] = x;
    __syncthreads();
    const unsigned s0 = w32[0], s1 = w32[1], s2 = w32[2], s3 = w32[3];
    __syncthreads();
    *total = s0 + s1 + s2 + s3;
    return x - v + (wv > 0 ? s0 : 0u) + (wv > 1 ? s1 : 0u) + (wv > 2 ? s2 : 0u);
  };
  auto give_up = [&]() { if (tid == 0) { needPass[unit] = 1; passList[atomicAdd(passCount, 1)] = unit; } };   // k_quadtree_list takes the unit

  // ---- 1. cell bases: the level's cells in (row, col) order == keyPointsToDistr (orbExtractor.cpp:584-590) ----------------------
  const int rootX1 = L.w - 2 * kBorder, rootY1 = L.h - 2 * kBorder;
  if (L.nCells > kQfMaxCells || rootX1 < 0 || rootY1 < 0 || rootX1 + rootY1 + 2 > tabLen) { give_up(); return; }
  const uint32_t* cnts = cellCount + (size_t)f * P.nCellsTotal + L.cellBegin;
  unsigned cc4[4], mine = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { cc4[k] = 4 * tid + k < L.nCells ? cnts[4 * tid + k] : 0u; mine += cc4[k]; }
  if (tid < 4) sRed[tid] = 0;
  if (tid == 0) { sTieN = 0; sFlag = 0; }
  if (tid < kQfMaxTied) sTieCnt[tid] = 0;
  unsigned nU;
  unsigned run = block_scan_excl(mine, &nU);
  const int n = (int)nU;
  if (tid == 0 && n > lvlMaxN[level]) atomicMax(&lvlMaxN[level], n);   // the host sizes the next launch from this
  if (n > CAP) { give_up(); return; }
  if (n == 0 || quota <= 0) {
    if (tid == 0) { needPass[unit] = 0; lvlCount[unit] = 0; }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { cellBase[4 * tid + k] = (uint16_t)run; run += cc4[k]; }   // cells beyond nCells: base = n
  if (tid == 0) cellBase[kQfMaxCells] = (uint16_t)n;
  for (int i = tid; i < kQfPyrU16 / 2; i += NT) hist32[i] = 0;
  for (int k = tid; k < quotaMax + 2; k += NT) { nodeLo[k] = 0; nodeHi[k] = 0; nodeBig[k] = 0; }
  // split digits of the reference's geometry (qt_center = ceil of the midpoint), most significant first, already spread to the key's
  // even (x) and odd (y) bit positions and complemented: bit set = "not left" / "not top"
  QT_LDS uint16_t* ys = xs + rootX1 + 1;
  for (int v = tid; v <= rootX1 + rootY1 + 1; v += NT) {
    const bool isY = v > rootX1;
    const uint32_t bits = qf_axis_digits(isY ? v - rootX1 - 1 : v, isY ? rootY1 : rootX1);
    xs[v] = (uint16_t)(isY ? bits << 1 : bits);
  }
  __syncthreads();
  const uint32_t* slots = cellCand + ((size_t)f * P.nCellsTotal + L.cellBegin) * P.cellCap;
  uint32_t* outKp = lvlKp + (size_t)f * P.sumQuota + L.kpOff;
  auto fetch = [&](int i) -> uint32_t {     // candidate i of the level's concatenated list
    int c = 0;
#pragma unroll
    for (int st = kQfMaxCells / 2; st >= 1; st >>= 1)
      if ((int)cellBase[c + st] <= i) c += st;
    return slots[(size_t)c * P.cellCap + (i - (int)cellBase[c])];
  };
  if (n == 1) {          // a single candidate: the root is the only node (list [root], P = 1, K = 1)
    if (tid == 0) { outKp[0] = fetch(0); needPass[unit] = 0; lvlCount[unit] = 1; }
    return;
  }
  if (quota <= 1) { give_up(); return; }   // P = 0: the whole level is one node (never with the reference's quotas)

  // ---- 2. candidates -> one register each (key | response << 16), histogram pyramid ------------------------------------------------
  // Groups of four candidates, straight-line inside a group (an index beyond n is clamped and contributes nothing), so that the four
  // binary searches / loads / table look-ups of a group are in flight together; a group beyond n is skipped (wave-uniform).
  static_assert(ITEMS % 4 == 0, "groups of four");
  uint32_t st[ITEMS];
#pragma unroll
  for (int g = 0; g < ITEMS / 4; g++) {
#pragma unroll
    for (int jj = 0; jj < 4; jj++) st[4 * g + jj] = 0;
    if (4 * g * NT < n) {
      int ii[4], cc[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) { ii[jj] = min((4 * g + jj) * NT + tid, n - 1); cc[jj] = 0; }
#pragma unroll
      for (int sp = kQfMaxCells / 2; sp >= 1; sp >>= 1)
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
          cc[jj] += (int)cellBase[cc[jj] + sp] <= ii[jj] ? sp : 0;
      uint32_t c[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) c[jj] = slots[(size_t)cc[jj] * P.cellCap + (ii[jj] - (int)cellBase[cc[jj]])];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const uint32_t key = qf_key(xs[qt_x(c[jj])], ys[qt_y(c[jj])]);
        st[4 * g + jj] = key | ((uint32_t)qt_r(c[jj]) << 16);
        const int idx = qf_off(D) + (int)key;
        const bool real = (4 * g + jj) * NT + tid < n;
        __hip_atomic_fetch_add(hist32 + (idx >> 1), real ? 1u << (16 * (idx & 1)) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  // parents from children; a parent with >= 2 candidates makes its occupied children list nodes of the next pass and its
  // single-candidate children leaves.  Packed 16-bit tallies: [nodes d=1..4][nodes 5..6][leaves 1..4][leaves 5..6]
  unsigned long long tn0 = 0, tn1 = 0, tl0 = 0, tl1 = 0;
  auto reduce_level = [&](int d, int b) {   // parent b of depth d from its four children of depth d + 1
    const uint2 q = *reinterpret_cast<const QT_LDS uint2*>(hist + qf_off(d + 1) + 4 * b);
    const unsigned c0 = q.x & 0xFFFFu, c1 = q.x >> 16, c2 = q.y & 0xFFFFu, c3 = q.y >> 16;
    const unsigned sum = c0 + c1 + c2 + c3;
    hist[qf_off(d) + b] = (uint16_t)sum;
    if (sum >= 2) {
      const unsigned long long nn = (c0 != 0) + (c1 != 0) + (c2 != 0) + (c3 != 0), ll = (c0 == 1) + (c1 == 1) + (c2 == 1) + (c3 == 1);
      const int dd = d + 1;   // depth of the children
      if (dd <= 4) { tn0 += nn << (16 * (dd - 1)); tl0 += ll << (16 * (dd - 1)); }
      else { tn1 += nn << (16 * (dd - 5)); tl1 += ll << (16 * (dd - 5)); }
    }
  };
#pragma unroll
  for (int q = 0; q < 4; q++) reduce_level(5, tid + NT * q);
  __syncthreads();
  reduce_level(4, tid);
  __syncthreads();
  if (wv == 0) {                            // depths 3 .. 0 on one wave (LDS accesses of a wave are in program order)
    reduce_level(3, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (lane < 16) reduce_level(2, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (lane < 4) reduce_level(1, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (lane == 0) reduce_level(0, 0);
  }
  tn0 = wave_sum_u64(tn0); tn1 = wave_sum_u64(tn1); tl0 = wave_sum_u64(tl0); tl1 = wave_sum_u64(tl1);
  if (lane == 0) {
    __hip_atomic_fetch_add(&sRed[0], tn0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&sRed[1], tn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&sRed[2], tl0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&sRed[3], tl1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  int Pn, K = 0;
  {
    int nodes[D + 1], leaves[D + 1];
    nodes[0] = 1; leaves[0] = 0;
    const unsigned long long a0 = sRed[0], a1 = sRed[1], b0 = sRed[2], b1 = sRed[3];
#pragma unroll
    for (int d = 1; d <= D; d++) {
      nodes[d] = (int)(((d <= 4 ? a0 : a1) >> (16 * ((d - 1) & 3))) & 0xFFFF);
      leaves[d] = (int)(((d <= 4 ? b0 : b1) >> (16 * ((d - 1) & 3))) & 0xFFFF);
    }
    Pn = qt_flat_passes(nodes, leaves, quota, &K, D);
  }
  const bool pairs = Pn < 1;               // the reference's loop would still be splitting below depth 6
  if (pairs && n > kQfSmall) { give_up(); return; }
  int nOut = K < quota ? K : quota;
  if (pairs) {
    // ---- 2b / 3b. few candidates, deep tree: pairwise common prefixes of 30-bit keys --------------------------------------------
    constexpr int DP = kQtPairDepth, PER = kQfSmall / NT;
    QT_LDS uint32_t* key30 = (QT_LDS uint32_t*)smem;                                  // [kQfSmall]
    QT_LDS unsigned long long* R64 = (QT_LDS unsigned long long*)(smem + 4 * kQfSmall);   // [kQfSmall]
    QT_LDS uint8_t* firstOf = (QT_LDS uint8_t*)(smem + 12 * kQfSmall);                // [kQfSmall]: 1 = no earlier candidate in the same node
    __shared__ int sDiff[kQtPairDepth + 3], sLeaf[kQtPairDepth + 2];
    const int n4 = (n + 3) & ~3;
    uint32_t kk[PER];
    __syncthreads();                        // every read of the pyramid is done
    if (tid < DP + 3) sDiff[tid] = 0;
    if (tid < DP + 2) sLeaf[tid] = 0;
#pragma unroll
    for (int t = 0; t < PER; t++) {
      const int i = t * NT + tid;
      kk[t] = i < n ? qt_path_key<kQtPairDepth>(fetch(i), rootX1, rootY1) : 0xFFFFFFFFu;   // padding: clz(x) = 0 -> depth 0, neutral
      if (i < n4) key30[i] = kk[t];
    }
    __syncthreads();
    int sA[PER], eA[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) { sA[t] = 0; eA[t] = 0; }
    for (int q = 0; q < n4; q += 4) {
      const uint4 kq = *reinterpret_cast<const QT_LDS uint4*>(key30 + q);
      const uint32_t ko[4] = {kq.x, kq.y, kq.z, kq.w};
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int t = 0; t < PER; t++) {
          const int i = t * NT + tid;
          const uint32_t x = kk[t] ^ ko[u];
          const int dpt = q + u == i ? 0 : (x ? ((__clz((int)x) - 2) >> 1) + 1 : DP + 1);
          sA[t] = max(sA[t], dpt);
          if (q + u < i) eA[t] = max(eA[t], dpt);
        }
    }
#pragma unroll
    for (int t = 0; t < PER; t++)
      if (t * NT + tid < n) {
        if (eA[t] <= DP) { atomicAdd(&sDiff[eA[t]], 1); atomicAdd(&sDiff[min(sA[t], DP) + 1], -1); }
        if (sA[t] <= DP) atomicAdd(&sLeaf[sA[t]], 1);
      }
    __syncthreads();
    {
      int nodes[DP + 1], leaves[DP + 1], run2 = 0;
#pragma unroll
      for (int d = 0; d <= DP; d++) { run2 += sDiff[d]; nodes[d] = run2; leaves[d] = sLeaf[d]; }
      Pn = qt_flat_passes(nodes, leaves, quota, &K, DP);
    }
    if (Pn < 1) { give_up(); return; }      // cannot happen for 12-bit coordinates; the pass kernel would cope
    nOut = K < quota ? K : quota;
    unsigned long long RR[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) {
      const int i = t * NT + tid, dA = min(Pn, sA[t]);
      RR[t] = ((unsigned long long)(Pn - dA) << (2 * Pn)) | (unsigned long long)((kk[t] >> (2 * (DP - dA))) ^ qt_flat_mask(dA));
      if (i < n4) R64[i] = i < n ? RR[t] : ~0ull;
    }
    __syncthreads();
    // sweep A: is the candidate the first of its node (no earlier candidate with the same rank)?
    bool fst[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) fst[t] = true;
    for (int q = 0; q < n4; q += 2) {
      const ulonglong2 o = *reinterpret_cast<const QT_LDS ulonglong2*>(R64 + q);
      const unsigned long long oo[2] = {o.x, o.y};
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int t = 0; t < PER; t++) fst[t] = fst[t] && !(oo[u] == RR[t] && q + u < t * NT + tid);
    }
#pragma unroll
    for (int t = 0; t < PER; t++) { const int i = t * NT + tid; if (i < n4) firstOf[i] = i < n && fst[t]; }
    __syncthreads();
    // sweep B: list position of the node = number of distinct smaller ranks; members of the node
    int kp[PER], mm[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) { kp[t] = 0; mm[t] = 0; }
    for (int q = 0; q < n4; q += 2) {
      const ulonglong2 o = *reinterpret_cast<const QT_LDS ulonglong2*>(R64 + q);
      const unsigned long long oo[2] = {o.x, o.y};
      const unsigned f2 = *reinterpret_cast<const QT_LDS uint16_t*>(firstOf + q);
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int t = 0; t < PER; t++) {
          kp[t] += (oo[u] < RR[t]) & ((f2 >> (8 * u)) & 1u);
          mm[t] += oo[u] == RR[t];
        }
    }
    // hand-over to the shared tail: the low half of the state = the node's list position (0xFFFF = beyond the quota)
#pragma unroll
    for (int t = 0; t < PER; t++) {
      const int i = t * NT + tid;
      if (i < n && kp[t] < nOut && mm[t] > 16) nodeBig[kp[t]] = 1;
    }
    static_assert(PER <= ITEMS, "the all-pairs candidates live in the first PER state registers");
#pragma unroll
    for (int t = 0; t < PER; t++) st[t] = (st[t] & 0xFFFF0000u) | (uint32_t)(kp[t] < nOut ? kp[t] : 0x7FFF);
    __syncthreads();                        // the key / rank arrays are dead: the region becomes the position table of step 4 (identity)
  }

  // ---- 3. depth at which each candidate's node stops splitting: the first depth where it is alone, else P -------------------------
#pragma unroll
  for (int g = 0; g < ITEMS / 4; g++) {
    if (!pairs && 4 * g * NT < n) {
      int key[4], slot[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) { key[jj] = (int)(st[4 * g + jj] & 0xFFFu); slot[jj] = qf_off(Pn) + (key[jj] >> (2 * (D - Pn))); }
      for (int d = Pn - 1; d >= 1; d--) {
        const int off = qf_off(d), sh = 2 * (D - d);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const int sl = off + (key[jj] >> sh);
          if (hist[sl] == 1) slot[jj] = sl;
        }
      }
      // from here on the low half is the pyramid slot of the candidate's final node
#pragma unroll
      for (int jj = 0; jj < 4; jj++) st[4 * g + jj] = (st[4 * g + jj] & 0xFFFF0000u) | (uint32_t)slot[jj];
    }
  }
  // list positions: quads (four siblings) of the groups d = P, P-1, .., 1 in list order, a contiguous run of quads per thread
  if (!pairs) {
    const int TQ = ((1 << (2 * Pn)) - 1) / 3;            // 4^(P-1) + .. + 1
    const int per = (TQ + NT - 1) / NT;                  // <= 6
    constexpr int MAXQ = ((1 << (2 * D)) - 1) / 3 / NT + 1;
    uint2 qv[MAXQ];
    int qslot[MAXQ];                                     // pyramid index of the quad's first child; -1: none
    unsigned flags = 0, cntF = 0;                        // 4 flag bits per quad, in LIST order (child 3 first)
#pragma unroll
    for (int t = 0; t < MAXQ; t++) {
      qslot[t] = -1; qv[t] = make_uint2(0, 0);
      const int q = tid * per + t;
      if (t < per && q < TQ) {
        int d = Pn, start = 0;                           // group of quad q: depth d holds 4^(d-1) quads
        while (q >= start + (1 << (2 * (d - 1)))) { start += 1 << (2 * (d - 1)); d--; }
        const int pq = q - start;                        // permuted parent index inside the group
        const int bq = pq ^ (int)(qt_flat_mask(d) >> 2); // natural parent index (depth d - 1)
        qslot[t] = qf_off(d) + 4 * bq;
        qv[t] = *reinterpret_cast<const QT_LDS uint2*>(hist + qslot[t]);
        const unsigned c[4] = {qv[t].x & 0xFFFFu, qv[t].x >> 16, qv[t].y & 0xFFFFu, qv[t].y >> 16};
        const unsigned sum = c[0] + c[1] + c[2] + c[3];
#pragma unroll
        for (int e = 0; e < 4; e++) {                    // list order inside a quad: child 3, 2, 1, 0 (the last digit is always complemented)
          const unsigned cn = c[3 - e];
          const bool fl = sum >= 2 && (d == Pn ? cn >= 1 : cn == 1);
          flags |= (unsigned)fl << (4 * t + e);
          cntF += fl;
        }
      }
    }
    unsigned total;
    unsigned pos = block_scan_excl(cntF, &total);        // (its barriers also end every read of the counts: step 3 above included)
    if ((int)total != K) { give_up(); return; }          // cannot happen: tallies and flags describe the same list
#pragma unroll
    for (int t = 0; t < MAXQ; t++) {
      if (qslot[t] >= 0) {
        const unsigned c[4] = {qv[t].x & 0xFFFFu, qv[t].x >> 16, qv[t].y & 0xFFFFu, qv[t].y >> 16};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned cn = c[3 - e];
          unsigned v = 0x7FFFu;
          if ((flags >> (4 * t + e)) & 1u) {
            if ((int)pos < nOut) {
              v = pos | (cn > 16 ? 0x8000u : 0u);
              if (cn > 16) nodeBig[pos] = 1;
            }
            pos++;
          }
          o[3 - e] = v;
        }
        *reinterpret_cast<QT_LDS uint2*>(hist + qslot[t]) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
      }
    }
  }
  __syncthreads();

  // ---- 4. per node: maximum response with the smallest and with the largest original index -------------------------------------
  // (a candidate beyond n or beyond the quota joins the maxima of a spare slot with the value 0: no branch)
#pragma unroll
  for (int g = 0; g < ITEMS / 4; g++) {
    if (4 * g * NT < n) {
      unsigned pv[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) pv[jj] = pairs ? st[4 * g + jj] & 0x7FFFu : hist[st[4 * g + jj] & 0xFFFFu] & 0x7FFFu;
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const int i = (4 * g + jj) * NT + tid;
        const bool ok = i < n && (int)pv[jj] < nOut;
        const unsigned k = ok ? pv[jj] : (unsigned)(quotaMax + 1);
        const uint32_t rs = st[4 * g + jj] & 0xFFFF0000u;
        __hip_atomic_fetch_max(nodeLo + k, ok ? rs | (0xFFFFu - (uint32_t)i) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(nodeHi + k, ok ? rs | (uint32_t)i : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        st[4 * g + jj] = rs | (ok ? k : 0xFFFFu);   // from here on the low half is the candidate's node (list position), 0xFFFF = none
      }
    }
  }
  __syncthreads();
  // ---- 5. output; nodes with a tied maximum and more than 16 members are set aside for the introsort replay --------------------
  for (int k = tid; k < nOut; k += NT) {
    const unsigned iLo = 0xFFFFu - (nodeLo[k] & 0xFFFFu), iHi = nodeHi[k] & 0xFFFFu;
    if (iLo == iHi || !nodeBig[k]) {
      outKp[k] = fetch((int)iLo);
    } else {
      const int s = atomicAdd(&sTieN, 1);
      if (s < kQfMaxTied) { sTieNode[s] = k; nodeBig[k] = (uint8_t)(2 + s); }
    }
  }
  __syncthreads();
  const int nTied = sTieN;
  if (nTied > kQfMaxTied) { give_up(); return; }
  if (nTied > 0) {
    // the pyramid is dead (every position was read before the barrier above): it becomes [keys u32 | original indices u32] of the tied nodes
    const int slotCap = kQfTieCap / nTied;
    QT_LDS uint32_t* tKey = (QT_LDS uint32_t*)smem;
    QT_LDS uint32_t* tVal = tKey + kQfTieCap;
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
      if (j * NT < n) {
        const unsigned k = st[j] & 0xFFFFu;
        if (k != 0xFFFFu) {
          const unsigned tb = nodeBig[k];
          if (tb >= 2) {
            const int s = (int)tb - 2, idx = atomicAdd(&sTieCnt[s], 1);
            if (idx < slotCap) { tKey[s * slotCap + idx] = (uint32_t)(j * NT + tid); tVal[s * slotCap + idx] = st[j] >> 16; }
            else sFlag = 1;
          }
        }
      }
    }
    __syncthreads();
    if (sFlag) { give_up(); return; }       // a tied node larger than its share of the buffer: pass kernel
    for (int s = 0; s < nTied; s++) {       // members in original order: rank = number of members with a smaller original index
      const int m = sTieCnt[s];
      constexpr int MAXM = (kQfTieCap + NT - 1) / NT;
      int rk[MAXM];
      uint32_t oi[MAXM], rr[MAXM];
#pragma unroll
      for (int t = 0; t < MAXM; t++) {
        rk[t] = -1; oi[t] = 0; rr[t] = 0;
        const int e = tid + NT * t;
        if (e < m) {
          oi[t] = tKey[s * slotCap + e];
          rr[t] = tVal[s * slotCap + e];
          int r = 0;
          for (int u = 0; u < m; u++) r += tKey[s * slotCap + u] < oi[t];
          rk[t] = r;
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < MAXM; t++)
        if (rk[t] >= 0) { tKey[s * slotCap + rk[t]] = (rr[t] << 16) | (uint32_t)rk[t]; tVal[s * slotCap + rk[t]] = oi[t]; }
      __syncthreads();
    }
    if (tid < nTied) {
      const int m = sTieCnt[tid];
      const int best = qt_sort_front(tKey + tid * slotCap, m);
      outKp[sTieNode[tid]] = fetch((int)tVal[tid * slotCap + best]);
    }
  }
  if (tid == 0) { needPass[unit] = 0; lvlCount[unit] = nOut; }
}

// ------------------------------------------------------------------------------------------------
// cv::GaussianBlur(level, 7x7, sigma 2, BORDER_REFLECT_101) in OpenCV's 8.8 fixed-point form
// (orbExtractor.cpp:385-386).  The pyramid's own 19-px reflect-101 pad supplies the border.
// 64x32 output tile per workgroup: LDS-staged source, separable: 16-bit horizontal sums (v_dot4_u32_u8), 32-bit vertical
// (v_dot2_u32_u16) — the same integers as the scalar form, (sum + 32768) >> 16.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, size_t pyrFrameStride, uint8_t* __restrict__ blur,
                                              size_t blurFrameStride, PlanDev P) {
  constexpr int TW = kBlurTW, TH = kBlurTH, SH = TH + 6, SWW = 18;  // source tile: 38 rows x 18 dwords (72 B: cols x0-3 .. x0+68)
  __shared__ uint32_t src[SH * SWW];
  __shared__ __align__(16) uint32_t hb[(SH / 2) * TW];   // horizontal sums, rows paired: (row 2p | row 2p+1 << 16)
  // grid.x runs over the tiles of ALL levels of a frame (a per-level grid launched 60 % empty workgroups: the kernel's waves
  // are ~50 instructions long, so wave launch rate matters)
  int f = blockIdx.y, tileId = blockIdx.x;
  if ((f | 7) < (int)gridDim.y) {   // XCD-aware (see k_fast_cells): XCD x handles every tile of frame 8*(f/8) + x
    tileId = (f & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    f = (f & ~7) | (blockIdx.x & 7);
  }
  if (tileId >= P.blurTileBegin[P.nLevels]) return;
  int level = 0;
#pragma unroll
  for (int l = 1; l < kMaxLevels; l++)
    if (l < P.nLevels && tileId >= P.blurTileBegin[l]) level = l;
  tileId -= P.blurTileBegin[level];
  const LevelDev L = P.lv[level];
  const int tilesX = (L.w + TW - 1) / TW;
  const int ty = tileId / tilesX, tx = tileId - ty * tilesX;
  const int x0 = tx * TW, y0 = ty * TH;
  // padded-row coordinates: level x <-> x + 19, so the tile's first source column x0-3 sits at byte x0+16: dword aligned
  // (pitch and padOff are multiples of 64), and the source loads are whole dwords.
  const uint8_t* padded = pyr + (size_t)f * pyrFrameStride + L.padOff;
  {  // 14 source rows of 18 dwords per pass (252 of the 256 threads): the row / dword split is computed once per thread
    const int r0 = threadIdx.x / SWW, wd = threadIdx.x - r0 * SWW;
    const int pcol = x0 + 16 + 4 * wd;
    const bool colOk = pcol < L.pitch;
    if (r0 < 14)
#pragma unroll
      for (int r = r0; r < SH; r += 14) {
        const int prow = min(y0 + r + kPad - 3, L.h + 2 * kPad - 1);
        src[r * SWW + wd] = colOk ? *reinterpret_cast<const uint32_t*>(padded + (size_t)prow * L.pitch + pcol) : 0u;
      }
  }
  __syncthreads();
  // horizontal pass, two source rows per thread: the 7 taps of an output are two v_dot4_u32_u8 over the byte windows
  // [k, k+4) and [k+4, k+8) of the row (v_alignbyte), and the two rows' results share a dword, (row 2p | row 2p+1 << 16), so
  // that the vertical pass can take two taps per v_dot2_u32_u16.  (The kernel is VALU-issue bound; this halves its VALU count.)
  for (int task = threadIdx.x; task < (SH / 2) * (TW / 4); task += 256) {
    const int rp = task >> 4, g = task & 15;
    uint32_t o[2][4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int r = 2 * rp + h;
      const uint32_t w0 = src[r * SWW + g], w1 = src[r * SWW + g + 1], w2 = src[r * SWW + g + 2];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t lo4 = k ? __builtin_amdgcn_alignbyte(w1, w0, k) : w0, hi4 = k ? __builtin_amdgcn_alignbyte(w2, w1, k) : w1;
        o[h][k] = __builtin_amdgcn_udot4(hi4, 0x00122230u, __builtin_amdgcn_udot4(lo4, 0x38302212u, 0u, false), false);   // 18 34 48 56 | 48 34 18
      }
    }
    *reinterpret_cast<uint4*>(&hb[rp * TW + g * 4]) =
        make_uint4(o[0][0] | (o[1][0] << 16), o[0][1] | (o[1][1] << 16), o[0][2] | (o[1][2] << 16), o[0][3] | (o[1][3] << 16));
  }
  __syncthreads();
  const int c4 = (threadIdx.x & 15) * 4;
#pragma unroll
  for (int r = threadIdx.x >> 4; r < TH; r += 16) {
    const int gy = y0 + r;
    if (!(gy < L.h && x0 + c4 < L.blurPitch)) continue;
    // rows r .. r+6 live in row pairs r/2 .. r/2+3; an odd r starts in the upper half of its first pair
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const bool odd = r & 1;
    const u16x2 k0 = odd ? u16x2{0, 18} : u16x2{18, 34}, k1 = odd ? u16x2{34, 48} : u16x2{48, 56};
    const u16x2 k2 = odd ? u16x2{56, 48} : u16x2{48, 34}, k3 = odd ? u16x2{34, 18} : u16x2{18, 0};
    const uint32_t* hp = &hb[(r >> 1) * TW + c4];
    const uint4 p0 = *reinterpret_cast<const uint4*>(hp), p1 = *reinterpret_cast<const uint4*>(hp + TW);
    const uint4 p2 = *reinterpret_cast<const uint4*>(hp + 2 * TW), p3 = *reinterpret_cast<const uint4*>(hp + 3 * TW);
    const uint32_t q0[4] = {p0.x, p0.y, p0.z, p0.w}, q1[4] = {p1.x, p1.y, p1.z, p1.w}, q2[4] = {p2.x, p2.y, p2.z, p2.w}, q3[4] = {p3.x, p3.y, p3.z, p3.w};
    uint32_t acc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      acc[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, q0[k]), k0, 32768u, false);
      acc[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, q1[k]), k1, acc[k], false);
      acc[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, q2[k]), k2, acc[k], false);
      acc[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, q3[k]), k3, acc[k], false);
    }
    const uint32_t out = (acc[0] >> 16) | ((acc[1] >> 16) << 8) | ((acc[2] >> 16) << 16) | ((acc[3] >> 16) << 24);
    *reinterpret_cast<uint32_t*>(blur + (size_t)f * blurFrameStride + L.blurOff + (size_t)gy * L.blurPitch + x0 + c4) = out;
  }
}

// ------------------------------------------------------------------------------------------------
// cv::fastAtan2 (float polynomial, degrees) — call site orbExtractor.cpp:419.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// Deterministic float cos/sin: fixed sequence of IEEE double operations (Cody-Waite by pi/2 + minimax
// kernels), rounded once to float.  The reference calls libm cosf/sinf (orbExtractor.cpp:424-425), which
// is platform code; DESIGN.md "steering trig" explains the contract.  The oracle restates the same
// sequence independently in oracle/oracle_trig.h.
__device__ __forceinline__ void sincos_det(float xf, float* sinOut, float* cosOut) {
  const double x = (double)xf;
  const double n = rint(__dmul_rn(x, 6.36619772367581382433e-01));
  const double r = __dsub_rn(__dsub_rn(x, __dmul_rn(n, 1.57079632673412561417e+00)), __dmul_rn(n, 6.07710050650619224932e-11));
  const double z = __dmul_rn(r, r);
  double p = __dmul_rn(z, 1.58969099521155010221e-10);
  p = __dadd_rn(p, -2.50507602534068634195e-08); p = __dmul_rn(p, z);
  p = __dadd_rn(p, 2.75573137070700676789e-06);  p = __dmul_rn(p, z);
  p = __dadd_rn(p, -1.98412698298579493134e-04); p = __dmul_rn(p, z);
  p = __dadd_rn(p, 8.33333333332248946124e-03);  p = __dmul_rn(p, z);
  p = __dadd_rn(p, -1.66666666666666324348e-01); p = __dmul_rn(p, z);
  p = __dmul_rn(p, r); p = __dadd_rn(p, r);
  double q = __dmul_rn(z, -1.13596475577881948265e-11);
  q = __dadd_rn(q, 2.08757232129817482790e-09);  q = __dmul_rn(q, z);
  q = __dadd_rn(q, -2.75573143513906633035e-07); q = __dmul_rn(q, z);
  q = __dadd_rn(q, 2.48015872894767294178e-05);  q = __dmul_rn(q, z);
  q = __dadd_rn(q, -1.38888888888741095749e-03); q = __dmul_rn(q, z);
  q = __dadd_rn(q, 4.16666666666666019037e-02);  q = __dmul_rn(q, z);
  q = __dmul_rn(q, z);
  double h = __dmul_rn(z, 0.5);
  h = __dsub_rn(h, q);
  h = __dsub_rn(1.0, h);
  const int quad = ((int)n) & 3;
  const double sv = quad == 0 ? p : quad == 1 ? h : quad == 2 ? -p : -h;
  const double cv = quad == 0 ? h : quad == 1 ? -p : quad == 2 ? -h : p;
  *sinOut = (float)sv;
  *cosOut = (float)cv;
}

__device__ __align__(16) const float kPatternDev[1024] = {   // (x0, y0, x1, y1) per test, already as float: the kernel is VALU-bound
#include "orb_pattern_data.inc"
};

// ------------------------------------------------------------------------------------------------
// One wave per keypoint: intensity-centroid orientation (orbExtractor.cpp:400-421, with the reference's
// own m_v_maxXcords table), then steered rBRIEF on the blurred level (:422-454) and the final
// cv::KeyPoint (:391-396, :595-601).  Lane l evaluates tests l, 64+l, 128+l, 192+l; a 64-bit ballot of
// (t0 < t1) is exactly 8 consecutive descriptor bytes.
// Everything that is the same for the whole wave (slot, level, keypoint, patch origin) is kept in scalar registers, and the
// pixel loads use a scalar base + non-negative 32-bit lane offset (the base is moved up-left of the patch).
// ------------------------------------------------------------------------------------------------
struct YdKeyPointDev { float x, y, size, angle, response; int octave, class_id; };

__global__ __launch_bounds__(256) void k_orient_describe(const uint8_t* __restrict__ pyr, size_t pyrFrameStride,
                                                         const uint8_t* __restrict__ blur, size_t blurFrameStride, PlanDev P,
                                                         const uint32_t* __restrict__ lvlKp, const int* __restrict__ lvlCount,
                                                         YdKeyPointDev* __restrict__ kps, uint8_t* __restrict__ desc, int cap,
                                                         int* __restrict__ nOut, float* __restrict__ lvlAngle) {
  const int lane = threadIdx.x & 63;
  int bx = blockIdx.x, f = blockIdx.y;
  if ((f | 7) < (int)gridDim.y) {   // XCD-aware (see k_fast_cells): every keypoint of a frame reads its patches through one L2
    bx = (f & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    f = (f & ~7) | (blockIdx.x & 7);
  }
  const int slot = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (slot >= P.sumQuota) return;
  // the lane's four test point pairs do not depend on the keypoint: fetch them first, under the keypoint look-up
  float4 pat[4];
#pragma unroll
  for (int t = 0; t < 4; t++) pat[t] = reinterpret_cast<const float4*>(kPatternDev)[t * 64 + lane];
  // slot -> level needs no memory (kpOff is plan data); the packed keypoint is fetched together with the level counts instead of
  // after them (one dependent round trip less; the slot is inside the array whether or not it holds a keypoint)
  int level = 0;
#pragma unroll
  for (int l = 1; l < kMaxLevels; l++)
    if (l < P.nLevels && slot >= P.lv[l].kpOff) level = l;
  const uint32_t pk = __builtin_amdgcn_readfirstlane(lvlKp[(size_t)f * P.sumQuota + slot]);
  int before = 0, tot = 0, mine = 0;
#pragma unroll
  for (int l = 0; l < kMaxLevels; l++) {
    const int c = l < P.nLevels ? lvlCount[f * kMaxLevels + l] : 0;
    if (l < level) before += c;
    if (l == level) mine = c;
    tot += c;
  }
  if (slot == 0 && lane == 0) nOut[f] = min(tot, cap);
  const LevelDev L = P.lv[level];
  const int k = slot - L.kpOff;
  if (k >= mine) return;
  const int outIdx = before + k;
  if (outIdx >= cap) return;
  const int kx = qt_x(pk) + kBorder, ky = qt_y(pk) + kBorder;
  const uint8_t* roi = pyr + (size_t)f * pyrFrameStride + L.padOff + (size_t)kPad * L.pitch + kPad;
  // --- orientation.  The reference's m_v_maxXcords is 0 for most rows (only the centre column counts there) and up to 26 for
  // the rest, so: one lane per single-column row (all of them in one step), then one step per wide row with u = lane - 32.
  const uint8_t* org = roi + (ptrdiff_t)(ky - 15) * L.pitch + (kx - 32);   // row ky-15, column kx-32: every offset below is >= 0
  const int u = lane - 32;
  unsigned narrow = 0;                     // bit v: row v has maxX == 0 (wave-uniform, scalar)
#pragma unroll
  for (int v = 1; v <= 15; v++) narrow |= (unsigned)(P.maxX[v] == 0) << v;
  int m10 = 0, m01 = 0;
  if (u >= -15 && u <= 15) m10 = u * org[(unsigned)(15 * L.pitch + lane)];
  if (lane >= 1 && lane <= 15 && ((narrow >> lane) & 1u)) {
    const int pos = org[(unsigned)((15 + lane) * L.pitch + 32)], neg = org[(unsigned)((15 - lane) * L.pitch + 32)];
    m01 = lane * (pos - neg);              // u == 0: no m10 term
  }
#pragma unroll
  for (int v = 1; v <= 15; v++) {
    const int d = P.maxX[v];
    if (d > 0 && u >= -d && u <= d) {
      const int pos = org[(unsigned)((15 + v) * L.pitch + lane)], neg = org[(unsigned)((15 - v) * L.pitch + lane)];
      m01 += v * (pos - neg);
      m10 += u * (pos + neg);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    m10 += __shfl_xor(m10, o, 64);
    m01 += __shfl_xor(m01, o, 64);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // --- steered BRIEF ---
  const float rad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.0));
  float sinB, cosA;
  sincos_det(rad, &sinB, &cosA);
  constexpr int kBias = 24;   // |rotated pattern coordinate| <= 13 * sqrt(2) + 0.5 < 24 (and the reference keeps 19 px of margin)
  const uint8_t* bl = blur + (size_t)f * blurFrameStride + L.blurOff + (ptrdiff_t)(ky - kBias) * L.blurPitch + (kx - kBias);
  unsigned long long words[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const float4 pt = pat[t];
    const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(pt.x, sinB), __fmul_rn(pt.y, cosA)));
    const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(pt.x, cosA), __fmul_rn(pt.y, sinB)));
    const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(pt.z, sinB), __fmul_rn(pt.w, cosA)));
    const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(pt.z, cosA), __fmul_rn(pt.w, sinB)));
    const int t0 = bl[(unsigned)((r0 + kBias) * L.blurPitch + c0 + kBias)], t1 = bl[(unsigned)((r1 + kBias) * L.blurPitch + c1 + kBias)];
    words[t] = __ballot(t0 < t1);
  }
  if (lane < 4) {
    unsigned long long w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
    reinterpret_cast<unsigned long long*>(desc + ((size_t)f * cap + outIdx) * 32)[lane] = w;
  }
  if (lane == 0) {
    YdKeyPointDev kp;
    kp.x = (float)kx;
    kp.y = (float)ky;
    if (level) { kp.x = __fmul_rn(kp.x, L.scale); kp.y = __fmul_rn(kp.y, L.scale); }
    kp.size = L.size;
    kp.angle = angle;
    kp.response = (float)qt_r(pk);
    kp.octave = level;
    kp.class_id = -1;
    kps[(size_t)f * cap + outIdx] = kp;
    lvlAngle[(size_t)f * P.sumQuota + slot] = angle;
  }
}


// The same kernel with KPW keypoints per wave, their phases interleaved: a keypoint is three dependent memory round trips (packed
// keypoint + level counts, the orientation patch, the rotated test points) with little arithmetic between them, and a CU holds at
// most 32 waves, so one keypoint per wave leaves the vector ALUs idle half of the time.  With KPW keypoints the loads of all of
// them are in flight together.  Arithmetic per keypoint is unchanged (same operations in the same order).
template <int KPW>
__global__ __launch_bounds__(256) void k_orient_describe_n(const uint8_t* __restrict__ pyr, size_t pyrFrameStride,
                                                           const uint8_t* __restrict__ blur, size_t blurFrameStride, PlanDev P,
                                                           const uint32_t* __restrict__ lvlKp, const int* __restrict__ lvlCount,
                                                           YdKeyPointDev* __restrict__ kps, uint8_t* __restrict__ desc, int cap,
                                                           int* __restrict__ nOut, float* __restrict__ lvlAngle) {
  const int lane = threadIdx.x & 63;
  int bx = blockIdx.x, f = blockIdx.y;
  if ((f | 7) < (int)gridDim.y) {   // XCD-aware (see k_fast_cells)
    bx = (f & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    f = (f & ~7) | (blockIdx.x & 7);
  }
  const int slot0 = (bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * KPW;
  if (slot0 >= P.sumQuota) return;
  float4 pat[4];
#pragma unroll
  for (int t = 0; t < 4; t++) pat[t] = reinterpret_cast<const float4*>(kPatternDev)[t * 64 + lane];
  int cnt[kMaxLevels], tot = 0;
#pragma unroll
  for (int l = 0; l < kMaxLevels; l++) { cnt[l] = l < P.nLevels ? lvlCount[f * kMaxLevels + l] : 0; tot += cnt[l]; }
  int level[KPW], outIdx[KPW];
  uint32_t pk[KPW];
  bool valid[KPW];
#pragma unroll
  for (int j = 0; j < KPW; j++) {
    const int slot = min(slot0 + j, P.sumQuota - 1);
    level[j] = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; l++)
      if (l < P.nLevels && slot >= P.lv[l].kpOff) level[j] = l;
    pk[j] = __builtin_amdgcn_readfirstlane(lvlKp[(size_t)f * P.sumQuota + slot]);
  }
  if (slot0 == 0 && lane == 0) nOut[f] = min(tot, cap);
  bool any = false;
#pragma unroll
  for (int j = 0; j < KPW; j++) {
    int before = 0, mine = 0;
#pragma unroll
    for (int l = 0; l < kMaxLevels; l++) {
      if (l < level[j]) before += cnt[l];
      if (l == level[j]) mine = cnt[l];
    }
    const int k = slot0 + j - P.lv[level[j]].kpOff;
    outIdx[j] = before + k;
    valid[j] = slot0 + j < P.sumQuota && k < mine && outIdx[j] < cap;
    any = any || valid[j];
  }
  if (!any) return;
  // a slot without a keypoint repeats a valid one of the wave (same addresses: no extra traffic) and stores nothing
  int firstValid = 0;
#pragma unroll
  for (int j = KPW - 1; j >= 0; j--) if (valid[j]) firstValid = j;
#pragma unroll
  for (int j = 0; j < KPW; j++)
    if (!valid[j]) {
#pragma unroll
      for (int i = 0; i < KPW; i++) if (i == firstValid) { pk[j] = pk[i]; level[j] = level[i]; }
    }
  int kx[KPW], ky[KPW], pitch[KPW];
  const uint8_t* org[KPW];
#pragma unroll
  for (int j = 0; j < KPW; j++) {
    const LevelDev& L = P.lv[level[j]];
    kx[j] = qt_x(pk[j]) + kBorder; ky[j] = qt_y(pk[j]) + kBorder;
    pitch[j] = L.pitch;
    const uint8_t* roi = pyr + (size_t)f * pyrFrameStride + L.padOff + (size_t)kPad * L.pitch + kPad;
    org[j] = roi + (ptrdiff_t)(ky[j] - 15) * L.pitch + (kx[j] - 32);
  }
  const int u = lane - 32;
  unsigned narrow = 0;
#pragma unroll
  for (int v = 1; v <= 15; v++) narrow |= (unsigned)(P.maxX[v] == 0) << v;
  int m10[KPW], m01[KPW];
#pragma unroll
  for (int j = 0; j < KPW; j++) {
    m10[j] = 0; m01[j] = 0;
    if (u >= -15 && u <= 15) m10[j] = u * org[j][(unsigned)(15 * pitch[j] + lane)];
    if (lane >= 1 && lane <= 15 && ((narrow >> lane) & 1u)) {
      const int pos = org[j][(unsigned)((15 + lane) * pitch[j] + 32)], neg = org[j][(unsigned)((15 - lane) * pitch[j] + 32)];
      m01[j] = lane * (pos - neg);
    }
  }
#pragma unroll
  for (int v = 1; v <= 15; v++) {
    const int d = P.maxX[v];
    if (d > 0 && u >= -d && u <= d) {
#pragma unroll
      for (int j = 0; j < KPW; j++) {
        const int pos = org[j][(unsigned)((15 + v) * pitch[j] + lane)], neg = org[j][(unsigned)((15 - v) * pitch[j] + lane)];
        m01[j] += v * (pos - neg);
        m10[j] += u * (pos + neg);
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
    for (int j = 0; j < KPW; j++) {
      m10[j] += __shfl_xor(m10[j], o, 64);
      m01[j] += __shfl_xor(m01[j], o, 64);
    }
  }
  constexpr int kBias = 24;
  float angle[KPW];
  int tA[KPW][4], tB[KPW][4];
#pragma unroll
  for (int j = 0; j < KPW; j++) {
    const LevelDev& L = P.lv[level[j]];
    angle[j] = fast_atan2_deg((float)m01[j], (float)m10[j]);
    const float rad = __fmul_rn(angle[j], (float)(3.14159265358979323846 / 180.0));
    float sinB, cosA;
    sincos_det(rad, &sinB, &cosA);
    const uint8_t* bl = blur + (size_t)f * blurFrameStride + L.blurOff + (ptrdiff_t)(ky[j] - kBias) * L.blurPitch + (kx[j] - kBias);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float4 pt = pat[t];
      const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(pt.x, sinB), __fmul_rn(pt.y, cosA)));
      const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(pt.x, cosA), __fmul_rn(pt.y, sinB)));
      const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(pt.z, sinB), __fmul_rn(pt.w, cosA)));
      const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(pt.z, cosA), __fmul_rn(pt.w, sinB)));
      tA[j][t] = bl[(unsigned)((r0 + kBias) * L.blurPitch + c0 + kBias)];
      tB[j][t] = bl[(unsigned)((r1 + kBias) * L.blurPitch + c1 + kBias)];
    }
  }
#pragma unroll
  for (int j = 0; j < KPW; j++) {
    unsigned long long words[4];
#pragma unroll
    for (int t = 0; t < 4; t++) words[t] = __ballot(tA[j][t] < tB[j][t]);
    if (!valid[j]) continue;
    const LevelDev& L = P.lv[level[j]];
    if (lane < 4) {
      unsigned long long w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
      reinterpret_cast<unsigned long long*>(desc + ((size_t)f * cap + outIdx[j]) * 32)[lane] = w;
    }
    if (lane == 0) {
      YdKeyPointDev kp;
      kp.x = (float)kx[j];
      kp.y = (float)ky[j];
      if (level[j]) { kp.x = __fmul_rn(kp.x, L.scale); kp.y = __fmul_rn(kp.y, L.scale); }
      kp.size = L.size;
      kp.angle = angle[j];
      kp.response = (float)qt_r(pk[j]);
      kp.octave = level[j];
      kp.class_id = -1;
      kps[(size_t)f * cap + outIdx[j]] = kp;
      lvlAngle[(size_t)f * P.sumQuota + slot0 + j] = angle[j];
    }
  }
}

}  // namespace ydorb
