// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into or called from the product library).
//
// CPU restatement of YDORBSLAM::Frame::computeStereoMatches (/root/reference/src/frame.cpp:362-477) on plain arrays.
// Parity unpinned: the reference holds no fixture or test for this function and its OpenCV dependency is absent
// here, so the restatement is anchored on the text of frame.cpp alone.
//
// Restated as written, including:
//  * the output / descriptor index `leftIdx` that only advances at the end of the outer if-body (:462), so every
//    `continue` (:415,:425,:437,:444) and every left keypoint whose row has no right keypoint (:389) leaves it behind
//    the keypoint being processed (flag bit0 set replaces that index by the keypoint's own, the evident intent);
//  * the SAD minimum kept in an `int` initialised to 256 (:419,:430-432): a column only "wins" below 256 and the
//    running minimum is truncated;
//  * the outlier loop that starts at the smallest distance and breaks on the first entry below the threshold
//    (:465-472), so it removes everything or nothing.
// Two situations are undefined in the reference and are given a defined outcome here and in the product (status bit
// per pair): a keypoint whose (int)pt.y lies outside [0, rows) (out-of-range vector index at :389) counts as a row
// without right keypoints, and a best right column closer than 10 px to the left image edge (negative colRange at
// :427, cv::Exception in the reference) is treated like the other window `continue`s.
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>
#include <algorithm>

extern "C" int yo_descriptor_distance(const uint8_t* a, const uint8_t* b);

namespace {
struct Kp {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
struct Level {
  const uint8_t* p;  // ROI origin
  int w, h, stride;
};

// cv::norm(leftPatch - leftCentre, rightPatch - rightCentre, NORM_L1) over an 11x11 window (:416-429): all terms are
// integers below 2^24, so the float result is exact in any summation order.
float sadL1(const Level& L, int lx, int ly, const Level& R, int rx, int ry) {
  const float lc = (float)L.p[(size_t)ly * L.stride + lx], rc = (float)R.p[(size_t)ry * R.stride + rx];
  double acc = 0.0;
  for (int dy = -5; dy <= 5; dy++)
    for (int dx = -5; dx <= 5; dx++) {
      const float a = (float)L.p[(size_t)(ly + dy) * L.stride + lx + dx] - lc;
      const float b = (float)R.p[(size_t)(ry + dy) * R.stride + rx + dx] - rc;
      acc += std::fabs((double)(a - b));
    }
  return (float)acc;
}
}  // namespace

extern "C" {

// levels: per side nLevels x {ptr,w,h,stride}; scale / invScale: m_v_scaleFactors / m_v_invScaleFactors.
// bf, b: m_flt_baseLineTimesFx, m_flt_baseLine.  rows: m_v_imagePyramid[0].rows.  flags bit0: index by keypoint.
// Outputs rightX / depth have nLeft entries; status bit0: out-of-range row seen, bit1: negative window seen.
// Returns the number of stereo measurements kept (entries of vDistIndices).
int yo_stereo_matches(const void* kpsLeft, const uint8_t* descLeft, int nLeft, const void* kpsRight, const uint8_t* descRight,
                      int nRight, const uint8_t* const* leftPtr, const uint8_t* const* rightPtr, const int* w, const int* h,
                      const int* strideLeft, const int* strideRight, int nLevels, const float* scale, const float* invScale,
                      float bf, float b, int flags, float* rightXOut, float* depthOut, int* status) {
  const Kp* KL = (const Kp*)kpsLeft;
  const Kp* KR = (const Kp*)kpsRight;
  const int rowsNum = h[0];
  int st = 0;
  for (int i = 0; i < nLeft; i++) {  // :363-364
    rightXOut[i] = -1.0f;
    depthOut[i] = -1.0f;
  }
  const int orbDistThd = (100 + 50) / 2;  // :365, OrbMatcher::m_int_highThd / m_int_lowThd (orbMatcher.cpp:7-8)
  std::vector<std::vector<int>> rowIdx(rowsNum);  // :368-379
  for (int j = 0; j < nRight; j++) {
    const float r = 2.0f * scale[KR[j].octave];
    const int lo = (int)std::max(std::floor(KR[j].y - r), 0.0f);
    const float hi = std::min(std::ceil(KR[j].y + r), (float)rowsNum - 1.0f);
    for (int iy = lo; iy <= hi; iy++) rowIdx[iy].push_back(j);
  }
  const float minD = 0.0f;        // :381
  const float maxD = bf / b;      // :382
  std::vector<std::pair<int, int>> distIdx;  // :384
  int leftIdx = 0;
  for (int k = 0; k < nLeft; k++) {
    const Kp& kp = KL[k];
    if (flags & 1) leftIdx = k;
    const int row = (int)kp.y;  // float -> index conversion at :389
    if (!(kp.y >= 0.0f) || row >= rowsNum) {
      st |= 1;
      continue;
    }
    if (rowIdx[row].empty() || !(kp.x >= minD)) continue;  // :389 (leftIdx not advanced)
    int bestDist = 256, bestRight = 0;  // :390-391
    for (int j : rowIdx[row]) {         // :394-402
      if (KR[j].octave >= kp.octave - 1 && KR[j].octave <= kp.octave + 1 && KR[j].x >= (kp.x - maxD) && KR[j].x <= (kp.x - minD)) {
        const int d = yo_descriptor_distance(descLeft + (size_t)leftIdx * 32, descRight + (size_t)j * 32);
        if (d < bestDist) {
          bestDist = d;
          bestRight = j;
        }
      }
    }
    if (bestDist < orbDistThd) {  // :406
      const int o = kp.octave;
      const float lsx = std::round(kp.x * invScale[o]);  // :408-410
      const float lsy = std::round(kp.y * invScale[o]);
      const float rsx = std::round(KR[bestRight].x * invScale[o]);
      const int W = 5, S = 5;
      if ((lsy - W) < 0 || (lsy + W + 1) >= h[o] || (lsx - W) < 0 || (lsx + W + 1) >= w[o]) continue;  // :414-416
      if ((rsx + S - W) < 0 || (rsx + S + W + 1) >= w[o]) continue;                                       // :424-426
      if ((rsx - S - W) < 0) {  // colRange(rsx-10, ...) raises in the reference
        st |= 2;
        continue;
      }
      const Level L{leftPtr[o], w[o], h[o], strideLeft[o]}, R{rightPtr[o], w[o], h[o], strideRight[o]};
      int sadBest = 256, bestCol = 0;  // :419-420
      float dists[2 * S + 1];
      for (int i = -S; i <= S; i++) {  // :427-436
        const float d = sadL1(L, (int)lsx, (int)lsy, R, (int)rsx + i, (int)lsy);
        if (d < sadBest) {
          sadBest = (int)d;
          bestCol = i;
        }
        dists[i + S] = d;
      }
      if (bestCol == -S || bestCol == S) continue;  // :437-439
      const float d1 = dists[S + bestCol - 1], d2 = dists[S + bestCol], d3 = dists[S + bestCol + 1];
      const float delta = (float)((double)(d1 - d3) / (2.0 * ((double)(d1 + d3) - 2.0 * (double)d2)));  // :441-443
      if (delta < -1 || delta > 1) continue;                                                                // :444-446
      float bestRightX = scale[o] * ((float)rsx + (float)delta + (float)bestCol);                         // :448
      float disparity = kp.x - bestRightX;
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;
          bestRightX = (float)((double)kp.x - 0.01);
        }
        depthOut[leftIdx] = bf / disparity;  // :455-457
        rightXOut[leftIdx] = bestRightX;
        distIdx.push_back(std::make_pair(sadBest, leftIdx));
      }
    }
    leftIdx++;  // :462
  }
  std::sort(distIdx.begin(), distIdx.end());  // :464-472
  for (const std::pair<int, int>& di : distIdx) {
    if (di.first >= 1.5 * 1.4 * distIdx[distIdx.size() / 2].first) {
      rightXOut[di.second] = -2;
      depthOut[di.second] = -2;
    } else {
      break;
    }
  }
  if (status) *status = st;
  return (int)distIdx.size();
}

}  // extern "C"
