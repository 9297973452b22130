// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of YDORBSLAM::OrbExtractor (reference src/orbExtractor.cpp) on plain
// uint8_t images, including the OpenCV primitives it calls.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load this; the product library
// (ydorbslam_amd/csrc) never links or calls it.
//
// PARITY STATUS: the arithmetic of cv::FAST / cv::resize / cv::GaussianBlur / cv::fastAtan2 /
// cvRound lives in OpenCV, which is neither vendored under /root/reference nor installed
// (src/CMakeLists.txt:6 `find_package(OpenCV REQUIRED)`, version unpinned).  Those five
// primitives are restated here from the published OpenCV 4.x algorithms and pinned by
// known-answer tests written from their definitions (tests/test_oracle_primitives.py):
// "OpenCV-version parity unpinned".  Everything that orbExtractor.cpp itself specifies
// (cell grid, quad-tree, orientation table, rBRIEF, level scaling) follows the cited lines.
//
// Contract decision (SURVEY.md §7 hard part 2): first-call semantics.  The reference
// push_back()s onto m_v_imagePyramid without clearing (orbExtractor.cpp:612), so a second
// call on the same object re-extracts the first frame; the oracle builds a fresh pyramid
// per call.  `yo_extractor_set_stale_pyramid(h,1)` reproduces the stale behaviour for the
// known-deviation test.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <vector>

#include "oracle_trig.h"

namespace {

struct KeyPoint {  // cv::KeyPoint layout: 7 x 4 bytes
  float x, y, size, angle, response;
  int octave, class_id;
};

// cvRound: round-half-to-even under the default rounding mode.
inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvFloorF(float v) {
  int i = (int)v;
  return i - (i > v);
}

// A level image: pointer to the ROI origin inside a padded buffer (negative offsets legal).
struct Level {
  int w = 0, h = 0, stride = 0;
  std::vector<uint8_t> buf;  // (h+38) x stride
  uint8_t* roi() { return buf.data() + 19 * stride + 19; }
  const uint8_t* roi() const { return buf.data() + 19 * stride + 19; }
};

// ---------------------------------------------------------------------------------------
// cv::resize(src, dst, sz, 0, 0, INTER_LINEAR) for 8UC1 (call site orbExtractor.cpp:614).
// OpenCV 4.x imgproc/resize.cpp generic path: float source coordinate, 11-bit fixed-point
// coefficients (INTER_RESIZE_COEF_BITS), int32 horizontal pass, two-stage shifted vertical pass.
// ---------------------------------------------------------------------------------------
void resizeLinearU8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                    int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * dw), beta(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloorF(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    alpha[2 * dx] = (short)cvRoundF((1.f - fx) * 2048);
    alpha[2 * dx + 1] = (short)cvRoundF(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloorF(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = (short)cvRoundF((1.f - fy) * 2048);
    beta[2 * dy + 1] = (short)cvRoundF(fy * 2048);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hpass = [&](int sy, std::vector<int>& out) {
    sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);  // clip(sy, 0, ssize.height)
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int s1 = sx + 1 < sw ? S[sx + 1] : 0;  // coefficient is 0 there (fx forced to 0)
      out[dx] = S[sx] * alpha[2 * dx] + s1 * alpha[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    hpass(yofs[dy], row0);
    hpass(yofs[dy] + 1, row1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; dx++)
      D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// cv::copyMakeBorder(..., 19,19,19,19, BORDER_REFLECT_101) around the ROI (orbExtractor.cpp:615,618).
void fillBorder101(Level& L) {
  uint8_t* r = L.roi();
  for (int y = -19; y < L.h + 19; y++) {
    int sy = reflect101(y, L.h);
    for (int x = -19; x < L.w + 19; x++) {
      if (y >= 0 && y < L.h && x >= 0 && x < L.w) continue;
      r[y * L.stride + x] = r[sy * L.stride + reflect101(x, L.w)];
    }
  }
}

// ---------------------------------------------------------------------------------------
// cv::FAST(img, kps, thr, true) == FAST-9/16 (call site orbExtractor.cpp:581,583).
// ---------------------------------------------------------------------------------------
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// OpenCV features2d/fast_score.cpp cornerScore<16>: min/max-over-arcs recurrence.
int cornerScore16(const int d[25], int threshold) {
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    a = std::min(a, d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, d[k + 4]);
    a = std::min(a, d[k + 5]);
    a = std::min(a, d[k + 6]);
    a = std::min(a, d[k + 7]);
    a = std::min(a, d[k + 8]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    b = std::max(b, d[k + 3]);
    b = std::max(b, d[k + 4]);
    b = std::max(b, d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, d[k + 6]);
    b = std::max(b, d[k + 7]);
    b = std::max(b, d[k + 8]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

// Segment test from the definition: >= 9 contiguous circle pixels all darker than v-thr or
// all brighter than v+thr (strict).
bool isCorner9(const int d[25], int thr) {
  int runD = 0, runB = 0;
  for (int k = 0; k < 25; k++) {
    runD = d[k] > thr ? runD + 1 : 0;   // v - p > thr  : p darker
    runB = d[k] < -thr ? runB + 1 : 0;  // p brighter
    if (runD >= 9 || runB >= 9) return true;
  }
  return false;
}

void fast9_16(const uint8_t* img, int stride, int w, int h, int thr, bool nms, std::vector<KeyPoint>& out) {
  out.clear();
  thr = std::min(std::max(thr, 0), 255);
  if (w < 7 || h < 7) return;
  std::vector<uint8_t> score((size_t)w * h, 0);
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const uint8_t* p = img + (size_t)y * stride + x;
      int v = p[0], d[25];
      for (int k = 0; k < 25; k++) d[k] = v - p[kCircle[k & 15][1] * stride + kCircle[k & 15][0]];
      if (!isCorner9(d, thr)) continue;
      score[(size_t)y * w + x] = (uint8_t)cornerScore16(d, thr);
    }
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int s = score[(size_t)y * w + x];
      if (!s) continue;
      if (nms) {
        const uint8_t* c = &score[(size_t)y * w + x];
        if (!(s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] && s > c[w - 1] &&
              s > c[w] && s > c[w + 1]))
          continue;
      }
      out.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
    }
}

// ---------------------------------------------------------------------------------------
// cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) for 8UC1 (orbExtractor.cpp:386).
// OpenCV 4.x bit-exact 8U path: 8.8 fixed-point kernel from getGaussianKernelBitExact +
// error-diffused quantisation = {18,34,48,56,48,34,18}/256; 16-bit horizontal, 32-bit vertical,
// (acc + 2^15) >> 16.  Border is reflect-101 of the UNPADDED clone (orbExtractor.cpp:385).
// ---------------------------------------------------------------------------------------
std::vector<int> gaussKernelFixed(int n, double sigma, int bits) {
  // float kernel exp(-x^2/(2 sigma^2)) normalised, then OpenCV's error-diffusion rounding.
  int n2 = n / 2;
  std::vector<double> k(n);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - n2;
    k[i] = std::exp(-0.5 * x * x / (sigma * sigma));
    sum += k[i];
  }
  for (double& v : k) v /= sum;
  std::vector<int> r(n);
  double err = 0;
  int s = 0;
  for (int i = 0; i < n2; i++) {
    double adj = k[i] * (1 << bits) + err;
    int v0 = (int)lrint(adj);
    err = adj - v0;
    r[i] = r[n - 1 - i] = v0;
    s += v0;
  }
  r[n2] = (1 << bits) - 2 * s;
  return r;
}

void gaussianBlur7x7s2(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  static const std::vector<int> K = gaussKernelFixed(7, 2.0, 8);
  std::vector<uint16_t> hbuf((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int acc = 0;
      for (int i = 0; i < 7; i++) acc += K[i] * src[(size_t)y * sstride + reflect101(x + i - 3, w)];
      hbuf[(size_t)y * w + x] = (uint16_t)acc;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int j = 0; j < 7; j++) acc += (uint32_t)K[j] * hbuf[(size_t)reflect101(y + j - 3, h) * w + x];
      uint32_t v = (acc + (1u << 15)) >> 16;
      dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

// cv::fastAtan2(y, x) (orbExtractor.cpp:419): OpenCV core/mathfuncs_core atan_f32, float ops.
float fastAtan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

static const int8_t kPattern[1024] = {
#include "orb_pattern_data.inc"
};

// ---------------------------------------------------------------------------------------
// QuadTreeNode (orbExtractor.hpp:21-29, orbExtractor.cpp:4-54)
// ---------------------------------------------------------------------------------------
struct P2i { int x = 0, y = 0; };
struct QNode {
  std::vector<KeyPoint> kps;
  P2i tl, tr, bl, br;
  bool indivisible = false;
  void divide(QNode& n1, QNode& n2, QNode& n3, QNode& n4) const {
    const int cx = (int)ceil(static_cast<float>(tr.x + tl.x) / 2.0);
    const int cy = (int)ceil(static_cast<float>(bl.y + tl.y) / 2.0);
    n1.tl = tl; n1.tr = {cx, tl.y}; n1.bl = {tl.x, cy}; n1.br = {cx, cy};
    n2.tl = n1.tr; n2.tr = tr; n2.bl = {cx, cy}; n2.br = {tr.x, cy};
    n3.tl = n1.bl; n3.tr = {cx, cy}; n3.bl = bl; n3.br = {cx, tl.y};  // sic: orbExtractor.cpp:27
    n4.tl = {cx, cy}; n4.tr = n2.br; n4.bl = n3.br; n4.br = br;      // inherits the line-27 value
    for (const KeyPoint& kp : kps) {
      if (kp.x < cx && kp.y < cy) n1.kps.push_back(kp);
      else if (kp.x < cx && kp.y >= cy) n3.kps.push_back(kp);
      else if (kp.x >= cx && kp.y < cy) n2.kps.push_back(kp);
      else n4.kps.push_back(kp);
    }
    n1.indivisible = n1.kps.size() == 1;
    n2.indivisible = n2.kps.size() == 1;
    n3.indivisible = n3.kps.size() == 1;
    n4.indivisible = n4.kps.size() == 1;
  }
};

struct Extractor {
  int nFeatures, nLevels, iniThr, minThr;
  float scaleFactor;
  std::vector<float> sf, sf2, isf, isf2;
  std::vector<int> perLevel, maxX;
  std::vector<Level> pyr;                       // fresh per call (contract) unless staleMode
  std::vector<std::vector<uint8_t>> blurred;    // per level, unpadded w*h
  std::vector<std::vector<KeyPoint>> cands;     // pre-quad-tree candidates (border-relative coords)
  std::vector<std::vector<KeyPoint>> levelKps;  // post-orientation, level coordinates
  bool libmTrig = false;                        // use libm cosf/sinf exactly as the reference build would
  bool staleMode = false;
  std::vector<Level> stalePyr;

  // orbExtractor.cpp:315-354
  Extractor(int n, float s, int L, int ini, int /*min*/)
      : nFeatures(n), nLevels(L), iniThr(ini), minThr(ini) /* sic :318 */, scaleFactor(s) {
    int per = (int)round(nFeatures * (1 - 1.0 / scaleFactor) / (1.0 - pow(1.0 / scaleFactor, nLevels)));
    int sum = 0;
    for (int i = 0; i < nLevels; i++) {
      sf.push_back((float)pow(scaleFactor, i));
      sf2.push_back((float)pow(scaleFactor, 2 * i));
      isf.push_back((float)pow(scaleFactor, -i));
      isf2.push_back((float)pow(scaleFactor, -2 * i));
      if (i < nLevels - 1) {
        perLevel.push_back(per);
        sum += per;
        per = (int)round((float)per / scaleFactor);
      } else {
        perLevel.push_back(std::max(nFeatures - sum, 0));
      }
    }
    const int half = 15;
    maxX.resize(half + 1);  // 16 zeros, then push_back below (sic :341-346)
    int maxY = (int)floor(half * sqrt(2.0) / 2.0 + 1.0);
    int minY = (int)ceil(half * sqrt(2.0) / 2.0);
    for (int v = 0; v <= maxY; v++) maxX.push_back((int)round(sqrt(pow(half, 2) + pow(v, 2))));
    for (int v = half, i = 0; v >= minY; v--) {
      while (maxX[i] == maxX[i + 1]) i++;
      maxX[v] = i;
      i++;
    }
  }

  // orbExtractor.cpp:605-621
  void computePyramid(const uint8_t* img, int w, int h, int stride) {
    pyr.assign(nLevels, Level());
    for (int l = 0; l < nLevels; l++) {
      float scale = isf[l];
      Level& L = pyr[l];
      L.w = cvRoundF((float)w * scale);
      L.h = cvRoundF((float)h * scale);
      L.stride = L.w + 38;
      L.buf.assign((size_t)(L.h + 38) * L.stride, 0);
      if (l) {
        resizeLinearU8(pyr[l - 1].roi(), pyr[l - 1].w, pyr[l - 1].h, pyr[l - 1].stride, L.roi(), L.w, L.h, L.stride);
      } else {
        for (int y = 0; y < h; y++) memcpy(L.roi() + (size_t)y * L.stride, img + (size_t)y * stride, w);
      }
      fillBorder101(L);
    }
  }

  // orbExtractor.cpp:455-544
  void distributeQuadTree(const std::vector<KeyPoint>& in, int minX, int maxXb, int minY, int maxYb, int desired,
                          std::vector<KeyPoint>& out) {
    std::list<QNode> nodes;
    QNode root;
    root.tl = {0, 0};
    root.tr = {maxXb - minX, 0};
    root.bl = {0, maxYb - minY};
    root.br = {maxXb - minX, maxYb - minY};
    root.kps = in;
    nodes.push_back(root);
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->kps.size() == 1) { it->indivisible = true; ++it; }
      else if (it->kps.empty()) it = nodes.erase(it);
      else ++it;
    }
    int last = 0;
    while ((int)nodes.size() > last && (int)nodes.size() < desired) {
      last = (int)nodes.size();
      // the `else` arm at :509-533 is unreachable (expandableNodesNum is 0 when tested at :483)
      for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->indivisible) { ++it; continue; }
        QNode c[4];
        it->divide(c[0], c[1], c[2], c[3]);
        for (QNode& n : c)
          if (!n.kps.empty()) {
            if (n.kps.size() == 1) n.indivisible = true;
            nodes.push_front(n);
          }
        it = nodes.erase(it);
      }
    }
    out.clear();
    for (const QNode& n : nodes) {
      std::vector<KeyPoint> v = n.kps;
      std::sort(v.begin(), v.end(), [](KeyPoint& a, KeyPoint& b) { return a.response > b.response; });
      out.push_back(v.front());
    }
    if ((int)out.size() > desired) out.resize(desired);
  }

  // orbExtractor.cpp:400-421
  void computeOrientation(const Level& L, std::vector<KeyPoint>& kps) {
    const uint8_t* img = L.roi();
    for (KeyPoint& kp : kps) {
      int m01 = 0, m10 = 0;
      const int cy = cvRoundF(kp.y), cx = cvRoundF(kp.x);
      for (int u = -15; u <= 15; u++) m10 += u * img[cy * L.stride + cx + u];
      for (int v = 1; v <= 15; v++) {
        int vSum = 0, d = maxX[v];
        for (int u = -d; u <= d; u++) {
          int pos = img[(cy + v) * L.stride + cx + u], neg = img[(cy - v) * L.stride + cx + u];
          vSum += pos - neg;
          m10 += u * (pos + neg);
        }
        m01 += v * vSum;
      }
      kp.angle = fastAtan2((float)m01, (float)m10);
    }
  }

  // orbExtractor.cpp:545-604
  void computeKeyPointsPyramid() {
    cands.assign(nLevels, {});
    levelKps.assign(nLevels, {});
    const std::vector<Level>& P = staleMode ? stalePyr : pyr;
    for (int l = 0; l < nLevels; l++) {
      const Level& L = P[l];
      const int minBX = 16, minBY = 16, maxBX = L.w - 16, maxBY = L.h - 16;
      const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
      const int colsNum = (int)(width / 30.0f), rowsNum = (int)(height / 30.0f);
      std::vector<KeyPoint>& toDistr = cands[l];
      if (colsNum > 0 && rowsNum > 0) {
        const int cellW = (int)ceil(width / colsNum), cellH = (int)ceil(height / rowsNum);
        for (int i = 0; i < rowsNum; i++) {
          const float iniY = (float)(minBY + i * cellH);
          float maxY = iniY + cellH + 6;
          if (iniY >= maxBY - 3) continue;
          if (maxY > maxBY) maxY = (float)maxBY;
          for (int j = 0; j < colsNum; j++) {
            const float iniX = (float)(minBX + j * cellW);
            float maxXc = iniX + cellW + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxXc > maxBX) maxXc = (float)maxBX;
            std::vector<KeyPoint> cell;
            const uint8_t* sub = L.roi() + (int)iniY * L.stride + (int)iniX;
            fast9_16(sub, L.stride, (int)maxXc - (int)iniX, (int)maxY - (int)iniY, iniThr, true, cell);
            if (cell.empty()) fast9_16(sub, L.stride, (int)maxXc - (int)iniX, (int)maxY - (int)iniY, minThr, true, cell);
            for (KeyPoint& kp : cell) {
              kp.x += j * cellW;
              kp.y += i * cellH;
              toDistr.push_back(kp);
            }
          }
        }
      }
      distributeQuadTree(toDistr, minBX, maxBX, minBY, maxBY, perLevel[l], levelKps[l]);
      const int scaledPatch = (int)(31 * sf[l]);
      for (KeyPoint& kp : levelKps[l]) {
        kp.x += minBX;
        kp.y += minBY;
        kp.octave = l;
        kp.size = (float)scaledPatch;
      }
      computeOrientation(L, levelKps[l]);
    }
  }

  // orbExtractor.cpp:422-454
  void computeDescriptors(const uint8_t* img, int stride, const std::vector<KeyPoint>& kps, uint8_t* desc) {
    const float deg2rad = (float)(M_PI / 180.0);
    for (size_t i = 0; i < kps.size(); i++) {
      float angle = kps[i].angle * deg2rad;
      float cosA, sinB;
      if (libmTrig) { cosA = cosf(angle); sinB = sinf(angle); }
      else { cosA = yd_trig::cosf_det(angle); sinB = yd_trig::sinf_det(angle); }
      const int ky = cvRoundF(kps[i].y), kx = cvRoundF(kps[i].x);
      auto tap = [&](int idx) -> int {
        const int px = kPattern[2 * idx], py = kPattern[2 * idx + 1];
        return img[(ky + cvRoundF(px * sinB + py * cosA)) * stride + kx + cvRoundF(px * cosA - py * sinB)];
      };
      for (int j = 0; j < 32; j++) {
        int val = 0;
        for (int b = 0; b < 8; b++) val |= (tap(j * 16 + 2 * b) < tap(j * 16 + 2 * b + 1)) << b;
        desc[i * 32 + j] = (uint8_t)val;
      }
    }
  }

  // orbExtractor.cpp:355-399
  int extract(const uint8_t* img, int w, int h, int stride, KeyPoint* outK, uint8_t* outD, int cap) {
    if (!img || w <= 0 || h <= 0) return 0;
    computePyramid(img, w, h, stride);
    if (staleMode && stalePyr.empty()) stalePyr = pyr;
    computeKeyPointsPyramid();
    const std::vector<Level>& P = staleMode ? stalePyr : pyr;
    blurred.assign(nLevels, {});
    int n = 0;
    for (int l = 0; l < nLevels; l++) {
      std::vector<KeyPoint>& kps = levelKps[l];
      if (kps.empty()) continue;
      const Level& L = P[l];
      blurred[l].assign((size_t)L.w * L.h, 0);
      gaussianBlur7x7s2(L.roi(), L.w, L.h, L.stride, blurred[l].data(), L.w);
      std::vector<uint8_t> d(kps.size() * 32);
      computeDescriptors(blurred[l].data(), L.w, kps, d.data());
      for (size_t i = 0; i < kps.size(); i++) {
        if (n >= cap) return -1;
        KeyPoint kp = kps[i];
        if (l) { kp.x *= sf[l]; kp.y *= sf[l]; }
        outK[n] = kp;
        memcpy(outD + (size_t)n * 32, d.data() + i * 32, 32);
        n++;
      }
    }
    return n;
  }
};

}  // namespace

extern "C" {
void* yo_extractor_create(int n, float s, int L, int ini, int mn) { return new Extractor(n, s, L, ini, mn); }
void yo_extractor_destroy(void* h) { delete (Extractor*)h; }
void yo_extractor_set_libm_trig(void* h, int on) { ((Extractor*)h)->libmTrig = on != 0; }
void yo_extractor_set_stale_pyramid(void* h, int on) { ((Extractor*)h)->staleMode = on != 0; }
int yo_extract(void* h, const uint8_t* img, int w, int hgt, int stride, void* kps, uint8_t* desc, int cap) {
  return ((Extractor*)h)->extract(img, w, hgt, stride, (KeyPoint*)kps, desc, cap);
}
void yo_extractor_tables(void* h, float* sf, float* isf, float* sf2, float* isf2, int* perLevel, int* maxX28) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nLevels; i++) {
    sf[i] = e->sf[i]; isf[i] = e->isf[i]; sf2[i] = e->sf2[i]; isf2[i] = e->isf2[i]; perLevel[i] = e->perLevel[i];
  }
  for (size_t i = 0; i < e->maxX.size() && i < 28; i++) maxX28[i] = e->maxX[i];
}
// stage access (valid after yo_extract)
int yo_level_dims(void* h, int l, int* w, int* hh, int* stride) {
  Extractor* e = (Extractor*)h;
  if (l < 0 || l >= (int)e->pyr.size()) return -1;
  *w = e->pyr[l].w; *hh = e->pyr[l].h; *stride = e->pyr[l].stride;
  return 0;
}
const uint8_t* yo_level_padded(void* h, int l) { return ((Extractor*)h)->pyr[l].buf.data(); }
const uint8_t* yo_level_blurred(void* h, int l) {
  Extractor* e = (Extractor*)h;
  return e->blurred[l].empty() ? nullptr : e->blurred[l].data();
}
int yo_level_candidates(void* h, int l, void* out, int cap) {
  Extractor* e = (Extractor*)h;
  int n = (int)e->cands[l].size();
  if (out) memcpy(out, e->cands[l].data(), sizeof(KeyPoint) * std::min(n, cap));
  return n;
}
int yo_level_keypoints(void* h, int l, void* out, int cap) {
  Extractor* e = (Extractor*)h;
  int n = (int)e->levelKps[l].size();
  if (out) memcpy(out, e->levelKps[l].data(), sizeof(KeyPoint) * std::min(n, cap));
  return n;
}
// primitive entry points for the known-answer tests
void yo_resize_linear_u8(const uint8_t* s, int sw, int sh, int ss, uint8_t* d, int dw, int dh, int ds) {
  resizeLinearU8(s, sw, sh, ss, d, dw, dh, ds);
}
int yo_fast9_16(const uint8_t* img, int stride, int w, int h, int thr, int nms, void* out, int cap) {
  std::vector<KeyPoint> v;
  fast9_16(img, stride, w, h, thr, nms != 0, v);
  memcpy(out, v.data(), sizeof(KeyPoint) * std::min((int)v.size(), cap));
  return (int)v.size();
}
int yo_corner_score16(const int* d25, int thr) { return cornerScore16(d25, thr); }
void yo_gaussian_blur_7x7_s2(const uint8_t* s, int w, int h, int ss, uint8_t* d, int ds) { gaussianBlur7x7s2(s, w, h, ss, d, ds); }
void yo_gauss_kernel_fixed(int n, double sigma, int bits, int* out) {
  std::vector<int> k = gaussKernelFixed(n, sigma, bits);
  memcpy(out, k.data(), sizeof(int) * n);
}
float yo_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int yo_cv_round(float v) { return cvRoundF(v); }
int yo_reflect101(int i, int n) { return reflect101(i, n); }
float yo_cosf_det(float x) { return yd_trig::cosf_det(x); }
float yo_sinf_det(float x) { return yd_trig::sinf_det(x); }
// Count inputs in [lo_bits, hi_bits] (float bit patterns) where the deterministic trig differs from libm.
long yo_trig_mismatch_count(uint32_t lo_bits, uint32_t hi_bits, uint32_t step, long* n_cos, long* n_sin) {
  long nc = 0, ns = 0, tot = 0;
  for (uint64_t b = lo_bits; b <= hi_bits; b += step) {
    uint32_t bb = (uint32_t)b;
    float x;
    memcpy(&x, &bb, 4);
    if (cosf(x) != yd_trig::cosf_det(x)) nc++;
    if (sinf(x) != yd_trig::sinf_det(x)) ns++;
    tot++;
  }
  *n_cos = nc; *n_sin = ns;
  return tot;
}
}
