// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of YDORBSLAM::OrbMatcher's search-by-projection / search-by-BoW family
// (reference src/orbMatcher.cpp) and of the Frame candidate generator it depends on
// (src/frame.cpp:249-264, 291-294, 327-361) on POD arrays.  The reference versions walk
// Frame / KeyFrame / MapPoint objects; here a "map point" is a query row (descriptor + projected
// position + flags) and `assigned[idx]` stands for frame.m_v_sptrMapPoints[idx] (query index or -1).
// The float geometry that produces the projected positions (cv::Mat 3x3 products) stays in the caller,
// exactly as the C ABI draws the boundary (include/ydorb/c_api.h, "matcher").
//
// PARITY STATUS: no reference test pins any matcher result and the reference cannot be
// built here (needs OpenCV + DBoW3): "parity unpinned" beyond the line-by-line restatement; M0 is
// cross-checked against a numpy popcount in tests/test_oracle_matcher.py.
#include <climits>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct KeyPoint {
  float x, y, size, angle, response;
  int octave, class_id;
};

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // orbMatcher.cpp:7-9
const int GRID_COLS = 64, GRID_ROWS = 48;                 // frame.hpp:137-138

// orbMatcher.cpp:11-23
int descriptorDistance(const uint8_t* a, const uint8_t* b) {
  const int32_t* pa = (const int32_t*)a;
  const int32_t* pb = (const int32_t*)b;
  int dist = 0;
  for (int i = 0; i < 8; i++, pa++, pb++) {
    unsigned int v = *pa ^ *pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

struct Frame {
  int n = 0;
  std::vector<KeyPoint> kps;
  std::vector<uint8_t> desc;
  std::vector<float> rightX;
  float minX, maxX, minY, maxY, gridWInv, gridHInv;
  std::vector<int> grid[GRID_COLS][GRID_ROWS];

  // frame.cpp:327-336 (rounds instead of flooring, and uses minX for y — both kept)
  bool locationInGrid(const KeyPoint& kp, int& lx, int& ly) const {
    lx = (int)roundf((kp.x - minX) * gridWInv);
    ly = (int)roundf((kp.y - minX) * gridHInv);
    return !(lx < 0 || lx >= GRID_COLS || ly < 0 || ly >= GRID_ROWS);
  }
  // frame.cpp:249-264
  void assignToGrid() {
    for (int i = 0; i < n; i++) {
      int lx, ly;
      if (locationInGrid(kps[i], lx, ly)) grid[lx][ly].push_back(i);
    }
  }
  // frame.cpp:337-361 (per-axis test `|dx| > r && |dy| < r` and the `octave < maxLevel` skip are as written)
  std::vector<int> keyPointsInArea(float x, float y, float r, int minLevel, int maxLevel) const {
    std::vector<int> out;
    const int minCellX = std::max(0, (int)floorf((x - minX - r) * gridWInv));
    const int maxCellX = std::min(GRID_COLS - 1, (int)ceilf((x - minX + r) * gridWInv));
    const int minCellY = std::max(0, (int)floorf((y - minY - r) * gridHInv));
    const int maxCellY = std::min(GRID_ROWS - 1, (int)ceilf((y - minY + r) * gridHInv));
    if (minCellX < GRID_COLS && maxCellX >= 0 && minCellY < GRID_ROWS && maxCellY >= 0) {
      for (int ix = minCellX; ix <= maxCellX; ix++)
        for (int iy = minCellY; iy <= maxCellY; iy++)
          for (int idx : grid[ix][iy]) {
            if (minLevel > 0 || maxLevel >= 0) {
              if (kps[idx].octave < minLevel || (maxLevel >= 0 && kps[idx].octave < maxLevel)) continue;
            }
            if (fabsf(kps[idx].x - x) > r && fabsf(kps[idx].y - y) < r) out.push_back(idx);
          }
    }
    return out;
  }
};

// orbMatcher.cpp:827-854
void threeMaxima(const std::vector<std::vector<int>>& hist, int L, int& i1, int& i2, int& i3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)hist[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
    else if (s > max3) { max3 = s; i3 = i; }
  }
  if (max2 < max1 / 10) { i3 = -1; i2 = -1; }
  else if (max3 < max1 / 10) { i3 = -1; }
}

// rotation-histogram bin, orbMatcher.cpp:121-130 (factor = 1/30, so only bins 0..12 are reachable)
int rotBin(float a1, float a2) {
  const float factor = 1.0 / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0;
  int bin = (int)round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// One projected query == one MapPoint seen by the search (all derived floats computed by the caller).
struct Query {
  float u, v;          // projected position
  float r;             // window radius handed to getKeyPointsInArea
  int minLevel, maxLevel;
  float ur;            // projected right-image x (stereo consistency)
  float rs;            // stereo tolerance
  float angle;         // keypoint angle of the source observation (rotation histogram)
  int level;           // predicted / source scale level (M1 acceptance compares levels of best & second)
  int flags;           // bit0 valid, bit1 the map point has observations > 0
};

}  // namespace

extern "C" {

int yo_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptorDistance(a, b); }

// MapPoint::computeDistinctiveDescriptors (reference src/mapPoint.cpp:191-213) for one map point: m descriptors in the order the
// member function collected them (:183-187).  Returns bestMedianIdx (0 when m == 0 never happens: the caller returns early, :188).
int yo_distinctive_descriptor(const uint8_t* desc, int m) {
  if (m <= 0) return -1;
  std::vector<float> distances((size_t)m * m);   // float distances[m][m], :193
  for (int i = 0; i < m; i++) {
    distances[(size_t)i * m + i] = 0;
    for (int j = i + 1; j < m; j++) {
      const int dij = descriptorDistance(desc + (size_t)i * 32, desc + (size_t)j * 32);
      distances[(size_t)i * m + j] = (float)dij;
      distances[(size_t)j * m + i] = (float)dij;
    }
  }
  int bestMedian = INT_MAX, bestMedianIdx = 0;   // :203-213
  for (int i = 0; i < m; i++) {
    std::vector<int> v(distances.begin() + (size_t)i * m, distances.begin() + (size_t)(i + 1) * m);
    std::sort(v.begin(), v.end());
    const int median = v[(size_t)(0.5 * m)];
    if (median < bestMedian) {
      bestMedian = median;
      bestMedianIdx = i;
    }
  }
  return bestMedianIdx;
}

void* yo_frame_create(const void* kps, int n, const uint8_t* desc, const float* rightX, float minX, float maxX, float minY,
                      float maxY) {
  Frame* f = new Frame();
  f->n = n;
  f->kps.assign((const KeyPoint*)kps, (const KeyPoint*)kps + n);
  f->desc.assign(desc, desc + (size_t)n * 32);
  if (rightX) f->rightX.assign(rightX, rightX + n);
  else f->rightX.assign(n, -1.f);
  f->minX = minX; f->maxX = maxX; f->minY = minY; f->maxY = maxY;
  f->gridWInv = static_cast<float>(GRID_COLS) / (maxX - minX);  // frame.cpp:99-100
  f->gridHInv = static_cast<float>(GRID_ROWS) / (maxY - minY);
  f->assignToGrid();
  return f;
}
void yo_frame_destroy(void* f) { delete (Frame*)f; }

int yo_frame_keypoints_in_area(void* fp, float x, float y, float r, int minLevel, int maxLevel, int* out, int cap) {
  std::vector<int> v = ((Frame*)fp)->keyPointsInArea(x, y, r, minLevel, maxLevel);
  memcpy(out, v.data(), sizeof(int) * std::min((int)v.size(), cap));
  return (int)v.size();
}

// mode 0: searchByProjectionInFrameAndMapPoint      orbMatcher.cpp:24-64   (best + second, level rule, <= TH_HIGH)
// mode 1: searchByProjectionInLastAndCurrentFrame   orbMatcher.cpp:65-155  (best only, < TH_HIGH, rotation histogram)
// mode 2: searchByProjectionInKeyFrameAndCurrentFrame  :156-239            (best only, <= orbDist, rotation histogram)
// mode 7: searchByProjectionInSim                      :240-302            (level window predicted-1..predicted, <= TH_LOW, no histogram)
// taken[idx] != 0  <=>  frame.m_v_sptrMapPoints[idx] holds a point with observations > 0 (modes 0,1) or any point (mode 2).
// assigned[idx]: query index written at :57 / :118 / :203 (or -1).  Returns matchNum.
int yo_search_by_projection(void* fp, int mode, const void* queries, const uint8_t* qdesc, int nq, float ratio, int orbDist,
                            int checkOrientation, uint8_t* taken, int* assigned) {
  Frame& F = *(Frame*)fp;
  const Query* Q = (const Query*)queries;
  int matchNum = 0;
  std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
  for (int q = 0; q < nq; q++) {
    if (!(Q[q].flags & 1)) continue;
    const std::vector<int> vIdx = F.keyPointsInArea(Q[q].u, Q[q].v, Q[q].r, Q[q].minLevel, Q[q].maxLevel);
    if (vIdx.empty()) continue;
    int bestDist = 256, bestLevel = -1, secondDist = 256, secondLevel = -1, bestIdx = -1;
    for (int idx : vIdx) {
      if (taken[idx]) continue;
      if (mode == 7) { if (!(F.kps[idx].octave >= Q[q].level - 1 && F.kps[idx].octave <= Q[q].level)) continue; }   // searchByProjectionInSim :283-285
      else if (mode != 2 && !(F.rightX[idx] <= 0 || fabsf(Q[q].ur - F.rightX[idx]) <= Q[q].rs)) continue;
      const int dist = descriptorDistance(qdesc + (size_t)q * 32, &F.desc[(size_t)idx * 32]);
      if (dist < bestDist) {
        secondDist = bestDist; bestDist = dist;
        secondLevel = bestLevel; bestLevel = F.kps[idx].octave;
        bestIdx = idx;
      } else if (dist < secondDist) {
        secondLevel = F.kps[idx].octave;
        secondDist = dist;
      }
    }
    bool accept;
    if (mode == 0) accept = bestDist <= TH_HIGH && (bestLevel != secondLevel || bestDist <= ratio * secondDist);
    else if (mode == 1) accept = bestDist < TH_HIGH;
    else if (mode == 7) accept = bestDist <= TH_LOW;   // :293
    else accept = bestDist <= orbDist;
    if (!accept) continue;
    assigned[bestIdx] = q;
    taken[bestIdx] = (mode == 2 || mode == 7) ? 1 : ((Q[q].flags & 2) ? 1 : 0);
    matchNum++;
    if (mode != 0 && mode != 7 && checkOrientation) rotHist[rotBin(Q[q].angle, F.kps[bestIdx].angle)].push_back(bestIdx);
  }
  if (mode != 0 && mode != 7 && checkOrientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != i1 && i != i2 && i != i3)
        for (int idx : rotHist[i]) { assigned[idx] = -1; matchNum--; }
  }
  return matchNum;
}

// The search half of OrbMatcher::fuseByProjection (orbMatcher.cpp:682-745; fuseBySim3 :746-807 uses the same test): for every map point
// that passed the caller-side predicates (:688, :704-708 — flags bit 0), the keyframe feature with the smallest descriptor distance
// among those in the projection window whose level is predicted-1 .. predicted (:715-716) and whose reprojection error passes the
// chi-square test (:711-718).  Query fields used: u, v (projection), r (radius), ur (projected right x), level (predicted level).
// best[q] = feature index when bestDist <= TH_LOW (:725), else -1.  The replace / add bookkeeping (:726-737) stays with the caller
// and runs in list order on these results.  invSigma2 = the keyframe's m_v_invScaleFactorSquares (all zeros: no chi-square test, which is
// fuseBySim3's search :746-807 and each direction of searchBySim3 :594-667, the latter with maxDist = TH_HIGH).
int yo_fuse_search(void* fp, const void* queries, const uint8_t* qdesc, int nq, const float* invSigma2, int maxDist, int* best) {
  Frame& F = *(Frame*)fp;
  const Query* Q = (const Query*)queries;
  int found = 0;
  for (int q = 0; q < nq; q++) {
    best[q] = -1;
    if (!(Q[q].flags & 1)) continue;
    const std::vector<int> vIdx = F.keyPointsInArea(Q[q].u, Q[q].v, Q[q].r, -1, -1);
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIdx) {
      const int level = F.kps[idx].octave;
      const float monoErr = pow(F.kps[idx].x - Q[q].u, 2.0) + pow(F.kps[idx].y - Q[q].v, 2.0);
      const float stereoErr = monoErr + pow(F.rightX[idx] - Q[q].ur, 2.0);
      if (level >= Q[q].level - 1 && level <= Q[q].level &&
          ((F.rightX[idx] >= 0 && stereoErr * invSigma2[level] <= 7.81) || (F.rightX[idx] < 0 && monoErr * invSigma2[level] <= 5.99))) {
        const int dist = descriptorDistance(qdesc + (size_t)q * 32, &F.desc[(size_t)idx * 32]);
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
      }
    }
    if (bestDist <= maxDist) { best[q] = bestIdx; found++; }   // TH_LOW for the fuse functions (:725, :789), TH_HIGH for searchBySim3 (:625, :662)
  }
  return found;
}

// DBoW3::FeatureVector as CSR: nodeIds ascending (std::map order), nodeStart[nNodes+1], feat[] in append order.
// mode 3: searchByBowInKeyFrameAndFrame  orbMatcher.cpp:303-379 — out[frameIdx] = keyframe feature index (stands for its MapPoint)
// mode 4: searchByBowInTwoKeyFrames      orbMatcher.cpp:380-462 — out[firstIdx] = second-keyframe feature index
// validA[i] != 0 <=> keyframe A's feature i has a good MapPoint; validB likewise (mode 4 only).
int yo_search_by_bow(int mode, const void* kpsA, const uint8_t* descA, int nA, const uint8_t* validA, const uint32_t* nodeIdsA,
                     const int* nodeStartA, int nNodesA, const int* featA, const void* kpsB, const uint8_t* descB, int nB,
                     const uint8_t* validB, const uint32_t* nodeIdsB, const int* nodeStartB, int nNodesB, const int* featB,
                     float ratio, int checkOrientation, int* out) {
  const KeyPoint* KA = (const KeyPoint*)kpsA;
  const KeyPoint* KB = (const KeyPoint*)kpsB;
  const int nOut = mode == 3 ? nB : nA;
  for (int i = 0; i < nOut; i++) out[i] = -1;
  std::vector<uint8_t> matchedB(nB, 0);
  std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
  int matchNum = 0;
  int a = 0, b = 0;
  while (a < nNodesA && b < nNodesB) {
    if (nodeIdsA[a] == nodeIdsB[b]) {
      for (int ia = nodeStartA[a]; ia < nodeStartA[a + 1]; ia++) {
        const int idxA = featA[ia];
        if (!validA[idxA]) continue;
        int bestDist = 256, secondDist = 256, bestIdx = -1;
        for (int ib = nodeStartB[b]; ib < nodeStartB[b + 1]; ib++) {
          const int idxB = featB[ib];
          if (matchedB[idxB]) continue;
          if (mode == 4 && !validB[idxB]) continue;
          const int dist = descriptorDistance(descA + (size_t)idxA * 32, descB + (size_t)idxB * 32);
          if (dist < bestDist) { secondDist = bestDist; bestDist = dist; bestIdx = idxB; }
          else if (dist < secondDist) secondDist = dist;
        }
        if (bestDist <= TH_LOW && static_cast<float>(bestDist) < ratio * static_cast<float>(secondDist)) {
          matchedB[bestIdx] = 1;
          if (mode == 3) out[bestIdx] = idxA; else out[idxA] = bestIdx;
          if (checkOrientation) rotHist[rotBin(KA[idxA].angle, KB[bestIdx].angle)].push_back(mode == 3 ? bestIdx : idxA);
          matchNum++;
        }
      }
      a++; b++;
    } else if (nodeIdsA[a] < nodeIdsB[b]) {
      a = (int)(std::lower_bound(nodeIdsA, nodeIdsA + nNodesA, nodeIdsB[b]) - nodeIdsA);
    } else {
      b = (int)(std::lower_bound(nodeIdsB, nodeIdsB + nNodesB, nodeIdsA[a]) - nodeIdsB);
    }
  }
  if (checkOrientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != i1 && i != i2 && i != i3)
        for (int idx : rotHist[i]) { out[idx] = -1; matchNum--; }
  }
  return matchNum;
}

// OrbMatcher::searchForTriangulation, orbMatcher.cpp:463-565 (+ isEpipolarLineDistCorrect :808-819): BoW-guided search between the
// still-unmatched features of two keyframes, kept when the second feature lies on the first one's epipolar line.
// hasMpA/B[i] != 0 <=> the feature already has a MapPoint; rightA/B = m_v_rightXcords (>= 0: stereo good).
// F: the 3x3 float matrix _fMatrix_first2second, row-major F[r*3+c] = at<float>(r,c).  (ex, ey): the first camera's centre projected
// into the second image (:465-470, computed by the caller from float cv::Mat products).  sfB / sf2B: the second keyframe's
// m_v_scaleFactors / m_v_scaleFactorSquares.  out[firstIdx] = second index or -1.  Returns matchNum.
int yo_search_for_triangulation(const void* kpsA, const uint8_t* descA, int nA, const uint8_t* hasMpA, const float* rightA,
                                const uint32_t* nodeIdsA, const int* nodeStartA, int nNodesA, const int* featA, const void* kpsB,
                                const uint8_t* descB, int nB, const uint8_t* hasMpB, const float* rightB, const uint32_t* nodeIdsB,
                                const int* nodeStartB, int nNodesB, const int* featB, const float* F, float ex, float ey, const float* sfB,
                                const float* sf2B, int stereoOnly, int checkOrientation, int* out) {
  const KeyPoint* KA = (const KeyPoint*)kpsA;
  const KeyPoint* KB = (const KeyPoint*)kpsB;
  for (int i = 0; i < nA; i++) out[i] = -1;
  std::vector<uint8_t> matchedB(nB, 0);
  std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
  int matchNum = 0, a = 0, b = 0;
  auto epipolarOk = [&](const KeyPoint& k1, const KeyPoint& k2) {   // :808-819, every operation a float operation as written
    const float la = F[0] * k1.x + F[3] * k1.y + F[6];
    const float lb = F[1] * k1.x + F[4] * k1.y + F[7];
    const float lc = F[2] * k1.x + F[5] * k1.y + F[8];
    const float den = la * la + lb * lb;
    if (den > 0) {
      const float num = la * k2.x + lb * k2.y + lc;
      return (double)((num * num) / (den * den)) < 3.841 * (double)sf2B[k2.octave];   // the squared denominator is the reference's
    }
    return false;
  };
  while (a < nNodesA && b < nNodesB) {
    if (nodeIdsA[a] == nodeIdsB[b]) {
      for (int ia = nodeStartA[a]; ia < nodeStartA[a + 1]; ia++) {
        const int i1 = featA[ia];
        const bool goodA = rightA[i1] >= 0;
        if (hasMpA[i1] || (stereoOnly && !goodA)) continue;
        int bestDist = TH_LOW, bestIdx = -1;
        for (int ib = nodeStartB[b]; ib < nodeStartB[b + 1]; ib++) {
          const int i2 = featB[ib];
          const bool goodB = rightB[i2] >= 0;
          if (matchedB[i2] || hasMpB[i2] || (stereoOnly && !goodB)) continue;
          const int dist = descriptorDistance(descA + (size_t)i1 * 32, descB + (size_t)i2 * 32);
          if (dist <= TH_LOW && dist <= bestDist &&
              (goodA || goodB ||
               pow((float)ex - (float)KB[i2].x, 2.0) + pow((float)ey - (float)KB[i2].y, 2.0) >= 100 * sfB[KB[i2].octave])) {
            if (epipolarOk(KA[i1], KB[i2])) { bestIdx = i2; bestDist = dist; }
          }
        }
        if (bestIdx >= 0) {
          matchedB[bestIdx] = 1;
          out[i1] = bestIdx;
          matchNum++;
          if (checkOrientation) rotHist[rotBin(KA[i1].angle, KB[bestIdx].angle)].push_back(i1);
        }
      }
      a++; b++;
    } else if (nodeIdsA[a] < nodeIdsB[b]) {
      a = (int)(std::lower_bound(nodeIdsA, nodeIdsA + nNodesA, nodeIdsB[b]) - nodeIdsA);
    } else {
      b = (int)(std::lower_bound(nodeIdsB, nodeIdsB + nNodesB, nodeIdsA[a]) - nodeIdsB);
    }
  }
  if (checkOrientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != i1 && i != i2 && i != i3)
        for (int idx : rotHist[i]) { out[idx] = -1; matchNum--; }
  }
  return matchNum;
}

// Brute-force top-2 over a candidate list: the best / second-best chain of the searches as written (orbMatcher.cpp:39-52; the same
// chain at :327-334, :404-417): both start at 256, `dist < best` shifts best into second, `else if dist < second` replaces second.
// out per query: best_dist, best_idx, second_dist, second_idx, best_rank, second_rank (rank = position in the list, -1 = none).
void yo_hamming_topk(const uint8_t* q, int nq, const uint8_t* t, int nt, const int* candOff, const int* candIdx, int* out) {
  for (int i = 0; i < nq; i++) {
    const int c0 = candOff ? candOff[i] : 0, c1 = candOff ? candOff[i + 1] : nt;
    int best = 256, second = 256, bestRank = -1, secondRank = -1;
    for (int r = 0; r < c1 - c0; r++) {
      const int idx = candIdx ? candIdx[c0 + r] : r;
      if (idx < 0 || idx >= nt) continue;
      const int d = descriptorDistance(q + (size_t)i * 32, t + (size_t)idx * 32);
      if (d < best) { second = best; secondRank = bestRank; best = d; bestRank = r; }
      else if (d < second) { second = d; secondRank = r; }
    }
    int* o = out + (size_t)i * 6;
    o[0] = best; o[1] = bestRank < 0 ? -1 : (candIdx ? candIdx[c0 + bestRank] : bestRank);
    o[2] = second; o[3] = secondRank < 0 ? -1 : (candIdx ? candIdx[c0 + secondRank] : secondRank);
    o[4] = bestRank; o[5] = secondRank;
  }
}

void yo_three_maxima(const int* sizes, int L, int* idx3) {
  std::vector<std::vector<int>> h(L);
  for (int i = 0; i < L; i++) h[i].assign(sizes[i], 0);
  idx3[0] = idx3[1] = idx3[2] = -1;
  threeMaxima(h, L, idx3[0], idx3[1], idx3[2]);
}
int yo_rot_bin(float a1, float a2) { return rotBin(a1, a2); }
}
