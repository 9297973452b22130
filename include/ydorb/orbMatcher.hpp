// Drop-in adapter for YDORBSLAM::OrbMatcher (reference src/orbMatcher.hpp:24-66, src/orbMatcher.cpp:11-854) on top of the ydorb C ABI.
//
// The adapter is a set of templates over the reference's own Frame / KeyFrame / MapPoint types (included by the
// translation unit that instantiates them), so Tracking / LocalMapping / LoopClosing keep calling
// OrbMatcher::searchBy...() unchanged: each reference method body becomes a one-line forward, e.g.
//   int OrbMatcher::searchByProjectionInLastAndCurrentFrame(Frame& c, Frame& l, const float th)
//   { return ydorb::adapter::searchByProjectionInLastAndCurrentFrame(ydorb::adapter::matcher(), c, l, th, m_b_isToCheckOrientation); }
// What stays on the host is exactly the float geometry the reference computes per map point with cv::Mat (projection,
// radius, level window); the candidate search, Hamming distances, best/second-best, greedy assignment and the rotation
// histogram run on the GPU.  The same holds for the rest of the class: searchForTriangulation, fuseByProjection,
// fuseBySim3, searchBySim3, searchByProjectionInSim, and for MapPoint::computeDistinctiveDescriptors (distinctiveDescriptorIndices).
#ifndef YDORB_ADAPTER_ORBMATCHER_HPP
#define YDORB_ADAPTER_ORBMATCHER_HPP

#include <cmath>
#include <cstring>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "c_api.h"

namespace ydorb {
namespace adapter {

inline ydorb_matcher_t* matcher(int device = 0) {  // one matcher per calling thread (tracking / local mapping / loop closing)
  thread_local ydorb_matcher_t* m = nullptr;
  if (!m && ydorb_matcher_create(device, &m) != YDORB_OK) throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  return m;
}

// MapPoint::computeDistinctiveDescriptors(), src/mapPoint.cpp:191-213: index of the distinctive descriptor of every map point of
// a batch.  groups[p] = the rows the member function collects at :183-187 (1 x 32 CV_8U each, in that order); an empty group
// gives -1.  In src/mapPoint.cpp the member function keeps its gathering (:170-190) and its clone under the mutex (:214-217) and
// takes bestMedianIdx from distinctiveDescriptorIndices({vDescriptors})[0]; a LocalMapping / LoopClosing pass that refreshes many
// map points gathers them all and makes ONE call (a single point is far below launch latency).
inline std::vector<int> distinctiveDescriptorIndices(const std::vector<std::vector<cv::Mat>>& groups, ydorb_matcher_t* m = nullptr) {
  std::vector<int32_t> offsets(groups.size() + 1, 0);
  for (size_t p = 0; p < groups.size(); p++) offsets[p + 1] = offsets[p] + (int32_t)groups[p].size();
  std::vector<uint8_t> desc((size_t)offsets.back() * 32 + 32);
  size_t at = 0;
  for (const std::vector<cv::Mat>& g : groups)
    for (const cv::Mat& row : g) { std::memcpy(desc.data() + at, row.ptr<uint8_t>(), 32); at += 32; }
  std::vector<int> best(groups.size(), -1);
  if (!groups.empty() && ydorb_distinctive_descriptors(m ? m : matcher(), desc.data(), offsets.data(), (int32_t)groups.size(), best.data()) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  return best;
}

// static int OrbMatcher::computeDescriptorsDistance(const cv::Mat&, const cv::Mat&), src/orbMatcher.cpp:11-23
inline int computeDescriptorsDistance(const cv::Mat& a, const cv::Mat& b) { return ydorb_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

template <class FrameT>
inline YdFrameView frameView(const FrameT& f) {
  static_assert(sizeof(cv::KeyPoint) == sizeof(YdKeyPoint), "cv::KeyPoint layout");
  YdFrameView v;
  v.kps = reinterpret_cast<const YdKeyPoint*>(f.m_v_keyPoints.data());
  v.desc = f.m_cvMat_descriptors.template ptr<uint8_t>();
  v.right_x = f.m_v_rightXcords.empty() ? nullptr : f.m_v_rightXcords.data();
  v.n = (int)f.m_v_keyPoints.size();
  v.min_x = FrameT::m_flt_minX; v.max_x = FrameT::m_flt_maxX; v.min_y = FrameT::m_flt_minY; v.max_y = FrameT::m_flt_maxY;
  return v;
}

// taken[idx] / assigned[idx] <-> frame.m_v_sptrMapPoints[idx]
template <class FrameT>
inline void takenMask(const FrameT& f, bool anyPoint, std::vector<uint8_t>& taken) {
  taken.assign(f.m_v_sptrMapPoints.size(), 0);
  for (size_t i = 0; i < taken.size(); i++)
    if (f.m_v_sptrMapPoints[i] && (anyPoint || f.m_v_sptrMapPoints[i]->getObservationsNum() > 0)) taken[i] = 1;
}

// searchByProjectionInFrameAndMapPoint, src/orbMatcher.cpp:24-64
template <class FrameT, class MapPointPtr>
int searchByProjectionInFrameAndMapPoint(ydorb_matcher_t* m, FrameT& frame, const std::vector<MapPointPtr>& mps, float thd, float ratio) {
  const int nq = (int)mps.size();
  std::vector<YdQuery> q(nq);
  cv::Mat qd(std::max(nq, 1), 32, CV_8U);
  for (int i = 0; i < nq; i++) {
    YdQuery& Q = q[i];
    std::memset(&Q, 0, sizeof(Q));
    const MapPointPtr& mp = mps[i];
    if (!(mp->m_b_isTrackInView && !mp->isBad())) continue;
    const int lvl = mp->m_int_trackScaleLevel;
    const float radius = thd * (mp->m_flt_trackViewCos > 0.998 ? 2.5f : 4.0f);  // getRadiusByViewCos, :820-826
    Q.u = mp->m_flt_trackProjX; Q.v = mp->m_flt_trackProjY;
    Q.r = radius * frame.m_v_scaleFactors[lvl];
    Q.min_level = lvl - 1; Q.max_level = lvl;
    Q.ur = mp->m_flt_trackProjRightX; Q.rs = radius * frame.m_v_scaleFactors[lvl];
    Q.level = lvl;
    Q.flags = 1 | (mp->getObservationsNum() > 0 ? 2 : 0);
    mp->getDescriptor().copyTo(qd.row(i));
  }
  std::vector<uint8_t> taken;
  takenMask(frame, false, taken);
  std::vector<int32_t> assigned(taken.size(), -1);
  int32_t n = 0;
  YdFrameView fv = frameView(frame);
  if (ydorb_search_by_projection(m, YDORB_SEARCH_FRAME_MAPPOINT, &fv, q.data(), qd.data, nq, ratio, 0, 0, taken.data(), assigned.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t i = 0; i < assigned.size(); i++)
    if (assigned[i] >= 0) frame.m_v_sptrMapPoints[i] = mps[assigned[i]];
  return n;
}

// searchByProjectionInLastAndCurrentFrame, src/orbMatcher.cpp:65-155
template <class FrameT>
int searchByProjectionInLastAndCurrentFrame(ydorb_matcher_t* m, FrameT& cur, FrameT& last, float thd, bool checkOri) {
  const cv::Mat Rcw = cur.m_cvMat_T_c2w.rowRange(0, 3).colRange(0, 3), tcw = cur.m_cvMat_T_c2w.rowRange(0, 3).col(3);
  const cv::Mat twc = -Rcw.t() * tcw;
  const cv::Mat Rlw = last.m_cvMat_T_c2w.rowRange(0, 3).colRange(0, 3), tlw = last.m_cvMat_T_c2w.rowRange(0, 3).col(3);
  const cv::Mat tlc = Rlw * twc + tlw;
  const bool fwd = tlc.at<float>(2) > FrameT::m_flt_baseLine, bwd = -tlc.at<float>(2) > FrameT::m_flt_baseLine;
  const int nq = (int)last.m_v_sptrMapPoints.size();
  std::vector<YdQuery> q(nq);
  cv::Mat qd(std::max(nq, 1), 32, CV_8U);
  for (int i = 0; i < nq; i++) {
    YdQuery& Q = q[i];
    std::memset(&Q, 0, sizeof(Q));
    auto& mp = last.m_v_sptrMapPoints[i];
    if (!(mp && !last.m_v_isOutliers[i])) continue;
    const cv::Mat Xc = Rcw * mp->getPosInWorld() + tcw;
    const float x = Xc.at<float>(0), y = Xc.at<float>(1), z = Xc.at<float>(2);
    const float u = FrameT::m_flt_fx * x / z + FrameT::m_flt_cx, v = FrameT::m_flt_fy * y / z + FrameT::m_flt_cy;
    if (!(z >= 0.0 && cur.isInImage(u, v))) continue;
    const int oct = last.m_v_keyPoints[i].octave;
    Q.u = u; Q.v = v; Q.r = thd * cur.m_v_scaleFactors[oct];
    if (fwd) { Q.min_level = oct; Q.max_level = -1; }
    else if (bwd) { Q.min_level = 0; Q.max_level = oct; }
    else { Q.min_level = oct - 1; Q.max_level = oct + 1; }
    Q.ur = u - FrameT::m_flt_baseLineTimesFx / z; Q.rs = Q.r;
    Q.angle = last.m_v_keyPoints[i].angle; Q.level = oct;
    Q.flags = 1 | (mp->getObservationsNum() > 0 ? 2 : 0);
    mp->getDescriptor().copyTo(qd.row(i));
  }
  std::vector<uint8_t> taken;
  takenMask(cur, false, taken);
  std::vector<int32_t> assigned(taken.size(), -2);  // -2: untouched, -1: cleared by the rotation-histogram cull
  int32_t n = 0;
  YdFrameView fv = frameView(cur);
  if (ydorb_search_by_projection(m, YDORB_SEARCH_LAST_CURRENT, &fv, q.data(), qd.data, nq, 0.f, 0, checkOri ? 1 : 0, taken.data(), assigned.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t i = 0; i < assigned.size(); i++) {
    if (assigned[i] >= 0) cur.m_v_sptrMapPoints[i] = last.m_v_sptrMapPoints[assigned[i]];
    else if (assigned[i] == -1) cur.m_v_sptrMapPoints[i].reset();
  }
  return n;
}

// searchByProjectionInKeyFrameAndCurrentFrame, src/orbMatcher.cpp:156-239 (camera centre -R*t as written at :160)
template <class FrameT, class KeyFramePtr, class MapPointSet>
int searchByProjectionInKeyFrameAndCurrentFrame(ydorb_matcher_t* m, FrameT& cur, KeyFramePtr kf, const MapPointSet& found, float thd, int orbDist,
                                                bool checkOri) {
  const cv::Mat Rcw = cur.m_cvMat_T_c2w.rowRange(0, 3).colRange(0, 3), tcw = cur.m_cvMat_T_c2w.rowRange(0, 3).col(3);
  const cv::Mat Ow = -Rcw * tcw;
  const auto mps = kf->getMatchedMapPointsVec();
  const int nq = (int)mps.size();
  std::vector<YdQuery> q(nq);
  cv::Mat qd(std::max(nq, 1), 32, CV_8U);
  for (int i = 0; i < nq; i++) {
    YdQuery& Q = q[i];
    std::memset(&Q, 0, sizeof(Q));
    auto& mp = mps[i];
    if (!(mp && !mp->isBad() && !found.count(mp))) continue;
    const cv::Mat Xw = mp->getPosInWorld(), Xc = Rcw * Xw + tcw;
    const float x = Xc.at<float>(0), y = Xc.at<float>(1), z = Xc.at<float>(2);
    const float u = FrameT::m_flt_fx * x / z + FrameT::m_flt_cx, v = FrameT::m_flt_fy * y / z + FrameT::m_flt_cy;
    const float dist3D = (float)cv::norm(Xw - Ow);
    const int lvl = mp->predictScaleLevel(dist3D, cur);
    if (!(z >= 0.0 && cur.isInImage(u, v) && dist3D >= mp->getMinDistanceInvariance() && dist3D <= mp->getMaxDistanceInvariance())) continue;
    Q.u = u; Q.v = v; Q.r = thd * cur.m_v_scaleFactors[lvl];
    Q.min_level = lvl - 1; Q.max_level = lvl + 1;
    Q.angle = kf->m_v_keyPoints[i].angle; Q.level = lvl; Q.flags = 3;
    mp->getDescriptor().copyTo(qd.row(i));
  }
  std::vector<uint8_t> taken;
  takenMask(cur, true, taken);
  std::vector<int32_t> assigned(taken.size(), -2);
  int32_t n = 0;
  YdFrameView fv = frameView(cur);
  if (ydorb_search_by_projection(m, YDORB_SEARCH_KEYFRAME_CURRENT, &fv, q.data(), qd.data, nq, 0.f, orbDist, checkOri ? 1 : 0, taken.data(), assigned.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t i = 0; i < assigned.size(); i++) {
    if (assigned[i] >= 0) cur.m_v_sptrMapPoints[i] = mps[assigned[i]];
    else if (assigned[i] == -1) cur.m_v_sptrMapPoints[i].reset();
  }
  return n;
}

// DBoW3::FeatureVector (std::map<node, std::vector<unsigned>>) -> CSR
template <class FeatureVectorT>
struct BowCsr {
  std::vector<uint32_t> ids;
  std::vector<int32_t> start, feat;
  explicit BowCsr(const FeatureVectorT& fv) {
    start.push_back(0);
    for (const auto& kv : fv) {
      ids.push_back(kv.first);
      for (auto f : kv.second) feat.push_back((int32_t)f);
      start.push_back((int32_t)feat.size());
    }
  }
  YdFeatureVector c() const { return YdFeatureVector{ids.data(), start.data(), feat.data(), (int32_t)ids.size()}; }
};

// searchByBowInKeyFrameAndFrame, src/orbMatcher.cpp:303-379
template <class KeyFramePtr, class FrameT, class MapPointPtr>
int searchByBowInKeyFrameAndFrame(ydorb_matcher_t* m, KeyFramePtr kf, FrameT& frame, std::vector<MapPointPtr>& matched, float ratio, bool checkOri) {
  const auto mps = kf->getMatchedMapPointsVec();
  matched.assign(frame.m_int_keyPointsNum, MapPointPtr());
  std::vector<uint8_t> valid(mps.size());
  for (size_t i = 0; i < mps.size(); i++) valid[i] = mps[i] && !mps[i]->isBad();
  BowCsr<decltype(kf->m_bow_keyPointsVec)> a(kf->m_bow_keyPointsVec);
  BowCsr<decltype(frame.m_bow_keyPointsVec)> b(frame.m_bow_keyPointsVec);
  YdBowSide A{reinterpret_cast<const YdKeyPoint*>(kf->m_v_keyPoints.data()), kf->m_cvMat_descriptors.template ptr<uint8_t>(), valid.data(), (int32_t)mps.size(), a.c()};
  YdBowSide B{reinterpret_cast<const YdKeyPoint*>(frame.m_v_keyPoints.data()), frame.m_cvMat_descriptors.template ptr<uint8_t>(), nullptr,
              (int32_t)frame.m_v_keyPoints.size(), b.c()};
  std::vector<int32_t> out(B.n, -1);
  int32_t n = 0;
  if (ydorb_search_by_bow(m, YDORB_SEARCH_BOW_KEYFRAME_FRAME, &A, &B, ratio, checkOri ? 1 : 0, out.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t i = 0; i < out.size(); i++)
    if (out[i] >= 0) matched[i] = mps[out[i]];
  return n;
}

// searchByBowInTwoKeyFrames, src/orbMatcher.cpp:380-462
template <class KeyFramePtr, class MapPointPtr>
int searchByBowInTwoKeyFrames(ydorb_matcher_t* m, KeyFramePtr kf1, KeyFramePtr kf2, std::vector<MapPointPtr>& matched, float ratio, bool checkOri) {
  const auto mp1 = kf1->getMatchedMapPointsVec(), mp2 = kf2->getMatchedMapPointsVec();
  matched.assign(mp1.size(), MapPointPtr());
  std::vector<uint8_t> v1(mp1.size()), v2(mp2.size());
  for (size_t i = 0; i < mp1.size(); i++) v1[i] = mp1[i] && !mp1[i]->isBad();
  for (size_t i = 0; i < mp2.size(); i++) v2[i] = mp2[i] && !mp2[i]->isBad();
  BowCsr<decltype(kf1->m_bow_keyPointsVec)> a(kf1->m_bow_keyPointsVec), b(kf2->m_bow_keyPointsVec);
  YdBowSide A{reinterpret_cast<const YdKeyPoint*>(kf1->m_v_keyPoints.data()), kf1->m_cvMat_descriptors.template ptr<uint8_t>(), v1.data(), (int32_t)mp1.size(), a.c()};
  YdBowSide B{reinterpret_cast<const YdKeyPoint*>(kf2->m_v_keyPoints.data()), kf2->m_cvMat_descriptors.template ptr<uint8_t>(), v2.data(), (int32_t)mp2.size(), b.c()};
  std::vector<int32_t> out(A.n, -1);
  int32_t n = 0;
  if (ydorb_search_by_bow(m, YDORB_SEARCH_BOW_TWO_KEYFRAMES, &A, &B, ratio, checkOri ? 1 : 0, out.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t i = 0; i < out.size(); i++)
    if (out[i] >= 0) matched[i] = mp2[out[i]];
  return n;
}

// fuseByProjection, src/orbMatcher.cpp:682-745.  Pass 1 (host, the reference's own float cv::Mat arithmetic): projection, predicted
// level and the predicates of :688 / :704-708 for every map point.  Pass 2 (GPU): the window search with the level and chi-square tests.
// Pass 3 (host, list order): the replace / add bookkeeping of :726-737 on the search results.  A replacement can make a LATER list entry
// bad or put it into the keyframe, so those two predicates are evaluated again at each entry's turn, as the reference does implicitly;
// it cannot change a later entry's search result (the survivor of a replacement is either already in the keyframe or already processed).
template <class KeyFramePtr, class MapPointPtr, class FrameT>
int fuseByProjection(ydorb_matcher_t* m, KeyFramePtr kf, const std::vector<MapPointPtr>& mps, float th) {
  const cv::Mat Rcw = kf->getRotation_c2w(), tcw = kf->getTranslation_c2w(), Ow = kf->getCameraOriginInWorld();
  std::vector<YdQuery> q(mps.size());
  cv::Mat desc((int)mps.size() + 1, 32, CV_8U);   // rows of skipped entries are never read (flags 0)
  for (size_t i = 0; i < mps.size(); i++) {
    YdQuery& Q = q[i];
    Q = YdQuery{};
    const MapPointPtr& mp = mps[i];
    if (!mp || mp->isBad() || mp->isInKeyFrame(kf)) continue;
    const cv::Mat Pw = mp->getPosInWorld();
    const cv::Mat Pc = Rcw * Pw + tcw;
    const float xc = Pc.template at<float>(0), yc = Pc.template at<float>(1), zc = Pc.template at<float>(2);
    const float u = FrameT::m_flt_fx * xc / zc + FrameT::m_flt_cx, v = FrameT::m_flt_fy * yc / zc + FrameT::m_flt_cy;
    const float ur = u - FrameT::m_flt_baseLineTimesFx / zc;
    const cv::Mat PO = Pw - Ow;
    const float dist3D = (float)cv::norm(PO);
    const int level = mp->predictScaleLevel(dist3D, kf);
    Q.u = u; Q.v = v; Q.ur = ur; Q.level = level; Q.min_level = -1; Q.max_level = -1;
    Q.r = th * kf->m_v_scaleFactors[level];
    const bool ok = zc >= 0.0f && kf->isInImage(u, v) && dist3D >= mp->getMinDistanceInvariance() && dist3D <= mp->getMaxDistanceInvariance() &&
                    PO.dot(mp->getNormal()) >= 0.5 * dist3D;
    Q.flags = ok ? 1 : 0;
    mp->getDescriptor().copyTo(desc.row((int)i));
  }
  YdFrameView V = frameView(*kf);
  std::vector<int32_t> best(mps.size() + 1, -1);
  int32_t found = 0;
  if (ydorb_fuse_search(m, &V, q.data(), desc.data, (int32_t)mps.size(), kf->m_v_invScaleFactorSquares.data(),
                        (int32_t)kf->m_v_invScaleFactorSquares.size(), best.data(), &found) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  int fuseNum = 0;
  for (size_t i = 0; i < mps.size(); i++) {
    const MapPointPtr& mp = mps[i];
    if (best[i] < 0 || !(q[i].flags & 1) || mp->isBad() || mp->isInKeyFrame(kf)) continue;
    MapPointPtr inKF = kf->getMapPoint(best[i]);
    if (inKF) {
      if (!inKF->isBad() && inKF->getObservationsNum() > mp->getObservationsNum()) mp->beReplacedBy(inKF);
      else if (!inKF->isBad() && inKF->getObservationsNum() <= mp->getObservationsNum()) inKF->beReplacedBy(mp);
    } else {
      mp->addObservation(kf, best[i]);
      kf->addMapPoint(mp, best[i]);
    }
    fuseNum++;
  }
  return fuseNum;
}

// Sim3 decomposition shared by searchByProjectionInSim / fuseBySim3 (orbMatcher.cpp:241-247, :747-753), the reference's own cv::Mat arithmetic
struct SimProjection {
  cv::Mat R, t, Ow;
  explicit SimProjection(const cv::Mat& S) {
    const cv::Mat sR = S.rowRange(0, 3).colRange(0, 3);
    const float scale = (float)std::sqrt(sR.row(0).dot(sR.row(0)));
    R = sR / scale;
    t = S.rowRange(0, 3).col(3);
    Ow = -R.t() * t;
  }
  // query for one map point (:255-280 / :759-778); returns false when the predicates of :278-282 fail
  template <class KeyFramePtr, class MapPointPtr, class FrameT>
  bool query(const KeyFramePtr& kf, const MapPointPtr& mp, float th, YdQuery& Q) const {
    const cv::Mat Pw = mp->getPosInWorld();
    const cv::Mat Pc = R * Pw + t;
    const float xc = Pc.template at<float>(0), yc = Pc.template at<float>(1), zc = Pc.template at<float>(2);
    const float u = FrameT::m_flt_fx * xc / zc + FrameT::m_flt_cx, v = FrameT::m_flt_fy * yc / zc + FrameT::m_flt_cy;
    const cv::Mat PO = Pw - Ow;
    const float dist = (float)cv::norm(PO);
    const int level = mp->predictScaleLevel(dist, kf);
    Q = YdQuery{};
    Q.u = u; Q.v = v; Q.level = level; Q.min_level = -1; Q.max_level = -1; Q.r = th * kf->m_v_scaleFactors[level];
    return zc >= 0.0f && kf->isInImage(u, v) && dist >= mp->getMinDistanceInvariance() && dist <= mp->getMaxDistanceInvariance() &&
           PO.dot(mp->getNormal()) >= 0.5 * dist;
  }
};

// searchByProjectionInSim, src/orbMatcher.cpp:240-302 (loop closing): matched[idx] is filled for the keyframe features that receive a point
template <class KeyFramePtr, class MapPointPtr, class FrameT>
int searchByProjectionInSim(ydorb_matcher_t* m, KeyFramePtr kf, const cv::Mat& S, const std::vector<MapPointPtr>& mps, std::vector<MapPointPtr>& matched, int th) {
  const SimProjection P(S);
  std::set<MapPointPtr> found(matched.begin(), matched.end());
  found.erase(MapPointPtr());
  std::vector<YdQuery> q(mps.size());
  cv::Mat desc((int)mps.size() + 1, 32, CV_8U);
  for (size_t i = 0; i < mps.size(); i++) {
    q[i] = YdQuery{};
    if (!mps[i] || mps[i]->isBad() || found.count(mps[i])) continue;
    q[i].flags = P.template query<KeyFramePtr, MapPointPtr, FrameT>(kf, mps[i], (float)th, q[i]) ? 3 : 0;
    mps[i]->getDescriptor().copyTo(desc.row((int)i));
  }
  std::vector<uint8_t> taken(matched.size());
  for (size_t i = 0; i < matched.size(); i++) taken[i] = matched[i] ? 1 : 0;
  std::vector<int32_t> assigned(matched.size(), -1);
  YdFrameView V = frameView(*kf);
  int32_t n = 0;
  if (ydorb_search_by_projection(m, YDORB_SEARCH_SIM_PROJECTION, &V, q.data(), desc.data, (int32_t)mps.size(), 0.f, 0, 0, taken.data(), assigned.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t i = 0; i < assigned.size(); i++)
    if (assigned[i] >= 0) matched[i] = mps[assigned[i]];
  return n;
}

// fuseBySim3, src/orbMatcher.cpp:746-807: the same search as fuseByProjection without the chi-square test (a zero inverse-sigma table
// makes that test pass for every finite error), then the reference's bookkeeping in list order
template <class KeyFramePtr, class MapPointPtr, class FrameT>
int fuseBySim3(ydorb_matcher_t* m, KeyFramePtr kf, const cv::Mat& S, const std::vector<MapPointPtr>& mps, float th) {
  const SimProjection P(S);
  const std::set<MapPointPtr> found = kf->getMatchedMapPointsSet();
  std::vector<YdQuery> q(mps.size());
  cv::Mat desc((int)mps.size() + 1, 32, CV_8U);
  for (size_t i = 0; i < mps.size(); i++) {
    q[i] = YdQuery{};
    if (!mps[i] || mps[i]->isBad() || found.count(mps[i])) continue;
    q[i].flags = P.template query<KeyFramePtr, MapPointPtr, FrameT>(kf, mps[i], th, q[i]) ? 1 : 0;
    mps[i]->getDescriptor().copyTo(desc.row((int)i));
  }
  YdFrameView V = frameView(*kf);
  V.right_x = nullptr;   // every feature takes the monocular branch of the (disabled) error test
  const float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int32_t> best(mps.size() + 1, -1);
  int32_t n = 0;
  if (ydorb_fuse_search(m, &V, q.data(), desc.data, (int32_t)mps.size(), zero, 8, best.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  int fuseNum = 0;
  for (size_t i = 0; i < mps.size(); i++) {
    if (best[i] < 0 || !(q[i].flags & 1) || mps[i]->isBad()) continue;   // the found-set is a snapshot taken before the loop, as in the reference
    MapPointPtr inKF = kf->getMapPoint(best[i]);
    if (inKF) {
      if (!inKF->isBad()) mps[i]->beReplacedBy(inKF);
    } else {
      mps[i]->addObservation(kf, best[i]);
      kf->addMapPoint(mps[i], best[i]);
    }
    fuseNum++;
  }
  return fuseNum;
}

// searchBySim3, src/orbMatcher.cpp:566-681.  As written, the reference projects with the keyframes' own poses (the Sim3 products of :572-574
// are computed and never used), each direction is an independent window search (level window, best distance <= TH_HIGH, nothing taken),
// and a pair is kept when the two directions agree (:668-679).  Two ydorb_window_search calls + the agreement loop.
template <class KeyFramePtr, class MapPointPtr, class FrameT>
int searchBySim3(ydorb_matcher_t* m, KeyFramePtr kf1, KeyFramePtr kf2, std::vector<MapPointPtr>& matched12, float th) {
  const std::vector<MapPointPtr> mp1 = kf1->getMatchedMapPointsVec(), mp2 = kf2->getMatchedMapPointsVec();
  std::vector<uint8_t> done1(mp1.size(), 0), done2(mp2.size(), 0);
  for (size_t i = 0; i < matched12.size() && i < mp1.size(); i++)
    if (matched12[i]) {
      done1[i] = 1;
      const int idx = matched12[i]->getIdxInKeyFrame(kf2);
      if (idx >= 0 && idx < (int)mp2.size()) done2[idx] = 1;
    }
  auto direction = [&](const std::vector<MapPointPtr>& src, const std::vector<uint8_t>& done, const KeyFramePtr& dst, std::vector<int32_t>& best) {
    const cv::Mat R = dst->getRotation_c2w(), t = dst->getTranslation_c2w();
    std::vector<YdQuery> q(src.size());
    cv::Mat desc((int)src.size() + 1, 32, CV_8U);
    for (size_t i = 0; i < src.size(); i++) {
      q[i] = YdQuery{};
      if (!src[i] || done[i] || src[i]->isBad()) continue;
      const cv::Mat Pc = R * src[i]->getPosInWorld() + t;
      const float xc = Pc.template at<float>(0), yc = Pc.template at<float>(1), zc = Pc.template at<float>(2);
      const float u = FrameT::m_flt_fx * xc / zc + FrameT::m_flt_cx, v = FrameT::m_flt_fy * yc / zc + FrameT::m_flt_cy;
      const float dist = (float)cv::norm(Pc);
      const int level = src[i]->predictScaleLevel(dist, dst);
      q[i].u = u; q[i].v = v; q[i].level = level; q[i].min_level = -1; q[i].max_level = -1; q[i].r = th * dst->m_v_scaleFactors[level];
      q[i].flags = (zc >= 0.0f && dst->isInImage(u, v) && dist >= src[i]->getMinDistanceInvariance() && dist <= src[i]->getMaxDistanceInvariance()) ? 1 : 0;
      src[i]->getDescriptor().copyTo(desc.row((int)i));
    }
    YdFrameView V = frameView(*dst);
    V.right_x = nullptr;
    const float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    best.assign(src.size() + 1, -1);
    int32_t n = 0;
    if (ydorb_window_search(m, &V, q.data(), desc.data, (int32_t)src.size(), zero, 8, 100, best.data(), &n) != YDORB_OK)
      throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  };
  std::vector<int32_t> m12, m21;
  direction(mp1, done1, kf2, m12);
  direction(mp2, done2, kf1, m21);
  int found = 0;
  for (size_t i = 0; i < mp1.size(); i++) {
    const int i2 = m12[i];
    if (i2 >= 0 && m21[i2] == (int)i) { matched12[i] = mp2[i2]; found++; }
  }
  return found;
}

// searchForTriangulation, src/orbMatcher.cpp:463-565: the epipole (:465-470) and the stereo / map-point flags are gathered here exactly as
// the reference reads them; pairs come back in first-index order like the loop at :557-563.
template <class KeyFramePtr, class FrameT>
int searchForTriangulation(ydorb_matcher_t* m, KeyFramePtr kf1, KeyFramePtr kf2, const cv::Mat& F12, std::vector<std::pair<int, int>>& pairs,
                           bool stereoOnly, bool checkOri) {
  const cv::Mat Cw = kf1->getCameraOriginInWorld(), R2w = kf2->getRotation_c2w(), t2w = kf2->getTranslation_c2w();
  const cv::Mat C2 = R2w * Cw + t2w;
  const float ex = FrameT::m_flt_fx * C2.at<float>(0) / C2.at<float>(2) + FrameT::m_flt_cx;
  const float ey = FrameT::m_flt_fy * C2.at<float>(1) / C2.at<float>(2) + FrameT::m_flt_cy;
  const int n1 = kf1->m_int_keyPointsNum, n2 = kf2->m_int_keyPointsNum;
  std::vector<uint8_t> mp1(n1), mp2(n2);
  for (int i = 0; i < n1; i++) mp1[i] = kf1->getMapPoint(i) ? 1 : 0;
  for (int i = 0; i < n2; i++) mp2[i] = kf2->getMapPoint(i) ? 1 : 0;
  float Fm[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) Fm[r * 3 + c] = F12.at<float>(r, c);
  BowCsr<decltype(kf1->m_bow_keyPointsVec)> a(kf1->m_bow_keyPointsVec), b(kf2->m_bow_keyPointsVec);
  YdTriSide A{reinterpret_cast<const YdKeyPoint*>(kf1->m_v_keyPoints.data()), kf1->m_cvMat_descriptors.template ptr<uint8_t>(), kf1->m_v_rightXcords.data(),
              mp1.data(), (int32_t)n1, a.c()};
  YdTriSide B{reinterpret_cast<const YdKeyPoint*>(kf2->m_v_keyPoints.data()), kf2->m_cvMat_descriptors.template ptr<uint8_t>(), kf2->m_v_rightXcords.data(),
              mp2.data(), (int32_t)n2, b.c()};
  std::vector<int32_t> out(n1, -1);
  int32_t n = 0;
  if (ydorb_search_for_triangulation(m, &A, &B, Fm, ex, ey, kf2->m_v_scaleFactors.data(), kf2->m_v_scaleFactorSquares.data(),
                                     (int32_t)kf2->m_v_scaleFactors.size(), stereoOnly ? 1 : 0, checkOri ? 1 : 0, out.data(), &n) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  pairs.clear();
  pairs.reserve(n);
  for (int i = 0; i < n1; i++)
    if (out[i] >= 0) pairs.push_back(std::make_pair(i, (int)out[i]));
  return n;
}

}  // namespace adapter
}  // namespace ydorb
#endif
