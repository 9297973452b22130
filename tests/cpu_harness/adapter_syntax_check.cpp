// TEST-ONLY: type-checks the adapter headers (include/ydorb/*.hpp) against mock declarations of the reference's Frame /
// KeyFrame / MapPoint / Map members they touch (names as in reference src/frame.hpp, keyFrame.hpp, mapPoint.hpp, map.hpp).
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <vector>
#include <opencv2/core.hpp>
namespace cv { struct MatCopy { }; }
#include "../../include/ydorb/orbMatcher.hpp"
#ifdef YDORB_CHECK_OPTIMIZER
#include "../../include/ydorb/optimizer.hpp"
#endif

struct KeyFrame;
struct Frame;
struct MapPoint {
  bool m_b_isTrackInView; int m_int_trackScaleLevel; float m_flt_trackViewCos, m_flt_trackProjX, m_flt_trackProjY, m_flt_trackProjRightX;
  long int m_int_localBAForKeyFrameID, m_int_globalBAforKeyFrameID; cv::Mat m_cvMat_posGlobalBA;
  bool isBad(); int getObservationsNum(); cv::Mat getDescriptor(); cv::Mat getPosInWorld();
  float getMaxDistanceInvariance(); float getMinDistanceInvariance(); int predictScaleLevel(const float&, const Frame&);
  std::map<std::shared_ptr<KeyFrame>, int> getObservations(); void eraseObservation(std::shared_ptr<KeyFrame>);
  void setPosInWorld(const cv::Mat&); void updateNormalAndDepth();
  bool isInKeyFrame(std::shared_ptr<KeyFrame>); int getIdxInKeyFrame(std::shared_ptr<KeyFrame>); cv::Mat getNormal(); int predictScaleLevel(const float&, std::shared_ptr<KeyFrame>);
  void beReplacedBy(std::shared_ptr<MapPoint>); void addObservation(std::shared_ptr<KeyFrame>, int);
};
typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;
struct Frame {
  std::vector<cv::KeyPoint> m_v_keyPoints; cv::Mat m_cvMat_descriptors; std::vector<float> m_v_rightXcords, m_v_scaleFactors;
  std::vector<std::shared_ptr<MapPoint>> m_v_sptrMapPoints; std::vector<bool> m_v_isOutliers; cv::Mat m_cvMat_T_c2w;
  int m_int_keyPointsNum; FeatureVector m_bow_keyPointsVec; std::vector<float> m_v_invScaleFactorSquares;
  cv::Mat getCameraPoseByTransform_c2w(); void setCameraPoseByTransform_c2w(cv::Mat);
  static float m_flt_minX, m_flt_maxX, m_flt_minY, m_flt_maxY, m_flt_fx, m_flt_fy, m_flt_cx, m_flt_cy, m_flt_baseLine, m_flt_baseLineTimesFx;
  bool isInImage(const float&, const float&) const;
};
struct KeyFrame {
  std::vector<cv::KeyPoint> m_v_keyPoints; cv::Mat m_cvMat_descriptors; FeatureVector m_bow_keyPointsVec; std::vector<float> m_v_rightXcords, m_v_invScaleFactorSquares;
  long int m_int_keyFrameID, m_int_localBAForKeyFrameID, m_int_fixedBAForKeyFrameID, m_int_globalBAForKeyFrameID; cv::Mat m_cvMat_T_c2w_GlobalBA;
  int m_int_keyPointsNum; std::vector<float> m_v_scaleFactors, m_v_scaleFactorSquares;
  std::set<std::shared_ptr<MapPoint>> getMatchedMapPointsSet(); std::shared_ptr<MapPoint> getMapPoint(const int&); void addMapPoint(std::shared_ptr<MapPoint>, const int&); bool isInImage(const float&, const float&) const;
  static float m_flt_minX, m_flt_maxX, m_flt_minY, m_flt_maxY; cv::Mat getCameraOriginInWorld(); cv::Mat getRotation_c2w(); cv::Mat getTranslation_c2w();
  std::vector<std::shared_ptr<MapPoint>> getMatchedMapPointsVec(); std::vector<std::shared_ptr<KeyFrame>> getOrderedConnectedKeyFrames();
  bool isBad(); cv::Mat getCameraPoseByTransform_c2w(); void setCameraPoseByTransform_c2w(cv::Mat); void eraseMatchedMapPoint(std::shared_ptr<MapPoint>);
};
struct Map {
  std::mutex m_mutex_updateMap;
  std::vector<std::shared_ptr<KeyFrame>> getAllKeyFrames(); std::vector<std::shared_ptr<MapPoint>> getAllMapPoints();
};

namespace ya = ydorb::adapter;
int check(Frame& a, Frame& b, std::shared_ptr<KeyFrame> kf, std::shared_ptr<KeyFrame> kf2, std::vector<std::shared_ptr<MapPoint>>& mps,
          const std::set<std::shared_ptr<MapPoint>>& found) {
  int n = ya::searchByProjectionInFrameAndMapPoint(ya::matcher(), a, mps, 3.f, 0.8f);
  n += ya::searchByProjectionInLastAndCurrentFrame(ya::matcher(), a, b, 15.f, true);
  n += ya::searchByProjectionInKeyFrameAndCurrentFrame(ya::matcher(), a, kf, found, 10.f, 100, true);
  n += ya::searchByBowInKeyFrameAndFrame(ya::matcher(), kf, a, mps, 0.7f, true);
  n += ya::searchByBowInTwoKeyFrames(ya::matcher(), kf, kf2, mps, 0.75f, true);
  n += ya::computeDescriptorsDistance(a.m_cvMat_descriptors, b.m_cvMat_descriptors);
  std::vector<std::pair<int, int>> pairs;
  n += ya::distinctiveDescriptorIndices({{a.m_cvMat_descriptors.row(0), b.m_cvMat_descriptors.row(1)}})[0];
  n += ya::fuseByProjection<std::shared_ptr<KeyFrame>, std::shared_ptr<MapPoint>, Frame>(ya::matcher(), kf, mps, 3.0f);
  n += ya::fuseBySim3<std::shared_ptr<KeyFrame>, std::shared_ptr<MapPoint>, Frame>(ya::matcher(), kf, a.m_cvMat_T_c2w, mps, 4.0f);
  n += ya::searchBySim3<std::shared_ptr<KeyFrame>, std::shared_ptr<MapPoint>, Frame>(ya::matcher(), kf, kf2, mps, 7.5f);
  n += ya::searchByProjectionInSim<std::shared_ptr<KeyFrame>, std::shared_ptr<MapPoint>, Frame>(ya::matcher(), kf, a.m_cvMat_T_c2w, mps, mps, 10);
  n += ya::searchForTriangulation<std::shared_ptr<KeyFrame>, Frame>(ya::matcher(), kf, kf2, a.m_cvMat_T_c2w, pairs, false, true);
#ifdef YDORB_CHECK_OPTIMIZER
  bool stop = false;
  ya::localBundleAdjustImpl<std::shared_ptr<KeyFrame>, std::shared_ptr<Map>, Frame>(kf, std::make_shared<Map>(), &stop);
  ya::bundleAdjustImpl<Frame>(std::vector<std::shared_ptr<KeyFrame>>{kf, kf2}, mps, 10, &stop, 0L, true);
  ya::globalBundleAdjustImpl<Frame>(std::make_shared<Map>(), 10, nullptr, 7L, false);
  n += ya::optimizePoseImpl(a);
#endif
  return n;
}
