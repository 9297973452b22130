// TEST-ONLY: type-checks include/ydorb/orbExtractor.hpp against the mock OpenCV declarations.
#include "../../include/ydorb/orbExtractor.hpp"
int check(cv::Mat& img, std::vector<cv::KeyPoint>& k, cv::Mat& d) {
  YDORBSLAM::OrbExtractor e(1000, 1.2f, 8, 20, 7);
  e.extractAndCompute(img, k, d);
  return e.getLevelsNum() + (int)e.getScaleFactors().size() + (int)e.m_v_imagePyramid.size() + e.getKeyPointsNum();
}

// Frame::computeStereoMatches body (include/ydorb/frame.hpp) against the members of reference src/frame.hpp it touches
#include <memory>
#include "../../include/ydorb/frame.hpp"
struct StereoFrame {
  std::shared_ptr<YDORBSLAM::OrbExtractor> m_sptr_leftOrbExtractor, m_sptr_rightOrbExtractor;
  std::vector<cv::KeyPoint> m_v_keyPoints, m_v_rightKeyPoints; cv::Mat m_cvMat_descriptors, m_cvMat_rightDescriptors;
  std::vector<float> m_v_rightXcords, m_v_depth; int m_int_keyPointsNum;
  static float m_flt_baseLineTimesFx, m_flt_baseLine;
};
int checkStereo(StereoFrame& f) { return ydorb::adapter::computeStereoMatchesImpl(f) + ydorb::adapter::computeStereoMatchesImpl(f, YDORB_STEREO_INDEX_BY_KEYPOINT); }

// DBoW3::Vocabulary subclass (include/ydorb/vocabulary.hpp) against the DBoW3 members it touches
#include "../../include/ydorb/vocabulary.hpp"
int checkVocabulary(const std::vector<cv::Mat>& features) {
  ydorb::adapter::GpuVocabulary voc("orbvoc.dbow3");
  DBoW3::BowVector v; DBoW3::FeatureVector fv;
  std::shared_ptr<DBoW3::Vocabulary> base = std::make_shared<ydorb::adapter::GpuVocabulary>("orbvoc.dbow3");
  base->transform(features, v, fv, 4);
  voc.transform(features, v, fv, 4);
  return (int)v.size() + (int)fv.size();
}
