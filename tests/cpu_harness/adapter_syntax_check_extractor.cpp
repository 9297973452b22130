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
