// TEST-ONLY: type-checks include/ydorb/orbExtractor.hpp against the mock OpenCV declarations.
#include "../../include/ydorb/orbExtractor.hpp"
int check(cv::Mat& img, std::vector<cv::KeyPoint>& k, cv::Mat& d) {
  YDORBSLAM::OrbExtractor e(1000, 1.2f, 8, 20, 7);
  e.extractAndCompute(img, k, d);
  return e.getLevelsNum() + (int)e.getScaleFactors().size() + (int)e.m_v_imagePyramid.size() + e.getKeyPointsNum();
}
