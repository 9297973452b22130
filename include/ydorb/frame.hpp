// Drop-in body for YDORBSLAM::Frame::computeStereoMatches() (reference src/frame.cpp:362-477) on top of the ydorb C ABI.
// In src/frame.cpp the member function becomes
//     void Frame::computeStereoMatches(){ ydorb::adapter::computeStereoMatchesImpl(*this); }
// with m_sptr_leftOrbExtractor / m_sptr_rightOrbExtractor being the adapter class of include/ydorb/orbExtractor.hpp (their
// pyramids stay in HBM: setPyramidDownload(false) is enough for this path).  Everything the function read is passed as it
// is: the undistorted keypoints of both images, both descriptor matrices, baseline and baseline x fx.
#ifndef YDORB_ADAPTER_FRAME_HPP
#define YDORB_ADAPTER_FRAME_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "c_api.h"
#include "orbMatcher.hpp"

namespace ydorb {
namespace adapter {

// flags = 0 reproduces the reference as written (including the left index that lags behind from the first keypoint that
// leaves the loop body early, frame.cpp:462); YDORB_STEREO_INDEX_BY_KEYPOINT indexes descriptor row and output slot by the
// keypoint itself.  Returns the status bits of ydorb_stereo_matches.
template <class FrameT>
int computeStereoMatchesImpl(FrameT& f, int flags = 0, ydorb_matcher_t* m = matcher()) {
  static_assert(sizeof(cv::KeyPoint) == sizeof(YdKeyPoint), "cv::KeyPoint must be the 28-byte POD the ABI mirrors");
  const int nl = (int)f.m_v_keyPoints.size(), nr = (int)f.m_v_rightKeyPoints.size();
  f.m_v_rightXcords.assign(f.m_int_keyPointsNum, -1.0f);   // :363-364
  f.m_v_depth.assign(f.m_int_keyPointsNum, -1.0f);
  if (nl == 0 || nr == 0) return 0;
  if (nl != f.m_int_keyPointsNum) throw std::runtime_error("ydorb: m_int_keyPointsNum differs from m_v_keyPoints.size()");   // frame.cpp:88 sets it so
  cv::Mat dl = f.m_cvMat_descriptors.isContinuous() ? f.m_cvMat_descriptors : f.m_cvMat_descriptors.clone();
  cv::Mat dr = f.m_cvMat_rightDescriptors.isContinuous() ? f.m_cvMat_rightDescriptors : f.m_cvMat_rightDescriptors.clone();
  const int32_t cl = nl, cr = nr;
  YdStereoSide L{f.m_sptr_leftOrbExtractor->handle(), 0, 1, reinterpret_cast<const YdKeyPoint*>(f.m_v_keyPoints.data()), dl.data, &cl, nl, 0};
  YdStereoSide R{f.m_sptr_rightOrbExtractor->handle(), 0, 1, reinterpret_cast<const YdKeyPoint*>(f.m_v_rightKeyPoints.data()), dr.data, &cr, nr, 0};
  int32_t kept = 0, status = 0;
  if (ydorb_stereo_matches(m, &L, &R, 1, FrameT::m_flt_baseLineTimesFx, FrameT::m_flt_baseLine, flags, f.m_v_rightXcords.data(), f.m_v_depth.data(),
                           &kept, &status, nullptr) != YDORB_OK)
    throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  return status;
}

}  // namespace adapter
}  // namespace ydorb
#endif
