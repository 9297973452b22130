// Drop-in adapter: YDORBSLAM::OrbExtractor with the reference's exact public surface
// (reference src/orbExtractor.hpp:31-74) on top of the ydorb C ABI.  Link libydorb.so instead of compiling
// src/orbExtractor.cpp; Frame (src/frame.cpp:84-87,129,366,412-427) and Tracking (src/tracking.cpp:48-54) stay unchanged.
// Needs OpenCV (the reference's own dependency) for cv::Mat / cv::KeyPoint only; no OpenCV algorithm is called.
#ifndef YDORB_ADAPTER_ORBEXTRACTOR_HPP
#define YDORB_ADAPTER_ORBEXTRACTOR_HPP

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "c_api.h"

namespace YDORBSLAM {

class OrbExtractor {
 public:
  OrbExtractor(int _keyPointsNum, float _scaleFactor, int _levelsNum, int _initFastThd, int _minFastThd, int _device = 0)
      : m_int_levelsNum(_levelsNum) {
    YdExtractorConfig cfg{_keyPointsNum, _scaleFactor, _levelsNum, _initFastThd, _minFastThd, _device, 1, 0};
    if (ydorb_extractor_create(&cfg, &m_handle) != YDORB_OK) throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
    m_v_scaleFactors.resize(_levelsNum); m_v_invScaleFactors.resize(_levelsNum);
    m_v_scaleFactorSquares.resize(_levelsNum); m_v_invScaleFactorSquares.resize(_levelsNum);
    ydorb_extractor_tables(m_handle, m_v_scaleFactors.data(), m_v_invScaleFactors.data(), m_v_scaleFactorSquares.data(),
                           m_v_invScaleFactorSquares.data(), nullptr);
    m_flt_scaleFactor = _scaleFactor;
  }
  ~OrbExtractor() { ydorb_extractor_destroy(m_handle); }
  OrbExtractor(const OrbExtractor&) = delete;
  OrbExtractor& operator=(const OrbExtractor&) = delete;

  // src/orbExtractor.cpp:355-399.  Empty image: silent return (:357-359); zero keypoints: descriptors released (:370-371).
  void extractAndCompute(const cv::InputArray& _image, std::vector<cv::KeyPoint>& _keyPoints, cv::OutputArray& _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    static_assert(sizeof(cv::KeyPoint) == sizeof(YdKeyPoint), "cv::KeyPoint must be the 28-byte POD the ABI mirrors");
    const int cap = ydorb_extractor_max_keypoints(m_handle);
    std::vector<YdKeyPoint> kps(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int32_t n = 0;
    if (ydorb_extract(m_handle, image.data, image.cols, image.rows, (int)image.step, kps.data(), desc.data, cap, &n) != YDORB_OK)
      throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
    _keyPoints.resize(n);
    if (n) std::memcpy(static_cast<void*>(_keyPoints.data()), kps.data(), sizeof(YdKeyPoint) * n);
    if (n == 0) _descriptors.release();
    else desc.rowRange(0, n).copyTo(_descriptors);
    if (m_b_downloadPyramid) downloadPyramid();
  }

  // getters, src/orbExtractor.hpp:42-49 (getKeyPointsNum() returns the level count in the reference too, :42)
  int getKeyPointsNum() { return m_int_levelsNum; }
  float getScaleFactor() { return m_flt_scaleFactor; }
  int getLevelsNum() { return m_int_levelsNum; }
  std::vector<float> getScaleFactors() { return m_v_scaleFactors; }
  std::vector<float> getInvScaleFactors() { return m_v_invScaleFactors; }
  std::vector<float> getScaleFactorSquares() { return m_v_scaleFactorSquares; }
  std::vector<float> getInvScaleFactorSquares() { return m_v_invScaleFactorSquares; }
  std::vector<cv::Mat> getImagePyramid() { return m_v_imagePyramid; }
  // public member read by Frame::computeStereoMatches (src/frame.cpp:366,412-427): ROI views into 19-px padded buffers,
  // refreshed after every call (first-call semantics: the reference's push_back-without-clear at :612 is not reproduced).
  std::vector<cv::Mat> m_v_imagePyramid;
  // RGB-D pipelines never read the pyramid: skip the device->host copy of the 8 levels.
  void setPyramidDownload(bool on) { m_b_downloadPyramid = on; }
  ydorb_extractor_t* handle() { return m_handle; }

 protected:
  void downloadPyramid() {
    m_v_imagePyramid.clear();
    std::vector<cv::Mat> full;
    std::vector<uint8_t*> ptrs;
    std::vector<size_t> bytes;
    std::vector<int32_t> ws, hs;
    for (int l = 0; l < m_int_levelsNum; l++) {
      int32_t w, h, stride;
      const uint8_t* d = nullptr;
      if (ydorb_extractor_pyramid(m_handle, 0, l, &d, &w, &h, &stride) != YDORB_OK) return;
      full.emplace_back(h + 38, w + 38, CV_8UC1);
      ptrs.push_back(full.back().data);
      bytes.push_back(full.back().total());
      ws.push_back(w); hs.push_back(h);
    }
    // one device-to-host transfer for all levels
    if (ydorb_extractor_read_pyramid(m_handle, 0, ptrs.data(), bytes.data(), m_int_levelsNum) != YDORB_OK)
      throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
    for (int l = 0; l < m_int_levelsNum; l++) m_v_imagePyramid.push_back(full[l](cv::Rect(19, 19, ws[l], hs[l])));
  }
  ydorb_extractor_t* m_handle = nullptr;
  int m_int_levelsNum;
  float m_flt_scaleFactor;
  bool m_b_downloadPyramid = true;
  std::vector<float> m_v_scaleFactors, m_v_invScaleFactors, m_v_scaleFactorSquares, m_v_invScaleFactorSquares;
};

}  // namespace YDORBSLAM
#endif
