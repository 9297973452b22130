// Shared host helpers of the C-ABI library (error text, device guard).
#pragma once
#include <cstdint>
struct ydorb_extractor;
namespace ydorb {
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// YDORB_OK if `device` is a usable gfx950 HIP device and is now current; YDORB_ERR_NO_DEVICE otherwise.
// The product has no CPU path: callers propagate the error.
int require_device(int device);
// Device view of the image pyramid an extractor built in its last call (m_v_imagePyramid, read by Frame::computeStereoMatches,
// reference src/frame.cpp:366,412-427): ROI origin of every level in frame 0, the distance between frames, level geometry and
// the scale tables.  Defined in orb_extractor.hip.
struct PyramidView {
  int device, nLevels, frames;
  long long frameStride;
  const uint8_t* roi[8];
  int w[8], h[8], pitch[8];
  float scale[8], invScale[8];
  int quota[8];   // keypoints per level the extractor returns at most (m_v_keyPointsNumsPerLevel)
  void* stream;
};
int extractor_pyramid_view(const ydorb_extractor* e, PyramidView* out);
}  // namespace ydorb
