// TEST-ONLY: EXECUTES the adapter templates of include/ydorb/*.hpp (the code a maintainer pastes into the reference's
// orbExtractor.cpp / orbMatcher.cpp / optimizer.cpp) against small stand-ins of the reference's Frame / KeyFrame / MapPoint / Map
// classes that carry real data, on the GPU, and dumps what the adapters did to those objects.  tests/test_adapter_exec.py builds
// the scenarios, states the reference's host-side logic independently (covisibility walk, vertex / edge assembly, projections)
// and compares with direct C-ABI calls on the same data.  OpenCV / Eigen are the functional mocks of tests/cpu_harness/mockrt.
//
//   adapter_run ba      scenario.bin out.bin     Optimizer::localBundleAdjust      (optimizer.cpp:138-352)
//   adapter_run proj    scenario.bin out.bin     searchByProjectionInLastAndCurrentFrame (orbMatcher.cpp:65-155)
//   adapter_run bow     scenario.bin out.bin     searchByBowInKeyFrameAndFrame     (orbMatcher.cpp:303-379)
//   adapter_run extract scenario.bin out.bin     OrbExtractor::extractAndCompute + m_v_imagePyramid (orbExtractor.cpp:355-399)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/ydorb/orbExtractor.hpp"
#include "../../include/ydorb/orbMatcher.hpp"
#include "../../include/ydorb/optimizer.hpp"

namespace {

struct Reader {
  FILE* f;
  explicit Reader(const char* path) : f(fopen(path, "rb")) { if (!f) { perror(path); exit(2); } }
  ~Reader() { fclose(f); }
  template <class T> T get() { T v; if (fread(&v, sizeof(T), 1, f) != 1) { fprintf(stderr, "scenario truncated\n"); exit(2); } return v; }
  template <class T> void get(T* p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "scenario truncated\n"); exit(2); } }
};
struct Writer {
  FILE* f;
  explicit Writer(const char* path) : f(fopen(path, "wb")) { if (!f) { perror(path); exit(2); } }
  ~Writer() { fclose(f); }
  template <class T> void put(const T& v) { fwrite(&v, sizeof(T), 1, f); }
  template <class T> void put(const T* p, size_t n) { if (n) fwrite(p, sizeof(T), n, f); }
};

// ---- stand-ins with real storage: only the members the adapters touch (names as in reference src/*.hpp) -------------------------
struct KeyFrame;
struct Frame;
struct MapPoint : std::enable_shared_from_this<MapPoint> {
  int index = -1;
  bool bad = false;
  cv::Mat pos, desc;                       // 3x1 CV_32F, 1x32 CV_8U
  int nObs = 0;                            // getObservationsNum() when no observation table is given
  std::map<std::shared_ptr<KeyFrame>, int> obs;
  long int m_int_localBAForKeyFrameID = 0, m_int_globalBAforKeyFrameID = 0;
  cv::Mat m_cvMat_posGlobalBA;
  int updates = 0;
  bool isBad() { return bad; }
  int getObservationsNum() { return obs.empty() ? nObs : (int)obs.size(); }
  cv::Mat getDescriptor() { return desc.clone(); }           // mapPoint.hpp:63-66 returns a clone
  cv::Mat getPosInWorld() { return pos.clone(); }
  std::map<std::shared_ptr<KeyFrame>, int> getObservations() { return obs; }
  void eraseObservation(std::shared_ptr<KeyFrame> kf) { obs.erase(kf); }
  void setPosInWorld(const cv::Mat& p) { pos = p.clone(); }
  void updateNormalAndDepth() { updates++; }
};
typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;
struct Frame {
  std::vector<cv::KeyPoint> m_v_keyPoints;
  cv::Mat m_cvMat_descriptors, m_cvMat_T_c2w;
  std::vector<float> m_v_rightXcords, m_v_scaleFactors, m_v_invScaleFactorSquares;
  std::vector<std::shared_ptr<MapPoint>> m_v_sptrMapPoints;
  std::vector<bool> m_v_isOutliers;
  int m_int_keyPointsNum = 0;
  FeatureVector m_bow_keyPointsVec;
  cv::Mat getCameraPoseByTransform_c2w() { return m_cvMat_T_c2w.clone(); }
  void setCameraPoseByTransform_c2w(cv::Mat T) { m_cvMat_T_c2w = T.clone(); }
  static float m_flt_minX, m_flt_maxX, m_flt_minY, m_flt_maxY, m_flt_fx, m_flt_fy, m_flt_cx, m_flt_cy, m_flt_baseLine, m_flt_baseLineTimesFx;
  bool isInImage(const float& x, const float& y) const { return x >= m_flt_minX && x < m_flt_maxX && y >= m_flt_minY && y < m_flt_maxY; }   // frame.cpp:291-294
};
float Frame::m_flt_minX, Frame::m_flt_maxX, Frame::m_flt_minY, Frame::m_flt_maxY, Frame::m_flt_fx, Frame::m_flt_fy, Frame::m_flt_cx, Frame::m_flt_cy,
    Frame::m_flt_baseLine, Frame::m_flt_baseLineTimesFx;
struct KeyFrame : std::enable_shared_from_this<KeyFrame> {
  int index = -1;
  bool bad = false;
  long int m_int_keyFrameID = 0, m_int_localBAForKeyFrameID = 0, m_int_fixedBAForKeyFrameID = 0, m_int_globalBAForKeyFrameID = 0;
  cv::Mat m_cvMat_T_c2w_GlobalBA, pose, m_cvMat_descriptors;
  std::vector<cv::KeyPoint> m_v_keyPoints;
  std::vector<float> m_v_rightXcords, m_v_invScaleFactorSquares, m_v_scaleFactors, m_v_scaleFactorSquares;
  int m_int_keyPointsNum = 0;
  FeatureVector m_bow_keyPointsVec;
  std::vector<std::shared_ptr<MapPoint>> mps;
  std::vector<std::shared_ptr<KeyFrame>> connected;
  std::vector<std::shared_ptr<MapPoint>> getMatchedMapPointsVec() { return mps; }
  std::vector<std::shared_ptr<KeyFrame>> getOrderedConnectedKeyFrames() { return connected; }
  bool isBad() { return bad; }
  cv::Mat getCameraPoseByTransform_c2w() { return pose.clone(); }
  void setCameraPoseByTransform_c2w(cv::Mat T) { pose = T.clone(); }
  void eraseMatchedMapPoint(std::shared_ptr<MapPoint> mp) {       // keyFrame.cpp: by the point's index in this keyframe
    for (auto& p : mps) if (p == mp) p.reset();
  }
};
struct Map { std::mutex m_mutex_updateMap; };

void readCamera(Reader& R) {
  float c[10];
  R.get(c, 10);
  Frame::m_flt_fx = c[0]; Frame::m_flt_fy = c[1]; Frame::m_flt_cx = c[2]; Frame::m_flt_cy = c[3]; Frame::m_flt_baseLine = c[4]; Frame::m_flt_baseLineTimesFx = c[5];
  Frame::m_flt_minX = c[6]; Frame::m_flt_maxX = c[7]; Frame::m_flt_minY = c[8]; Frame::m_flt_maxY = c[9];
}
cv::Mat readMat32(Reader& R, int rows, int cols) { cv::Mat m(rows, cols, CV_32F); R.get(m.ptr<float>(), (size_t)rows * cols); return m; }
void readKeypoints(Reader& R, int n, std::vector<cv::KeyPoint>& kps, cv::Mat& desc, std::vector<float>& rightX) {
  kps.resize(n);
  R.get(reinterpret_cast<unsigned char*>(kps.data()), (size_t)n * sizeof(cv::KeyPoint));
  desc.create(std::max(n, 1), 32, CV_8U);
  R.get(desc.data, (size_t)n * 32);
  rightX.resize(n);
  R.get(rightX.data(), n);
}
void readFeatureVector(Reader& R, FeatureVector& fv) {
  const int nodes = R.get<int32_t>();
  for (int i = 0; i < nodes; i++) {
    const unsigned id = R.get<uint32_t>();
    const int m = R.get<int32_t>();
    std::vector<unsigned> f(m);
    R.get(f.data(), m);
    fv[id] = f;
  }
}

int runBa(const char* in, const char* out) {
  Reader R(in);
  readCamera(R);
  const int nKF = R.get<int32_t>(), nMP = R.get<int32_t>(), kf0 = R.get<int32_t>(), nConn = R.get<int32_t>();
  std::vector<std::shared_ptr<KeyFrame>> kfs(nKF);
  std::vector<std::shared_ptr<MapPoint>> mps(nMP);
  for (int i = 0; i < nMP; i++) { mps[i] = std::make_shared<MapPoint>(); mps[i]->index = i; }
  std::vector<std::vector<int>> kpMp(nKF);
  for (int k = 0; k < nKF; k++) {
    auto kf = std::make_shared<KeyFrame>();
    kf->index = k;
    kf->m_int_keyFrameID = R.get<int32_t>(); kf->bad = R.get<int32_t>() != 0;
    const int nkp = R.get<int32_t>();
    kf->pose = readMat32(R, 4, 4);
    kf->m_v_invScaleFactorSquares.resize(8);
    R.get(kf->m_v_invScaleFactorSquares.data(), 8);
    readKeypoints(R, nkp, kf->m_v_keyPoints, kf->m_cvMat_descriptors, kf->m_v_rightXcords);
    kpMp[k].resize(nkp);
    R.get(kpMp[k].data(), nkp);
    kf->mps.resize(nkp);
    for (int i = 0; i < nkp; i++) if (kpMp[k][i] >= 0) kf->mps[i] = mps[kpMp[k][i]];
    kfs[k] = kf;
  }
  for (int i = 0; i < nConn; i++) kfs[kf0]->connected.push_back(kfs[R.get<int32_t>()]);
  for (int i = 0; i < nMP; i++) {
    R.get<int32_t>();   // id (informational)
    mps[i]->bad = R.get<int32_t>() != 0;
    const int nobs = R.get<int32_t>();
    mps[i]->pos = readMat32(R, 3, 1);
    for (int o = 0; o < nobs; o++) { const int k = R.get<int32_t>(), idx = R.get<int32_t>(); mps[i]->obs[kfs[k]] = idx; }
  }
  bool stop = false;
  auto map = std::make_shared<Map>();
  ydorb::adapter::localBundleAdjustImpl<std::shared_ptr<KeyFrame>, std::shared_ptr<Map>, Frame>(kfs[kf0], map, &stop);
  Writer W(out);
  for (int k = 0; k < nKF; k++) {
    W.put(kfs[k]->pose.ptr<float>(), 16);
    W.put<int32_t>((int32_t)kfs[k]->m_int_localBAForKeyFrameID); W.put<int32_t>((int32_t)kfs[k]->m_int_fixedBAForKeyFrameID);
    for (size_t i = 0; i < kfs[k]->mps.size(); i++) W.put<int32_t>(kfs[k]->mps[i] ? kfs[k]->mps[i]->index : -1);
  }
  for (int i = 0; i < nMP; i++) {
    W.put(mps[i]->pos.ptr<float>(), 3);
    W.put<int32_t>(mps[i]->updates);
    W.put<int32_t>((int32_t)mps[i]->obs.size());
    std::vector<std::pair<int, int>> o;
    for (auto& kv : mps[i]->obs) o.push_back({kv.first->index, kv.second});
    std::sort(o.begin(), o.end());
    for (auto& p : o) { W.put<int32_t>(p.first); W.put<int32_t>(p.second); }
  }
  return 0;
}

void readFrame(Reader& R, Frame& f, std::vector<std::shared_ptr<MapPoint>>& mps) {
  f.m_cvMat_T_c2w = readMat32(R, 4, 4);
  const int n = R.get<int32_t>();
  f.m_int_keyPointsNum = n;
  readKeypoints(R, n, f.m_v_keyPoints, f.m_cvMat_descriptors, f.m_v_rightXcords);
  std::vector<int32_t> mp(n), outl(n);
  R.get(mp.data(), n); R.get(outl.data(), n);
  f.m_v_sptrMapPoints.resize(n); f.m_v_isOutliers.resize(n);
  for (int i = 0; i < n; i++) { if (mp[i] >= 0) f.m_v_sptrMapPoints[i] = mps[mp[i]]; f.m_v_isOutliers[i] = outl[i] != 0; }
}

int runProj(const char* in, const char* out) {
  Reader R(in);
  readCamera(R);
  std::vector<float> sf(8);
  R.get(sf.data(), 8);
  const int nMP = R.get<int32_t>();
  std::vector<std::shared_ptr<MapPoint>> mps(nMP);
  for (int i = 0; i < nMP; i++) {
    mps[i] = std::make_shared<MapPoint>();
    mps[i]->index = i;
    mps[i]->pos = readMat32(R, 3, 1);
    mps[i]->nObs = R.get<int32_t>();
    mps[i]->desc.create(1, 32, CV_8U);
    R.get(mps[i]->desc.data, 32);
  }
  Frame last, cur;
  readFrame(R, last, mps);
  readFrame(R, cur, mps);
  last.m_v_scaleFactors = cur.m_v_scaleFactors = sf;
  const float th = R.get<float>();
  const int checkOri = R.get<int32_t>();
  const int n = ydorb::adapter::searchByProjectionInLastAndCurrentFrame(ydorb::adapter::matcher(), cur, last, th, checkOri != 0);
  Writer W(out);
  W.put<int32_t>(n);
  for (auto& p : cur.m_v_sptrMapPoints) W.put<int32_t>(p ? p->index : -1);
  return 0;
}

int runBow(const char* in, const char* out) {
  Reader R(in);
  const int nMP = R.get<int32_t>();
  std::vector<std::shared_ptr<MapPoint>> mps(nMP);
  for (int i = 0; i < nMP; i++) { mps[i] = std::make_shared<MapPoint>(); mps[i]->index = i; mps[i]->bad = R.get<int32_t>() != 0; }
  auto kf = std::make_shared<KeyFrame>();
  const int na = R.get<int32_t>();
  kf->m_int_keyPointsNum = na;
  readKeypoints(R, na, kf->m_v_keyPoints, kf->m_cvMat_descriptors, kf->m_v_rightXcords);
  std::vector<int32_t> mp(na);
  R.get(mp.data(), na);
  kf->mps.resize(na);
  for (int i = 0; i < na; i++) if (mp[i] >= 0) kf->mps[i] = mps[mp[i]];
  readFeatureVector(R, kf->m_bow_keyPointsVec);
  Frame f;
  const int nb = R.get<int32_t>();
  f.m_int_keyPointsNum = nb;
  readKeypoints(R, nb, f.m_v_keyPoints, f.m_cvMat_descriptors, f.m_v_rightXcords);
  readFeatureVector(R, f.m_bow_keyPointsVec);
  const float ratio = R.get<float>();
  const int checkOri = R.get<int32_t>();
  std::vector<std::shared_ptr<MapPoint>> matched;
  const int n = ydorb::adapter::searchByBowInKeyFrameAndFrame(ydorb::adapter::matcher(), kf, f, matched, ratio, checkOri != 0);
  Writer W(out);
  W.put<int32_t>(n);
  W.put<int32_t>((int32_t)matched.size());
  for (auto& p : matched) W.put<int32_t>(p ? p->index : -1);
  return 0;
}

int runExtract(const char* in, const char* out) {
  Reader R(in);
  const int w = R.get<int32_t>(), h = R.get<int32_t>(), nf = R.get<int32_t>();
  cv::Mat img(h, w, CV_8UC1);
  R.get(img.data, (size_t)w * h);
  YDORBSLAM::OrbExtractor ex(nf, 1.2f, 8, 20, 7);
  std::vector<cv::KeyPoint> kps;
  cv::Mat desc;
  ex.extractAndCompute(img, kps, desc);
  Writer W(out);
  W.put<int32_t>((int32_t)kps.size());
  W.put(reinterpret_cast<const unsigned char*>(kps.data()), kps.size() * sizeof(cv::KeyPoint));
  for (size_t i = 0; i < kps.size(); i++) W.put(desc.ptr<unsigned char>((int)i), 32);
  W.put<int32_t>((int32_t)ex.m_v_imagePyramid.size());
  for (const cv::Mat& lv : ex.m_v_imagePyramid) {        // ROI views: rows at the padded buffer's pitch
    W.put<int32_t>(lv.cols); W.put<int32_t>(lv.rows);
    for (int y = 0; y < lv.rows; y++) W.put(lv.ptr<unsigned char>(y), lv.cols);
  }
  W.put<int32_t>(ex.getLevelsNum()); W.put<int32_t>(ex.getKeyPointsNum());
  const std::vector<float> s = ex.getScaleFactors();
  W.put(s.data(), s.size());
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: adapter_run ba|proj|bow|extract scenario.bin out.bin\n"); return 2; }
  try {
    const std::string what = argv[1];
    if (what == "ba") return runBa(argv[2], argv[3]);
    if (what == "proj") return runProj(argv[2], argv[3]);
    if (what == "bow") return runBow(argv[2], argv[3]);
    if (what == "extract") return runExtract(argv[2], argv[3]);
  } catch (const std::exception& e) {
    fprintf(stderr, "adapter_run: %s\n", e.what());
    return 1;
  }
  return 2;
}
