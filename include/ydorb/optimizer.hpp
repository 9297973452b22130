// Drop-in adapter for YDORBSLAM::Optimizer::localBundleAdjust (reference src/optimizer.hpp:34, src/optimizer.cpp:138-352)
// on top of the ydorb C ABI.  Template over the reference's KeyFrame / MapPoint / Map types: the body of
//   void Optimizer::localBundleAdjust(std::shared_ptr<KeyFrame> kf, std::shared_ptr<Map> map, bool& stop)
// becomes  ydorb::adapter::localBundleAdjust(kf, map, &stop == nullptr ? nullptr : &stop);
// The host keeps the covisibility walk (:140-173), the float->double conversions of Converter (converter.cpp:12-19) and the
// write-back under Map::m_mutex_updateMap (:336-351); residuals, Jacobians, Huber, Schur, Cholesky, LM run on the GPU.
#ifndef YDORB_ADAPTER_OPTIMIZER_HPP
#define YDORB_ADAPTER_OPTIMIZER_HPP

#include <cmath>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core.hpp>

#include "c_api.h"

namespace ydorb {
namespace adapter {

// Converter::transform_cvMat_SE3Quat (converter.cpp:12-19): float 4x4 -> (t, unit quaternion) in double
inline void poseToSE3Quat(const cv::Mat& T, double* p7) {
  Eigen::Matrix3d R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R(i, j) = T.at<float>(i, j);
  Eigen::Quaterniond q(R);
  if (q.w() < 0) q.coeffs() *= -1;
  q.normalize();
  p7[0] = T.at<float>(0, 3); p7[1] = T.at<float>(1, 3); p7[2] = T.at<float>(2, 3);
  p7[3] = q.x(); p7[4] = q.y(); p7[5] = q.z(); p7[6] = q.w();
}
// Converter::transform_SE3_cvMat (converter.cpp:20-38): back to a float 4x4
inline cv::Mat se3QuatToPose(const double* p7) {
  const Eigen::Matrix3d R = Eigen::Quaterniond(p7[6], p7[3], p7[4], p7[5]).toRotationMatrix();
  cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T.at<float>(i, j) = (float)R(i, j);
    T.at<float>(i, 3) = (float)p7[i];
  }
  return T;
}

template <class KeyFramePtr, class MapPtr, class FrameT>
void localBundleAdjustImpl(KeyFramePtr kf0, MapPtr map, const volatile bool* stop) {
  typedef decltype(kf0->getMatchedMapPointsVec()) MapPointVec;
  typedef typename MapPointVec::value_type MapPointPtr;
  const long int tag = kf0->m_int_keyFrameID;
  // local keyframes = current + its covisible ones; local points = what they see; fixed keyframes = other observers (:140-173)
  std::list<KeyFramePtr> localKFs{kf0};
  kf0->m_int_localBAForKeyFrameID = tag;
  for (const KeyFramePtr& c : kf0->getOrderedConnectedKeyFrames()) {
    c->m_int_localBAForKeyFrameID = tag;
    if (!c->isBad()) localKFs.push_back(c);
  }
  std::list<MapPointPtr> localMPs;
  for (const KeyFramePtr& k : localKFs)
    for (const MapPointPtr& mp : k->getMatchedMapPointsVec())
      if (mp && !mp->isBad() && mp->m_int_localBAForKeyFrameID != tag) { localMPs.push_back(mp); mp->m_int_localBAForKeyFrameID = tag; }
  std::list<KeyFramePtr> fixedKFs;
  for (const MapPointPtr& mp : localMPs)
    for (const auto& ob : mp->getObservations())
      if (ob.first->m_int_localBAForKeyFrameID != tag && ob.first->m_int_fixedBAForKeyFrameID != tag) {
        ob.first->m_int_fixedBAForKeyFrameID = tag;
        if (!ob.first->isBad()) fixedKFs.push_back(ob.first);
      }
  // flat graph (:185-283)
  std::vector<KeyFramePtr> poseKF;
  std::map<long int, int> poseIndex;
  std::vector<double> poses;
  std::vector<uint8_t> fixed;
  long int maxKFid = 0;
  auto addPose = [&](const KeyFramePtr& k, bool fix) {
    poseIndex[k->m_int_keyFrameID] = (int)poseKF.size();
    poseKF.push_back(k);
    poses.resize(poses.size() + 7);
    poseToSE3Quat(k->getCameraPoseByTransform_c2w(), &poses[poses.size() - 7]);
    fixed.push_back(fix ? 1 : 0);
    if (k->m_int_keyFrameID > maxKFid) maxKFid = k->m_int_keyFrameID;
  };
  for (const KeyFramePtr& k : localKFs) if (!k->isBad()) addPose(k, k->m_int_keyFrameID == 0);
  const size_t nLocal = poseKF.size();
  for (const KeyFramePtr& k : fixedKFs) addPose(k, true);
  std::vector<MapPointPtr> pointMP(localMPs.begin(), localMPs.end());
  std::vector<double> points(3 * pointMP.size());
  std::vector<int32_t> ePose, ePoint;
  std::vector<double> eMeas, eInfo;
  std::vector<KeyFramePtr> edgeKF;
  for (size_t p = 0; p < pointMP.size(); p++) {
    const cv::Mat X = pointMP[p]->getPosInWorld();
    for (int d = 0; d < 3; d++) points[3 * p + d] = X.at<float>(d);
    for (const auto& ob : pointMP[p]->getObservations()) {
      if (ob.first->isBad() || ob.first->m_int_keyFrameID > maxKFid) continue;
      const auto it = poseIndex.find(ob.first->m_int_keyFrameID);
      if (it == poseIndex.end()) continue;
      const cv::KeyPoint& kp = ob.first->m_v_keyPoints[ob.second];
      const float ur = ob.first->m_v_rightXcords[ob.second];
      ePose.push_back(it->second); ePoint.push_back((int32_t)p);
      eMeas.push_back(kp.pt.x); eMeas.push_back(kp.pt.y); eMeas.push_back(ur < 0 ? -1.0 : (double)ur);
      eInfo.push_back(ob.first->m_v_invScaleFactorSquares[kp.octave]);
      edgeKF.push_back(ob.first);
    }
  }
  if (stop && *stop) return;  // :284-286
  YdBaProblem P{};
  P.n_poses = (int32_t)poseKF.size(); P.n_points = (int32_t)pointMP.size(); P.n_edges = (int32_t)ePose.size();
  P.poses = poses.data(); P.pose_fixed = fixed.data(); P.points = points.data();
  P.edge_pose = ePose.data(); P.edge_point = ePoint.data(); P.edge_meas = eMeas.data(); P.edge_inv_sigma2 = eInfo.data();
  P.fx = FrameT::m_flt_fx; P.fy = FrameT::m_flt_fy; P.cx = FrameT::m_flt_cx; P.cy = FrameT::m_flt_cy; P.bf = FrameT::m_flt_baseLineTimesFx;
  P.stop = reinterpret_cast<const volatile uint8_t*>(stop);
  std::vector<uint8_t> outlier(ePose.size() + 1, 0);
  YdBaResult res{};
  res.edge_outlier = outlier.data();
  if (ydorb_ba_solve(&P, nullptr, &res) != YDORB_OK) throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  // write-back (:336-351)
  std::unique_lock<std::mutex> lock(map->m_mutex_updateMap);
  for (size_t e = 0; e < ePose.size(); e++)
    if (outlier[e] && !pointMP[ePoint[e]]->isBad()) {
      edgeKF[e]->eraseMatchedMapPoint(pointMP[ePoint[e]]);
      pointMP[ePoint[e]]->eraseObservation(edgeKF[e]);
    }
  for (size_t k = 0; k < nLocal; k++) poseKF[k]->setCameraPoseByTransform_c2w(se3QuatToPose(&poses[7 * k]));
  for (size_t p = 0; p < pointMP.size(); p++) {
    cv::Mat X(3, 1, CV_32F);
    for (int d = 0; d < 3; d++) X.at<float>(d) = (float)points[3 * p + d];
    pointMP[p]->setPosInWorld(X);
    pointMP[p]->updateNormalAndDepth();
  }
}

// Optimizer::bundleAdjust (optimizer.cpp:7-137): every given keyframe (id 0 fixed) and map point, ONE optimize(iterNum) call, Huber
// optional; results go to the vertices' GBA fields when loopKeyFrameID != 0 (:113-136).  Same C ABI call with
// YDORB_BA_SINGLE_STAGE; the body of
//   void Optimizer::bundleAdjust(kfs, mps, iterNum, stop, loopKFid, robust)
// becomes  ydorb::adapter::bundleAdjustImpl<Frame>(kfs, mps, iterNum, &stop == nullptr ? nullptr : &stop, loopKFid, robust);
template <class FrameT, class KeyFramePtr, class MapPointPtr>
void bundleAdjustImpl(const std::vector<KeyFramePtr>& kfs, const std::vector<MapPointPtr>& mps, int iterNum, const volatile bool* stop,
                      long int loopKeyFrameID, bool robust) {
  std::vector<KeyFramePtr> poseKF;
  std::map<long int, int> poseIndex;
  std::vector<double> poses;
  std::vector<uint8_t> fixed;
  long int maxKFid = 0;
  for (const KeyFramePtr& k : kfs) {
    if (k->isBad()) continue;
    poseIndex[k->m_int_keyFrameID] = (int)poseKF.size();
    poseKF.push_back(k);
    poses.resize(poses.size() + 7);
    poseToSE3Quat(k->getCameraPoseByTransform_c2w(), &poses[poses.size() - 7]);
    fixed.push_back(k->m_int_keyFrameID == 0 ? 1 : 0);
    if (k->m_int_keyFrameID > maxKFid) maxKFid = k->m_int_keyFrameID;
  }
  std::vector<double> points(3 * mps.size(), 0.0);
  std::vector<uint8_t> excluded(mps.size(), 1);   // bad points and points without a usable observation stay untouched (:103-110)
  std::vector<int32_t> ePose, ePoint;
  std::vector<double> eMeas, eInfo;
  for (size_t p = 0; p < mps.size(); p++) {
    if (!mps[p] || mps[p]->isBad()) continue;
    const cv::Mat X = mps[p]->getPosInWorld();
    for (int d = 0; d < 3; d++) points[3 * p + d] = X.at<float>(d);
    for (const auto& ob : mps[p]->getObservations()) {
      if (ob.first->isBad() || ob.first->m_int_keyFrameID > maxKFid) continue;
      const auto it = poseIndex.find(ob.first->m_int_keyFrameID);
      if (it == poseIndex.end()) continue;   // g2o: optimizer.vertex(id) == nullptr -> the edge cannot be added
      const cv::KeyPoint& kp = ob.first->m_v_keyPoints[ob.second];
      const float ur = ob.first->m_v_rightXcords[ob.second];
      ePose.push_back(it->second); ePoint.push_back((int32_t)p);
      eMeas.push_back(kp.pt.x); eMeas.push_back(kp.pt.y); eMeas.push_back(ur < 0 ? -1.0 : (double)ur);
      eInfo.push_back(ob.first->m_v_invScaleFactorSquares[kp.octave]);
      excluded[p] = 0;
    }
  }
  YdBaProblem P{};
  P.n_poses = (int32_t)poseKF.size(); P.n_points = (int32_t)mps.size(); P.n_edges = (int32_t)ePose.size();
  P.poses = poses.data(); P.pose_fixed = fixed.data(); P.points = points.data();
  P.edge_pose = ePose.data(); P.edge_point = ePoint.data(); P.edge_meas = eMeas.data(); P.edge_inv_sigma2 = eInfo.data();
  P.fx = FrameT::m_flt_fx; P.fy = FrameT::m_flt_fy; P.cx = FrameT::m_flt_cx; P.cy = FrameT::m_flt_cy; P.bf = FrameT::m_flt_baseLineTimesFx;
  P.stop = reinterpret_cast<const volatile uint8_t*>(stop);
  YdBaOptions O;
  ydorb_ba_default_options(&O);
  O.iters1 = iterNum; O.iters2 = 0;
  O.delta_mono = (double)(float)std::sqrt(5.99);   // `const float monoDelta = sqrt(5.99)`, :37
  O.flags = YDORB_BA_SINGLE_STAGE | (robust ? 0 : YDORB_BA_NO_ROBUST);
  YdBaResult res{};
  if (ydorb_ba_solve(&P, &O, &res) != YDORB_OK) throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t k = 0; k < poseKF.size(); k++) {   // :113-124
    const cv::Mat T = se3QuatToPose(&poses[7 * k]);
    if (loopKeyFrameID == 0) {
      poseKF[k]->setCameraPoseByTransform_c2w(T);
    } else {
      poseKF[k]->m_cvMat_T_c2w_GlobalBA.create(4, 4, CV_32F);
      T.copyTo(poseKF[k]->m_cvMat_T_c2w_GlobalBA);
      poseKF[k]->m_int_globalBAForKeyFrameID = loopKeyFrameID;
    }
  }
  for (size_t p = 0; p < mps.size(); p++) {     // :126-136
    if (!mps[p] || mps[p]->isBad() || excluded[p]) continue;
    cv::Mat X(3, 1, CV_32F);
    for (int d = 0; d < 3; d++) X.at<float>(d) = (float)points[3 * p + d];
    if (loopKeyFrameID == 0) {
      mps[p]->setPosInWorld(X);
      mps[p]->updateNormalAndDepth();
    } else {
      mps[p]->m_cvMat_posGlobalBA.create(3, 1, CV_32F);
      X.copyTo(mps[p]->m_cvMat_posGlobalBA);
      mps[p]->m_int_globalBAforKeyFrameID = loopKeyFrameID;
    }
  }
}

// Optimizer::optimizePose (optimizer.cpp:358-501): pose-only refinement of one frame against its matched map points; returns the
// number of inliers and updates Frame::m_v_isOutliers and the frame pose exactly where the reference does.  The body of
//   int Optimizer::optimizePose(Frame& frame)   becomes   return ydorb::adapter::optimizePoseImpl(frame);
// (A tracker that refines several frames at once — replay, multi-camera rigs — can hand them to ydorb_pose_optimize as one batch:
// one workgroup per frame, 64 frames cost about as much as one.)
template <class FrameT>
int optimizePoseImpl(FrameT& frame) {
  std::vector<int> idx;
  std::vector<double> X, z, w;
  for (int i = 0; i < frame.m_int_keyPointsNum; i++) {
    if (!frame.m_v_sptrMapPoints[i]) continue;
    frame.m_v_isOutliers[i] = false;                       // :392
    const cv::KeyPoint& kp = frame.m_v_keyPoints[i];
    const cv::Mat Xw = frame.m_v_sptrMapPoints[i]->getPosInWorld();
    for (int d = 0; d < 3; d++) X.push_back(Xw.at<float>(d));
    const float ur = frame.m_v_rightXcords[i];
    z.push_back(kp.pt.x); z.push_back(kp.pt.y); z.push_back(ur < 0 ? -1.0 : (double)ur);
    w.push_back(frame.m_v_invScaleFactorSquares[kp.octave]);
    idx.push_back(i);
  }
  if (idx.size() < 3) return 0;                            // :443-445
  double pose[7];
  poseToSE3Quat(frame.getCameraPoseByTransform_c2w(), pose);
  const int32_t start[2] = {0, (int32_t)idx.size()};
  YdPoseBatch B{};
  B.n_frames = 1; B.device = 0; B.edge_start = start; B.poses = pose; B.points = X.data(); B.meas = z.data(); B.inv_sigma2 = w.data();
  B.fx = FrameT::m_flt_fx; B.fy = FrameT::m_flt_fy; B.cx = FrameT::m_flt_cx; B.cy = FrameT::m_flt_cy; B.bf = FrameT::m_flt_baseLineTimesFx;
  std::vector<uint8_t> outlier(idx.size());
  int32_t inliers = 0;
  if (ydorb_pose_optimize(&B, outlier.data(), &inliers, nullptr, nullptr) != YDORB_OK) throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  for (size_t k = 0; k < idx.size(); k++) frame.m_v_isOutliers[idx[k]] = outlier[k] != 0;
  frame.setCameraPoseByTransform_c2w(se3QuatToPose(pose));  // :498-499
  return inliers;
}

// Optimizer::globalBundleAdjust (optimizer.cpp:353-357)
template <class FrameT, class MapPtr>
void globalBundleAdjustImpl(MapPtr map, int iterNum, const volatile bool* stop, long int loopKeyFrameID, bool robust) {
  bundleAdjustImpl<FrameT>(map->getAllKeyFrames(), map->getAllMapPoints(), iterNum, stop, loopKeyFrameID, robust);
}

}  // namespace adapter
}  // namespace ydorb
#endif
