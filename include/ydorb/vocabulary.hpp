// Drop-in vocabulary: DBoW3::Vocabulary whose transform(features, BowVector&, FeatureVector&, levelsup) runs on the MI355X
// (reference thirdParty/DBow3/src/Vocabulary.cpp:752-824, called by Frame::computeBoW / KeyFrame::computeBoW, src/frame.cpp:265-272).
// System creates `std::make_shared<ydorb::adapter::GpuVocabulary>(path)` where it created a DBoW3::Vocabulary; everything that holds
// a std::shared_ptr<DBoW3::Vocabulary> (Frame, KeyFrame, KeyFrameDatabase's copy constructor, scoring) keeps working on the base class.
// Loading, scoring, the inverted file and the other transform overloads stay DBoW3's; only the per-frame transform is replaced.
#ifndef YDORB_ADAPTER_VOCABULARY_HPP
#define YDORB_ADAPTER_VOCABULARY_HPP

#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "DBoW3/DBoW3.h"
#include "c_api.h"

namespace ydorb {
namespace adapter {

class GpuVocabulary : public DBoW3::Vocabulary {
 public:
  explicit GpuVocabulary(const std::string& filename, int device = 0) : DBoW3::Vocabulary(filename), m_device(device) {}
  ~GpuVocabulary() override { ydorb_vocabulary_destroy(m_handle); }

  // Batched form: one call for several descriptor sets (e.g. the two key frames of a loop candidate, or a replayed sequence).
  void transformBatch(const std::vector<const std::vector<cv::Mat>*>& sets, std::vector<DBoW3::BowVector>& vs, std::vector<DBoW3::FeatureVector>& fvs,
                      int levelsup) const {
    vs.assign(sets.size(), DBoW3::BowVector());
    fvs.assign(sets.size(), DBoW3::FeatureVector());
    if (sets.empty() || empty()) return;   // :758-761
    upload();
    int cap = 1;
    for (const auto* s : sets) cap = std::max<int>(cap, (int)s->size());
    const size_t F = sets.size();
    std::vector<uint8_t> desc(F * cap * 32);
    std::vector<int32_t> n(F);
    for (size_t f = 0; f < F; f++) {
      n[f] = (int32_t)sets[f]->size();
      for (int i = 0; i < n[f]; i++) std::memcpy(&desc[(f * cap + i) * 32], (*sets[f])[i].ptr<uint8_t>(), 32);
    }
    std::vector<int32_t> bw(F * cap), fn(F * cap), fs(F * (cap + 1)), ff(F * cap), nw(F), nn(F);
    std::vector<double> bv(F * cap);
    if (ydorb_vocabulary_transform(m_handle, desc.data(), n.data(), (int32_t)F, cap, levelsup, bw.data(), bv.data(), nw.data(), fn.data(), fs.data(), ff.data(),
                                   nn.data(), nullptr) != YDORB_OK)
      throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
    for (size_t f = 0; f < F; f++) {
      DBoW3::BowVector& v = vs[f];
      for (int e = 0; e < nw[f]; e++) v.insert(v.end(), std::make_pair((DBoW3::WordId)bw[f * cap + e], bv[f * cap + e]));   // ascending: O(1) each
      DBoW3::FeatureVector& fv = fvs[f];
      for (int e = 0; e < nn[f]; e++) {
        const int32_t* a = &ff[f * cap + fs[f * (cap + 1) + e]];
        const int32_t* b = &ff[f * cap + fs[f * (cap + 1) + e + 1]];
        fv.insert(fv.end(), std::make_pair((DBoW3::NodeId)fn[f * cap + e], std::vector<unsigned int>(a, b)));
      }
    }
  }

  // virtual void transform(const std::vector<cv::Mat>&, BowVector&, FeatureVector&, int) const, Vocabulary.h:154
  void transform(const std::vector<cv::Mat>& features, DBoW3::BowVector& v, DBoW3::FeatureVector& fv, int levelsup) const override {
    std::vector<DBoW3::BowVector> vs;
    std::vector<DBoW3::FeatureVector> fvs;
    transformBatch({&features}, vs, fvs, levelsup);
    v.swap(vs[0]);
    fv.swap(fvs[0]);
  }

 private:
  void upload() const {   // flatten m_nodes once (thread-safe: tracking and loop closing both call computeBoW)
    std::lock_guard<std::mutex> lock(m_mutex);
    if (m_handle) return;
    const int nn = (int)m_nodes.size();
    std::vector<int32_t> begin(nn + 1, 0), ids, word(nn, 0);
    std::vector<uint8_t> desc((size_t)nn * 32, 0);
    std::vector<double> weight(nn, 0.0);
    for (int i = 0; i < nn; i++) {
      for (DBoW3::NodeId c : m_nodes[i].children) ids.push_back((int32_t)c);
      begin[i + 1] = (int32_t)ids.size();
      if (i > 0 && !m_nodes[i].descriptor.empty()) std::memcpy(&desc[(size_t)i * 32], m_nodes[i].descriptor.ptr<uint8_t>(), 32);
      weight[i] = m_nodes[i].weight;
      word[i] = (int32_t)m_nodes[i].word_id;
    }
    DBoW3::LNorm norm;
    const bool must = m_scoring_object->mustNormalize(norm);   // Vocabulary.cpp:766
    YdVocabularyTree t{nn, m_L, begin.data(), ids.data(), desc.data(), weight.data(), word.data(), (int32_t)m_weighting,
                       must ? (norm == DBoW3::L1 ? 1 : 2) : 0};
    if (ydorb_vocabulary_create(&t, m_device, &m_handle) != YDORB_OK) throw std::runtime_error(std::string("ydorb: ") + ydorb_last_error());
  }
  int m_device;
  mutable std::mutex m_mutex;
  mutable ydorb_vocabulary_t* m_handle = nullptr;
};

}  // namespace adapter
}  // namespace ydorb
#endif
