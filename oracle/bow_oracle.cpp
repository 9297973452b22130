// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into or called from the product library).
//
// CPU restatement of DBoW3::Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
// (/root/reference/thirdParty/DBow3/src/Vocabulary.cpp:752-824) with the per-feature tree descent (:836-874), BowVector::addWeight /
// addIfNotExist / normalize (BowVector.cpp:31-88) and FeatureVector::addFeature (FeatureVector.cpp:31-45); call site
// Frame::computeBoW, /root/reference/src/frame.cpp:265-272 (levelsup = 4).
// Parity unpinned: the reference's vocabulary file is a missing blob and DBoW3 needs OpenCV (absent), so the restatement is anchored
// on the text of those files and exercised on synthetic trees.
//
// The tree crosses the boundary as flat arrays: node i's children are child_ids[child_begin[i] .. child_begin[i+1]) in the order of
// Node::children (the order decides ties, :858-865: first minimum); a node without children is a leaf (a word).
// One case is undefined in the reference: `NodeId nid` (:777) is only assigned when the descent passes level L - levelsup (:868-869);
// for a leaf above that level it is read uninitialised.  Here and in the product such a feature is filed under the leaf's own
// node id and status bit 0 is set.
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <vector>

extern "C" int yo_descriptor_distance(const uint8_t* a, const uint8_t* b);

extern "C" {

// weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (Vocabulary.h WeightingType order); norm: 0 none, 1 L1, 2 L2 (what
// GeneralScoring::mustNormalize returns).  node_weight already holds the per-word value transform() reads (:873).
// Outputs sized for n entries: bow_word / bow_value, fv_node / fv_start (n + 1) / fv_feat.  Returns status bits.
int yo_bow_transform(int L, const int32_t* child_begin, const int32_t* child_ids, const uint8_t* node_desc, const double* node_weight,
                     const int32_t* node_word, const uint8_t* desc, int n, int levelsup, int weighting, int norm, int32_t* bow_word,
                     double* bow_value, int32_t* n_words, int32_t* fv_node, int32_t* fv_start, int32_t* fv_feat, int32_t* n_fv_nodes) {
  std::map<unsigned, double> v;
  std::map<unsigned, std::vector<unsigned>> fv;
  int status = 0;
  const int nidLevel = L - levelsup;   // :845
  for (int f = 0; f < n; f++) {
    const uint8_t* feature = desc + (size_t)f * 32;
    int finalId = 0, level = 0, nid = -1;
    if (nidLevel <= 0) nid = 0;   // :846
    do {                          // :851-871
      ++level;
      double best = std::numeric_limits<double>::max();
      for (int c = child_begin[finalId], e = child_begin[finalId + 1], parent = finalId; c < e; c++) {
        (void)parent;
        const int id = child_ids[c];
        const double d = (double)yo_descriptor_distance(feature, node_desc + (size_t)id * 32);
        if (d < best) {
          best = d;
          finalId = id;
        }
      }
      if (level == nidLevel) nid = finalId;
    } while (child_begin[finalId + 1] > child_begin[finalId]);
    if (nid < 0) {
      nid = finalId;
      status |= 1;
    }
    const unsigned id = (unsigned)node_word[finalId];
    const double w = node_weight[finalId];
    if (weighting == 0 || weighting == 1) {   // TF_IDF / TF, :769-787
      if (w > 0) {
        auto it = v.lower_bound(id);          // BowVector::addWeight
        if (it != v.end() && !(id < it->first)) it->second += w;
        else v.insert(it, std::make_pair(id, w));
        fv[(unsigned)nid].push_back((unsigned)f);   // FeatureVector::addFeature
      }
    } else {                                  // IDF / BINARY, :798-816
      if (w > 0) {
        auto it = v.lower_bound(id);          // BowVector::addIfNotExist
        if (it == v.end() || id < it->first) v.insert(it, std::make_pair(id, w));
        fv[(unsigned)nid].push_back((unsigned)f);
      }
    }
  }
  if ((weighting == 0 || weighting == 1) && !v.empty() && norm == 0) {   // :789-795
    const double nd = (double)v.size();
    for (auto& kv : v) kv.second /= nd;
  }
  if (norm != 0) {   // BowVector::normalize
    double s = 0.0;
    if (norm == 1) for (auto& kv : v) s += std::fabs(kv.second);
    else {
      for (auto& kv : v) s += kv.second * kv.second;
      s = std::sqrt(s);
    }
    if (s > 0.0) for (auto& kv : v) kv.second /= s;
  }
  int i = 0;
  for (auto& kv : v) { bow_word[i] = (int32_t)kv.first; bow_value[i] = kv.second; i++; }
  *n_words = i;
  int j = 0, at = 0;
  for (auto& kv : fv) {
    fv_node[j] = (int32_t)kv.first;
    fv_start[j] = at;
    for (unsigned idx : kv.second) fv_feat[at++] = (int32_t)idx;
    j++;
  }
  fv_start[j] = at;
  *n_fv_nodes = j;
  return status;
}

}  // extern "C"
