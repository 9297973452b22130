// TEST-ONLY declarations of the DBoW3 names include/ydorb/vocabulary.hpp touches (member names as in reference
// thirdParty/DBow3/src/{Vocabulary.h,BowVector.h,FeatureVector.h,ScoringObject.h}), so that g++ -fsyntax-only can type-check the
// adapter without DBoW3 / OpenCV.  Never linked, never shipped.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <opencv2/core.hpp>
namespace DBoW3 {
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
enum LNorm { L1, L2 };
enum WeightingType { TF_IDF, TF, IDF, BINARY };
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
class GeneralScoring { public: virtual bool mustNormalize(LNorm& norm) const = 0; virtual ~GeneralScoring() {} };
class Vocabulary {
 public:
  Vocabulary(int k = 10, int L = 5, WeightingType weighting = TF_IDF, ScoringType scoring = L1_NORM);
  Vocabulary(const std::string& filename);
  virtual ~Vocabulary();
  virtual bool empty() const;
  virtual void transform(const std::vector<cv::Mat>& features, BowVector& v, FeatureVector& fv, int levelsup) const;
  void load(const std::string& filename);
 protected:
  struct Node {
    NodeId id; WordValue weight; std::vector<NodeId> children; NodeId parent; cv::Mat descriptor; WordId word_id;
    bool isLeaf() const { return children.empty(); }
  };
  int m_k, m_L; WeightingType m_weighting; ScoringType m_scoring; GeneralScoring* m_scoring_object;
  std::vector<Node> m_nodes; std::vector<Node*> m_words;
};
}  // namespace DBoW3
