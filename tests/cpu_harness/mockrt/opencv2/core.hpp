// TEST-ONLY stand-in for <opencv2/core.hpp> WITH REAL STORAGE, so that the adapter templates of include/ydorb/*.hpp can be
// instantiated and EXECUTED in a container without OpenCV (tests/cpp_host/adapter_run.cpp).  It implements only the members the
// adapters call, with fixed, documented arithmetic: element types CV_8U and CV_32F, row-major, views share their parent's buffer,
// matrix products accumulate in float in ascending k (one rounded multiply and one rounded add per term; compile with
// -ffp-contract=off), cv::norm accumulates squares in double.  Never linked into the product, never shipped.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) do { if (!(x)) throw std::runtime_error("CV_Assert: " #x); } while (0)
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Rect { int x, y, width, height; Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {} };
struct _OutputArray;
struct Mat {
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;
  int type_ = CV_8U;
  std::shared_ptr<std::vector<unsigned char>> buf;
  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  static size_t esz(int t) { return t == CV_32F ? 4 : 1; }
  void create(int r, int c, int t) {
    if (r == rows && c == cols && t == type_ && data) return;
    rows = r; cols = c; type_ = t; step = (size_t)c * esz(t);
    buf = std::make_shared<std::vector<unsigned char>>((size_t)r * step + 64, 0);
    data = buf->data();
  }
  int type() const { return type_; }
  bool empty() const { return rows == 0 || cols == 0 || !data; }
  size_t total() const { return (size_t)rows * cols; }
  bool isContinuous() const { return step == (size_t)cols * esz(type_); }
  Mat view(int r0, int r1, int c0, int c1) const {
    Mat m; m.rows = r1 - r0; m.cols = c1 - c0; m.type_ = type_; m.step = step; m.buf = buf; m.data = data + (size_t)r0 * step + (size_t)c0 * esz(type_);
    return m;
  }
  Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
  Mat colRange(int a, int b) const { return view(0, rows, a, b); }
  Mat row(int r) const { return view(r, r + 1, 0, cols); }
  Mat col(int c) const { return view(0, rows, c, c + 1); }
  Mat operator()(const Rect& r) const { return view(r.y, r.y + r.height, r.x, r.x + r.width); }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }       // vectors only (as the adapters use it)
  template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  Mat clone() const { Mat m(rows, cols, type_); for (int r = 0; r < rows; r++) std::memcpy(m.ptr<unsigned char>(r), ptr<unsigned char>(r), (size_t)cols * esz(type_)); return m; }
  void copyTo(Mat dst) const {   // dst is a view (e.g. a row of a larger matrix) or an allocated matrix of the same size
    if (dst.rows != rows || dst.cols != cols || dst.type_ != type_) throw std::runtime_error("mock cv::Mat::copyTo: size mismatch");
    for (int r = 0; r < rows; r++) std::memcpy(dst.ptr<unsigned char>(r), ptr<unsigned char>(r), (size_t)cols * esz(type_));
  }
  void copyTo(const _OutputArray& o) const;
  Mat t() const { Mat m(cols, rows, type_); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) m.at<float>(c, r) = at<float>(r, c); return m; }
  double dot(const Mat& o) const { double s = 0; for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) s += (double)at<float>(r, c) * (double)o.at<float>(r, c); return s; }
  static Mat eye(int r, int c, int t) { Mat m(r, c, t); for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.f; return m; }
  void release() { *this = Mat(); }
};
typedef Mat MatExpr;
inline Mat operator*(const Mat& a, const Mat& b) {
  if (a.cols != b.rows) throw std::runtime_error("mock cv::Mat: product size mismatch");
  Mat m(a.rows, b.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < b.cols; j++) {
      float acc = 0.f;
      for (int k = 0; k < a.cols; k++) { const float p = a.at<float>(i, k) * b.at<float>(k, j); acc = acc + p; }
      m.at<float>(i, j) = acc;
    }
  return m;
}
inline Mat operator+(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j); return m; }
inline Mat operator-(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j); return m; }
inline Mat operator-(const Mat& a) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = -a.at<float>(i, j); return m; }
inline Mat operator/(const Mat& a, double d) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = (float)(a.at<float>(i, j) / d); return m; }
inline double norm(const Mat& a) { double s = 0; for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) s += (double)a.at<float>(i, j) * (double)a.at<float>(i, j); return std::sqrt(s); }
struct _InputArray { Mat m; _InputArray(const Mat& m_) : m(m_) {} bool empty() const { return m.empty(); } Mat getMat() const { return m; } };
struct _OutputArray { Mat* m; _OutputArray(Mat& m_) : m(&m_) {} void release() const { m->release(); } };
inline void Mat::copyTo(const _OutputArray& o) const { *o.m = clone(); }
typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray;
}  // namespace cv
