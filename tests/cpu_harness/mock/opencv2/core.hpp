// TEST-ONLY declarations (no definitions) of the handful of OpenCV names the adapter headers use, so that
// `g++ -fsyntax-only` can type-check include/ydorb/*.hpp in a container without OpenCV.  Never linked, never shipped.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) ((void)(x))
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Rect { Rect(int, int, int, int); };
struct MatExpr;
struct _OutputArray;
struct Mat {
  Mat(); Mat(int, int, int); Mat(const MatExpr&);
  int rows, cols; unsigned char* data; size_t step;
  int type() const; bool empty() const; size_t total() const; bool isContinuous() const; Mat clone() const;
  Mat rowRange(int, int) const; Mat colRange(int, int) const; Mat col(int) const; Mat row(int) const; MatExpr t() const;
  Mat operator()(const Rect&) const;
  template <class T> T& at(int); template <class T> const T& at(int) const;
  template <class T> T& at(int, int); template <class T> const T& at(int, int) const;
  template <class T> T* ptr(); template <class T> const T* ptr() const;
  void copyTo(Mat) const;
  void create(int, int, int);
  double dot(const Mat&) const;
  void copyTo(const _OutputArray&) const;
  static MatExpr eye(int, int, int);
};
struct MatExpr { operator Mat() const; template <class T> T& at(int); };
MatExpr operator*(const Mat&, const Mat&); MatExpr operator*(const MatExpr&, const Mat&); MatExpr operator+(const MatExpr&, const Mat&);
MatExpr operator-(const Mat&); MatExpr operator-(const Mat&, const Mat&); MatExpr operator-(const MatExpr&); MatExpr operator/(const Mat&, double);
double norm(const MatExpr&); double norm(const Mat&);
struct _InputArray { _InputArray(const Mat&); bool empty() const; Mat getMat() const; };
struct _OutputArray { _OutputArray(Mat&); void release() const; };
typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray;
inline void copyHelper(const Mat&, OutputArray) {}
}  // namespace cv
