// DECLARATION-ONLY stand-in for the few OpenCV 4.x types the drop-in code of geoflowslam_amd/host/gfs_reference_dropins.hpp touches.
// Test infrastructure for a SYNTAX / layout check (tests/test_host_logic.py::test_reference_dropins_compile, g++ -fsyntax-only):
// nothing here is ever linked or run, and nothing in the product includes it.  Signatures follow OpenCV 4.5's core/mat.hpp, types.hpp.
#pragma once
#include <cstddef>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
template <typename T> struct Point_ { T x, y; Point_(); Point_(T, T); };
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
template <typename T> struct Size_ { T width, height; Size_(); Size_(T, T); };
typedef Size_<int> Size;
class _OutputArray;
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type);
  int type() const;
  bool empty() const;
  Mat rowRange(int start, int end) const;
  void create(int rows, int cols, int type);
  void copyTo(const _OutputArray& dst) const;
  uchar* data;
  int rows, cols;
  struct MatStep { operator size_t() const; } step;
};
class _InputArray {
 public:
  _InputArray(const Mat&);
  bool empty() const;
  Mat getMat(int idx = -1) const;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat&);
  void release() const;
  void create(int rows, int cols, int type) const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
class KeyPoint {
 public:
  KeyPoint();
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
class DMatch {
 public:
  DMatch();
  DMatch(int queryIdx, int trainIdx, int imgIdx, float distance);
  int queryIdx, trainIdx, imgIdx;
  float distance;
};
}  // namespace cv
