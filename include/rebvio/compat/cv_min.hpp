// Minimal stand-in for the cv::Mat subset rebvio's public API exposes (a ref-counted dense 2-D image container,
// 8UC1 / 32FC1 / 32SC1, ptr<T>(row), at<T>, convertTo with a scale). Used ONLY when <opencv2/core.hpp> is not
// installed; with OpenCV on the include path the real header is used and this file is not compiled.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>

#define CV_8UC1 0
#define CV_32SC1 4
#define CV_32FC1 5

namespace cv {

class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;

  Mat() = default;
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* external, size_t step_bytes = 0)
      : rows(r), cols(c), data(static_cast<unsigned char*>(external)), type_(type) {
    step = step_bytes ? step_bytes : (size_t)c * elem_size(type);
  }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type;
    step = (size_t)c * elem_size(type);
    store_ = std::shared_ptr<unsigned char>(new unsigned char[step * (size_t)r], std::default_delete<unsigned char[]>());
    data = store_.get();
  }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  template <typename T>
  T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <typename T>
  const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
  template <typename T>
  T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <typename T>
  const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  // convertTo(dst, CV_32FC1, alpha): dst = saturate(src * alpha); only u8/f32 -> f32 is needed by rebvio.cpp:43
  void convertTo(Mat& dst, int rtype, double alpha = 1.0) const {
    Mat out(rows, cols, rtype);
    for (int r = 0; r < rows; ++r) {
      float* o = out.ptr<float>(r);
      if (type_ == CV_8UC1) {
        const unsigned char* s = ptr<unsigned char>(r);
        for (int c = 0; c < cols; ++c) o[c] = (float)((double)s[c] * alpha);
      } else {
        const float* s = ptr<float>(r);
        for (int c = 0; c < cols; ++c) o[c] = (float)((double)s[c] * alpha);
      }
    }
    dst = out;
  }
  Mat clone() const {
    Mat out(rows, cols, type_);
    for (int r = 0; r < rows; ++r) std::memcpy(out.data + (size_t)r * out.step, data + (size_t)r * step, out.step);
    return out;
  }

 private:
  static size_t elem_size(int type) { return type == CV_8UC1 ? 1 : 4; }
  int type_ = CV_8UC1;
  std::shared_ptr<unsigned char> store_;
};

}  // namespace cv
