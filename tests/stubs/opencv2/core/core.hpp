// TEST INFRASTRUCTURE -- stand-in for the sliver of OpenCV's core module that include/orbslam3_shim.hpp touches
// (cv::Mat as a byte matrix, cv::KeyPoint with the real 28-byte layout, Input/OutputArray proxies).  Declarations with
// just enough behaviour for the shim's toy-map test; not OpenCV.
#pragma once

#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct Rect { int x, y, width, height; Rect(int a, int b, int w, int h) : x(a), y(b), width(w), height(h) {} };

struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int id = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(id) {}
};

class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, 0); }
    void create(int r, int c, int /*type*/) { buf_ = std::make_shared<std::vector<unsigned char> >((size_t)r * c, 0); rows = r; cols = c; step = (size_t)c; data = buf_->data(); }
    void release() { buf_.reset(); rows = cols = 0; data = nullptr; step = 0; }
    int type() const { return CV_8UC1; }
    bool empty() const { return rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols; }
    Mat clone() const { Mat m(rows, cols, 0); for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols); return m; }
    Mat operator()(const Rect& q) const { Mat m; m.buf_ = buf_; m.rows = q.height; m.cols = q.width; m.step = step; m.data = data + (size_t)q.y * step + q.x; return m; }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }

private:
    std::shared_ptr<std::vector<unsigned char> > buf_;
};

class _InputArray {
public:
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }

protected:
    Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) : _InputArray(m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
