// Value types of the reference's public signatures (cv::Mat_<T>, cv::Rect, cv::Point_, cv::Size,
// cv::KeyPoint) for hosts built without OpenCV.  Define SLAM_HOST_WITH_OPENCV to use the real ones:
// the wrappers in slam_host.hpp only touch rows / cols / step / data / ROI / clone, which both agree on.
#pragma once

#ifdef SLAM_HOST_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else

#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
    bool operator==(const Point_& o) const { return x == o.x && y == o.y; }
    bool operator!=(const Point_& o) const { return !(*this == o); }
};
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point2i Point;

struct Vec3f {
    float val[3];
    Vec3f() : val{0.f, 0.f, 0.f} {}
    Vec3f(float a, float b, float c) : val{a, b, c} {}
    float operator[](int i) const { return val[i]; }
    float& operator[](int i) { return val[i]; }
};

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

// 2-D array with a row pitch; copies share the buffer, operator()(Rect) is a view (like cv::Mat_).
template <typename T> class Mat_ {
public:
    int rows, cols;
    size_t step;      // bytes between rows
    T* data;
    Mat_() : rows(0), cols(0), step(0), data(nullptr) {}
    Mat_(int r, int c) { create(r, c); }
    Mat_(int r, int c, const T& v) { create(r, c); for (int i = 0; i < r * c; i++) data[i] = v; }
    explicit Mat_(Size s) { create(s.height, s.width); }
    Mat_(int r, int c, T* ext, size_t step_bytes) : rows(r), cols(c), step(step_bytes), data(ext) {}   // caller-owned
    void create(int r, int c) {
        rows = r; cols = c; step = (size_t)c * sizeof(T);
        buf_.reset(new std::vector<T>((size_t)r * c));
        data = buf_->data();
    }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    Size size() const { return Size(cols, rows); }
    T* ptr(int r) { return reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(data) + (size_t)r * step); }
    const T* ptr(int r) const { return reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(data) + (size_t)r * step); }
    T& operator()(int r, int c) { return ptr(r)[c]; }
    const T& operator()(int r, int c) const { return ptr(r)[c]; }
    Mat_ operator()(const Rect& roi) const {
        Mat_ v;
        v.rows = roi.height; v.cols = roi.width; v.step = step; v.buf_ = buf_;
        v.data = const_cast<T*>(ptr(roi.y)) + roi.x;
        return v;
    }
    Mat_ clone() const {
        Mat_ c(rows, cols);
        for (int r = 0; r < rows; r++) std::memcpy(c.ptr(r), ptr(r), (size_t)cols * sizeof(T));
        return c;
    }
private:
    std::shared_ptr<std::vector<T>> buf_;
};

struct Mat;   // only ever passed as a (NULL) pointer: Feature::track's debug display argument

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

}  // namespace cv
#endif
