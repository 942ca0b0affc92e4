// oracle/refbuild/include/opencv2/opencv.hpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A minimal stand-in for the slice of OpenCV 2.4 that the reference's hot-path sources use, so that the REAL
// reference sources (/root/reference/core/{cnn_softam.h,maxloss.h,types.h,Hypothesis.*,properties.*,thread_rand.*,
// util.*,lua_calls.h}) compile where they lie into oracle/_ref/libdsac_ref.so (see oracle/refbuild/Makefile).
// OpenCV itself is an un-vendored third-party dependency (core/CMakeLists.txt:19 find_package(OpenCV REQUIRED)).
//
// What is emulated faithfully, because the reference relies on it:
//   * cv::Mat header/buffer semantics: copy construction and Mat = Mat SHARE the buffer; row()/col()/rowRange()/
//     colRange() are views; assigning an expression (a*b, -a, a.t(), zeros(), ...) to a Mat that already has the
//     same size and type writes INTO its buffer (that is how `rmat.row(1) = -rmat.row(1)` (types.h:193) and
//     `dPdR.row(0).colRange(0,3) = ...` (cnn_softam.h:499) work), otherwise it re-allocates;
//   * Mat_<T> converts element types on construction/assignment from a Mat of another type (maxloss.h:41);
//   * at<T>(r, c) is unchecked pointer arithmetic (Hypothesis.cpp:283 reads a 3x1 vector as (0,1), (0,2));
//   * OutputArray binds const Mat& (cnn_softam.h:508 writes the rotation matrix back through a const reference);
//   * Vec/Point conversions use saturate_cast (round-half-even for float -> short; cnn_softam.h:265).
// What is NOT OpenCV's code: Rodrigues, projectPoints, solvePnP(P3P / ITERATIVE with guess) forward to the
// restated algorithms of oracle/cvlike.h (PARITY UNPINNED for those three, see that header); inv()/determinant()/
// SVD are plain textbook routines.
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>
#include <climits>
#include <cassert>
#include <iostream>
#include <memory>
#include <vector>
#include <map>
#include <string>
#include <algorithm>
#include <stdexcept>
#include <utility>

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_PI 3.1415926535897932384626433832795
// solvePnP method flags (OpenCV 2.4 calib3d.hpp)
#define CV_ITERATIVE 0
#define CV_EPNP 1
#define CV_P3P 2

namespace cv {

static inline int cvRound(double v) { return (int)std::nearbyint(v); }  // round-half-even in the default FP mode

template <typename T> static inline T saturate_cast(double v) { return (T)v; }
template <> inline uchar saturate_cast<uchar>(double v) { int i = cvRound(v); return (uchar)(i < 0 ? 0 : i > 255 ? 255 : i); }
template <> inline short saturate_cast<short>(double v) { int i = cvRound(v); return (short)(i < SHRT_MIN ? SHRT_MIN : i > SHRT_MAX ? SHRT_MAX : i); }
template <> inline ushort saturate_cast<ushort>(double v) { int i = cvRound(v); return (ushort)(i < 0 ? 0 : i > USHRT_MAX ? USHRT_MAX : i); }
template <> inline int saturate_cast<int>(double v) { return cvRound(v); }
template <> inline signed char saturate_cast<signed char>(double v) { int i = cvRound(v); return (signed char)(i < -128 ? -128 : i > 127 ? 127 : i); }

// ------------------------------------------------------------------------------------------------ Vec
template <typename T, int cn>
struct Vec {
    T val[cn];
    Vec() { for (int i = 0; i < cn; i++) val[i] = T(0); }
    Vec(T a, T b) { static_assert(cn >= 2, ""); for (int i = 0; i < cn; i++) val[i] = T(0); val[0] = a; val[1] = b; }
    Vec(T a, T b, T c) { static_assert(cn >= 3, ""); for (int i = 0; i < cn; i++) val[i] = T(0); val[0] = a; val[1] = b; val[2] = c; }
    Vec(T a, T b, T c, T d) { static_assert(cn >= 4, ""); for (int i = 0; i < cn; i++) val[i] = T(0); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
    T& operator()(int i) { return val[i]; }
    const T& operator()(int i) const { return val[i]; }
    template <typename T2> operator Vec<T2, cn>() const {
        Vec<T2, cn> r;
        for (int i = 0; i < cn; i++) r.val[i] = saturate_cast<T2>((double)val[i]);
        return r;
    }
};
template <typename T, int cn> Vec<T, cn> operator*(const Vec<T, cn>& a, double s) { Vec<T, cn> r; for (int i = 0; i < cn; i++) r.val[i] = saturate_cast<T>(a.val[i] * s); return r; }
template <typename T, int cn> Vec<T, cn> operator*(const Vec<T, cn>& a, int s) { Vec<T, cn> r; for (int i = 0; i < cn; i++) r.val[i] = saturate_cast<T>(a.val[i] * s); return r; }
template <typename T, int cn> Vec<T, cn> operator*(const Vec<T, cn>& a, float s) { Vec<T, cn> r; for (int i = 0; i < cn; i++) r.val[i] = saturate_cast<T>(a.val[i] * s); return r; }
template <typename T, int cn> Vec<T, cn> operator+(const Vec<T, cn>& a, const Vec<T, cn>& b) { Vec<T, cn> r; for (int i = 0; i < cn; i++) r.val[i] = saturate_cast<T>(a.val[i] + b.val[i]); return r; }
template <typename T, int cn> Vec<T, cn> operator-(const Vec<T, cn>& a, const Vec<T, cn>& b) { Vec<T, cn> r; for (int i = 0; i < cn; i++) r.val[i] = saturate_cast<T>(a.val[i] - b.val[i]); return r; }
typedef Vec<uchar, 3> Vec3b;
typedef Vec<short, 3> Vec3s;
typedef Vec<int, 3> Vec3i;
typedef Vec<float, 3> Vec3f;
typedef Vec<double, 3> Vec3d;
typedef Vec<double, 4> Scalar_base;
struct Scalar : public Vec<double, 4> {
    Scalar() {}
    Scalar(double a, double b = 0, double c = 0, double d = 0) : Vec<double, 4>(a, b, c, d) {}
};

// ------------------------------------------------------------------------------------------------ Point
template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename T2> operator Point_<T2>() const { return Point_<T2>(saturate_cast<T2>((double)x), saturate_cast<T2>((double)y)); }
};
template <typename T> Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(saturate_cast<T>(a.x - b.x), saturate_cast<T>(a.y - b.y)); }
template <typename T> Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(saturate_cast<T>(a.x + b.x), saturate_cast<T>(a.y + b.y)); }
template <typename T> double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;

template <typename T>
struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    explicit Point3_(const Vec<T, 3>& v) : x(v[0]), y(v[1]), z(v[2]) {}
    template <typename T2> operator Point3_<T2>() const { return Point3_<T2>(saturate_cast<T2>((double)x), saturate_cast<T2>((double)y), saturate_cast<T2>((double)z)); }
};
template <typename T> Point3_<T> operator-(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>(saturate_cast<T>(a.x - b.x), saturate_cast<T>(a.y - b.y), saturate_cast<T>(a.z - b.z)); }
template <typename T> Point3_<T> operator+(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>(saturate_cast<T>(a.x + b.x), saturate_cast<T>(a.y + b.y), saturate_cast<T>(a.z + b.z)); }
template <typename T> Point3_<T> operator-(const Point3_<T>& a) { return Point3_<T>(saturate_cast<T>(-a.x), saturate_cast<T>(-a.y), saturate_cast<T>(-a.z)); }
template <typename T> Point3_<T>& operator+=(Point3_<T>& a, const Point3_<T>& b) { a.x = saturate_cast<T>(a.x + b.x); a.y = saturate_cast<T>(a.y + b.y); a.z = saturate_cast<T>(a.z + b.z); return a; }
template <typename T> Point3_<T>& operator-=(Point3_<T>& a, const Point3_<T>& b) { a.x = saturate_cast<T>(a.x - b.x); a.y = saturate_cast<T>(a.y - b.y); a.z = saturate_cast<T>(a.z - b.z); return a; }
template <typename T> Point3_<T>& operator*=(Point3_<T>& a, double s) { a.x = saturate_cast<T>(a.x * s); a.y = saturate_cast<T>(a.y * s); a.z = saturate_cast<T>(a.z * s); return a; }
template <typename T> Point3_<T> operator*(const Point3_<T>& a, double s) { return Point3_<T>(saturate_cast<T>(a.x * s), saturate_cast<T>(a.y * s), saturate_cast<T>(a.z * s)); }
template <typename T> double norm(const Point3_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z); }
typedef Point3_<int> Point3i;
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size& o) const { return !(*this == o); }
};

// ------------------------------------------------------------------------------------------------ type traits
template <typename T> struct DataType;
template <> struct DataType<uchar> { enum { depth = CV_8U, channels = 1, type = CV_MAKETYPE(CV_8U, 1) }; };
template <> struct DataType<signed char> { enum { depth = CV_8S, channels = 1, type = CV_MAKETYPE(CV_8S, 1) }; };
template <> struct DataType<ushort> { enum { depth = CV_16U, channels = 1, type = CV_MAKETYPE(CV_16U, 1) }; };
template <> struct DataType<short> { enum { depth = CV_16S, channels = 1, type = CV_MAKETYPE(CV_16S, 1) }; };
template <> struct DataType<int> { enum { depth = CV_32S, channels = 1, type = CV_MAKETYPE(CV_32S, 1) }; };
template <> struct DataType<float> { enum { depth = CV_32F, channels = 1, type = CV_MAKETYPE(CV_32F, 1) }; };
template <> struct DataType<double> { enum { depth = CV_64F, channels = 1, type = CV_MAKETYPE(CV_64F, 1) }; };
template <typename T, int cn> struct DataType<Vec<T, cn>> { enum { depth = DataType<T>::depth, channels = cn, type = CV_MAKETYPE(DataType<T>::depth, cn) }; };
template <typename T> struct DataType<Point_<T>> { enum { depth = DataType<T>::depth, channels = 2, type = CV_MAKETYPE(DataType<T>::depth, 2) }; };
template <typename T> struct DataType<Point3_<T>> { enum { depth = DataType<T>::depth, channels = 3, type = CV_MAKETYPE(DataType<T>::depth, 3) }; };

static inline int depth_size(int depth) {
    switch (depth) { case CV_8U: case CV_8S: return 1; case CV_16U: case CV_16S: return 2; case CV_32S: case CV_32F: return 4; default: return 8; }
}

// ------------------------------------------------------------------------------------------------ Mat
class MatExpr;
class Mat {
public:
    int flags_type;  // CV_MAKETYPE value
    int rows, cols;
    size_t step;     // bytes per row
    uchar* data;
    std::shared_ptr<std::vector<uchar>> buf;

    Mat() : flags_type(0), rows(0), cols(0), step(0), data(nullptr) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(int r, int c, int type, const Scalar& s) : Mat() { create(r, c, type); setTo(s[0]); }
    Mat(Size sz, int type) : Mat() { create(sz.height, sz.width, type); }
    Mat(const Mat&) = default;             // shares the buffer
    Mat& operator=(const Mat&) = default;  // shares the buffer
    inline Mat(const MatExpr& e);
    inline Mat& operator=(const MatExpr& e);
    template <typename T> explicit Mat(const Point3_<T>& p) : Mat() { create(3, 1, DataType<T>::type); at<T>(0, 0) = p.x; at<T>(1, 0) = p.y; at<T>(2, 0) = p.z; }
    template <typename T> explicit Mat(const Point_<T>& p) : Mat() { create(2, 1, DataType<T>::type); at<T>(0, 0) = p.x; at<T>(1, 0) = p.y; }

    int type() const { return flags_type; }
    int depth() const { return CV_MAT_DEPTH(flags_type); }
    int channels() const { return CV_MAT_CN(flags_type); }
    size_t elemSize() const { return (size_t)depth_size(depth()) * channels(); }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return data == nullptr || total() == 0; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return step == (size_t)cols * elemSize() || rows <= 1; }

    // same size and type -> keep the buffer (views stay views); otherwise allocate a fresh zero-initialised one
    void create(int r, int c, int type) {
        if (data && rows == r && cols == c && flags_type == type) return;
        flags_type = type; rows = r; cols = c;
        step = (size_t)c * elemSize();
        // 64 zero bytes of padding: after a failed PnP the reference reads a 1x3 zero matrix as (1,0), (2,0)
        // (types.h:191 on the matrices of cnn_softam.h:68-69), i.e. past the buffer.  Real OpenCV returns its
        // reference counter / heap bytes there (denormal garbage); here those reads are deterministic zeros.
        buf = std::make_shared<std::vector<uchar>>(step * r + 64, 0);
        data = (r > 0 && c > 0) ? buf->data() : nullptr;
    }
    template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }

    // generic scalar access (channel 0 granularity: every channel is one "column" of width channels())
    double getd(int r, int j) const {  // j indexes scalars along the row (col * channels + ch)
        const uchar* p = data + (size_t)r * step + (size_t)j * depth_size(depth());
        switch (depth()) {
            case CV_8U: return *p; case CV_8S: return *(const signed char*)p; case CV_16U: return *(const ushort*)p;
            case CV_16S: return *(const short*)p; case CV_32S: return *(const int*)p; case CV_32F: return *(const float*)p;
            default: return *(const double*)p;
        }
    }
    void setd(int r, int j, double v) {
        uchar* p = data + (size_t)r * step + (size_t)j * depth_size(depth());
        switch (depth()) {
            case CV_8U: *p = saturate_cast<uchar>(v); break; case CV_8S: *(signed char*)p = saturate_cast<signed char>(v); break;
            case CV_16U: *(ushort*)p = saturate_cast<ushort>(v); break; case CV_16S: *(short*)p = saturate_cast<short>(v); break;
            case CV_32S: *(int*)p = saturate_cast<int>(v); break; case CV_32F: *(float*)p = (float)v; break;
            default: *(double*)p = v;
        }
    }
    int scalars_per_row() const { return cols * channels(); }
    void setTo(double v) { for (int r = 0; r < rows; r++) for (int j = 0; j < scalars_per_row(); j++) setd(r, j, v); }

    Mat view(int r0, int r1, int c0, int c1) const {
        Mat m(*this);
        m.rows = r1 - r0; m.cols = c1 - c0;
        m.data = data + (size_t)r0 * step + (size_t)c0 * elemSize();
        return m;
    }
    Mat row(int r) const { return view(r, r + 1, 0, cols); }
    Mat col(int c) const { return view(0, rows, c, c + 1); }
    Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }

    // element-wise copy with depth conversion into an existing header of the same shape
    void copy_elems_to(Mat& dst) const {
        assert(dst.rows == rows && dst.cols == cols && dst.channels() == channels());
        if (dst.flags_type == flags_type) {
            if (dst.data == data && dst.step == step) return;
            // overlapping views are never produced by the reference code paths compiled here
            for (int r = 0; r < rows; r++) std::memmove(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * elemSize());
        } else {
            for (int r = 0; r < rows; r++) for (int j = 0; j < scalars_per_row(); j++) dst.setd(r, j, getd(r, j));
        }
    }
    Mat clone() const { Mat m; m.create(rows, cols, flags_type); if (!empty()) copy_elems_to(m); return m; }
    void copyTo(Mat& dst) const { dst.create(rows, cols, flags_type); if (!empty()) copy_elems_to(dst); }
    void copyTo(Mat&& dst) const { dst.create(rows, cols, flags_type); if (!empty()) copy_elems_to(dst); }
    void convertTo(Mat& dst, int rtype) const {
        const int t = CV_MAKETYPE(CV_MAT_DEPTH(rtype), channels());
        if (t == flags_type) { if (&dst != this) copyTo(dst); return; }
        Mat tmp; tmp.create(rows, cols, t);
        if (!empty()) copy_elems_to(tmp);
        if (dst.data && dst.rows == rows && dst.cols == cols && dst.flags_type == t) tmp.copy_elems_to(dst); else dst = tmp;
    }

    inline MatExpr t() const;
    inline MatExpr inv() const;
    static inline MatExpr zeros(int r, int c, int type);
    static inline MatExpr zeros(Size s, int type);
    static inline MatExpr ones(int r, int c, int type);
    static inline MatExpr eye(int r, int c, int type);

    inline Mat& operator+=(const Mat& b);
    inline Mat& operator-=(const Mat& b);
    inline Mat& operator*=(double s);
};

// An evaluated expression: the only difference to Mat is what assignment does with it (see Mat::operator=).
class MatExpr : public Mat {
public:
    MatExpr() {}
    explicit MatExpr(const Mat& m) : Mat(m) {}
};
inline Mat::Mat(const MatExpr& e) : Mat(static_cast<const Mat&>(e)) {}
inline Mat& Mat::operator=(const MatExpr& e) {
    if (data && rows == e.rows && cols == e.cols && flags_type == e.flags_type) { if (!e.empty()) e.copy_elems_to(*this); }
    else *this = static_cast<const Mat&>(e);
    return *this;
}

static inline MatExpr make_filled(int r, int c, int type, double v, bool eye) {
    Mat m; m.create(r, c, type);
    if (v != 0 && !eye) m.setTo(v);
    if (eye) for (int i = 0; i < std::min(r, c); i++) m.setd(i, i * m.channels(), v);
    return MatExpr(m);
}
inline MatExpr Mat::zeros(int r, int c, int type) { return make_filled(r, c, type, 0, false); }
inline MatExpr Mat::zeros(Size s, int type) { return make_filled(s.height, s.width, type, 0, false); }
inline MatExpr Mat::ones(int r, int c, int type) { return make_filled(r, c, type, 1, false); }
inline MatExpr Mat::eye(int r, int c, int type) { return make_filled(r, c, type, 1, true); }

// ---- arithmetic (single-channel real matrices; CV_32F and CV_64F keep their type, accumulate in double) ----
static inline void check_fp(const Mat& a) { if (a.channels() != 1 || (a.depth() != CV_32F && a.depth() != CV_64F)) throw std::runtime_error("mini-cv: arithmetic on a non-float matrix"); }
static inline MatExpr operator*(const Mat& a, const Mat& b) {
    check_fp(a); check_fp(b);
    if (a.cols != b.rows || a.type() != b.type()) throw std::runtime_error("mini-cv: gemm shape/type mismatch");
    Mat c; c.create(a.rows, b.cols, a.type());
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < b.cols; j++) {
        double s = 0;
        for (int k = 0; k < a.cols; k++) s += a.getd(i, k) * b.getd(k, j);
        c.setd(i, j, s);
    }
    return MatExpr(c);
}
template <typename F> static inline MatExpr zip(const Mat& a, const Mat& b, F f) {
    if (a.rows != b.rows || a.cols != b.cols || a.type() != b.type()) throw std::runtime_error("mini-cv: element-wise shape/type mismatch");
    Mat c; c.create(a.rows, a.cols, a.type());
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.scalars_per_row(); j++) c.setd(i, j, f(a.getd(i, j), b.getd(i, j)));
    return MatExpr(c);
}
template <typename F> static inline MatExpr map1(const Mat& a, F f) {
    Mat c; c.create(a.rows, a.cols, a.type());
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.scalars_per_row(); j++) c.setd(i, j, f(a.getd(i, j)));
    return MatExpr(c);
}
static inline MatExpr operator+(const Mat& a, const Mat& b) { return zip(a, b, [](double x, double y) { return x + y; }); }
static inline MatExpr operator-(const Mat& a, const Mat& b) { return zip(a, b, [](double x, double y) { return x - y; }); }
static inline MatExpr operator-(const Mat& a) { return map1(a, [](double x) { return -x; }); }
static inline MatExpr operator*(const Mat& a, double s) { return map1(a, [s](double x) { return x * s; }); }
static inline MatExpr operator*(double s, const Mat& a) { return map1(a, [s](double x) { return x * s; }); }
static inline MatExpr operator/(const Mat& a, double s) { return map1(a, [s](double x) { return x / s; }); }
static inline MatExpr operator!=(const Mat& a, const Mat& b) {
    if (a.rows != b.rows || a.cols != b.cols || a.type() != b.type()) throw std::runtime_error("mini-cv: compare shape/type mismatch");
    Mat c; c.create(a.rows, a.cols, CV_MAKETYPE(CV_8U, a.channels()));
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.scalars_per_row(); j++) c.setd(i, j, a.getd(i, j) != b.getd(i, j) ? 255 : 0);
    return MatExpr(c);
}
inline Mat& Mat::operator+=(const Mat& b) { zip(*this, b, [](double x, double y) { return x + y; }).copy_elems_to(*this); return *this; }
inline Mat& Mat::operator-=(const Mat& b) { zip(*this, b, [](double x, double y) { return x - y; }).copy_elems_to(*this); return *this; }
inline Mat& Mat::operator*=(double s) { map1(*this, [s](double x) { return x * s; }).copy_elems_to(*this); return *this; }
// (the member operators are also what runs for temporaries: jacobean.colRange(a, b) += ...  cnn_softam.h:641)

inline MatExpr Mat::t() const {
    Mat c; c.create(cols, rows, flags_type);
    const int cn = channels();
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) for (int k = 0; k < cn; k++) c.setd(j, i * cn + k, getd(i, j * cn + k));
    return MatExpr(c);
}

static inline double determinant(const Mat& m) {
    check_fp(m);
    const int n = m.rows;
    if (m.cols != n) throw std::runtime_error("mini-cv: determinant of a non-square matrix");
    std::vector<double> a((size_t)n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) a[(size_t)i * n + j] = m.getd(i, j);
    if (n == 2) return a[0] * a[3] - a[1] * a[2];
    if (n == 3) return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    double det = 1;
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(a[(size_t)r * n + c]) > std::fabs(a[(size_t)p * n + c])) p = r;
        if (a[(size_t)p * n + c] == 0) return 0;
        if (p != c) { for (int j = 0; j < n; j++) std::swap(a[(size_t)p * n + j], a[(size_t)c * n + j]); det = -det; }
        det *= a[(size_t)c * n + c];
        for (int r = c + 1; r < n; r++) {
            const double f = a[(size_t)r * n + c] / a[(size_t)c * n + c];
            for (int j = c; j < n; j++) a[(size_t)r * n + j] -= f * a[(size_t)c * n + j];
        }
    }
    return det;
}

// DECOMP_LU semantics: closed forms for n <= 3 (as cv::invert does), Gauss-Jordan with partial pivoting above;
// a singular matrix gives all zeros.
inline MatExpr Mat::inv() const {
    check_fp(*this);
    const int n = rows;
    if (cols != n) throw std::runtime_error("mini-cv: inv of a non-square matrix");
    Mat out; out.create(n, n, flags_type);
    std::vector<double> a((size_t)n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) a[(size_t)i * n + j] = getd(i, j);
    if (n == 3) {
        const double d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
        if (d != 0) {
            const double id = 1. / d;
            const double t[9] = {(a[4] * a[8] - a[5] * a[7]) * id, (a[2] * a[7] - a[1] * a[8]) * id, (a[1] * a[5] - a[2] * a[4]) * id,
                                 (a[5] * a[6] - a[3] * a[8]) * id, (a[0] * a[8] - a[2] * a[6]) * id, (a[2] * a[3] - a[0] * a[5]) * id,
                                 (a[3] * a[7] - a[4] * a[6]) * id, (a[1] * a[6] - a[0] * a[7]) * id, (a[0] * a[4] - a[1] * a[3]) * id};
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out.setd(i, j, t[i * 3 + j]);
        }
        return MatExpr(out);
    }
    std::vector<double> b((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) b[(size_t)i * n + i] = 1;
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(a[(size_t)r * n + c]) > std::fabs(a[(size_t)p * n + c])) p = r;
        if (std::fabs(a[(size_t)p * n + c]) < 1e-300) return MatExpr(out);
        if (p != c) for (int j = 0; j < n; j++) { std::swap(a[(size_t)p * n + j], a[(size_t)c * n + j]); std::swap(b[(size_t)p * n + j], b[(size_t)c * n + j]); }
        const double id = 1. / a[(size_t)c * n + c];
        for (int j = 0; j < n; j++) { a[(size_t)c * n + j] *= id; b[(size_t)c * n + j] *= id; }
        for (int r = 0; r < n; r++) if (r != c) {
            const double f = a[(size_t)r * n + c];
            if (f != 0) for (int j = 0; j < n; j++) { a[(size_t)r * n + j] -= f * a[(size_t)c * n + j]; b[(size_t)r * n + j] -= f * b[(size_t)c * n + j]; }
        }
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) out.setd(i, j, b[(size_t)i * n + j]);
    return MatExpr(out);
}

static inline Scalar sum(const Mat& m) {
    Scalar s;
    const int cn = m.channels();
    for (int i = 0; i < m.rows; i++) for (int j = 0; j < m.cols; j++) for (int k = 0; k < cn && k < 4; k++) s[k] += m.getd(i, j * cn + k);
    return s;
}
static inline Scalar trace(const Mat& m) {
    Scalar s;
    for (int i = 0; i < std::min(m.rows, m.cols); i++) s[0] += m.getd(i, i * m.channels());
    return s;
}
static inline double norm(const Mat& m) {
    double s = 0;
    for (int i = 0; i < m.rows; i++) for (int j = 0; j < m.scalars_per_row(); j++) s += m.getd(i, j) * m.getd(i, j);
    return std::sqrt(s);
}
static inline std::ostream& operator<<(std::ostream& os, const Mat& m) {
    os << "[";
    for (int i = 0; i < m.rows; i++) {
        for (int j = 0; j < m.scalars_per_row(); j++) os << (j ? ", " : "") << m.getd(i, j);
        os << (i + 1 < m.rows ? ";\n " : "");
    }
    return os << "]";
}

// ------------------------------------------------------------------------------------------------ Mat_<T>
template <typename T>
class Mat_ : public Mat {
public:
    typedef T value_type;
    Mat_() { flags_type = DataType<T>::type; }
    Mat_(int r, int c) { create(r, c, DataType<T>::type); }
    explicit Mat_(Size s) { create(s.height, s.width, DataType<T>::type); }
    Mat_(const Mat_&) = default;
    Mat_& operator=(const Mat_&) = default;
    Mat_(const Mat& m) { assign_converted(m); }
    Mat_(const MatExpr& e) { assign_converted(e); }
    Mat_& operator=(const Mat& m) { assign_converted(m); return *this; }
    Mat_& operator=(const MatExpr& e) {
        if (e.type() == DataType<T>::type) Mat::operator=(e);
        else if (data && rows == e.rows && cols == e.cols) e.copy_elems_to(*this);
        else assign_converted(e);
        return *this;
    }
    T& operator()(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    const T& operator()(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    Mat_ row(int r) const { return Mat_(Mat::row(r), 0); }
    Mat_ col(int c) const { return Mat_(Mat::col(c), 0); }
    Mat_ rowRange(int a, int b) const { return Mat_(Mat::rowRange(a, b), 0); }
    Mat_ colRange(int a, int b) const { return Mat_(Mat::colRange(a, b), 0); }
    Mat_ clone() const { return Mat_(Mat::clone(), 0); }
    // views assigned from expressions write through (cnn_softam.h:499-502)
    static MatExpr zeros(int r, int c) { return Mat::zeros(r, c, DataType<T>::type); }
    static MatExpr zeros(Size s) { return Mat::zeros(s, DataType<T>::type); }
    static MatExpr ones(int r, int c) { return Mat::ones(r, c, DataType<T>::type); }
    static MatExpr eye(int r, int c) { return Mat::eye(r, c, DataType<T>::type); }

private:
    Mat_(const Mat& same_type, int) : Mat(same_type) {}
    void assign_converted(const Mat& m) {
        if (m.type() == DataType<T>::type || m.data == nullptr) { Mat::operator=(m); flags_type = DataType<T>::type; return; }
        if (m.channels() != DataType<T>::channels) throw std::runtime_error("mini-cv: Mat_ channel mismatch");
        Mat tmp; tmp.create(m.rows, m.cols, DataType<T>::type);
        m.copy_elems_to(tmp);
        Mat::operator=(tmp);
    }
};

// ------------------------------------------------------------------------------------------------ arrays as arguments
// OutputArray binds Mat& and (as in OpenCV 2.4) const Mat&.
class OutputArray {
public:
    Mat* m;
    OutputArray() : m(nullptr) {}
    OutputArray(Mat& x) : m(&x) {}
    OutputArray(const Mat& x) : m(const_cast<Mat*>(&x)) {}
    bool needed() const { return m != nullptr; }
};
static inline OutputArray noArray() { return OutputArray(); }

class SVD {
public:
    Mat u, w, vt;
    SVD() {}
    explicit SVD(const Mat& a) { compute(a); }
    // one-sided Jacobi (Hestenes) on A^T, singular values sorted descending; A = u * diag(w) * vt
    void compute(const Mat& a) {
        check_fp(a);
        const int m = a.rows, n = a.cols;
        if (m < n) throw std::runtime_error("mini-cv: SVD expects rows >= cols");
        std::vector<double> U((size_t)m * n), V((size_t)n * n, 0.0), W(n);
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) U[(size_t)i * n + j] = a.getd(i, j);
        for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1;
        for (int sweep = 0; sweep < 60; sweep++) {
            bool changed = false;
            for (int p = 0; p < n - 1; p++) for (int q = p + 1; q < n; q++) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < m; i++) { al += U[(size_t)i * n + p] * U[(size_t)i * n + p]; be += U[(size_t)i * n + q] * U[(size_t)i * n + q]; ga += U[(size_t)i * n + p] * U[(size_t)i * n + q]; }
                if (std::fabs(ga) <= 1e-300 || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
                changed = true;
                const double zeta = (be - al) / (2 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double c = 1 / std::sqrt(1 + t * t), s = c * t;
                for (int i = 0; i < m; i++) { const double x = U[(size_t)i * n + p], y = U[(size_t)i * n + q]; U[(size_t)i * n + p] = c * x - s * y; U[(size_t)i * n + q] = s * x + c * y; }
                for (int i = 0; i < n; i++) { const double x = V[(size_t)i * n + p], y = V[(size_t)i * n + q]; V[(size_t)i * n + p] = c * x - s * y; V[(size_t)i * n + q] = s * x + c * y; }
            }
            if (!changed) break;
        }
        std::vector<int> order(n);
        for (int j = 0; j < n; j++) { double s = 0; for (int i = 0; i < m; i++) s += U[(size_t)i * n + j] * U[(size_t)i * n + j]; W[j] = std::sqrt(s); order[j] = j; }
        std::sort(order.begin(), order.end(), [&](int x, int y) { return W[x] > W[y]; });
        u.create(m, n, a.type()); w.create(n, 1, a.type()); vt.create(n, n, a.type());
        for (int jj = 0; jj < n; jj++) {
            const int j = order[jj];
            w.setd(jj, 0, W[j]);
            for (int i = 0; i < m; i++) u.setd(i, jj, W[j] > 0 ? U[(size_t)i * n + j] / W[j] : 0);
            for (int i = 0; i < n; i++) vt.setd(jj, i, V[(size_t)i * n + j]);
        }
    }
};

// ------------------------------------------------------------------------------------------------ calib3d stand-ins
// Implemented in oracle/refbuild/minicv_calib.cpp on top of oracle/cvlike.h.
void Rodrigues(const Mat& src, OutputArray dst, OutputArray jacobian = OutputArray());
void projectPoints(const std::vector<Point3f>& objectPoints, const Mat& rvec, const Mat& tvec, const Mat& cameraMatrix, const Mat& distCoeffs,
                   std::vector<Point2f>& imagePoints);
bool solvePnP(const std::vector<Point3f>& objectPoints, const std::vector<Point2f>& imagePoints, const Mat& cameraMatrix, const Mat& distCoeffs,
              Mat& rvec, Mat& tvec, bool useExtrinsicGuess = false, int flags = CV_ITERATIVE);

}  // namespace cv
