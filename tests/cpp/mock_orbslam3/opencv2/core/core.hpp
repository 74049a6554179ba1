// TEST INFRASTRUCTURE: the sliver of cv:: the integration glue uses (see ../../README.md).  cv::Mat here is a small dense matrix with value
// semantics (CV_32F or CV_8U); float products accumulate in double and round once (OpenCV's small-matrix GEMM does the same).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cstdlib>
#include <utility>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) do { if (!(expr)) std::abort(); } while (0)
namespace cv {
enum { DECOMP_LU = 0, DECOMP_SVD = 1 };
struct Rect { int x, y, width, height; Rect(int x_ = 0, int y_ = 0, int w_ = 0, int h_ = 0) : x(x_), y(y_), width(w_), height(h_) {} };
struct Point2f { float x, y; Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {} };
struct Point3f { float x, y, z; Point3f(float x_ = 0, float y_ = 0, float z_ = 0) : x(x_), y(y_), z(z_) {} };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;     // bytes per row (dense)
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type), buf_((size_t)r * c * esz(), 0) { data = buf_.data(); step = (size_t)c * esz(); }
    // cv::Mat(rows, cols, type, void* data, size_t step): a header over memory somebody else owns (no allocation, no copy; copies of the header share it)
    Mat(int r, int c, int type, void* ext, size_t step_) : rows(r), cols(c), step(step_), data((unsigned char*)ext), type_(type), ext_(true) {}
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), step(o.step), type_(o.type_), buf_(o.buf_), ext_(o.ext_) { data = ext_ ? o.data : (buf_.empty() ? nullptr : buf_.data()); }
    Mat& operator=(const Mat& o) { rows = o.rows; cols = o.cols; step = o.step; type_ = o.type_; buf_ = o.buf_; ext_ = o.ext_; data = ext_ ? o.data : (buf_.empty() ? nullptr : buf_.data()); return *this; }
    Mat operator()(const Rect& r) const { return block(r.y, r.y + r.height, r.x, r.x + r.width); }   // a COPY of the region (value semantics here)
    void create(int r, int c, int type) { *this = Mat(r, c, type); }
    void release() { *this = Mat(); }
    static Mat eye(int r, int c, int type) { Mat m(r, c, type); for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.f; return m; }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    int type() const { return type_; }
    Mat clone() const { return *this; }
    template <class T> T& at(int r, int c) { return ((T*)data)[(size_t)r * cols + c]; }
    template <class T> const T& at(int r, int c) const { return ((const T*)data)[(size_t)r * cols + c]; }
    template <class T> T& at(int i) { return ((T*)data)[i]; }
    template <class T> const T& at(int i) const { return ((const T*)data)[i]; }
    bool isContinuous() const { return step == (size_t)cols * esz(); }
    unsigned char* ptr(int r = 0) { return data + (size_t)r * step; }
    const unsigned char* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <class T> T* ptr(int r = 0) { return (T*)data + (size_t)r * cols; }
    template <class T> const T* ptr(int r = 0) const { return (const T*)data + (size_t)r * cols; }
    Mat rowRange(int a, int b) const { return block(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return block(0, rows, a, b); }
    Mat row(int r) const { return block(r, r + 1, 0, cols); }
    Mat col(int c) const { return block(0, rows, c, c + 1); }
    Mat t() const { Mat m(cols, rows, CV_32F); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) m.at<float>(c, r) = at<float>(r, c); return m; }
    double dot(const Mat& o) const { double s = 0; for (int i = 0; i < rows * cols; i++) s += (double)at<float>(i) * (double)o.at<float>(i); return s; }
    Mat inv(int method) const {   // inv(cv::DECOMP_SVD) of a well-conditioned n x n block: Gauss-Jordan with partial pivoting in double, rounded once
        (void)method;
        if (rows == 3 && cols == 3) return inv();
        const int n = rows;
        std::vector<double> a((size_t)n * 2 * n, 0.0);
        for (int r = 0; r < n; r++) { for (int c = 0; c < n; c++) a[(size_t)r * 2 * n + c] = at<float>(r, c); a[(size_t)r * 2 * n + n + r] = 1.0; }
        for (int c = 0; c < n; c++) {
            int piv = c;
            for (int r = c + 1; r < n; r++) if (std::fabs(a[(size_t)r * 2 * n + c]) > std::fabs(a[(size_t)piv * 2 * n + c])) piv = r;
            if (piv != c) for (int k = 0; k < 2 * n; k++) std::swap(a[(size_t)c * 2 * n + k], a[(size_t)piv * 2 * n + k]);
            const double d = a[(size_t)c * 2 * n + c];
            for (int k = 0; k < 2 * n; k++) a[(size_t)c * 2 * n + k] /= d;
            for (int r = 0; r < n; r++) {
                if (r == c) continue;
                const double f = a[(size_t)r * 2 * n + c];
                if (f != 0.0) for (int k = 0; k < 2 * n; k++) a[(size_t)r * 2 * n + k] -= f * a[(size_t)c * 2 * n + k];
            }
        }
        Mat m(n, n, CV_32F);
        for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) m.at<float>(r, c) = (float)a[(size_t)r * 2 * n + n + c];
        return m;
    }
    Mat inv() const {   // 3x3 (cofactors in double, rounded once) or a rigid 4x4 [R t; 0 1] (Converter / write-back use)
        if (rows == 3 && cols == 3) {
            const double a = at<float>(0, 0), b = at<float>(0, 1), c = at<float>(0, 2), d = at<float>(1, 0), e = at<float>(1, 1), f = at<float>(1, 2),
                         g = at<float>(2, 0), h = at<float>(2, 1), i = at<float>(2, 2);
            const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
            const double v[9] = {e * i - f * h, c * h - b * i, b * f - c * e, f * g - d * i, a * i - c * g, c * d - a * f, d * h - e * g, b * g - a * h, a * e - b * d};
            Mat m(3, 3, CV_32F);
            for (int k = 0; k < 9; k++) m.at<float>(k / 3, k % 3) = (float)(v[k] / det);
            return m;
        }
        Mat m = eye(4, 4, CV_32F);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m.at<float>(r, c) = at<float>(c, r);
        for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)at<float>(k, r) * (double)at<float>(k, 3); m.at<float>(r, 3) = (float)-s; }
        return m;
    }
private:
    int type_ = CV_32F;
    std::vector<unsigned char> buf_;
    bool ext_ = false;
    size_t esz() const { return type_ == CV_32F ? 4 : 1; }
    Mat block(int r0, int r1, int c0, int c1) const {
        Mat m(r1 - r0, c1 - c0, type_);
        for (int r = r0; r < r1; r++) memcpy(m.data + (size_t)(r - r0) * m.cols * esz(), data + (size_t)r * step + (size_t)c0 * esz(), (size_t)(c1 - c0) * esz());
        return m;
    }
};
inline Mat operator*(const Mat& a, const Mat& b) {
    Mat m(a.rows, b.cols, CV_32F);
    for (int r = 0; r < a.rows; r++) for (int c = 0; c < b.cols; c++) { double s = 0; for (int k = 0; k < a.cols; k++) s += (double)a.at<float>(r, k) * (double)b.at<float>(k, c); m.at<float>(r, c) = (float)s; }
    return m;
}
inline Mat operator+(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows * a.cols; i++) m.at<float>(i) = a.at<float>(i) + b.at<float>(i); return m; }
inline Mat operator-(const Mat& a, const Mat& b) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows * a.cols; i++) m.at<float>(i) = a.at<float>(i) - b.at<float>(i); return m; }
inline Mat operator-(const Mat& a) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows * a.cols; i++) m.at<float>(i) = -a.at<float>(i); return m; }
inline Mat operator*(double s, const Mat& a) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows * a.cols; i++) m.at<float>(i) = (float)(s * (double)a.at<float>(i)); return m; }
inline Mat operator/(const Mat& a, float s) { Mat m(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows * a.cols; i++) m.at<float>(i) = a.at<float>(i) / s; return m; }
inline double norm(const Mat& a) { return std::sqrt(a.dot(a)); }
// InputArray / OutputArray as the extractor's operator() uses them: a view of one Mat (getMat() of the output proxy hands out the Mat itself)
class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat& getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
