// TEST INFRASTRUCTURE: stand-in for the OpenCV types the reference's headers mention.  With explicit
// line_segments (line3D.cc:174-186) only cv::Mat::cols/rows and cv::Vec4f are ever touched; every image
// processing entry point aborts if reached.
#ifndef L3D_REF_SHIM_OPENCV_
#define L3D_REF_SHIM_OPENCV_
#include <cstdio>
#include <cstdlib>
#include <vector>
#define L3D_SHIM_UNREACHABLE(what) do { std::fprintf(stderr, "[ref_shim] %s is outside the oracle's path\n", what); std::abort(); } while (0)
#define CV_8U 0
#define CV_8UC3 16
#define CV_64FC1 6
#define CV_RGB2GRAY 7
namespace cv {
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Point { int x, y; Point() : x(0), y(0) {} Point(int a, int b) : x(a), y(b) {} template <class A, class B> Point(A a, B b) : x((int)a), y((int)b) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; } };
class Mat {
public:
    int cols, rows;
    Mat() : cols(0), rows(0) {}
    Mat(int r, int c, int, const Scalar& = Scalar()) : cols(c), rows(r) {}
    int type() const { return CV_8U; }
    Mat clone() const { return *this; }
    Size size() const { return Size(cols, rows); }
    template <class T> T& at(int) { L3D_SHIM_UNREACHABLE("cv::Mat::at"); }
    template <class T> T& at(int, int) { L3D_SHIM_UNREACHABLE("cv::Mat::at"); }
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
    static Mat zeros(Size s, int t) { return Mat(s.height, s.width, t); }
};
template <class T> class Mat_ : public Mat { public: static Mat eye(int r, int c) { return Mat(r, c, 0); } static Mat zeros(int r, int c) { return Mat(r, c, 0); } };
struct Vec4f { float v[4]; Vec4f() { v[0] = v[1] = v[2] = v[3] = 0; } Vec4f(float a, float b, float c, float d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
    float& operator()(int i) { return v[i]; } const float& operator()(int i) const { return v[i]; } float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
template <class T> class Ptr { public: Ptr() : p_(0) {} Ptr(T* p) : p_(p) {} T* operator->() const { if (!p_) L3D_SHIM_UNREACHABLE("cv::Ptr"); return p_; } private: T* p_; };
enum { LSD_REFINE_NONE = 0, LSD_REFINE_STD = 1, LSD_REFINE_ADV = 2, INTER_LINEAR = 1, BORDER_CONSTANT = 0 };
class LineSegmentDetector { public: virtual ~LineSegmentDetector() {} virtual void detect(const Mat&, std::vector<Vec4f>&) { L3D_SHIM_UNREACHABLE("LSD"); } };
template <class... A> Ptr<LineSegmentDetector> createLineSegmentDetector(A...) { return Ptr<LineSegmentDetector>(); }
template <class... A> Ptr<LineSegmentDetector> createLineSegmentDetectorPtr(A...) { return Ptr<LineSegmentDetector>(); }
template <class... A> void line(A...) { L3D_SHIM_UNREACHABLE("cv::line"); }
template <class... A> void resize(A...) { L3D_SHIM_UNREACHABLE("cv::resize"); }
template <class... A> void remap(A...) { L3D_SHIM_UNREACHABLE("cv::remap"); }
template <class... A> void initUndistortRectifyMap(A...) { L3D_SHIM_UNREACHABLE("cv::initUndistortRectifyMap"); }
template <class... A> void cvtColor(A...) { L3D_SHIM_UNREACHABLE("cv::cvtColor"); }
}
#endif
