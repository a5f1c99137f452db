// TEST INFRASTRUCTURE: cv::imread for the reference's front ends: the image a name stands for is 640 x 480 when a file of
// that name exists (the tests create empty files), empty otherwise -- only the size is ever looked at
#ifndef L3D_REF_SHIM_FRONT_HIGHGUI_
#define L3D_REF_SHIM_FRONT_HIGHGUI_
#include <string>
#include <sys/stat.h>
#include "opencv2/core.hpp"
#ifndef CV_LOAD_IMAGE_GRAYSCALE
#define CV_LOAD_IMAGE_GRAYSCALE 0
#endif
namespace cv {
inline Mat imread(const std::string& name, int = 0) {
    struct stat st;
    if (::stat(name.c_str(), &st) != 0) return Mat();
    return Mat(480, 640, 0);
}
}
#endif
