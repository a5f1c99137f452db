// TEST INFRASTRUCTURE (oracle/ref_shim_front): a RECORDER with the public interface of L3DPP::Line3D
// (/root/reference/line3D.h:80-173), found first on the include path when the reference's own front ends -- main_vsfm.cpp,
// main_colmap.cpp, main_bundler.cpp -- are compiled in place (oracle/Makefile: _ref/libl3d_ref_front.so, with
// -DLine3D=Line3DFront so that no symbol of the real class is shadowed; their main() renamed in the object file).  Every call the front
// end makes is appended to a log; tests/test_front_ends_pinned.py holds the library's own readers (l3d_io.hip,
// line3dpp_amd/io.py) against that log: what the reference's parsers hand to addImage for a given SfM file.
// rotationFromQ is NOT restated here: it is the reference's own static method, reached through a function pointer the
// test takes from oracle/_ref/libl3d_ref.so (ref_driver.cpp: lo_ref_rotation_from_q).
#ifndef L3D_REF_SHIM_FRONT_LINE3D_H_
#define L3D_REF_SHIM_FRONT_LINE3D_H_
// (the standard headers the real line3D.h brings along and the front ends rely on)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "eigen3/Eigen/Eigen"
#include "opencv2/core.hpp"
#include "commons.h"      // the reference's: L3D_DEF_* defaults, L3D_EPS
#include "segment3D.h"    // the reference's: FinalLine3D

namespace l3d_front {
typedef void (*rotation_fn)(const double q[4], double R[9]);
struct Log {
    std::ostringstream js;   // a JSON array of events, built as the calls come
    bool first;
    rotation_fn rot;
    Log() : first(true), rot(0) {}
    std::ostringstream& begin(const char* what) {
        js << (first ? "" : ",\n") << "{\"call\": \"" << what << "\"";
        first = false;
        js.precision(17);
        return js;
    }
};
inline Log& log() { static Log l; return l; }
inline void put3x3(std::ostringstream& o, const char* key, const Eigen::Matrix3d& M) {
    o << ", \"" << key << "\": [";
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o << (r + c ? ", " : "") << M(r, c);
    o << "]";
}
}  // namespace l3d_front

namespace L3DPP {
class Line3D {
public:
    Line3D(const std::string& output_folder, const bool load_segments = L3D_DEF_LOAD_AND_STORE_SEGMENTS,
           const int max_img_width = L3D_DEF_MAX_IMG_WIDTH, const unsigned int max_line_segments = L3D_DEF_MAX_NUM_SEGMENTS,
           const bool neighbors_by_worldpoints = true, const bool use_GPU = true) {
        l3d_front::log().begin("Line3D") << ", \"output_folder\": \"" << output_folder << "\", \"load_segments\": " << (load_segments ? 1 : 0)
                                         << ", \"max_img_width\": " << max_img_width << ", \"max_line_segments\": " << max_line_segments
                                         << ", \"neighbors_by_worldpoints\": " << (neighbors_by_worldpoints ? 1 : 0)
                                         << ", \"use_GPU\": " << (use_GPU ? 1 : 0) << "}";
    }
    void addImage(const unsigned int camID, cv::Mat& image, const Eigen::Matrix3d& K, const Eigen::Matrix3d& R,
                  const Eigen::Vector3d& t, const float median_depth, const std::list<unsigned int>& wps_or_neighbors,
                  const std::vector<cv::Vec4f>& line_segments = std::vector<cv::Vec4f>()) {
        std::ostringstream& o = l3d_front::log().begin("addImage");
        o << ", \"camID\": " << camID << ", \"cols\": " << image.cols << ", \"rows\": " << image.rows;
        l3d_front::put3x3(o, "K", K); l3d_front::put3x3(o, "R", R);
        o << ", \"t\": [" << t(0) << ", " << t(1) << ", " << t(2) << "], \"median_depth\": " << (double)median_depth << ", \"wps\": [";
        bool f = true;
        for (std::list<unsigned int>::const_iterator it = wps_or_neighbors.begin(); it != wps_or_neighbors.end(); ++it, f = false)
            o << (f ? "" : ", ") << *it;
        o << "], \"n_segments\": " << line_segments.size() << "}";
    }
    static void undistortImage(const cv::Mat& inImg, cv::Mat& outImg, const Eigen::Vector3d& radial_coeffs,
                               const Eigen::Vector2d& tangential_coeffs, const Eigen::Matrix3d& K) {
        std::ostringstream& o = l3d_front::log().begin("undistortImage");
        o << ", \"radial\": [" << radial_coeffs(0) << ", " << radial_coeffs(1) << ", " << radial_coeffs(2) << "], \"tangential\": ["
          << tangential_coeffs(0) << ", " << tangential_coeffs(1) << "]";
        l3d_front::put3x3(o, "K", K);
        o << "}";
        outImg = inImg;
    }
    static Eigen::Matrix3d rotationFromQ(const double Qw, const double Qx, const double Qy, const double Qz) {
        const double q[4] = {Qw, Qx, Qy, Qz};
        double R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (!l3d_front::log().rot) { std::fprintf(stderr, "ref_shim_front: no rotationFromQ set\n"); std::abort(); }
        l3d_front::log().rot(q, R);
        Eigen::Matrix3d M;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M(r, c) = R[3 * r + c];
        return M;
    }
    void matchImages(const float sigma_position = L3D_DEF_SCORING_POS_REGULARIZER, const float sigma_angle = L3D_DEF_SCORING_ANG_REGULARIZER,
                     const unsigned int num_neighbors = L3D_DEF_MATCHING_NEIGHBORS, const float epipolar_overlap = L3D_DEF_EPIPOLAR_OVERLAP,
                     const int kNN = L3D_DEF_KNN, const float const_regularization_depth = -1.0f) {
        l3d_front::log().begin("matchImages") << ", \"sigma_position\": " << (double)sigma_position << ", \"sigma_angle\": " << (double)sigma_angle
                                              << ", \"num_neighbors\": " << num_neighbors << ", \"epipolar_overlap\": " << (double)epipolar_overlap
                                              << ", \"kNN\": " << kNN << ", \"const_regularization_depth\": " << (double)const_regularization_depth << "}";
    }
    void reconstruct3Dlines(const unsigned int visibility_t = L3D_DEF_MIN_VISIBILITY_T, const bool perform_diffusion = L3D_DEF_PERFORM_RDD,
                            const float collinearity_t = L3D_DEF_COLLINEARITY_T, const bool use_CERES = L3D_DEF_USE_CERES,
                            const unsigned int max_iter_CERES = L3D_DEF_CERES_MAX_ITER) {
        l3d_front::log().begin("reconstruct3Dlines") << ", \"visibility_t\": " << visibility_t << ", \"perform_diffusion\": " << (perform_diffusion ? 1 : 0)
                                                     << ", \"collinearity_t\": " << (double)collinearity_t << ", \"use_CERES\": " << (use_CERES ? 1 : 0)
                                                     << ", \"max_iter_CERES\": " << max_iter_CERES << "}";
    }
    void get3Dlines(std::vector<L3DPP::FinalLine3D>& result) { result.clear(); l3d_front::log().begin("get3Dlines") << "}"; }
    void saveResultAsSTL(const std::string& f) { l3d_front::log().begin("saveResultAsSTL") << ", \"folder\": \"" << f << "\"}"; }
    void saveResultAsOBJ(const std::string& f) { l3d_front::log().begin("saveResultAsOBJ") << ", \"folder\": \"" << f << "\"}"; }
    void save3DLinesAsTXT(const std::string& f) { l3d_front::log().begin("save3DLinesAsTXT") << ", \"folder\": \"" << f << "\"}"; }
    void save3DLinesAsBIN(const std::string& f) { l3d_front::log().begin("save3DLinesAsBIN") << ", \"folder\": \"" << f << "\"}"; }
};
}  // namespace L3DPP
#endif
