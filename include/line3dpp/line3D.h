// line3dpp/line3D.h -- header-only C++ facade over the C-ABI (include/l3dpp_hip.h) that mirrors the
// public interface of the reference's class L3DPP::Line3D (line3D.h:61-424) for the hot path:
//
//     Line3D(output_folder, load_segments, max_img_width, max_line_segments,
//            neighbors_by_worldpoints=false, use_GPU=true)
//     addImage(camID, image, K, R, t, median_depth, wps_or_neighbors, line_segments)
//     matchImages(sigma_position, sigma_angle, num_neighbors, epipolar_overlap, kNN, const_regularization_depth)
//     computeAffinityMatrix()            // the affinity part of reconstruct3Dlines()
//
// Same names, argument order, defaults (commons.h:40-70) and error behaviour as the reference: errors
// are printed with the "[L3D++] ERROR:" prefix and the call returns (void), no exceptions
// (line3D.cc:119-126, 385-391).  The matrix/vector/image types are template parameters, so Eigen
// (Matrix3d / Vector3d: operator()(r,c), operator()(i)) and OpenCV (cv::Mat: .cols/.rows; cv::Vec4f:
// operator[]) objects can be passed exactly as to the reference without this header depending on
// either library.  After matchImages()/computeAffinityMatrix() the results are available in the
// reference's own container types -- see matches(), estimatedPosition3D(), affinity() -- so the
// reference's clusterSegments()/optimizeClusters() stages can consume them unchanged
// (INTEGRATION.md shows the ten-line patch).
#ifndef L3DPP_HIP_FACADE_LINE3D_H_
#define L3DPP_HIP_FACADE_LINE3D_H_

#include <cstdint>
#include <iostream>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "../l3dpp_hip.h"

namespace L3DPP_HIP {

// defaults, commons.h:40-70
constexpr float L3D_DEF_SCORING_POS_REGULARIZER = 2.5f;
constexpr float L3D_DEF_SCORING_ANG_REGULARIZER = 10.0f;
constexpr unsigned L3D_DEF_MATCHING_NEIGHBORS = 10;
constexpr float L3D_DEF_EPIPOLAR_OVERLAP = 0.25f;
constexpr int L3D_DEF_KNN = 10;

struct ImageSize { int cols, rows; };  // stand-in for cv::Mat when OpenCV is not around

class Line3D {
public:
    Line3D(const std::string& output_folder, const bool load_segments = true, const int max_img_width = -1,
           const unsigned int max_line_segments = 3000, const bool neighbors_by_worldpoints = false,
           const bool use_GPU = true, const int device = 0, void* hip_stream = nullptr)
        : prefix_("[L3D++] "), prefix_err_("[L3D++] ERROR: ") {
        (void)output_folder; (void)load_segments; (void)max_line_segments; (void)use_GPU;
        max_img_width_ = max_img_width;
        // addImage's list is a worldpoint list; neighbours from the worldpoint overlap at every matchImages
        // (Line3D::findVisualNeighborsFromWPs, line3D.cc:578-699 -> l3d_add_view_worldpoints)
        neighbors_by_worldpoints_ = neighbors_by_worldpoints;
        ctx_ = l3d_create(device, hip_stream);
        if (!ctx_) std::cout << prefix_err_ << l3d_last_error() << std::endl;
    }
    ~Line3D() { l3d_destroy(ctx_); }
    Line3D(const Line3D&) = delete;
    Line3D& operator=(const Line3D&) = delete;

    // void Line3D::addImage(...), line3D.h:104-108.  Image: anything with .cols/.rows; Mat3: K(r,c);
    // Vec3: t(i); Seg: s[0..3] (cv::Vec4f).  `line_segments` must be given (LSD detection is outside
    // the accelerated path).  [multithreading safe like the reference]
    template <class Image, class Mat3, class Vec3, class Seg>
    void addImage(const unsigned int camID, const Image& image, const Mat3& K, const Mat3& R, const Vec3& t,
                  const float median_depth, const std::list<unsigned int>& wps_or_neighbors,
                  const std::vector<Seg>& line_segments) {
        double k[9], r[9], tt[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) { k[3 * i + j] = K(i, j); r[3 * i + j] = R(i, j); }
            tt[i] = t(i);
        }
        std::vector<float> segs(4 * line_segments.size());
        for (size_t i = 0; i < line_segments.size(); ++i)
            for (int j = 0; j < 4; ++j) segs[4 * i + j] = line_segments[i][j];
        std::vector<uint32_t> nb(wps_or_neighbors.begin(), wps_or_neighbors.end());
        const int rc = (neighbors_by_worldpoints_ ? l3d_add_view_worldpoints : l3d_add_view)(
            ctx_, camID, segs.data(), (uint32_t)line_segments.size(), k, r, tt, (uint32_t)image.cols, (uint32_t)image.rows,
            median_depth, nb.data(), (uint32_t)nb.size());
        if (rc != L3D_OK) std::cout << prefix_err_ << "view [" << camID << "]: " << l3d_last_error() << std::endl;
        else num_lines_[camID] = (uint32_t)line_segments.size();
    }

    // void Line3D::matchImages(...), line3D.h:143-148
    void matchImages(const float sigma_position = L3D_DEF_SCORING_POS_REGULARIZER,
                     const float sigma_angle = L3D_DEF_SCORING_ANG_REGULARIZER,
                     const unsigned int num_neighbors = L3D_DEF_MATCHING_NEIGHBORS,
                     const float epipolar_overlap = L3D_DEF_EPIPOLAR_OVERLAP, const int kNN = L3D_DEF_KNN,
                     const float const_regularization_depth = -1.0f) {
        std::cout << std::endl << prefix_ << "[2] LINE MATCHING ================================" << std::endl;
        l3d_match_params p{sigma_position, sigma_angle, num_neighbors, epipolar_overlap, kNN, const_regularization_depth};
        const int rc = l3d_match_images(ctx_, &p);
        if (rc != L3D_OK) std::cout << prefix_err_ << l3d_last_error() << std::endl;
    }

    // the affinity part of void Line3D::reconstruct3Dlines(...), line3D.cc:1749-1778
    void computeAffinityMatrix() {
        const int rc = l3d_compute_affinity(ctx_);
        if (rc != L3D_OK) std::cout << prefix_err_ << l3d_last_error() << std::endl;
    }

    // void Line3D::reconstruct3Dlines(...), line3D.h:162-166: affinity matrix, graph clustering, 3D line per
    // cluster, collinear 3D segments; perform_diffusion runs the replicator-dynamics diffusion (performRDD) on the
    // GPU like a CUDA build of the reference; no Ceres (like a reference build without it)
    void reconstruct3Dlines(const unsigned int visibility_t = 3, const bool perform_diffusion = false,
                            const float collinearity_t = -1.0f, const bool use_CERES = false,
                            const unsigned int max_iter_CERES = 250) {
        std::cout << std::endl << prefix_ << "[3] RECONSTRUCTION ===============================" << std::endl;
        if (use_CERES) std::cout << prefix_err_ << "CERES was not found! no optimization will be performed..." << std::endl;
        const int rc = l3d_reconstruct_3d_lines(ctx_, visibility_t, perform_diffusion, collinearity_t, use_CERES, max_iter_CERES);
        if (rc != L3D_OK) std::cout << prefix_err_ << l3d_last_error() << std::endl;
    }

    // void Line3D::get3Dlines(std::vector<FinalLine3D>&), line3D.h:173.  FinalLine3D / LineCluster3D with the reference's member
    // and accessor names (segment3D.h:120-178), so that a consumer written against them compiles unchanged:
    // line.underlyingCluster_.seg3D(), .residuals(), .size(), .reference_view(); the 3D segments and 2D segment ids are the
    // C-ABI's PODs (reference layouts: l3d_segment3d = Segment3D's P1, P2, dir; l3d_segment2d = camID, segID)
    class LineCluster3D {
    public:
        LineCluster3D() : reference_view_(0) {}
        LineCluster3D(const l3d_segment3d& seg3D, const std::list<l3d_segment2d>& residuals, const unsigned int ref_view)
            : seg3D_(seg3D), residuals_(residuals), reference_view_(ref_view) {}
        l3d_segment3d seg3D() const { return seg3D_; }
        const std::list<l3d_segment2d>* residuals() const { return &residuals_; }
        size_t size() const { return residuals_.size(); }
        unsigned int reference_view() const { return reference_view_; }
        void update3Dline(const l3d_segment3d& seg3D) { seg3D_ = seg3D; }
    private:
        l3d_segment3d seg3D_{};
        std::list<l3d_segment2d> residuals_;
        unsigned int reference_view_;
    };
    struct FinalLine3D {
        std::list<l3d_segment3d> collinear3Dsegments_;
        LineCluster3D underlyingCluster_;
    };
    void get3Dlines(std::vector<FinalLine3D>& result) {
        result.clear();
        uint32_t nl = 0, ns = 0, nr = 0;
        if (l3d_num_3d_lines(ctx_, &nl, &ns, &nr) != L3D_OK) return;
        std::vector<uint32_t> so(nl + 1), ro(nl + 1), rv(nl);
        std::vector<l3d_segment3d> segs(ns), cl(nl);
        std::vector<l3d_segment2d> res(nr);
        l3d_get_3d_lines(ctx_, so.data(), segs.data(), ro.data(), res.data(), cl.data(), rv.data());
        result.resize(nl);
        for (uint32_t i = 0; i < nl; ++i) {
            result[i].collinear3Dsegments_.assign(segs.begin() + so[i], segs.begin() + so[i + 1]);
            result[i].underlyingCluster_ = LineCluster3D(cl[i], std::list<l3d_segment2d>(res.begin() + ro[i], res.begin() + ro[i + 1]), rv[i]);
        }
    }

    // void Line3D::save3DLinesAsTXT(const std::string& output_folder), line3D.h:176
    void save3DLinesAsTXT(const std::string& output_folder) {
        if (l3d_save_3d_lines_txt(ctx_, output_folder.c_str(), max_img_width_) != L3D_OK)
            std::cout << prefix_ << "WARNING: " << l3d_last_error() << std::endl;
    }

    // void Line3D::save3DLinesAsBIN(const std::string& output_folder), line3D.h:185
    void save3DLinesAsBIN(const std::string& output_folder) {
        if (l3d_save_3d_lines_bin(ctx_, output_folder.c_str(), max_img_width_) != L3D_OK)
            std::cout << prefix_ << "WARNING: " << l3d_last_error() << std::endl;
    }

    // std::string Line3D::createOutputFilename(), line3D.h:226
    std::string createOutputFilename() {
        char buf[512];
        return l3d_output_filename(ctx_, max_img_width_, buf, sizeof(buf)) == L3D_OK ? std::string(buf) : std::string();
    }
    // getSegmentCoords2D(camID, segID), line3D.h:195-197: (x1, y1, x2, y2)
    struct Coords2D { float v[4]; float operator()(int i) const { return v[i]; } };
    Coords2D getSegmentCoords2D(const unsigned int camID, const unsigned int segID) {
        Coords2D c{};
        l3d_get_segment_coords2d(ctx_, camID, segID, c.v);
        return c;
    }

    // saveResultAsSTL / saveResultAsOBJ, line3D.h:174-175
    void saveResultAsSTL(const std::string& output_folder) {
        if (l3d_save_result_stl(ctx_, output_folder.c_str(), max_img_width_) != L3D_OK)
            std::cout << prefix_ << "WARNING: " << l3d_last_error() << std::endl;
    }
    void saveResultAsOBJ(const std::string& output_folder) {
        if (l3d_save_result_obj(ctx_, output_folder.c_str(), max_img_width_) != L3D_OK)
            std::cout << prefix_ << "WARNING: " << l3d_last_error() << std::endl;
    }

    size_t numImages() const { return num_lines_.size(); }

    // matches_[camID] rebuilt in the reference's container type (line3D.h:348)
    std::vector<std::list<l3d_match>> matches(const unsigned int camID) {
        std::vector<std::list<l3d_match>> out;
        auto f = num_lines_.find(camID);
        if (f == num_lines_.end()) return out;
        uint64_t n = 0;
        std::vector<uint32_t> off(f->second + 1);
        if (l3d_get_matches(ctx_, camID, nullptr, 0, off.data(), &n) != L3D_OK) return out;
        std::vector<l3d_match> flat(n);
        if (n) l3d_get_matches(ctx_, camID, flat.data(), n, off.data(), &n);
        out.resize(f->second);
        for (uint32_t s = 0; s < f->second; ++s) out[s].assign(flat.begin() + off[s], flat.begin() + off[s + 1]);
        return out;
    }

    // estimated_position3D_ + entry_map_ (line3D.h:352-356): (Segment2D, Segment3D members, best Match)
    struct Hypothesis { l3d_segment2d seg2D; l3d_segment3d seg3D; l3d_match match; };
    std::vector<Hypothesis> estimatedPosition3D() {
        uint32_t n = 0;
        l3d_num_best(ctx_, &n);
        std::vector<l3d_segment2d> a(n); std::vector<l3d_segment3d> b(n); std::vector<l3d_match> m(n);
        std::vector<Hypothesis> out(n);
        if (n && l3d_get_best(ctx_, a.data(), b.data(), m.data()) == L3D_OK)
            for (uint32_t i = 0; i < n; ++i) out[i] = Hypothesis{a[i], b[i], m[i]};
        return out;
    }

    // A_ (std::list<CLEdge>), local2global_, as clusterSegments() consumes them (line3D.cc:2079-2090)
    void affinity(std::list<l3d_cledge>& A, std::map<int, l3d_segment2d>& local2global) {
        A.clear(); local2global.clear();
        uint32_t ne = 0, nr = 0;
        if (l3d_num_affinity(ctx_, &ne, &nr) != L3D_OK) return;
        std::vector<l3d_cledge> e(ne); std::vector<l3d_segment2d> l(nr);
        float msdl = 0;
        l3d_get_affinity(ctx_, e.data(), l.data(), &msdl);
        A.assign(e.begin(), e.end());
        for (uint32_t i = 0; i < nr; ++i) local2global[(int)i] = l[i];
    }

    l3d_ctx* handle() { return ctx_; }

private:
    l3d_ctx* ctx_ = nullptr;
    int max_img_width_ = -1;
    bool neighbors_by_worldpoints_ = false;
    std::map<unsigned int, uint32_t> num_lines_;
    std::string prefix_, prefix_err_;
};

}  // namespace L3DPP_HIP

#endif  // L3DPP_HIP_FACADE_LINE3D_H_
