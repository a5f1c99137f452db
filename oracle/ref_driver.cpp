// =============================================================================
// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C interface (same lo_* names as l3d_oracle.cpp, so oracle/oracle.py can load either library) over the
// REFERENCE'S OWN line3D.cc / view.cc / clustering.cc, compiled in place from /root/reference against the
// thin shim headers in oracle/ref_shim/ (Eigen subset with real arithmetic; Boost/OpenCV stand-ins for code
// that the explicit-segments path never reaches).  Built by oracle/Makefile into oracle/_ref/ (git-ignored;
// no reference source is copied into this repository).
//
// Purpose: pin the restatement (l3d_oracle.cpp) against the reference's real control flow and arithmetic,
// and serve as bench.py's cpu_baseline of kind "reference" (its OpenMP CPU path).
//
// Private members of L3DPP::Line3D are read through the usual test-only access hack; it does not change
// the class layout or the code generated for the reference's translation units.
// =============================================================================
#include <sstream>
#include <iostream>
#define private public
#define protected public
#include "line3D.h"
#undef private
#undef protected

#include <algorithm>
#include <cstdint>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Match40 {  // commons.h:186-203
    uint32_t src_camID_, src_segID_, tgt_camID_, tgt_segID_;
    float overlap_score_, score3D_, depth_p1_, depth_p2_, depth_q1_, depth_q2_;
};
static_assert(sizeof(L3DPP::Match) == 40, "Match layout");
struct CLEdgeOut { int i_, j_; float w_; };

struct Ref {
    L3DPP::Line3D* l3d = nullptr;
    std::map<uint32_t, uint32_t> M;
    std::streambuf* old_cout = nullptr;
    std::ostringstream sink;
};

struct Quiet {  // the reference prints progress to std::cout
    std::streambuf* old;
    std::ostringstream sink;
    Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~Quiet() { std::cout.rdbuf(old); }
};

}  // namespace

extern "C" {

void* lo_create() {
    Quiet q;
    Ref* r = new Ref();
    // output folder only receives an (unused) L3D++_data directory; load_segments=false;
    // neighbors_by_worldpoints=false (explicit neighbour lists); use_GPU=false
    r->l3d = new L3DPP::Line3D("/tmp", false, -1, 3000, false, false);
    return r;
}
// the same instance with neighbors_by_worldpoints=true: addImage's list is a worldpoint list, matchImages finds the
// neighbours itself (Line3D::findVisualNeighborsFromWPs) -- the pin of line3dpp_amd/csrc/l3d_neighbors.hip
void* lo_create_worldpoints() {
    Quiet q;
    Ref* r = new Ref();
    r->l3d = new L3DPP::Line3D("/tmp", false, -1, 3000, true, false);
    return r;
}
// visual_neighbors_[cam] after matchImages, ascending; returns its size
uint32_t lo_get_visual_neighbors(void* p, uint32_t cam, uint32_t* out, uint32_t cap) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    std::map<unsigned int, std::set<unsigned int> >::const_iterator it = l->visual_neighbors_.find(cam);
    if (it == l->visual_neighbors_.end()) return 0;
    uint32_t n = 0;
    for (std::set<unsigned int>::const_iterator s = it->second.begin(); s != it->second.end(); ++s, ++n)
        if (out && n < cap) out[n] = *s;
    return n;
}
void lo_destroy(void* p) { Ref* r = (Ref*)p; { Quiet q; delete r->l3d; } delete r; }
void lo_set_record_scored(void*, int) {}
void lo_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#endif
}
int lo_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
int lo_is_reference() { return 1; }

int lo_add_view(void* p, uint32_t camID, const float* segs4, uint32_t M, const double* K, const double* R,
                const double* t, uint32_t width, uint32_t height, float median_depth, const uint32_t* nbrs,
                uint32_t n_nbrs) {
    Ref* r = (Ref*)p;
    Quiet q;
    Eigen::Matrix3d Km, Rm; Eigen::Vector3d tv;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { Km(i, j) = K[3 * i + j]; Rm(i, j) = R[3 * i + j]; } tv(i) = t[i]; }
    cv::Mat image; image.cols = (int)width; image.rows = (int)height;
    std::list<unsigned int> nb(nbrs, nbrs + n_nbrs);
    std::vector<cv::Vec4f> segs(M);
    for (uint32_t i = 0; i < M; ++i) segs[i] = cv::Vec4f(segs4[4 * i], segs4[4 * i + 1], segs4[4 * i + 2], segs4[4 * i + 3]);
    const size_t before = r->l3d->views_.size();
    r->l3d->addImage(camID, image, Km, Rm, tv, median_depth, nb, segs);
    if (r->l3d->views_.size() == before) return 1;
    r->M[camID] = M;
    return 0;
}

void lo_match_images(void* p, float sigma_p, float sigma_a, uint32_t num_neighbors, float epi_overlap, int kNN,
                     float const_reg_depth) {
    Quiet q;
    ((Ref*)p)->l3d->matchImages(sigma_p, sigma_a, num_neighbors, epi_overlap, kNN, const_reg_depth);
}

// the part of Line3D::reconstruct3Dlines up to the affinity matrix, line3D.cc:1749-1778 -- statement for
// statement, calling the reference's own (private) functions
void lo_compute_affinity(void* p) {
    Quiet q;
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    if (l->estimated_position3D_.size() == 0) return;
    l->translate();
    std::vector<float> scene_depths_lines;
    for (std::map<unsigned int, L3DPP::View*>::const_iterator vit = l->views_.begin(); vit != l->views_.end(); ++vit)
        if (vit->second->median_depth() > L3D_EPS) scene_depths_lines.push_back(vit->second->median_depth());
    if (scene_depths_lines.size() > 0) {
        std::sort(scene_depths_lines.begin(), scene_depths_lines.end());
        l->med_scene_depth_lines_ = scene_depths_lines[scene_depths_lines.size() / 2];
    } else {
        l->med_scene_depth_lines_ = 0.0f;
    }
    l->computingAffinityMatrix();
    l->untranslate();
}

// collinearity_t_ as reconstruct3Dlines sets it (line3D.cc:1725-1726, 1752-1756): the per-view collinear lists
// are (re)computed by the reference's own View::findCollinearSegments (CPU path)
void lo_set_collinearity(void* p, float t) {
    Quiet q;
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    l->collinearity_t_ = t;
    if (t > L3D_EPS) l->findCollinearSegments();
}
uint32_t lo_get_collinear(void* p, uint32_t cam, uint32_t* offsets, uint32_t* idx, uint32_t cap) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    L3DPP::View* v = l->views_[cam];
    uint32_t n = 0;
    const uint32_t M = (uint32_t)v->num_lines();
    for (uint32_t s = 0; s < M; ++s) {
        if (offsets) offsets[s] = n;
        if (l->collinearity_t_ > L3D_EPS) {
            std::list<unsigned int> c = v->collinearSegments(s);
            for (std::list<unsigned int>::const_iterator it = c.begin(); it != c.end(); ++it) { if (idx && n < cap) idx[n] = *it; ++n; }
        }
    }
    if (offsets) offsets[M] = n;
    return n;
}

// ---- accessors (same contracts as l3d_oracle.cpp) ----------------------------------------------------------
uint64_t lo_get_matches(void* p, uint32_t cam, Match40* out, uint64_t cap, uint32_t* offsets) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    const std::vector<std::list<L3DPP::Match> >& rows = l->matches_[cam];
    uint64_t n = 0;
    for (size_t r = 0; r < rows.size(); ++r) {
        if (offsets) offsets[r] = (uint32_t)n;
        for (std::list<L3DPP::Match>::const_iterator it = rows[r].begin(); it != rows[r].end(); ++it) {
            if (out && n < cap) std::memcpy(&out[n], &(*it), 40);
            ++n;
        }
    }
    if (offsets) offsets[rows.size()] = (uint32_t)n;
    return n;
}
uint64_t lo_get_scored(void*, uint32_t, Match40*, uint64_t, uint32_t*) { return 0; }
uint32_t lo_num_best(void* p) { return (uint32_t)((Ref*)p)->l3d->estimated_position3D_.size(); }
// estimated_position3D_ is filled in thread-timing order under OpenMP (line3D.cc:1639-1646): hand it out
// ordered by (camID, segID)
void lo_get_best(void* p, uint32_t* camseg2, double* p1p2dir9, float* length, Match40* m) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    const size_t n = l->estimated_position3D_.size();
    std::vector<size_t> ord(n);
    for (size_t i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
        const L3DPP::Match& ma = l->estimated_position3D_[a].second; const L3DPP::Match& mb = l->estimated_position3D_[b].second;
        return ma.src_camID_ < mb.src_camID_ || (ma.src_camID_ == mb.src_camID_ && ma.src_segID_ < mb.src_segID_); });
    for (size_t i = 0; i < n; ++i) {
        const L3DPP::Segment3D& s = l->estimated_position3D_[ord[i]].first;
        const L3DPP::Match& mm = l->estimated_position3D_[ord[i]].second;
        camseg2[2 * i] = mm.src_camID_; camseg2[2 * i + 1] = mm.src_segID_;
        double* o = p1p2dir9 + 9 * i;
        const Eigen::Vector3d P1 = s.P1(), P2 = s.P2(), d = s.dir();
        o[0] = P1.x(); o[1] = P1.y(); o[2] = P1.z(); o[3] = P2.x(); o[4] = P2.y(); o[5] = P2.z();
        o[6] = d.x(); o[7] = d.y(); o[8] = d.z();
        length[i] = s.length();
        std::memcpy(&m[i], &mm, 40);
    }
}
void lo_view_info(void* p, uint32_t cam, float* k, float* median_depth, double* C3, double* t3) {
    L3DPP::View* v = ((Ref*)p)->l3d->views_[cam];
    *k = v->k(); *median_depth = v->median_depth();
    const Eigen::Vector3d C = v->C(), t = v->t();
    for (int i = 0; i < 3; ++i) { C3[i] = C(i); t3[i] = t(i); }
}
void lo_translation(void* p, double* t3) {
    const Eigen::Vector3d t = ((Ref*)p)->l3d->translation_;
    for (int i = 0; i < 3; ++i) t3[i] = t(i);
}
float lo_med_scene_depth_lines(void* p) { return ((Ref*)p)->l3d->med_scene_depth_lines_; }
uint32_t lo_num_edges(void* p) { return (uint32_t)((Ref*)p)->l3d->A_.size(); }
uint32_t lo_num_rows(void* p) { return (uint32_t)((Ref*)p)->l3d->local2global_.size(); }
void lo_get_affinity(void* p, CLEdgeOut* edges, uint32_t* local2global2) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    size_t i = 0;
    for (std::list<L3DPP::CLEdge>::const_iterator it = l->A_.begin(); it != l->A_.end(); ++it, ++i) {
        edges[i].i_ = it->i_; edges[i].j_ = it->j_; edges[i].w_ = it->w_;
    }
    for (std::map<int, L3DPP::Segment2D>::const_iterator it = l->local2global_.begin(); it != l->local2global_.end(); ++it) {
        local2global2[2 * it->first] = it->second.camID();
        local2global2[2 * it->first + 1] = it->second.segID();
    }
}
uint32_t lo_num_pairs(void* p) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    uint32_t n = 0;
    for (auto& kv : l->matched_) n += (uint32_t)kv.second.size();
    return n / 2;
}
void lo_get_pairs(void*, uint32_t*, uint32_t*) {}
uint64_t lo_pair_tests(void* p) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    uint64_t n = 0;
    for (auto& kv : l->matched_)
        for (unsigned int t : kv.second)
            if (kv.first < t) n += (uint64_t)l->views_[kv.first]->num_lines() * l->views_[t]->num_lines();
    return n;
}
// Line3D::reconstruct3Dlines (graph clustering, no diffusion, no collinearity, no Ceres) and get3Dlines,
// run by the reference's own code.  lines3D_ flattened like include/l3dpp_hip.h: l3d_get_3d_lines.
void lo_reconstruct(void* p, uint32_t visibility_t) {
    Quiet q;
    ((Ref*)p)->l3d->reconstruct3Dlines(visibility_t, false, -1.0f, false, 250);
}
void lo_reconstruct_collin(void* p, uint32_t visibility_t, float collinearity_t) {
    Quiet q;
    ((Ref*)p)->l3d->reconstruct3Dlines(visibility_t, false, collinearity_t, false, 250);
}
// the reference's own writer (Line3D::save3DLinesAsTXT): <folder>/<createOutputFilename()>.txt
void lo_save_txt(void* p, const char* folder) {
    Quiet q;
    ((Ref*)p)->l3d->save3DLinesAsTXT(std::string(folder));
}
void lo_num_lines(void* p, uint32_t* n_lines, uint32_t* n_segments, uint32_t* n_residuals) {
    std::vector<L3DPP::FinalLine3D> r;
    ((Ref*)p)->l3d->get3Dlines(r);
    uint32_t ns = 0, nr = 0;
    for (size_t i = 0; i < r.size(); ++i) { ns += (uint32_t)r[i].collinear3Dsegments_.size(); nr += (uint32_t)r[i].underlyingCluster_.size(); }
    *n_lines = (uint32_t)r.size(); *n_segments = ns; *n_residuals = nr;
}
// segments: 9 doubles (P1,P2,dir) each; cluster_lines: 9 doubles per line
void lo_get_lines(void* p, uint32_t* seg_offsets, double* segments9, uint32_t* res_offsets, uint32_t* residuals2,
                  double* cluster_lines9, uint32_t* reference_views) {
    std::vector<L3DPP::FinalLine3D> r;
    ((Ref*)p)->l3d->get3Dlines(r);
    auto put = [](double* o, const L3DPP::Segment3D& s) {
        const Eigen::Vector3d a = s.P1(), b = s.P2(), d = s.dir();
        o[0] = a.x(); o[1] = a.y(); o[2] = a.z(); o[3] = b.x(); o[4] = b.y(); o[5] = b.z(); o[6] = d.x(); o[7] = d.y(); o[8] = d.z();
    };
    uint32_t ns = 0, nr = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        seg_offsets[i] = ns; res_offsets[i] = nr;
        for (std::list<L3DPP::Segment3D>::const_iterator it = r[i].collinear3Dsegments_.begin(); it != r[i].collinear3Dsegments_.end(); ++it) put(segments9 + 9 * (ns++), *it);
        const std::list<L3DPP::Segment2D>* res = r[i].underlyingCluster_.residuals();
        for (std::list<L3DPP::Segment2D>::const_iterator it = res->begin(); it != res->end(); ++it) { residuals2[2 * nr] = it->camID(); residuals2[2 * nr + 1] = it->segID(); ++nr; }
        put(cluster_lines9 + 9 * i, r[i].underlyingCluster_.seg3D());
        reference_views[i] = r[i].underlyingCluster_.reference_view();
    }
    seg_offsets[r.size()] = ns; res_offsets[r.size()] = nr;
}

// ---- linear algebra of the reference as compiled here (i.e. through oracle/ref_shim's Eigen subset), handed out so
// that tests can bound its distance from a real linear-algebra library (numpy / LAPACK): tests/test_shim_vs_lapack.py
void lo_ref_view_matrices(void* p, uint32_t cam, double* Kinv9, double* RtKinv9) {
    L3DPP::View* v = ((Ref*)p)->l3d->views_[cam];
    const Eigen::Matrix3d a = v->Kinv(), b = v->RtKinv();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Kinv9[3 * i + j] = a(i, j); RtKinv9[3 * i + j] = b(i, j); }
}
// Line3D::getFundamentalMatrix (line3D.cc:861-897) of the views as they stand (call between addImage and matchImages:
// untranslated frame; F does not depend on the frame beyond rounding)
void lo_ref_fundamental(void* p, uint32_t src, uint32_t tgt, double* F9) {
    L3DPP::Line3D* l = ((Ref*)p)->l3d;
    const Eigen::Matrix3d F = l->getFundamentalMatrix(l->views_[src], l->views_[tgt]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F9[3 * i + j] = F(i, j);
}
// Line3D::rotationFromQ (line3D.cc:2730-2754), the reference's own: q = (w, x, y, z).  The front-end pin
// (ref_front_driver.cpp) routes the rotationFromQ calls of main_colmap.cpp / main_vsfm.cpp here.
void lo_ref_rotation_from_q(const double* q, double* R9) {
    const Eigen::Matrix3d R = L3DPP::Line3D::rotationFromQ(q[0], q[1], q[2], q[3]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R9[3 * i + j] = R(i, j);
}
// what get3DlineFromCluster (line3D.cc:2196-2211) does with its 3x3 scatter matrix: JacobiSVD, column of the largest
// singular value, normalised
void lo_ref_principal_direction(const double* S9, double* dir3) {
    Eigen::MatrixXd Scat(3, 3);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Scat(i, j) = S9[3 * i + j];
    Eigen::JacobiSVD<Eigen::MatrixXd> svd(Scat, Eigen::ComputeThinU);
    Eigen::MatrixXd U; Eigen::VectorXd S;
    U = svd.matrixU(); S = svd.singularValues();
    int maxPos; S.maxCoeff(&maxPos);
    Eigen::Vector3d dir = Eigen::Vector3d(U(0, maxPos), U(1, maxPos), U(2, maxPos));
    dir.normalize();
    dir3[0] = dir.x(); dir3[1] = dir.y(); dir3[2] = dir.z();
}

#ifdef L3DPP_CUDA
// ---- the build with the reference's CUDA path as host code (oracle/ref_shim_cuda, libl3d_ref_cuda.so) ---------------
}  // extern "C"
thread_local uint3 blockIdx, threadIdx;          // the variables L3D_SHIM_LAUNCH sets for the "kernels"
thread_local dim3 blockDim, gridDim;
extern "C" {
int lo_has_cuda_path() { return 1; }
// The replicator-dynamics diffusion of an arbitrary affinity list by the reference's OWN code: A_ and the matrix size
// (global2local_.size(), the only use performRDD makes of that map) are set, then Line3D::performRDD
// (line3D.cc:2026-2076) runs: SparseMatrix (sparsematrix.cc:8-139), replicator_dynamics_diffusion_GPU
// (cudawrapper.cu:708-766) with K_sparseMat_row_normalization / K_sparseMat_diffusion_step (:432-544) executed thread
// by thread on the host, and the min(w12, w21) symmetrisation.  out: room for 2 n edges; returns the count.
uint32_t lo_ref_rdd(const CLEdgeOut* in, uint32_t n, uint32_t n_rows, CLEdgeOut* out) {
    Quiet q;
    L3DPP::Line3D* l = new L3DPP::Line3D("/tmp", false, -1, 3000, false, false);
    l->A_.clear();
    for (uint32_t i = 0; i < n; ++i) { L3DPP::CLEdge e; e.i_ = in[i].i_; e.j_ = in[i].j_; e.w_ = in[i].w_; l->A_.push_back(e); }
    l->global2local_.clear();
    for (uint32_t i = 0; i < n_rows; ++i) l->global2local_[L3DPP::Segment2D(0, i)] = (int)i;
    l->performRDD();
    uint32_t k = 0;
    for (std::list<L3DPP::CLEdge>::const_iterator it = l->A_.begin(); it != l->A_.end(); ++it, ++k) { out[k].i_ = it->i_; out[k].j_ = it->j_; out[k].w_ = it->w_; }
    delete l;
    return k;
}
// Line3D::reconstruct3Dlines with perform_diffusion = true (the flag only exists in a CUDA build, line3D.cc:1728-1736)
void lo_reconstruct_rdd(void* p, uint32_t visibility_t) {
    Quiet q;
    ((Ref*)p)->l3d->reconstruct3Dlines(visibility_t, true, -1.0f, false, 250);
}
#endif

// stage-level entry points exist only in the restatement
void lo_begin_match(void*, float, float, uint32_t, float, int, float) {}
void lo_end_match(void*) {}
uint64_t lo_match_pair(void*, uint32_t, uint32_t, Match40*, uint64_t, uint32_t*) { return 0; }
void lo_fundamental(void*, uint32_t, uint32_t, double*) {}

}  // extern "C"
