// l3d_output.hip -- output layer of the C-ABI: Line3D::get3Dlines (line3D.cc:2455-2463), createOutputFilename
// (:2853-2893), getSegmentCoords2D (:2757-2772) and the result writers save3DLinesAsTXT (:2631-2688),
// save3DLinesAsBIN (:2690-2711), saveResultAsSTL (:2465-2531), saveResultAsOBJ (:2579-2628).
#include "l3d_ctx.h"

namespace l3d {

// Line3D::createOutputFilename, line3D.cc:2853-2893 (stream formatting of the float parameters as there)
std::string output_filename(l3d_ctx* c, int max_image_width) {
    std::stringstream str;
    str << "Line3D++__";
    if (max_image_width > 0) str << "W_" << max_image_width << "__";
    else str << "W_FULL__";
    str << "N_" << c->num_neighbors << "__";
    str << "sigmaP_" << c->sigma_p << "__";
    str << "sigmaA_" << c->sigma_a << "__";
    str << "epiOverlap_" << c->epipolar_overlap << "__";
    if (c->kNN > 0) str << "kNN_" << c->kNN << "__";
    if (c->collinearity_t > (float)kEps) str << "COLLIN_" << c->collinearity_t << "__";
    if (c->fixed3Dregularizer) {
        str << "FXD_SIGMA_P__";
        if (c->const_regularization_depth > 0.0f) str << "REG_DEPTH_" << c->const_regularization_depth << "__";
    }
    if (c->perform_rdd) str << "DIFFUSION__";
    str << "vis_" << c->visibility_t;     // (no "OPTIMIZED__": Ceres is not part of this library)
    return str.str();
}

}  // namespace l3d

using namespace l3d;

extern "C" {

int l3d_output_filename(l3d_ctx* c, int max_image_width, char* buf, uint32_t cap) {
    if (!c || !buf || !cap) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const std::string n = output_filename(c, max_image_width);
    if (n.size() + 1 > cap) return fail(L3D_ERR_ARG, "buffer too small for the output file name");
    std::memcpy(buf, n.c_str(), n.size() + 1);
    return L3D_OK;
}

// Line3D::save3DLinesAsTXT, line3D.cc:2631-2688: one text line per 3D line --
//   #segments  (P1.x P1.y P1.z P2.x P2.y P2.z)*  #residuals  (camID segID x1 y1 x2 y2)*
// written with the stream defaults the reference uses (6 significant digits), so files can be diffed
int l3d_save_3d_lines_txt(l3d_ctx* c, const char* output_folder, int max_image_width) {
    if (!c || !output_folder) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->lines_done || c->lines3D.empty()) return fail(L3D_ERR_STATE, "no 3D lines to save!");   // :2636-2642
    const std::string filename = std::string(output_folder) + "/" + output_filename(c, max_image_width) + ".txt";
    std::ofstream file(filename.c_str());
    if (!file) return fail(L3D_ERR_ARG, "cannot open " + filename);
    for (const ReconLine& L : c->lines3D) {
        if (L.collinear.empty()) continue;
        file << L.collinear.size() << " ";
        for (const ReconSeg3D& sg : L.collinear) {
            file << sg.P1.x << " " << sg.P1.y << " " << sg.P1.z << " ";
            file << sg.P2.x << " " << sg.P2.y << " " << sg.P2.z << " ";
        }
        file << L.residuals.size() << " ";
        for (const auto& r : L.residuals) {
            file << r.first << " " << r.second << " ";
            float co[4] = {0, 0, 0, 0};                                   // getSegmentCoords2D, line3D.cc:2608-2628
            auto f = c->views.find(r.first);
            if (f != c->views.end() && r.second < f->second->M)
                for (int k = 0; k < 4; ++k) co[k] = f->second->segs[4 * (size_t)r.second + k];
            file << co[0] << " " << co[1] << " " << co[2] << " " << co[3] << " ";
        }
        file << std::endl;
    }
    file.close();
    return file.fail() ? fail(L3D_ERR_ARG, "writing " + filename + " failed") : L3D_OK;
}

// Line3D::save3DLinesAsBIN, line3D.cc:2690-2711: serializeToFile(filename, lines3D_) (serialization.h:38-45) =
// boost::archive::binary_oarchive of std::vector<FinalLine3D>.  Written here without Boost, byte for byte as the
// Boost behind the reference's own fixtures does (archive library version 10; layout in line3dpp_amd/io.py, which
// re-creates testdata/Line3D++_ref/*vis_3.bin bit-identically -- tests/test_bin_format.py): every class writes its
// 5-byte header (tracking 0, version 0) at its first occurrence only; collections write u64 count + u32 item_version.
int l3d_save_3d_lines_bin(l3d_ctx* c, const char* output_folder, int max_image_width) {
    if (!c || !output_folder) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->lines_done || c->lines3D.empty()) return fail(L3D_ERR_STATE, "no 3D lines to save!");   // :2695-2701
    const std::string filename = std::string(output_folder) + "/" + output_filename(c, max_image_width) + ".bin";
    std::ofstream os(filename.c_str(), std::ios::binary);
    if (!os) return fail(L3D_ERR_ARG, "cannot open " + filename);
    auto put = [&](const void* p, size_t n) { os.write((const char*)p, (std::streamsize)n); };
    auto u64 = [&](uint64_t v) { put(&v, 8); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    bool seen[7] = {};
    enum { VEC, FINAL, LIST3, SEG3, CLUSTER, LIST2, SEG2 };
    auto hdr = [&](int cls) { if (!seen[cls]) { seen[cls] = true; const char z[5] = {0, 0, 0, 0, 0}; put(z, 5); } };
    auto seg3 = [&](const ReconSeg3D& sg) {               // Segment3D::serialize, segment3D.h:99-115
        hdr(SEG3);
        const float len = sg.length; const uint8_t valid = sg.valid ? 1 : 0;
        put(&len, 4); put(&valid, 1);
        const double g[9] = {sg.P1.x, sg.P1.y, sg.P1.z, sg.P2.x, sg.P2.y, sg.P2.z, sg.dir.x, sg.dir.y, sg.dir.z};
        put(g, 72);
    };
    const char sig[] = "serialization::archive";
    u64(22); put(sig, 22);
    const uint16_t lib_version = 10; put(&lib_version, 2);
    const uint8_t sizes[4] = {4, 8, 4, 8}; put(sizes, 4); u32(1);
    hdr(VEC); u64(c->lines3D.size()); u32(0);
    for (const ReconLine& L : c->lines3D) {
        hdr(FINAL);                                        // FinalLine3D::serialize, segment3D.h:165-178
        hdr(LIST3); u64(L.collinear.size()); u32(0);
        for (const ReconSeg3D& sg : L.collinear) seg3(sg);
        hdr(CLUSTER);                                      // LineCluster3D::serialize, segment3D.h:152-160
        seg3(L.cluster_seg);
        hdr(LIST2); u64(L.residuals.size()); u32(0);
        for (const auto& r : L.residuals) { hdr(SEG2); u32(r.first); u32(r.second); }   // commons.h:123-130
        u32(L.reference_view);
    }
    os.close();
    return os.fail() ? fail(L3D_ERR_ARG, "writing " + filename + " failed") : L3D_OK;
}

// Line3D::getSegmentCoords2D, line3D.cc:2757-2772: (0,0,0,0) for an unknown camera / segment
int l3d_get_segment_coords2d(l3d_ctx* c, uint32_t camID, uint32_t segID, float coords[4]) {
    if (!c || !coords) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    for (int k = 0; k < 4; ++k) coords[k] = 0.0f;
    auto f = c->views.find(camID);
    if (f != c->views.end() && segID < f->second->M)
        for (int k = 0; k < 4; ++k) coords[k] = f->second->segs[4 * (size_t)segID + k];
    return L3D_OK;
}

// Line3D::saveResultAsSTL (line3D.cc:2465-2531) / saveResultAsOBJ (:2579-2628)
int l3d_save_result_stl(l3d_ctx* c, const char* output_folder, int max_image_width) {
    if (!c || !output_folder) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->lines_done || c->lines3D.empty()) return fail(L3D_ERR_STATE, "no 3D lines to save!");
    const std::string filename = std::string(output_folder) + "/" + output_filename(c, max_image_width) + ".stl";
    std::ofstream file(filename.c_str());
    if (!file) return fail(L3D_ERR_ARG, "cannot open " + filename);
    file << "solid lineModel" << std::endl;
    for (const ReconLine& L : c->lines3D)
        for (const ReconSeg3D& sg : L.collinear) {
            char a[6][50];
            const double v[6] = {sg.P1.x, sg.P1.y, sg.P1.z, sg.P2.x, sg.P2.y, sg.P2.z};
            for (int k = 0; k < 6; ++k) std::snprintf(a[k], sizeof(a[k]), "%e", v[k]);
            file << " facet normal 1.0e+000 0.0e+000 0.0e+000" << std::endl;
            file << "  outer loop" << std::endl;
            file << "   vertex " << a[0] << " " << a[1] << " " << a[2] << std::endl;
            file << "   vertex " << a[3] << " " << a[4] << " " << a[5] << std::endl;
            file << "   vertex " << a[0] << " " << a[1] << " " << a[2] << std::endl;
            file << "  endloop" << std::endl;
            file << " endfacet" << std::endl;
        }
    file << "endsolid lineModel" << std::endl;
    file.close();
    return file.fail() ? fail(L3D_ERR_ARG, "writing " + filename + " failed") : L3D_OK;
}

int l3d_save_result_obj(l3d_ctx* c, const char* output_folder, int max_image_width) {
    if (!c || !output_folder) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->lines_done || c->lines3D.empty()) return fail(L3D_ERR_STATE, "no 3D lines to save!");
    const std::string filename = std::string(output_folder) + "/" + output_filename(c, max_image_width) + ".obj";
    std::ofstream file(filename.c_str());
    if (!file) return fail(L3D_ERR_ARG, "cannot open " + filename);
    size_t n_segments = 0;
    for (const ReconLine& L : c->lines3D)
        for (const ReconSeg3D& sg : L.collinear) {
            file << "v " << sg.P1.x << " " << sg.P1.y << " " << sg.P1.z << std::endl;
            file << "v " << sg.P2.x << " " << sg.P2.y << " " << sg.P2.z << std::endl;
            ++n_segments;
        }
    for (size_t k = 0; k < n_segments; ++k) file << "l " << 2 * k + 1 << " " << 2 * k + 2 << std::endl;
    file.close();
    return file.fail() ? fail(L3D_ERR_ARG, "writing " + filename + " failed") : L3D_OK;
}

int l3d_num_3d_lines(l3d_ctx* c, uint32_t* n_lines, uint32_t* n_segments, uint32_t* n_residuals) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->lines_done) return fail(L3D_ERR_STATE, "l3d_reconstruct_3d_lines has not run");
    uint32_t ns = 0, nr = 0;
    for (auto& L : c->lines3D) { ns += (uint32_t)L.collinear.size(); nr += (uint32_t)L.residuals.size(); }
    if (n_lines) *n_lines = (uint32_t)c->lines3D.size();
    if (n_segments) *n_segments = ns;
    if (n_residuals) *n_residuals = nr;
    return L3D_OK;
}

int l3d_get_3d_lines(l3d_ctx* c, uint32_t* seg_offsets, l3d_segment3d* segments, uint32_t* res_offsets,
                     l3d_segment2d* residuals, l3d_segment3d* cluster_lines, uint32_t* reference_views) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->lines_done) return fail(L3D_ERR_STATE, "l3d_reconstruct_3d_lines has not run");
    auto put = [](l3d_segment3d& o, const ReconSeg3D& s) {
        o.P1[0] = s.P1.x; o.P1[1] = s.P1.y; o.P1[2] = s.P1.z; o.P2[0] = s.P2.x; o.P2[1] = s.P2.y; o.P2[2] = s.P2.z;
        o.dir[0] = s.dir.x; o.dir[1] = s.dir.y; o.dir[2] = s.dir.z; o.length_ = s.length; o.valid_ = s.valid ? 1u : 0u;
    };
    uint32_t ns = 0, nr = 0;
    for (size_t i = 0; i < c->lines3D.size(); ++i) {
        const ReconLine& L = c->lines3D[i];
        if (seg_offsets) seg_offsets[i] = ns;
        if (res_offsets) res_offsets[i] = nr;
        for (auto& s : L.collinear) { if (segments) put(segments[ns], s); ++ns; }
        for (auto& r : L.residuals) { if (residuals) { residuals[nr].camID_ = r.first; residuals[nr].segID_ = r.second; } ++nr; }
        if (cluster_lines) put(cluster_lines[i], L.cluster_seg);
        if (reference_views) reference_views[i] = L.reference_view;
    }
    if (seg_offsets) seg_offsets[c->lines3D.size()] = ns;
    if (res_offsets) res_offsets[c->lines3D.size()] = nr;
    return L3D_OK;
}

}  // extern "C"
