// facade_smoke.cpp -- drives the hot path through the C++ facade (include/line3dpp/line3D.h) exactly as a
// reference main_*.cpp drives L3DPP::Line3D (main_mavmap.cpp:311-325), on a scene file written by the
// Python test, and prints result counts/checksums for comparison with the Python front-end.
#include <cstdio>
#include <cstdlib>
#include <list>
#include <string>
#include <vector>

#include "line3dpp/line3D.h"

struct Mat3 { double m[9]; double operator()(int r, int c) const { return m[3 * r + c]; } };
struct Vec3 { double v[3]; double operator()(int i) const { return v[i]; } };
struct Vec4f { float v[4]; float operator[](int i) const { return v[i]; } };

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    uint32_t nv = 0;
    if (fread(&nv, 4, 1, f) != 1) return 4;
    // argv[2] == "worldpoints": the per-view list of the scene file is a worldpoint list and the instance is constructed
    // with neighbors_by_worldpoints = true, as main_vsfm.cpp:140-141 does
    const bool by_wps = argc > 2 && std::string(argv[2]) == "worldpoints";
    L3DPP_HIP::Line3D l3d("/tmp", false, -1, 3000, by_wps, true);
    std::vector<uint32_t> cams;
    for (uint32_t i = 0; i < nv; ++i) {
        uint32_t hdr[5];  // cam, M, width, height, n_nb
        Mat3 K, R; Vec3 t; float md;
        if (fread(hdr, 4, 5, f) != 5 || fread(K.m, 8, 9, f) != 9 || fread(R.m, 8, 9, f) != 9 ||
            fread(t.v, 8, 3, f) != 3 || fread(&md, 4, 1, f) != 1) return 5;
        std::vector<uint32_t> nb(hdr[4]);
        if (fread(nb.data(), 4, hdr[4], f) != hdr[4]) return 6;
        std::vector<Vec4f> segs(hdr[1]);
        if (fread(segs.data(), 16, hdr[1], f) != hdr[1]) return 7;
        L3DPP_HIP::ImageSize img{(int)hdr[2], (int)hdr[3]};
        l3d.addImage(hdr[0], img, K, R, t, md, std::list<unsigned int>(nb.begin(), nb.end()), segs);
        cams.push_back(hdr[0]);
    }
    fclose(f);
    l3d.matchImages();
    l3d.computeAffinityMatrix();
    size_t n_matches = 0; double score_sum = 0;
    for (uint32_t c : cams)
        for (auto& lst : l3d.matches(c)) { n_matches += lst.size(); for (auto& m : lst) score_sum += m.score3D_; }
    auto hyp = l3d.estimatedPosition3D();
    std::list<l3d_cledge> A; std::map<int, l3d_segment2d> l2g;
    l3d.affinity(A, l2g);
    double wsum = 0; for (auto& e : A) wsum += e.w_;
    l3d.reconstruct3Dlines(3);
    std::vector<L3DPP_HIP::Line3D::FinalLine3D> lines;
    l3d.get3Dlines(lines);
    size_t nseg3 = 0; for (auto& L : lines) nseg3 += L.collinear3Dsegments_.size();
    // a consumer written against the reference's FinalLine3D / LineCluster3D (segment3D.h:120-178) reads the cluster through
    // the same accessors
    size_t nres = 0; double len = 0; unsigned refsum = 0;
    for (auto& L : lines) {
        const l3d_segment3d s3 = L.underlyingCluster_.seg3D();
        const std::list<l3d_segment2d>* res = L.underlyingCluster_.residuals();
        nres += res->size(); refsum += L.underlyingCluster_.reference_view();
        if (res->size() != L.underlyingCluster_.size()) return 3;
        const double* q = reinterpret_cast<const double*>(&s3);   // P1[3], P2[3], dir[3]
        double d2 = 0; for (int k = 0; k < 3; ++k) d2 += (q[k] - q[3 + k]) * (q[k] - q[3 + k]);
        len += d2;
    }
    printf("LINES lines=%zu segments=%zu residuals=%zu refsum=%u len2=%.6f\n", lines.size(), nseg3, nres, refsum, len);
    printf("RESULT images=%zu matches=%zu score_sum=%.6f hypotheses=%zu edges=%zu rows=%zu wsum=%.6f\n",
           l3d.numImages(), n_matches, score_sum, hyp.size(), A.size(), l2g.size(), wsum);
    return 0;
}
