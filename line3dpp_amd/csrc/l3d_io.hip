// l3d_io.hip -- the input side of the matching path behind the C-ABI (host code; the Python twin is
// line3dpp_amd/io.py, and tests/test_input_formats.py holds the two against each other):
//   * VisualSfM .nvm as main_vsfm.cpp:144-250 reads it (two ignored lines, the number of cameras, one line per camera:
//     file name, focal length, quaternion w x y z, centre, radial distortion; an ignored line, the number of points, one
//     line per point: position, colour, number of measurements, then camera index, feature index, x, y each) with what
//     main_vsfm.cpp derives per camera: R from the quaternion (:188-199), t = -R C, the ids of the points it sees
//     (= the worldpoint list handed to addImage) and the median of their distances (:300-303)
//   * the segment cache of Line3D::detectLineSegments (line3D.cc:295-309, 362-366): the boost binary archive of a
//     one-row L3DPP::DataArray<float4> (dataArray.h:352-374; host row padded to 32 bytes, :111-122)
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "l3d_ctx.h"

struct l3d_nvm {
    struct Cam {
        std::string filename;
        float focal = 0, distortion = 0, median_depth = 0;
        double R[9], t[3], C[3];
        std::vector<uint32_t> worldpoints;
        std::vector<float> depths;
    };
    std::vector<Cam> cams;
};

extern "C" {

int l3d_nvm_open(const char* path, l3d_nvm** out) {
    if (!path || !out) return fail(L3D_ERR_ARG, "null argument");
    *out = nullptr;
    std::ifstream f(path);
    if (!f) return fail(L3D_ERR_ARG, std::string("cannot open ") + path);
    std::string line;
    std::getline(f, line); std::getline(f, line);             // two ignored lines
    std::getline(f, line);
    unsigned n_cams = 0;
    { std::stringstream s(line); s >> n_cams; }
    if (n_cams == 0) return fail(L3D_ERR_NO_VIEWS, "No aligned cameras in NVM file!");   // main_vsfm.cpp:157-161
    auto nvm = std::make_unique<l3d_nvm>();
    nvm->cams.resize(n_cams);
    for (unsigned i = 0; i < n_cams; ++i) {
        std::getline(f, line);
        std::stringstream s(line);
        double focal = 0, qw = 0, qx = 0, qy = 0, qz = 0, cx = 0, cy = 0, cz = 0, dist = 0;
        l3d_nvm::Cam& c = nvm->cams[i];
        s >> c.filename >> focal >> qw >> qx >> qy >> qz >> cx >> cy >> cz >> dist;
        c.focal = (float)focal; c.distortion = (float)dist;   // kept in float vectors there
        double* R = c.R;
        R[0] = 1.0 - 2.0 * qy * qy - 2.0 * qz * qz; R[1] = 2.0 * qx * qy - 2.0 * qz * qw; R[2] = 2.0 * qx * qz + 2.0 * qy * qw;
        R[3] = 2.0 * qx * qy + 2.0 * qz * qw; R[4] = 1.0 - 2.0 * qx * qx - 2.0 * qz * qz; R[5] = 2.0 * qy * qz - 2.0 * qx * qw;
        R[6] = 2.0 * qx * qz - 2.0 * qy * qw; R[7] = 2.0 * qy * qz + 2.0 * qx * qw; R[8] = 1.0 - 2.0 * qx * qx - 2.0 * qy * qy;
        c.C[0] = cx; c.C[1] = cy; c.C[2] = cz;
        const d3 rc = mul33(R, d3{cx, cy, cz});
        c.t[0] = -rc.x; c.t[1] = -rc.y; c.t[2] = -rc.z;
    }
    std::getline(f, line);                                    // ignored
    std::getline(f, line);
    unsigned n_pts = 0;
    { std::stringstream s(line); s >> n_pts; }
    for (unsigned i = 0; i < n_pts; ++i) {
        if (!std::getline(f, line)) break;
        std::istringstream s(line);
        double px = 0, py = 0, pz = 0, cr, cg, cb;
        unsigned n_meas = 0;
        s >> px >> py >> pz >> cr >> cg >> cb >> n_meas;
        for (unsigned j = 0; j < n_meas; ++j) {
            unsigned cam = 0, feat = 0; float x, y;
            s >> cam >> feat >> x >> y;
            if (!s || cam >= n_cams) return fail(L3D_ERR_ARG, "malformed measurement in NVM file");
            l3d_nvm::Cam& c = nvm->cams[cam];
            c.worldpoints.push_back(i);
            c.depths.push_back((float)norm(d3{px, py, pz} - d3{c.C[0], c.C[1], c.C[2]}));
        }
    }
    for (auto& c : nvm->cams)
        if (!c.depths.empty()) {                              // main_vsfm.cpp:300-303
            std::sort(c.depths.begin(), c.depths.end());
            c.median_depth = c.depths[c.depths.size() / 2];
        }
    *out = nvm.release();
    return L3D_OK;
}

uint32_t l3d_nvm_num_cameras(const l3d_nvm* n) { return n ? (uint32_t)n->cams.size() : 0u; }

int l3d_nvm_get_camera(const l3d_nvm* n, uint32_t i, l3d_nvm_camera* out) {
    if (!n || !out || i >= n->cams.size()) return fail(L3D_ERR_ARG, "bad argument");
    const l3d_nvm::Cam& c = n->cams[i];
    out->filename = c.filename.c_str();
    out->focal = c.focal; out->distortion = c.distortion; out->median_depth = c.median_depth;
    std::memcpy(out->R, c.R, 72); std::memcpy(out->t, c.t, 24); std::memcpy(out->C, c.C, 24);
    out->n_worldpoints = (uint32_t)c.worldpoints.size();
    return L3D_OK;
}

int l3d_nvm_get_worldpoints(const l3d_nvm* n, uint32_t i, uint32_t* out, uint32_t cap) {
    if (!n || i >= n->cams.size() || (cap && !out)) return fail(L3D_ERR_ARG, "bad argument");
    const auto& w = n->cams[i].worldpoints;
    if (cap && !w.empty()) std::memcpy(out, w.data(), 4 * (size_t)std::min<size_t>(cap, w.size()));
    return w.size() > cap ? fail(L3D_ERR_LIMIT, "worldpoint buffer too small") : L3D_OK;
}

void l3d_nvm_close(l3d_nvm* n) { delete n; }

// K as main_vsfm.cpp:272-282 builds it: principal point at the image centre, in float arithmetic
void l3d_nvm_intrinsics(float focal, uint32_t width, uint32_t height, double K[9]) {
    const float px = float(width) / 2.0f, py = float(height) / 2.0f;
    for (int i = 0; i < 9; ++i) K[i] = 0.0;
    K[0] = focal; K[4] = focal; K[2] = px; K[5] = py; K[8] = 1.0;
}

// ---- segment cache ----------------------------------------------------------------------------------------------------
static const char kArchiveSig[] = "serialization::archive";            // 22 characters
static const unsigned char kPlatform[8] = {4, 8, 4, 8, 1, 0, 0, 0};   // sizes of int, long, float, double; endianness
static const unsigned char kClassHdr[5] = {0, 0, 0, 0, 0};

int l3d_segment_cache_name(uint32_t camID, uint32_t width, uint32_t height, uint32_t max_segments, char* out, uint32_t cap) {
    if (!out) return fail(L3D_ERR_ARG, "null argument");
    const int n = std::snprintf(out, cap, "segments_L3D++_%u_%ux%u_%u.bin", camID, width, height, max_segments);   // line3D.cc:300
    return (n < 0 || (uint32_t)n >= cap) ? fail(L3D_ERR_LIMIT, "name buffer too small") : L3D_OK;
}

int l3d_write_segment_cache(const char* path, const float* segs4, uint32_t n) {
    if (!path || (n && !segs4)) return fail(L3D_ERR_ARG, "null argument");
    const uint32_t real = n + (((uint64_t)n * 16) % 32 ? 1u : 0u);     // host row padded to 32 bytes
    std::string b;
    auto put = [&](const void* p, size_t k) { b.append((const char*)p, k); };
    const uint64_t sig_len = sizeof(kArchiveSig) - 1; const uint16_t lib_version = 10;
    put(&sig_len, 8); put(kArchiveSig, sig_len); put(&lib_version, 2); put(kPlatform, 8);
    put(kClassHdr, 5);
    const uint32_t w = n, h = 1; const uint64_t pitch = (uint64_t)real * 16, stride = real, zero = 0;
    put(&w, 4); put(&h, 4); put(&real, 4); put(&pitch, 8); put(&stride, 8); put(&zero, 8); put(&zero, 8);
    if (real) put(kClassHdr, 5);
    if (n) put(segs4, (size_t)n * 16);
    if (real > n) { const float pad[4] = {0, 0, 0, 0}; put(pad, 16); }
    std::ofstream f(path, std::ios::binary);
    if (!f || !f.write(b.data(), (std::streamsize)b.size())) return fail(L3D_ERR_ARG, std::string("cannot write ") + path);
    return L3D_OK;
}

int l3d_read_segment_cache(const char* path, float* segs4, uint32_t cap, uint32_t* n) {
    if (!path || !n) return fail(L3D_ERR_ARG, "null argument");
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(L3D_ERR_ARG, std::string("cannot open ") + path);
    std::string b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t p = 0;
    auto take = [&](void* dst, size_t k) { if (p + k > b.size()) return false; std::memcpy(dst, b.data() + p, k); p += k; return true; };
    auto expect = [&](const void* want, size_t k) { if (p + k > b.size() || std::memcmp(b.data() + p, want, k)) return false; p += k; return true; };
    uint64_t sig_len = 0; uint16_t lib_version = 0;
    if (!take(&sig_len, 8) || sig_len != sizeof(kArchiveSig) - 1 || !expect(kArchiveSig, sig_len) || !take(&lib_version, 2))
        return fail(L3D_ERR_ARG, "not a boost binary archive");
    if (!expect(kPlatform, 8)) return fail(L3D_ERR_ARG, "written on a platform with other type sizes / endianness");
    uint32_t w = 0, h = 0, real = 0; uint64_t pitch = 0, stride = 0, pg = 0, sg = 0;
    if (!expect(kClassHdr, 5) || !take(&w, 4) || !take(&h, 4) || !take(&real, 4) || !take(&pitch, 8) || !take(&stride, 8) ||
        !take(&pg, 8) || !take(&sg, 8))
        return fail(L3D_ERR_ARG, "truncated DataArray<float4> header");
    if (h != 1 || real < w || pitch != (uint64_t)real * 16 || stride != real)
        return fail(L3D_ERR_ARG, "not a one-row DataArray<float4>");
    if (real && !expect(kClassHdr, 5)) return fail(L3D_ERR_ARG, "unexpected class header for float4");
    if (p + (size_t)real * 16 != b.size()) return fail(L3D_ERR_ARG, "wrong archive length");
    *n = w;
    if (segs4) std::memcpy(segs4, b.data() + p, 16 * (size_t)std::min(cap, w));
    return (segs4 && w > cap) ? fail(L3D_ERR_LIMIT, "segment buffer too small") : L3D_OK;
}

}  // extern "C"
