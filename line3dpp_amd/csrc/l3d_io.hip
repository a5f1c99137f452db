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

// ---- COLMAP text result and bundler file: one handle type, one record per image ----------------------------------------
// l3d_sfm_open_colmap: cameras.txt / images.txt / points3D.txt of a COLMAP result folder as main_colmap.cpp:136-348 reads
// them -- lines that start with '#' are skipped in the first two files; camera models SIMPLE_PINHOLE, PINHOLE,
// SIMPLE_RADIAL, RADIAL, OPENCV, FULL_OPENCV (:177-219, anything else is an error: -3 there); images.txt alternates an
// image line (id, quaternion w x y z, t, camera id, name) with a line of 2D points (x y POINT3D_ID ...), of which the
// non-negative ids are the worldpoint list handed to addImage; an image whose camera is unknown is dropped (:281);
// R = rotationFromQ (line3D.cc:2730-2754: s = 2 / |q|^2, 0 for a vanishing quaternion), C = R^T (-t); points3D.txt: EVERY
// line is parsed as "id X Y Z" (the comment test there looks at the last line of images.txt, :333 -- a line that does
// not parse is left alone here); median depth = sorted float distances |C - X| [n / 2] over the image's list (:390-406,
// a point without a points3D entry sits at the origin like in the reference's map); images are kept in file order
// (img_seq), an image without worldpoints has n_worldpoints == 0 and is skipped by the caller like :389-410 does.
// l3d_sfm_open_bundler: bundle.rd.out as main_bundler.cpp:147-252 reads it -- an ignored line, "num_cams num_points",
// per camera focal / two radial coefficients, three rotation rows and the translation with the y and z rows / entries
// negated (:192-212), C = R^T (-t); per point a position line, an ignored colour line and the view list
// (count, then camera, key, x, y each); camera index = camID; median depth as above (:371-373).  K is built by the
// caller from the image size (l3d_nvm_intrinsics: the same construction, :341-350).
struct l3d_sfm {
    struct Img {
        uint32_t id = 0, cam = 0, width = 0, height = 0;
        std::string name;
        double K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, R[9], t[3], C[3], radial[3] = {0, 0, 0}, tangential[2] = {0, 0};
        float focal = 0, median_depth = 0;
        std::vector<uint32_t> worldpoints;
    };
    std::vector<Img> imgs;
};

// Line3D::rotationFromQ, line3D.cc:2730-2754
static void rotation_from_q(double w, double x, double y, double z, double R[9]) {
    const double n = w * w + x * x + y * y + z * z;
    const double s = std::fabs(n) < kEps ? 0.0 : 2.0 / n;
    const double wx = s * w * x, wy = s * w * y, wz = s * w * z, xx = s * x * x, xy = s * x * y, xz = s * x * z;
    const double yy = s * y * y, yz = s * y * z, zz = s * z * z;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
}
static void centre_from(const double R[9], const double t[3], double C[3]) {   // R^T (-1 t), Eigen's product order
    const M3 Rm{{R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7], R[8]}};
    const d3 c = mul33(m3_t(Rm).m, d3{-1.0 * t[0], -1.0 * t[1], -1.0 * t[2]});
    C[0] = c.x; C[1] = c.y; C[2] = c.z;
}
static float median_of(std::vector<float>& d) { std::sort(d.begin(), d.end()); return d[d.size() / 2]; }


int l3d_sfm_open_colmap(const char* folder, l3d_sfm** out) {
    if (!folder || !out) return fail(L3D_ERR_ARG, "null argument");
    *out = nullptr;
    const std::string dir(folder);
    std::ifstream fc(dir + "/cameras.txt"), fi(dir + "/images.txt"), fp(dir + "/points3D.txt");
    if (!fc || !fi || !fp) return fail(L3D_ERR_ARG, "at least one of the colmap result files does not exist in sfm folder: " + dir);
    struct Cam { double K[9], radial[3], tangential[2]; uint32_t w, h; };
    std::map<uint32_t, Cam> cams;
    std::string line;
    while (std::getline(fc, line)) {
        if (line.substr(0, 1) == "#") continue;
        std::stringstream s(line);
        uint32_t id = 0, w = 0, h = 0; std::string model;
        s >> id >> model >> w >> h;
        double fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, k3 = 0, p1 = 0, p2 = 0;
        if (model == "SIMPLE_PINHOLE") { s >> fx >> cx >> cy; fy = fx; }
        else if (model == "PINHOLE") s >> fx >> fy >> cx >> cy;
        else if (model == "SIMPLE_RADIAL") { s >> fx >> cx >> cy >> k1; fy = fx; }
        else if (model == "RADIAL") { s >> fx >> cx >> cy >> k1 >> k2; fy = fx; }
        else if (model == "OPENCV") s >> fx >> fy >> cx >> cy >> k1 >> k2 >> p1 >> p2;
        else if (model == "FULL_OPENCV") s >> fx >> fy >> cx >> cy >> k1 >> k2 >> p1 >> p2 >> k3;
        else return fail(L3D_ERR_ARG, "camera model " + model + " unknown!");
        Cam c{{fx, 0, cx, 0, fy, cy, 0, 0, 1}, {k1, k2, k3}, {p1, p2}, w, h};
        cams[id] = c;
    }
    auto sfm = std::make_unique<l3d_sfm>();
    std::map<uint32_t, size_t> by_id;                          // image id -> record (a repeated id overwrites, like the maps there)
    std::map<uint32_t, d3> wps;                                // worldpoints seen in images.txt (origin until points3D.txt says otherwise)
    bool first = true;
    uint32_t img_id = 0, cam_id = 0;
    while (std::getline(fi, line)) {
        if (line.substr(0, 1) == "#") continue;
        std::stringstream s(line);
        if (first) {
            double qw = 0, qx = 0, qy = 0, qz = 0, tx = 0, ty = 0, tz = 0; std::string name;
            s >> img_id >> qw >> qx >> qy >> qz >> tx >> ty >> tz >> cam_id >> name;
            auto ci = cams.find(cam_id);
            if (ci != cams.end()) {
                l3d_sfm::Img im;
                im.id = img_id; im.cam = cam_id; im.name = name; im.width = ci->second.w; im.height = ci->second.h;
                std::memcpy(im.K, ci->second.K, 72); std::memcpy(im.radial, ci->second.radial, 24); std::memcpy(im.tangential, ci->second.tangential, 16);
                rotation_from_q(qw, qx, qy, qz, im.R);
                im.t[0] = tx; im.t[1] = ty; im.t[2] = tz;
                centre_from(im.R, im.t, im.C);
                im.focal = (float)im.K[0];
                auto it = by_id.find(img_id);
                if (it == by_id.end()) { by_id[img_id] = sfm->imgs.size(); sfm->imgs.push_back(im); }
                else { im.worldpoints = sfm->imgs[it->second].worldpoints; sfm->imgs[it->second] = im; sfm->imgs.push_back(im); by_id[img_id] = sfm->imgs.size() - 1; }
            }
            first = false;
        } else {
            if (cams.count(cam_id)) {
                std::vector<uint32_t> list;
                for (;;) {
                    double x, y; std::string wp;
                    s >> x >> y >> wp;
                    if (wp.empty()) break;
                    const int id = std::atoi(wp.c_str());
                    if (id >= 0) { list.push_back((uint32_t)id); wps[(uint32_t)id] = d3{0, 0, 0}; }
                }
                auto it = by_id.find(img_id);
                if (it != by_id.end()) sfm->imgs[it->second].worldpoints = list;
            }
            first = true;
        }
    }
    while (std::getline(fp, line)) {
        std::stringstream s(line);
        uint32_t id = 0; double X = 0, Y = 0, Z = 0;
        s >> id >> X >> Y >> Z;
        if (!s) continue;                                      // (a comment or malformed line)
        auto it = wps.find(id);
        if (it != wps.end()) it->second = d3{X, Y, Z};
    }
    // a repeated image id: img_seq holds it twice, both entries see the LAST pose and list (the maps are keyed by id)
    for (auto& im : sfm->imgs) { const l3d_sfm::Img& last = sfm->imgs[by_id[im.id]]; if (&last != &im) im = last; }
    for (auto& im : sfm->imgs) {
        std::vector<float> depths;
        for (uint32_t w : im.worldpoints) depths.push_back((float)norm(d3{im.C[0], im.C[1], im.C[2]} - wps[w]));
        if (!depths.empty()) im.median_depth = median_of(depths);
    }
    *out = sfm.release();
    return L3D_OK;
}

int l3d_sfm_open_bundler(const char* path, l3d_sfm** out) {
    if (!path || !out) return fail(L3D_ERR_ARG, "null argument");
    *out = nullptr;
    std::ifstream f(path);
    if (!f) return fail(L3D_ERR_ARG, std::string("bundle file '") + path + "' does not exist!");
    std::string line;
    std::getline(f, line); std::getline(f, line);             // first line ignored
    uint32_t n_cams = 0, n_pts = 0;
    { std::stringstream s(line); s >> n_cams >> n_pts; }
    if (n_cams == 0 || n_pts == 0) return fail(L3D_ERR_NO_VIEWS, "No cameras and/or points in bundle file!");   // main_bundler.cpp:160-164
    auto sfm = std::make_unique<l3d_sfm>();
    sfm->imgs.resize(n_cams);
    auto next = [&](std::stringstream& s) { std::getline(f, line); s.str(""); s.clear(); s.str(line); };
    std::stringstream s;
    for (uint32_t i = 0; i < n_cams; ++i) {
        l3d_sfm::Img& im = sfm->imgs[i];
        im.id = i; im.cam = i;
        double focal = 0, d1 = 0, d2 = 0;
        next(s); s >> focal >> d1 >> d2;
        im.focal = (float)focal; im.radial[0] = (double)(float)d1; im.radial[1] = (double)(float)d2;   // (float pairs there, :170-176)
        for (int j = 0; j < 3; ++j) { next(s); s >> im.R[3 * j] >> im.R[3 * j + 1] >> im.R[3 * j + 2]; }
        for (int k = 3; k < 9; ++k) im.R[k] *= -1.0;          // flip 2nd and 3rd line
        next(s); s >> im.t[0] >> im.t[1] >> im.t[2];
        im.t[1] *= -1.0; im.t[2] *= -1.0;
        centre_from(im.R, im.t, im.C);
    }
    std::vector<std::vector<float>> depths(n_cams);
    for (uint32_t i = 0; i < n_pts; ++i) {
        double px = 0, py = 0, pz = 0;
        if (!std::getline(f, line)) break;
        { std::istringstream p(line); p >> px >> py >> pz; }
        std::getline(f, line);                                // colour
        std::getline(f, line);
        std::istringstream v(line);
        uint32_t n_views = 0;
        v >> n_views;
        for (uint32_t j = 0; j < n_views; ++j) {
            uint32_t cam = 0, key = 0; float x, y;
            v >> cam >> key >> x >> y;
            if (!v || cam >= n_cams) return fail(L3D_ERR_ARG, "malformed view list in bundle file");
            sfm->imgs[cam].worldpoints.push_back(i);
            depths[cam].push_back((float)norm(d3{px, py, pz} - d3{sfm->imgs[cam].C[0], sfm->imgs[cam].C[1], sfm->imgs[cam].C[2]}));
        }
    }
    for (uint32_t i = 0; i < n_cams; ++i) if (!depths[i].empty()) sfm->imgs[i].median_depth = median_of(depths[i]);
    *out = sfm.release();
    return L3D_OK;
}

uint32_t l3d_sfm_num_images(const l3d_sfm* s) { return s ? (uint32_t)s->imgs.size() : 0u; }

int l3d_sfm_get_image(const l3d_sfm* s, uint32_t i, l3d_sfm_image* out) {
    if (!s || !out || i >= s->imgs.size()) return fail(L3D_ERR_ARG, "bad argument");
    const l3d_sfm::Img& im = s->imgs[i];
    out->id = im.id; out->camera = im.cam; out->width = im.width; out->height = im.height; out->name = im.name.c_str();
    out->focal = im.focal; out->median_depth = im.median_depth; out->n_worldpoints = (uint32_t)im.worldpoints.size();
    std::memcpy(out->K, im.K, 72); std::memcpy(out->R, im.R, 72); std::memcpy(out->t, im.t, 24); std::memcpy(out->C, im.C, 24);
    std::memcpy(out->radial, im.radial, 24); std::memcpy(out->tangential, im.tangential, 16);
    return L3D_OK;
}

int l3d_sfm_get_worldpoints(const l3d_sfm* s, uint32_t i, uint32_t* out, uint32_t cap) {
    if (!s || i >= s->imgs.size() || (cap && !out)) return fail(L3D_ERR_ARG, "bad argument");
    const auto& w = s->imgs[i].worldpoints;
    if (cap && !w.empty()) std::memcpy(out, w.data(), 4 * (size_t)std::min<size_t>(cap, w.size()));
    return w.size() > cap ? fail(L3D_ERR_LIMIT, "worldpoint buffer too small") : L3D_OK;
}

void l3d_sfm_close(l3d_sfm* s) { delete s; }



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
