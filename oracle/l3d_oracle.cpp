// =============================================================================
// oracle/l3d_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// CPU restatement of the Line3D++ *CPU/OpenMP* hot path (reference repository
// manhofer/Line3Dpp): pairwise 2D-segment matching -> orientation filter ->
// 3D scoring -> inverse matches -> depth-hypothesis collapse -> affinity fill.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; the shipped HIP path (line3dpp_amd/csrc) never links or calls it.
//
// PARITY STATUS: "parity unpinned" by the reference itself -- the reference has
// no tests and no golden vectors for intermediates, its
// testdata poses (testdata/vsfm_result.nvm) are missing, and its sources cannot
// be compiled in this image (Eigen3 / Boost / OpenCV absent).  This file is
// therefore a line-by-line restatement; every function cites the reference
// file:line it follows (paths relative to /root/reference).
//
// Arithmetic conventions: the reference promotes float
// inputs to double (Eigen Vector3d/Matrix3d) and stores float results; every
// place where the reference holds a `float` is a `float` here.  3-vector dot /
// squaredNorm use Eigen's unrolled reduction order a0 + (a1 + a2); 3x3 products
// accumulate left to right.  Build with -ffp-contract=off (no FMA contraction).
// =============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <queue>
#include <set>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---- constants: commons.h:40-100 -------------------------------------------
constexpr double L3D_EPS = 1e-12;
constexpr float L3D_PI_1_32 = 0.098174771f;
constexpr float L3D_PI_31_32 = 3.043417886f;
constexpr float L3D_DEF_MIN_SIMILARITY_3D = 0.50f;
constexpr float L3D_DEF_MIN_BEST_SCORE_3D = 0.75f;
constexpr float L3D_DEF_MIN_BEST_SCORE_PERC = 0.10f;
constexpr float L3D_DEF_MIN_AFFINITY = 0.50f;

// ---- minimal double 3-vector / 3x3 algebra (stands in for Eigen) -------------
struct V3 { double x, y, z; };
struct M3 { double m[9]; };  // row-major

inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline V3 cross(const V3& a, const V3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(const V3& a) { double n = norm(a); return {a.x / n, a.y / n, a.z / n}; }
inline V3 mul(const M3& A, const V3& v) {
    return {(A.m[0] * v.x + A.m[1] * v.y) + A.m[2] * v.z,
            (A.m[3] * v.x + A.m[4] * v.y) + A.m[5] * v.z,
            (A.m[6] * v.x + A.m[7] * v.y) + A.m[8] * v.z};
}
inline M3 mul(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[3 * i + j] = (A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j]) + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
inline M3 transpose(const M3& A) {
    M3 T;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.m[3 * i + j] = A.m[3 * j + i];
    return T;
}
// cofactor inverse, as Eigen's fixed-size 3x3 inverse (compute_inverse_size3)
inline M3 inverse(const M3& A) {
    const double* a = A.m;
    double c00 = a[4] * a[8] - a[5] * a[7];
    double c10 = a[5] * a[6] - a[3] * a[8];   // cofactor (1,0) of A  -> used for inv(0,1)
    double c20 = a[3] * a[7] - a[4] * a[6];
    double det = a[0] * c00 + (a[1] * c10 + a[2] * c20);
    double id = 1.0 / det;
    M3 R;
    R.m[0] = c00 * id;
    R.m[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    R.m[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    R.m[3] = c10 * id;
    R.m[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    R.m[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    R.m[6] = c20 * id;
    R.m[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    R.m[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return R;
}

// ---- PODs: commons.h:186-203 (Match), segment3D.h:35-116 (Segment3D) ---------
struct Match {
    uint32_t src_camID_, src_segID_, tgt_camID_, tgt_segID_;
    float overlap_score_, score3D_;
    float depth_p1_, depth_p2_, depth_q1_, depth_q2_;
};
static_assert(sizeof(Match) == 40, "Match layout (commons.h:186-203)");

struct Segment3D {  // segment3D.h:47-66
    V3 P1{0, 0, 0}, P2{0, 0, 0}, dir{0, 0, 0};
    float length = 0.0f;
    bool valid = false;
    Segment3D() {}
    Segment3D(const V3& a, const V3& b) {
        length = (float)norm(a - b);
        if (length > L3D_EPS) {
            P1 = a; P2 = b; dir = normalized(b - a); valid = true;
        } else {
            length = 0.0f; valid = false;
        }
    }
    // segment3D.h:69-73
    // `P1_ + (dir_ * ((P - P1_).transpose()) * dir_)`: Eigen evaluates the outer product dir*v^T
    // into a 3x3 temporary and multiplies it with dir (row sums left to right).
    float distance_Point2Line(const V3& P) const {
        V3 v = P - P1;
        V3 h{P1.x + (((dir.x * v.x) * dir.x + (dir.x * v.y) * dir.y) + (dir.x * v.z) * dir.z),
             P1.y + (((dir.y * v.x) * dir.x + (dir.y * v.y) * dir.y) + (dir.y * v.z) * dir.z),
             P1.z + (((dir.z * v.x) * dir.x + (dir.z * v.y) * dir.y) + (dir.z * v.z) * dir.z)};
        return (float)norm(h - P);
    }
};

struct CLEdge { int i_, j_; float w_; };  // clustering.h:47-51

// kNN comparator: commons.h:217-228
struct Match_kNN {
    bool operator()(const Match& l, const Match& r) const { return l.overlap_score_ < r.overlap_score_; }
};

// ---- per-view camera model: view.cc:6-42 -------------------------------------
struct View {
    uint32_t id = 0;
    std::vector<float> segs;  // 4*M  (x1,y1,x2,y2)  DataArray<float4> width=M height=1
    uint32_t M = 0;
    M3 K, R, Kinv, Rt, RtKinv;
    V3 t, C, pp;
    uint32_t width = 0, height = 0;
    float initial_median_depth = 0, k = 0, median_depth = 0, median_sigma = 0;

    V3 ray(const V3& p) const { return normalized(mul(RtKinv, p)); }  // view.cc:317-321
    V3 p1(uint32_t s) const { return {(double)segs[4 * s], (double)segs[4 * s + 1], 1.0}; }
    V3 p2(uint32_t s) const { return {(double)segs[4 * s + 2], (double)segs[4 * s + 3], 1.0}; }
    // view.cc:356-371
    Segment3D unprojectSegment(uint32_t s, float d1, float d2) const {
        if (s >= M) return Segment3D();
        return Segment3D(C + ray(p1(s)) * (double)d1, C + ray(p2(s)) * (double)d2);
    }
    // view.cc:307-314
    float getSpecificSpatialReg(float r) const {
        V3 pps = pp + V3{(double)r, 0.0, 0.0};
        V3 a = ray(pp), b = ray(pps);
        double alpha = std::acos(std::fmin(std::fmax(dot(a, b), -1.0), 1.0));
        return (float)std::sin(alpha);
    }
    // view.cc:445-448
    float regularizerFrom3Dpoint(const V3& P) const { return (float)(norm(P - C) * (double)k); }
    // view.cc:466-484
    double segmentQualityAngle(const Segment3D& s3, uint32_t s) const {
        if (s >= M) return 0.0;
        double ax = segs[4 * s], ay = segs[4 * s + 1], bx = segs[4 * s + 2], by = segs[4 * s + 3];
        V3 p{0.5 * (ax + bx), 0.5 * (ay + by), 1.0};
        V3 r1 = ray(p);
        return std::acos(std::fmin(std::fmax(dot(r1, s3.dir), -1.0), 1.0));
    }
    // view.cc:510-514
    void translate(const V3& d) {
        C = C + d;
        V3 rc = mul(R, C);
        t = {-rc.x, -rc.y, -rc.z};
    }
    // view.h:108-121
    void update_median_depth(float d, float sigmaP, float med_scene_depth) {
        median_depth = d;
        if (sigmaP > 0.0f) k = sigmaP / med_scene_depth;
        median_sigma = k * median_depth;
    }
};

struct Best {  // one entry of estimated_position3D_ (line3D.cc:1637-1646)
    Segment3D seg3D;
    Match m;
};

struct Ctx {
    std::map<uint32_t, View> views;                      // views_ (ascending camID)
    std::map<uint32_t, std::vector<uint32_t>> fixed_nbrs;  // fixed_visual_neighbors_
    std::map<uint32_t, std::set<uint32_t>> visual_nbrs;  // visual_neighbors_
    std::map<uint32_t, std::set<uint32_t>> matched;      // matched_
    std::map<uint32_t, std::vector<std::list<Match>>> matches;  // matches_
    std::map<uint32_t, bool> processed;
    std::vector<float> views_avg_depths;
    std::vector<Best> best;                              // estimated_position3D_
    std::map<std::pair<uint32_t, uint32_t>, size_t> entry_map;  // entry_map_
    std::vector<CLEdge> A;                               // A_
    std::vector<std::pair<uint32_t, uint32_t>> local2global;
    // per-view snapshot of matches after scoring, before inverse/filter (debug / tests)
    bool record_scored = false;
    std::map<uint32_t, std::vector<std::vector<Match>>> scored;
    // fresh matches per directed pair, before orientation filter (debug / tests)
    std::vector<std::pair<uint32_t, uint32_t>> pair_list;
    // params
    float sigma_p = 2.5f, sigma_a = 10.0f, two_sigA_sqr = 200.0f, epipolar_overlap = 0.25f;
    float const_regularization_depth = -1.0f, med_scene_depth = (float)L3D_EPS, med_scene_depth_lines = 0.0f;
    int kNN = 10, num_neighbors = 10;
    bool fixed3Dregularizer = false;
    V3 translation{0, 0, 0};
    uint64_t pair_tests = 0;
};

// line3D.cc:500-536
void translate(Ctx& c) {
    if (c.views.empty()) return;
    double tr[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        std::vector<double> coords;
        for (auto& kv : c.views) {
            double val = i == 0 ? kv.second.C.x : (i == 1 ? kv.second.C.y : kv.second.C.z);
            if (std::fabs(val) > L3D_EPS) coords.push_back(val);
        }
        if (!coords.empty()) {
            std::sort(coords.begin(), coords.end());
            tr[i] = coords[coords.size() / 2];
        }
    }
    c.translation = {tr[0], tr[1], tr[2]};
    V3 neg{-tr[0], -tr[1], -tr[2]};
    for (auto& kv : c.views) kv.second.translate(neg);  // performTranslation(-translation_)
}
// line3D.cc:539-545
void untranslate(Ctx& c) {
    for (auto& kv : c.views) kv.second.translate(c.translation);
}

// line3D.cc:861-897 (cache omitted: every unordered pair is matched once)
M3 fundamental(const View& src, const View& tgt) {
    M3 R = mul(tgt.R, transpose(src.R));
    V3 Rt1 = mul(R, src.t);
    V3 t = tgt.t - Rt1;
    M3 T = {{0.0, -t.z, t.y, t.z, 0.0, -t.x, -t.y, t.x, 0.0}};
    M3 E = mul(T, R);
    return mul(mul(inverse(transpose(tgt.K)), E), inverse(src.K));
}

// line3D.cc:1077-1083
inline bool pointOnSegment(const V3& x, const V3& p1, const V3& p2) {
    double v1x = p1.x - x.x, v1y = p1.y - x.y, v2x = p2.x - x.x, v2y = p2.y - x.y;
    return (v1x * v2x + v1y * v2y) < L3D_EPS;
}

// line3D.cc:1086-1165
float mutualOverlap(const V3 cp[4]) {
    float overlap = 0.0f;
    const V3 &p1 = cp[0], &p2 = cp[1], &q1 = cp[2], &q2 = cp[3];
    if (pointOnSegment(p1, q1, q2) || pointOnSegment(p2, q1, q2) || pointOnSegment(q1, p1, p2) ||
        pointOnSegment(q2, p1, p2)) {
        float max_dist = 0.0f;
        size_t outer1 = 0, inner1 = 1, inner2 = 2, outer2 = 3;
        for (size_t i = 0; i < 3; ++i)
            for (size_t j = i + 1; j < 4; ++j) {
                float dist = (float)norm(cp[i] - cp[j]);
                if (dist > max_dist) { max_dist = dist; outer1 = i; outer2 = j; }
            }
        if (max_dist < 1.0f) return 0.0f;
        if (outer1 == 0) {
            if (outer2 == 1) { inner1 = 2; inner2 = 3; }
            else if (outer2 == 2) { inner1 = 1; inner2 = 3; }
            else { inner1 = 1; inner2 = 2; }
        } else if (outer1 == 1) {
            inner1 = 0;
            inner2 = (outer2 == 2) ? 3 : 2;
        } else { inner1 = 0; inner2 = 1; }
        overlap = (float)(norm(cp[inner1] - cp[inner2]) / (double)max_dist);
    }
    return overlap;
}

// line3D.cc:1168-1193
inline void triangulationDepths(const View& vs, const V3& p1, const V3& p2, const View& vt, const V3& q1,
                                const V3& q2, double& d1, double& d2) {
    V3 C1 = vs.C;
    V3 rp1 = vs.ray(p1), rp2 = vs.ray(p2);
    V3 C2 = vt.C;
    V3 rq1 = vt.ray(q1), rq2 = vt.ray(q2);
    V3 n = normalized(cross(rq1, rq2));
    if (std::fabs(dot(rp1, n)) < L3D_EPS || std::fabs(dot(rp2, n)) < L3D_EPS) { d1 = -1; d2 = -1; return; }
    d1 = (dot(C2, n) - dot(n, C1)) / dot(n, rp1);
    d2 = (dot(C2, n) - dot(n, C1)) / dot(n, rp2);
}

// one (src seg r, tgt seg c) test: line3D.cc:931-995.  returns true + filled M on acceptance
inline bool pair_test(const Ctx& c, const View& vs, const View& vt, const M3& F, uint32_t r, uint32_t cc,
                      const V3& p1, const V3& p2, const V3& epi_p1, const V3& epi_p2, Match& M) {
    V3 q1 = vt.p1(cc), q2 = vt.p2(cc);
    V3 l2 = cross(q1, q2);
    V3 p1_proj = cross(l2, epi_p1), p2_proj = cross(l2, epi_p2);
    if (std::fabs(p1_proj.z) > L3D_EPS && std::fabs(p2_proj.z) > L3D_EPS) {
        p1_proj = {p1_proj.x / p1_proj.z, p1_proj.y / p1_proj.z, p1_proj.z / p1_proj.z};
        p2_proj = {p2_proj.x / p2_proj.z, p2_proj.y / p2_proj.z, p2_proj.z / p2_proj.z};
        V3 cp[4] = {p1_proj, p2_proj, q1, q2};
        float score = mutualOverlap(cp);
        if (score > c.epipolar_overlap) {
            double ds1, ds2, dt1, dt2;
            triangulationDepths(vs, p1, p2, vt, q1, q2, ds1, ds2);
            triangulationDepths(vt, q1, q2, vs, p1, p2, dt1, dt2);
            if (ds1 > L3D_EPS && ds2 > L3D_EPS && dt1 > L3D_EPS && dt2 > L3D_EPS) {
                M.src_camID_ = vs.id; M.src_segID_ = r; M.tgt_camID_ = vt.id; M.tgt_segID_ = cc;
                M.overlap_score_ = score; M.score3D_ = 0.0f;
                M.depth_p1_ = (float)ds1; M.depth_p2_ = (float)ds2;
                M.depth_q1_ = (float)dt1; M.depth_q2_ = (float)dt2;
                return true;
            }
        }
    }
    return false;
}

// line3D.cc:900-1015.  `out` receives the fresh matches of row r (appended).
void matching_rows(const Ctx& c, const View& vs, const View& vt, const M3& F,
                   std::vector<std::list<Match>>& rows) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int r = 0; r < (int)vs.M; ++r) {
        int new_matches = 0;
        V3 p1 = vs.p1(r), p2 = vs.p2(r);
        V3 epi_p1 = mul(F, p1), epi_p2 = mul(F, p2);
        std::priority_queue<Match, std::vector<Match>, Match_kNN> scored;
        for (uint32_t cc = 0; cc < vt.M; ++cc) {
            Match M;
            if (pair_test(c, vs, vt, F, (uint32_t)r, cc, p1, p2, epi_p1, epi_p2, M)) {
                if (c.kNN > 0) scored.push(M);
                else { rows[r].push_back(M); ++new_matches; }
            }
        }
        if (c.kNN > 0) {
            while (new_matches < c.kNN && !scored.empty()) {
                rows[r].push_back(scored.top());
                scored.pop();
                ++new_matches;
            }
        }
    }
}

inline Segment3D unprojectMatchSrc(const Ctx& c, const Match& m) {  // line3D.cc:1556-1568 (src=true)
    const View& v = c.views.at(m.src_camID_);
    return v.unprojectSegment(m.src_segID_, m.depth_p1_, m.depth_p2_);
}

// line3D.cc:811-858
void checkMatchOrientation(Ctx& c, uint32_t src) {
    auto& rows = c.matches[src];
    const View& v = c.views.at(src);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < (int)rows.size(); ++i) {
        std::list<Match> remaining;
        for (const Match& m : rows[i]) {
            Segment3D s3 = unprojectMatchSrc(c, m);
            double ang = c.views.at(m.src_camID_).segmentQualityAngle(s3, m.src_segID_);
            if (ang > L3D_PI_1_32 && ang < L3D_PI_31_32) remaining.push_back(m);
        }
        rows[i] = remaining;
    }
    (void)v;
}

// line3D.cc:1571-1583
inline float angleBetweenSeg3D(const Segment3D& s1, const Segment3D& s2) {
    float dot_p = (float)dot(s1.dir, s2.dir);
    float angle = (float)(std::acos((double)std::fmax(std::fmin(dot_p, 1.0f), -1.0f)) / M_PI * 180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    return angle;
}

// line3D.cc:1417-1446
inline float similarityForScoring(const Ctx& c, const Match& m1, const Match& m2, const Segment3D& seg3D1,
                                  float reg1, float reg2) {
    Segment3D seg3D2 = unprojectMatchSrc(c, m2);
    if (seg3D1.length < L3D_EPS || seg3D2.length < L3D_EPS) return 0.0f;
    float angle = angleBetweenSeg3D(seg3D1, seg3D2);
    float sim_a = expf(-angle * angle / c.two_sigA_sqr);
    float sim_p = 0.0f;
    if (m1.src_camID_ == m2.src_camID_ && m1.src_segID_ == m2.src_segID_) {
        float d1 = m1.depth_p1_ - m2.depth_p1_;
        float d2 = m1.depth_p2_ - m2.depth_p2_;
        sim_p = std::fmin(expf(-d1 * d1 / reg1), expf(-d2 * d2 / reg2));
    }
    float sim = std::fmin(sim_a, sim_p);
    return sim > L3D_DEF_MIN_SIMILARITY_3D ? sim : 0.0f;
}

// line3D.cc:1208-1294
void scoringCPU(Ctx& c, uint32_t src, float& valid_f) {
    const View& v = c.views.at(src);
    float k = v.k;
    auto& rows = c.matches[src];
    unsigned num_valid = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : num_valid)
    for (int i = 0; i < (int)rows.size(); ++i) {
        bool valid_match_exists = false;
        for (auto it = rows[i].begin(); it != rows[i].end(); ++it) {
            Match M = *it;
            float score3D = 0.0f;
            std::map<unsigned, float> score_per_cam;
            Segment3D M3D = v.unprojectSegment(M.src_segID_, M.depth_p1_, M.depth_p2_);
            float sig1 = M.depth_p1_ * k, sig2 = M.depth_p2_ * k;
            float reg1 = 2.0f * sig1 * sig1, reg2 = 2.0f * sig2 * sig2;
            const View& vt = c.views.at(M.tgt_camID_);
            float sig1_tgt = vt.regularizerFrom3Dpoint(M3D.P1);
            float sig2_tgt = vt.regularizerFrom3Dpoint(M3D.P2);
            reg1 = 0.5f * (reg1 + 2.0f * sig1_tgt * sig1_tgt);
            reg2 = 0.5f * (reg2 + 2.0f * sig2_tgt * sig2_tgt);
            for (auto it2 = rows[i].begin(); it2 != rows[i].end(); ++it2) {
                const Match& M2 = *it2;
                if (M.tgt_camID_ != M2.tgt_camID_) {
                    float sim = similarityForScoring(c, M, M2, M3D, reg1, reg2);
                    auto f = score_per_cam.find(M2.tgt_camID_);
                    if (f != score_per_cam.end()) {
                        if (sim > f->second) {
                            score3D -= f->second;
                            score3D += sim;
                            f->second = sim;
                        }
                    } else {
                        score3D += sim;
                        score_per_cam[M2.tgt_camID_] = sim;
                    }
                }
            }
            it->score3D_ = score3D;
            if (score3D > L3D_DEF_MIN_BEST_SCORE_3D) valid_match_exists = true;
        }
        if (valid_match_exists) ++num_valid;
    }
    valid_f = v.M ? float(num_valid) / float(v.M) : 0.0f;
}

// line3D.cc:1672-1699
void storeInverseMatches(Ctx& c, uint32_t src) {
    auto& rows = c.matches[src];
    for (size_t i = 0; i < rows.size(); ++i)
        for (const Match& m : rows[i]) {
            if (m.score3D_ > 0.0f && !c.processed[m.tgt_camID_]) {
                Match inv = m;
                inv.src_camID_ = m.tgt_camID_; inv.src_segID_ = m.tgt_segID_;
                inv.tgt_camID_ = m.src_camID_; inv.tgt_segID_ = m.src_segID_;
                inv.depth_p1_ = m.depth_q1_; inv.depth_p2_ = m.depth_q2_;
                inv.depth_q1_ = m.depth_p1_; inv.depth_q2_ = m.depth_p2_;
                inv.score3D_ = 0.0f;
                c.matches[m.tgt_camID_][m.tgt_segID_].push_back(inv);
            }
        }
}

// line3D.cc:1586-1669.  estimated_position3D_ is filled in ascending segment order (the
// reference's order under OpenMP is thread-timing dependent; consumers key by (cam,seg)).
void filterMatches(Ctx& c, uint32_t src) {
    auto& rows = c.matches[src];
    std::vector<float> depths;
    float max_score = 0.0f;
    for (auto& row : rows)
        for (const Match& m : row) max_score = std::fmax(max_score, m.score3D_);
    float score_lim = L3D_DEF_MIN_BEST_SCORE_PERC * max_score;
    std::vector<Match> bests(rows.size());
    std::vector<char> has_best(rows.size(), 0);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < (int)rows.size(); ++i) {
        Match best_match;
        std::memset(&best_match, 0, sizeof(best_match));
        best_match.score3D_ = 0.0f;
        std::list<Match> ms = rows[i];
        rows[i].clear();
        for (const Match& m : ms) {
            if (m.score3D_ > 0.0f && m.score3D_ > score_lim) {
                rows[i].push_back(m);
                if (m.score3D_ > best_match.score3D_) best_match = m;
            }
        }
        if (best_match.score3D_ > L3D_DEF_MIN_BEST_SCORE_3D) {
            bests[i] = best_match;
            has_best[i] = 1;
        } else {
            rows[i].clear();
        }
    }
    for (size_t i = 0; i < rows.size(); ++i) {
        if (!has_best[i]) continue;
        Best b;
        b.m = bests[i];
        b.seg3D = unprojectMatchSrc(c, b.m);
        c.entry_map[{src, (uint32_t)i}] = c.best.size();
        c.best.push_back(b);
        depths.push_back(b.m.depth_p1_);
        depths.push_back(b.m.depth_p2_);
    }
    float med_depth = (float)L3D_EPS;
    if (!depths.empty()) {
        std::sort(depths.begin(), depths.end());
        med_depth = depths[depths.size() / 2];
    }
    View& v = c.views.at(src);
    if (!c.fixed3Dregularizer) v.update_median_depth(med_depth, -1.0f, c.med_scene_depth);
    else v.update_median_depth(med_depth, c.sigma_p, c.med_scene_depth);
}

// line3D.cc:702-778
void computeMatches(Ctx& c) {
    c.pair_list.clear();
    c.pair_tests = 0;
    for (auto& kv : c.visual_nbrs) {
        uint32_t src = kv.first;
        for (uint32_t tgt : kv.second) {
            if (c.matched[src].find(tgt) == c.matched[src].end()) {
                const View& vs = c.views.at(src);
                const View& vt = c.views.at(tgt);
                M3 F = fundamental(vs, vt);
                std::vector<std::list<Match>> rows(vs.M);
                matching_rows(c, vs, vt, F, rows);
                for (uint32_t r = 0; r < vs.M; ++r)
                    c.matches[src][r].insert(c.matches[src][r].end(), rows[r].begin(), rows[r].end());
                c.pair_list.push_back({src, tgt});
                c.pair_tests += (uint64_t)vs.M * vt.M;
                c.matched[src].insert(tgt);
                c.matched[tgt].insert(src);
            }
        }
        checkMatchOrientation(c, src);  // L3D_DEF_CHECK_MATCH_ORIENTATION true (commons.h:56)
        float valid_f;
        scoringCPU(c, src, valid_f);
        if (c.record_scored) {
            auto& snap = c.scored[src];
            snap.assign(c.matches[src].size(), {});
            for (size_t i = 0; i < snap.size(); ++i) snap[i].assign(c.matches[src][i].begin(), c.matches[src][i].end());
        }
        storeInverseMatches(c, src);
        filterMatches(c, src);
        c.processed[src] = true;
    }
}

// param clamps + driver: line3D.cc:375-497
void begin_match(Ctx& c, float sigma_position, float sigma_angle, unsigned num_neighbors,
                 float epipolar_overlap, int kNN, float const_regularization_depth) {
    c.num_neighbors = std::max(int(num_neighbors), 2);
    c.sigma_p = sigma_position;
    c.sigma_a = std::fmin(std::fabs(sigma_angle), 90.0f);
    c.two_sigA_sqr = 2.0f * c.sigma_a * c.sigma_a;
    c.epipolar_overlap = std::fmin(std::fabs(epipolar_overlap), 0.99f);
    c.kNN = kNN;
    c.const_regularization_depth = const_regularization_depth;
    if (c.sigma_p < 0.0f) { c.fixed3Dregularizer = true; c.sigma_p = std::fabs(c.sigma_p); }
    else { c.fixed3Dregularizer = false; c.sigma_p = std::fmax(0.1f, c.sigma_p); }
    c.matched.clear();
    c.best.clear();
    c.entry_map.clear();
    c.scored.clear();
    c.med_scene_depth = c.const_regularization_depth;
    if (c.const_regularization_depth < 0.0f && c.fixed3Dregularizer && !c.views_avg_depths.empty()) {
        std::sort(c.views_avg_depths.begin(), c.views_avg_depths.end());
        c.med_scene_depth = c.views_avg_depths[c.views_avg_depths.size() / 2];
    }
    translate(c);
    for (auto& kv : c.views) {
        View& v = kv.second;
        if (!c.fixed3Dregularizer) v.k = v.getSpecificSpatialReg(c.sigma_p);  // computeSpatialRegularizer
        else v.k = c.sigma_p / c.med_scene_depth;                             // update_k, view.h:124-127
        c.matches[kv.first] = std::vector<std::list<Match>>(v.M);
        c.processed[kv.first] = false;
    }
    // fixed neighbours (neighbors_by_worldpoints=false): line3D.cc:467-479
    for (auto& kv : c.views) {
        uint32_t cam = kv.first;
        auto f = c.fixed_nbrs.find(cam);
        if (f != c.fixed_nbrs.end() && c.visual_nbrs[cam].empty())
            for (uint32_t n : f->second)
                if (c.views.find(n) != c.views.end()) c.visual_nbrs[cam].insert(n);
    }
}

// line3D.cc:1467-1553 (truncate=false)
float similarity(const Ctx& c, const Segment3D& s1, const Match& m1, uint32_t cam2, uint32_t seg2) {
    auto f = c.entry_map.find({cam2, seg2});
    if (f == c.entry_map.end()) return 0.0f;
    const Best& d2 = c.best[f->second];
    const Segment3D& s2 = d2.seg3D;
    const Match& m2 = d2.m;
    if (s1.length < L3D_EPS || s2.length < L3D_EPS) return 0.0f;
    const View& v1 = c.views.at(m1.src_camID_);
    const View& v2 = c.views.at(m2.src_camID_);
    float angle = angleBetweenSeg3D(s1, s2);
    float sim_a = expf(-angle * angle / c.two_sigA_sqr);
    float cutoff1 = v1.median_depth, cutoff2 = v2.median_depth;
    if (c.med_scene_depth_lines > L3D_EPS) {
        cutoff1 = std::fmin(cutoff1, c.med_scene_depth_lines);
        cutoff2 = std::fmin(cutoff2, c.med_scene_depth_lines);
    }
    float d11 = s2.distance_Point2Line(s1.P1), d12 = s2.distance_Point2Line(s1.P2);
    float d21 = s1.distance_Point2Line(s2.P1), d22 = s1.distance_Point2Line(s2.P2);
    float sig11 = (m1.depth_p1_ > cutoff1) ? cutoff1 * v1.k : m1.depth_p1_ * v1.k;
    float sig12 = (m1.depth_p2_ > cutoff1) ? cutoff1 * v1.k : m1.depth_p2_ * v1.k;
    float reg11 = 2.0f * sig11 * sig11, reg12 = 2.0f * sig12 * sig12;
    float sig21 = (m2.depth_p1_ > cutoff2) ? cutoff2 * v2.k : m2.depth_p1_ * v2.k;
    float sig22 = (m2.depth_p2_ > cutoff2) ? cutoff2 * v2.k : m2.depth_p2_ * v2.k;
    float reg21 = 2.0f * sig21 * sig21, reg22 = 2.0f * sig22 * sig22;
    float sim_p1 = std::fmin(expf(-d11 * d11 / reg11), expf(-d12 * d12 / reg12));
    float sim_p2 = std::fmin(expf(-d21 * d21 / reg21), expf(-d22 * d22 / reg22));
    float sim_p = std::fmin(sim_p1, sim_p2);
    return std::fmin(sim_a, sim_p);
}

// reconstruct3Dlines up to the affinity matrix: line3D.cc:1749-1778, 1852-2023 (collinearity off)
void compute_affinity(Ctx& c) {
    c.A.clear();
    c.local2global.clear();
    if (c.best.empty()) return;
    translate(c);
    std::vector<float> sd;
    for (auto& kv : c.views)
        if (kv.second.median_depth > L3D_EPS) sd.push_back(kv.second.median_depth);
    if (!sd.empty()) { std::sort(sd.begin(), sd.end()); c.med_scene_depth_lines = sd[sd.size() / 2]; }
    else c.med_scene_depth_lines = 0.0f;
    std::map<std::pair<uint32_t, uint32_t>, int> g2l;
    std::set<std::pair<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint32_t>>> used;
    auto localID = [&](std::pair<uint32_t, uint32_t> s) {  // line3D.cc:2005-2023
        auto f = g2l.find(s);
        if (f != g2l.end()) return f->second;
        int id = (int)c.local2global.size();
        g2l[s] = id;
        c.local2global.push_back(s);
        return id;
    };
    for (size_t i = 0; i < c.best.size(); ++i) {
        const Segment3D& seg3D = c.best[i].seg3D;
        const Match& m = c.best[i].m;
        std::pair<uint32_t, uint32_t> a{m.src_camID_, m.src_segID_};
        int id1 = -1;
        for (const Match& m2 : c.matches[m.src_camID_][m.src_segID_]) {
            std::pair<uint32_t, uint32_t> b{m2.tgt_camID_, m2.tgt_segID_};
            float sim = similarity(c, seg3D, m, b.first, b.second);
            if (sim > L3D_DEF_MIN_AFFINITY) {
                // unused(): line3D.cc:1982-2002
                if (used.find({a, b}) != used.end()) continue;
                used.insert({a, b});
                used.insert({b, a});
                if (id1 < 0) id1 = localID(a);
                int id2 = localID(b);
                c.A.push_back({id1, id2, sim});
                c.A.push_back({id2, id1, sim});
            }
        }
    }
    untranslate(c);
}

}  // namespace

// =============================================================================
// C interface (ctypes).  Prefix lo_ = "Line3D++ oracle".
// =============================================================================
extern "C" {

void* lo_create() { return new Ctx(); }
void lo_destroy(void* p) { delete (Ctx*)p; }
void lo_set_record_scored(void* p, int on) { ((Ctx*)p)->record_scored = on != 0; }
void lo_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
int lo_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Line3D::addImage with explicit segments + explicit neighbours: line3D.cc:112-227, view.cc:6-42
int lo_add_view(void* p, uint32_t camID, const float* segs4, uint32_t M, const double* K, const double* R,
                const double* t, uint32_t width, uint32_t height, float median_depth, const uint32_t* nbrs,
                uint32_t n_nbrs) {
    Ctx& c = *(Ctx*)p;
    if (std::max(width, height) < 800) return 1;       // L3D_DEF_MIN_IMG_WIDTH
    if (c.views.find(camID) != c.views.end()) return 2;  // ID in use
    if (n_nbrs == 0) return 3;
    if (M == 0) return 4;
    View v;
    v.id = camID; v.M = M;
    v.segs.assign(segs4, segs4 + 4 * (size_t)M);
    std::memcpy(v.K.m, K, 72); std::memcpy(v.R.m, R, 72);
    v.t = {t[0], t[1], t[2]};
    v.width = width; v.height = height;
    v.initial_median_depth = (float)std::fmax(std::fabs(median_depth), L3D_EPS);
    v.pp = {v.K.m[2], v.K.m[5], 1.0};
    v.Kinv = inverse(v.K);
    v.Rt = transpose(v.R);
    v.RtKinv = mul(v.Rt, v.Kinv);
    v.C = mul(v.Rt, V3{-1.0 * v.t.x, -1.0 * v.t.y, -1.0 * v.t.z});
    c.views[camID] = v;
    c.matches[camID] = std::vector<std::list<Match>>(M);
    c.processed[camID] = false;
    c.visual_nbrs[camID] = {};
    c.views_avg_depths.push_back((float)std::fmax(median_depth, L3D_EPS));
    c.fixed_nbrs[camID].assign(nbrs, nbrs + n_nbrs);
    return 0;
}

// Line3D::matchImages: line3D.cc:375-497
void lo_match_images(void* p, float sigma_p, float sigma_a, uint32_t num_neighbors, float epi_overlap, int kNN,
                     float const_reg_depth) {
    Ctx& c = *(Ctx*)p;
    if (c.views.empty()) return;
    begin_match(c, sigma_p, sigma_a, num_neighbors, epi_overlap, kNN, const_reg_depth);
    computeMatches(c);
    untranslate(c);
}

// split-phase helpers for stage-level tests: begin (clamp/translate/k/neighbours), one pair, end
void lo_begin_match(void* p, float sigma_p, float sigma_a, uint32_t num_neighbors, float epi_overlap, int kNN,
                    float const_reg_depth) {
    begin_match(*(Ctx*)p, sigma_p, sigma_a, num_neighbors, epi_overlap, kNN, const_reg_depth);
}
void lo_end_match(void* p) { untranslate(*(Ctx*)p); }

// matchingCPU for one directed pair in the current (translated) frame.  out: up to cap Matches in row
// order; offsets[Ms+1].  Returns total count (may exceed cap -> nothing beyond cap is written).
uint64_t lo_match_pair(void* p, uint32_t src, uint32_t tgt, Match* out, uint64_t cap, uint32_t* offsets) {
    Ctx& c = *(Ctx*)p;
    const View& vs = c.views.at(src);
    const View& vt = c.views.at(tgt);
    M3 F = fundamental(vs, vt);
    std::vector<std::list<Match>> rows(vs.M);
    matching_rows(c, vs, vt, F, rows);
    uint64_t n = 0;
    for (uint32_t r = 0; r < vs.M; ++r) {
        offsets[r] = (uint32_t)n;
        for (const Match& m : rows[r]) { if (n < cap) out[n] = m; ++n; }
    }
    offsets[vs.M] = (uint32_t)n;
    return n;
}
void lo_fundamental(void* p, uint32_t src, uint32_t tgt, double* F9) {
    Ctx& c = *(Ctx*)p;
    M3 F = fundamental(c.views.at(src), c.views.at(tgt));
    std::memcpy(F9, F.m, 72);
}

void lo_compute_affinity(void* p) { compute_affinity(*(Ctx*)p); }

// ---- accessors ---------------------------------------------------------------
uint32_t lo_num_pairs(void* p) { return (uint32_t)((Ctx*)p)->pair_list.size(); }
void lo_get_pairs(void* p, uint32_t* src, uint32_t* tgt) {
    Ctx& c = *(Ctx*)p;
    for (size_t i = 0; i < c.pair_list.size(); ++i) { src[i] = c.pair_list[i].first; tgt[i] = c.pair_list[i].second; }
}
uint64_t lo_pair_tests(void* p) { return ((Ctx*)p)->pair_tests; }

static uint64_t dump_rows(const std::vector<std::list<Match>>& rows, Match* out, uint64_t cap, uint32_t* offsets) {
    uint64_t n = 0;
    for (size_t r = 0; r < rows.size(); ++r) {
        if (offsets) offsets[r] = (uint32_t)n;
        for (const Match& m : rows[r]) { if (out && n < cap) out[n] = m; ++n; }
    }
    if (offsets) offsets[rows.size()] = (uint32_t)n;
    return n;
}
// surviving matches_[cam] (after matchImages): CSR over segments
uint64_t lo_get_matches(void* p, uint32_t cam, Match* out, uint64_t cap, uint32_t* offsets) {
    Ctx& c = *(Ctx*)p;
    return dump_rows(c.matches.at(cam), out, cap, offsets);
}
// matches of view `cam` right after scoring (before inverse/filter); needs lo_set_record_scored(1)
uint64_t lo_get_scored(void* p, uint32_t cam, Match* out, uint64_t cap, uint32_t* offsets) {
    Ctx& c = *(Ctx*)p;
    auto f = c.scored.find(cam);
    if (f == c.scored.end()) return 0;
    uint64_t n = 0;
    for (size_t r = 0; r < f->second.size(); ++r) {
        if (offsets) offsets[r] = (uint32_t)n;
        for (const Match& m : f->second[r]) { if (out && n < cap) out[n] = m; ++n; }
    }
    if (offsets) offsets[f->second.size()] = (uint32_t)n;
    return n;
}
uint32_t lo_num_best(void* p) { return (uint32_t)((Ctx*)p)->best.size(); }
// estimated_position3D_: per entry (cam,seg), 9 doubles (P1,P2,dir; translated frame), float length, Match
void lo_get_best(void* p, uint32_t* camseg2, double* p1p2dir9, float* length, Match* m) {
    Ctx& c = *(Ctx*)p;
    for (size_t i = 0; i < c.best.size(); ++i) {
        const Best& b = c.best[i];
        camseg2[2 * i] = b.m.src_camID_; camseg2[2 * i + 1] = b.m.src_segID_;
        double* o = p1p2dir9 + 9 * i;
        o[0] = b.seg3D.P1.x; o[1] = b.seg3D.P1.y; o[2] = b.seg3D.P1.z;
        o[3] = b.seg3D.P2.x; o[4] = b.seg3D.P2.y; o[5] = b.seg3D.P2.z;
        o[6] = b.seg3D.dir.x; o[7] = b.seg3D.dir.y; o[8] = b.seg3D.dir.z;
        length[i] = b.seg3D.length;
        m[i] = b.m;
    }
}
// per-view state: k, median_depth, C (current frame), t
void lo_view_info(void* p, uint32_t cam, float* k, float* median_depth, double* C3, double* t3) {
    const View& v = ((Ctx*)p)->views.at(cam);
    *k = v.k; *median_depth = v.median_depth;
    C3[0] = v.C.x; C3[1] = v.C.y; C3[2] = v.C.z;
    t3[0] = v.t.x; t3[1] = v.t.y; t3[2] = v.t.z;
}
void lo_translation(void* p, double* t3) {
    Ctx& c = *(Ctx*)p;
    t3[0] = c.translation.x; t3[1] = c.translation.y; t3[2] = c.translation.z;
}
float lo_med_scene_depth_lines(void* p) { return ((Ctx*)p)->med_scene_depth_lines; }
uint32_t lo_num_edges(void* p) { return (uint32_t)((Ctx*)p)->A.size(); }
uint32_t lo_num_rows(void* p) { return (uint32_t)((Ctx*)p)->local2global.size(); }
void lo_get_affinity(void* p, CLEdge* edges, uint32_t* local2global2) {
    Ctx& c = *(Ctx*)p;
    std::memcpy(edges, c.A.data(), c.A.size() * sizeof(CLEdge));
    for (size_t i = 0; i < c.local2global.size(); ++i) {
        local2global2[2 * i] = c.local2global[i].first;
        local2global2[2 * i + 1] = c.local2global[i].second;
    }
}

}  // extern "C"
