// l3d_dev.h -- shared host/device PODs and double-precision geometry helpers of the HIP path.
//
// Arithmetic contract (DESIGN.md §"precision"): everything that decides set membership is
// evaluated in double exactly as the reference CPU path does (float inputs promoted to double,
// float where the reference stores float), with the operation order written out here and FMA
// contraction disabled at compile time (-ffp-contract=off).  fp32 is used only for the
// conservative pre-filter of the all-pairs loop, never for a value that is stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define L3D_HD __host__ __device__ __forceinline__

namespace l3d {

constexpr double kEps = 1e-12;              // L3D_EPS, commons.h:97
constexpr float kPi_1_32 = 0.098174771f;    // L3D_PI_1_32, commons.h:101
constexpr float kPi_31_32 = 3.043417886f;   // L3D_PI_31_32, commons.h:102
constexpr float kMinBestScore3D = 0.75f;    // L3D_DEF_MIN_BEST_SCORE_3D, commons.h:60
constexpr float kMinBestScorePerc = 0.10f;  // L3D_DEF_MIN_BEST_SCORE_PERC, commons.h:61
constexpr float kMinAffinity = 0.50f;       // L3D_DEF_MIN_AFFINITY, commons.h:68
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

struct d3 { double x, y, z; };

L3D_HD d3 operator+(const d3& a, const d3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
L3D_HD d3 operator-(const d3& a, const d3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
L3D_HD d3 operator*(const d3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
// Eigen's unrolled 3-term reduction: a0 + (a1 + a2)
L3D_HD double dot(const d3& a, const d3& b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
L3D_HD d3 cross(const d3& a, const d3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
L3D_HD double norm(const d3& a) { return sqrt(dot(a, a)); }
L3D_HD d3 normalized(const d3& a) { double n = norm(a); return {a.x / n, a.y / n, a.z / n}; }
// row-major 3x3 times vector, accumulated left to right
L3D_HD d3 mul33(const double* A, const d3& v) {
    return {(A[0] * v.x + A[1] * v.y) + A[2] * v.z, (A[3] * v.x + A[4] * v.y) + A[5] * v.z,
            (A[6] * v.x + A[7] * v.y) + A[8] * v.z};
}

// ---- IEEE double division and square root without the range scaffolding ------------------------------------------------
// hipcc expands a / b into v_div_scale x2, v_rcp_f64, two Newton steps, q = a*y, r = fma(-b,q,a), v_div_fmas, v_div_fixup
// (59 issue cycles per wave, tools/valu_calib.hip) and sqrt into v_rsq_f64 + Goldschmidt with ldexp scaling and a class
// select (86).  The scale / fixup instructions only act on operands whose exponents are extreme (|exponent| beyond
// several hundred: quotients or intermediate reciprocals that would leave the normal range); for everything else
// they pass their inputs through and the sequence below IS the compiler's sequence, bit for bit.  The `fast` forms are
// used where the host has checked the ranges of what enters the arithmetic (PairDesc::flags kPairFastMath: finite F
// with entries in [1e-30, 1e30] or zero, pixel coordinates and camera centres below 1e7 / 1e30) and the reference's own
// guards bound the denominators from below (|x.z| > L3D_EPS, |n.r| >= L3D_EPS); a reciprocal is shared by the divisions
// that have the same denominator.  Host code (tests, restated helpers) always takes the plain operators.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rcp_refined(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0); y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0); y = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ double div_by(double a, double b, double y) {   // a / b, y = rcp_refined(b)
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
__device__ __forceinline__ double sqrt_unscaled(double x) {   // x == 0 or x in the normal range far from its ends
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    return x == 0.0 ? x : g;
}
#define L3D_DIV2(fast, a1, a2, b, o1, o2) do { if (fast) { const double y_ = rcp_refined(b); o1 = div_by(a1, b, y_); o2 = div_by(a2, b, y_); } else { o1 = (a1) / (b); o2 = (a2) / (b); } } while (0)
#define L3D_DIV(fast, a, b) ((fast) ? div_by(a, b, rcp_refined(b)) : (a) / (b))
#define L3D_SQRT(fast, x) ((fast) ? sqrt_unscaled(x) : sqrt(x))
#else
#define L3D_DIV2(fast, a1, a2, b, o1, o2) do { o1 = (a1) / (b); o2 = (a2) / (b); } while (0)
#define L3D_DIV(fast, a, b) ((a) / (b))
#define L3D_SQRT(fast, x) sqrt(x)
#endif
constexpr uint32_t kPairFastMath = 1u;   // PairDesc::flags: the operands of this pair's exact tests are inside the range above

// ---- device-resident per-segment records ------------------------------------------------
// exact (double) invariants of one 2D segment of one view; built by k_prep_view after
// translate().  rays = normalize(RtKinv * p) (view.cc:317-321); n = normalize(r1 x r2) and
// cn = C.n are the plane terms Line3D::triangulationDepths (line3D.cc:1180-1191) recomputes
// for every pair; rm = ray through the segment mid-point (view.cc:466-481).
struct SegX {
    double r1[3], r2[3], n[3], cn, rm[3];
};  // 13 doubles = 104 B

// the part of SegX the depth test needs (its first ten doubles: a SegX can be read as a SegD); k_cull_prepare keeps a
// copy of the target view's records in this form in the order in which the match kernel walks them
struct SegD {
    double r1[3], r2[3], n[3], cn;
};  // 80 B
static_assert(sizeof(SegD) == 80 && sizeof(SegX) == 104, "SegD is the prefix of SegX");

// fp32 copy of what the depth DECISION needs (rays and plane normal: unit vectors, so their float roundings are absolute
// errors of 6e-8 per component); depths_positive32 below.  48 bytes: three 16-byte loads instead of the five of a SegD.
struct __attribute__((aligned(16))) SegD32 {
    float r1[3], r2[3], n[3], pad[3];
};
static_assert(sizeof(SegD32) == 48, "SegD32 is 48 bytes");

// fp32 pre-filter record of a segment in the target role: end point 1 relative to the image
// centre and the end-point difference.
struct __attribute__((aligned(16))) SegF {
    float qx, qy, dx, dy;
};

// phase-A / exchange slot (include/l3dpp_hip.h: l3d_slot)
struct __attribute__((aligned(16))) Slot {
    uint32_t tgt_seg;
    float overlap;
    float dp1, dp2, dq1, dq2;
    float score3D;
    uint32_t flags;
};
static_assert(sizeof(Slot) == 32, "slot is 32 bytes");
constexpr uint32_t kSlotAlive = 1u;     // survived the src view's orientation filter
constexpr uint32_t kSlotInvAlive = 2u;  // inverse (tgt-view) copy survived the tgt view's orientation filter

// reference Match (commons.h:186-203)
struct Match {
    uint32_t src_cam, src_seg, tgt_cam, tgt_seg;
    float overlap, score3D, dp1, dp2, dq1, dq2;
};
static_assert(sizeof(Match) == 40, "Match is 40 bytes");

// per-view constants + device pointers (device copy lives in an array indexed by view index)
struct ViewDev {
    double C[3];       // camera centre, translated frame
    double RtKinv[9];
    const float4* seg4;  // raw segments
    const SegF* segf;
    const SegX* segx;
    const SegD32* segd32;
    uint32_t M;
    uint32_t cam;
    float k;           // View::k_
    float cx, cy;      // image centre used by the fp32 pre-filter
    uint32_t pad;
};

// one directed view pair (line3D.cc:719-741)
struct PairDesc {
    double F[9];        // getFundamentalMatrix(src,tgt), row-major
    uint32_t src, tgt;  // view indices
    uint32_t Ms, Mt;
    uint32_t K;         // slots per src segment
    uint32_t row_off;   // first row of this pair in the per-row count array (kNN <= 0 mode)
    uint64_t slot_off;  // first slot of this pair in the slot buffer
    float cc_dist;      // |C_src - C_tgt| rounded up: bounds |P - C_other| of a hypothesis from its depth (k_lists.hip)
    uint32_t flags;     // kPairFastMath
    float B[3];         // C_tgt - C_src (double difference, rounded): the numerators of both triangulations are +-n.B
    float tolB;         // |n.B| above this: the float evaluation has the sign of the exact numerator (inf: never trusted)
};
static_assert(sizeof(PairDesc) == 128, "PairDesc is 128 bytes");

// ---- phase-B records ------------------------------------------------------------------------
// transposed index entry: slot (pair, src_row, j) is a potential inverse hypothesis of its target segment
struct InvRef {
    uint32_t src_view, src_row, pair, j;
};
static_assert(sizeof(InvRef) == 16, "InvRef is 16 bytes");

// the same hypothesis at its canonical position, with what scoring needs.  The unprojected 3D direction is NOT
// stored: it is recomputed from the owning segment's rays and (dp1, dp2) where needed (entry_dir, k_views.hip)
struct DEntry {
    uint32_t ref;          // slot index (l3d_match_begin limits the slot buffer to 2^32 slots)
    float dp1, dp2, reg1, reg2, score3D;
    uint32_t tgt_view;
    uint32_t flags : 8;
    uint32_t pair : 24;    // index of the directed pair the slot belongs to (< 2^24, checked by l3d_match_begin)
};
static_assert(sizeof(DEntry) == 32, "DEntry is 32 bytes");
constexpr uint32_t kDInverse = 1u;   // inverse hypothesis (exists only if the source view scored it > 0)
constexpr uint32_t kDZeroLen = 2u;   // unprojected segment shorter than L3D_EPS
constexpr uint32_t kDPresent = 8u;   // took part in scoring
constexpr uint32_t kDKeep = 16u;     // survives filterMatches

// one entry of estimated_position3D_: Segment3D + Match (line3D.cc:1637-1646)
struct HypRec {
    double P1[3], P2[3], dir[3];
    float length;
    uint32_t valid;
    Match m;
    uint32_t view, pad;
};
static_assert(sizeof(HypRec) == 128, "HypRec is 128 bytes");

// constants of similarityForScoring (line3D.cc:1417-1446) and the thresholds of its decision form (k_views.hip)
struct SimConst {
    float two_sigA_sqr;
    float y_thr;        // expf(y) > L3D_DEF_MIN_SIMILARITY_3D  <=>  y > y_thr
    float x_hi, x_lo;   // angular component > L3D_DEF_MIN_SIMILARITY_3D  <=>  x >= x_hi || x <= x_lo
};

// per-view scalars the affinity kernels read
struct ViewAff {
    float k;              // View::k_
    uint32_t pad;
};

// ---- the exact pair test (reference CPU semantics) -----------------------------------------

// Line3D::pointOnSegment, line3D.cc:1077-1083 (2D part of homogeneous points)
L3D_HD bool point_on_segment(double xx, double xy, double p1x, double p1y, double p2x, double p2y) {
    double v1x = p1x - xx, v1y = p1y - xy, v2x = p2x - xx, v2y = p2y - xy;
    return (v1x * v2x + v1y * v2y) < kEps;
}

// Line3D::mutualOverlap, line3D.cc:1086-1165, on 4 collinear points with z == 1.
// The reference takes float(norm) of all 6 point pairs and keeps the first strict maximum.  float(sqrt(x))
// is monotone in x, so when the largest squared distance exceeds every other one by more than 2^-21
// relative (=> their square roots differ by more than one float ulp) the winner is known without the
// other five square roots; only near-ties walk the reference's loop literally.  Same result bit for bit.
L3D_HD float mutual_overlap(const double px[4], const double py[4], bool fast = false) {
    if (!(point_on_segment(px[0], py[0], px[2], py[2], px[3], py[3]) ||
          point_on_segment(px[1], py[1], px[2], py[2], px[3], py[3]) ||
          point_on_segment(px[2], py[2], px[0], py[0], px[1], py[1]) ||
          point_on_segment(px[3], py[3], px[0], py[0], px[1], py[1])))
        return 0.0f;
    // pair order of the reference's double loop: (0,1),(0,2),(0,3),(1,2),(1,3),(2,3); the inner pair of
    // outer pair k is pair 5-k (line3D.cc:1125-1159)
    double d2[6];
    {
        double dx, dy;
        dx = px[0] - px[1]; dy = py[0] - py[1]; d2[0] = dx * dx + dy * dy;
        dx = px[0] - px[2]; dy = py[0] - py[2]; d2[1] = dx * dx + dy * dy;
        dx = px[0] - px[3]; dy = py[0] - py[3]; d2[2] = dx * dx + dy * dy;
        dx = px[1] - px[2]; dy = py[1] - py[2]; d2[3] = dx * dx + dy * dy;
        dx = px[1] - px[3]; dy = py[1] - py[3]; d2[4] = dx * dx + dy * dy;
        dx = px[2] - px[3]; dy = py[2] - py[3]; d2[5] = dx * dx + dy * dy;
    }
    double m2 = d2[0], inner2 = d2[5], second = 0.0;
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        if (d2[k] > m2) { second = m2; m2 = d2[k]; inner2 = d2[5 - k]; }
        else if (d2[k] > second) second = d2[k];
    }
    if (second < m2 * (1.0 - 4.76837158203125e-7)) {   // 2^-21: no float tie possible
        const float max_dist = (float)L3D_SQRT(fast, m2);
        if (max_dist < 1.0f) return 0.0f;
        return (float)L3D_DIV(fast, L3D_SQRT(fast, inner2), (double)max_dist);
    }
    // near-tie: the reference's loop, literally
    float max_dist = 0.0f;
    int outer = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float dist = (float)sqrt(d2[k]);
        if (dist > max_dist) { max_dist = dist; outer = k; }
    }
    if (max_dist < 1.0f) return 0.0f;
    double in2 = d2[5];
#pragma unroll
    for (int k = 1; k < 6; ++k) in2 = (outer == k) ? d2[5 - k] : in2;
    return (float)(sqrt(in2) / (double)max_dist);
}

// ---- fp32 pre-filter of the match kernel: "could overlap(src, tgt) exceed thr?" --------------------------------------
// (e?x, e?y): unit normals of the source segment's two epipolar lines in the target image, e?z their offsets w.r.t. the
// image centre; (qx, qy) = q1 - centre, (qz, qw) = q1 - q2 of the target segment.  s_i = a_i / d_i is the position of
// the intersection of epipolar line i with the target line in units of the target segment (q1 -> 0, q2 -> 1), and for
// intervals [s1, s2] that meet [0, 1] at all
//     inner = |clamp(s1) - clamp(s2)|,   inner + outer = |s1 - s2| + 1          (overlap = inner / outer)
// The test "inner > t * outer" with t = thr - kKappa0 - kKappa / min|d_i| (the slack covers the fp32 rounding of a and
// d, in pixels) is evaluated WITHOUT the two reciprocals (v_rcp_f32: 8 issue cycles each, a quarter of the round-2
// form's cost per target): multiplied through by |d1 d2| * min|d_i| > 0 it reads
//     X * (dmin + T) - T * Y > 0,    X = |A1 d2 - A2 d1|,  A_i = d_i * clamp(a_i / d_i) = median(0, a_i, d_i) (either
//     sign of d_i),  Y = |a1 d2 - a2 d1| + |d1 d2|,  T = t * dmin = (thr - kKappa0) * dmin - kKappa
// For intervals that miss [0, 1] the true overlap is zero and any answer is conservative.  A degenerate d (zero, NaN)
// gives u = 0 or NaN: "candidate", sorted out by the exact test.  tests/cpp/prefilter_cover.cpp pins, on the host,
// that no pair whose exact overlap exceeds thr is ever rejected.
constexpr float kKappa = 1.0e-2f;   // slack, px
constexpr float kKappa0 = 2.0e-4f;
L3D_HD float med3_zero(float a, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(0.0f, a, d);
#else
    const float lo = d < 0.0f ? d : 0.0f, hi = d < 0.0f ? 0.0f : d;
    return a < lo ? lo : (a > hi ? hi : a);   // (NaN a: stays NaN -> u NaN -> candidate)
#endif
}
L3D_HD bool prefilter_products(float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                               float qx, float qy, float qz, float qw, float thr) {
    const float a1 = __builtin_fmaf(e1x, qx, __builtin_fmaf(e1y, qy, e1z));
    const float a2 = __builtin_fmaf(e2x, qx, __builtin_fmaf(e2y, qy, e2z));
    const float d1 = __builtin_fmaf(e1x, qz, e1y * qw);
    const float d2 = __builtin_fmaf(e2x, qz, e2y * qw);
    const float A1 = med3_zero(a1, d1), A2 = med3_zero(a2, d2);
    const float X = __builtin_fmaf(A1, d2, -(A2 * d1));
    const float V = __builtin_fmaf(a1, d2, -(a2 * d1));
    const float Y = __builtin_fabsf(V) + __builtin_fabsf(d1 * d2);
    const float dmin = __builtin_fminf(__builtin_fabsf(d1), __builtin_fabsf(d2));
    const float T = __builtin_fmaf(thr - kKappa0, dmin, -kKappa);
    const float u = __builtin_fmaf(__builtin_fabsf(X), dmin + T, -(T * Y));
    return !(u < 0.0f);
}

struct PairResult {
    float overlap;
    float dp1, dp2, dq1, dq2;
};

// epipolar-overlap part of Line3D::matchingCPU, line3D.cc:919-958.  s = src segment, t = tgt
// segment (raw float pixels), F row-major.  Returns the overlap score (0 when the epipolar
// intersection is invalid).
L3D_HD float exact_overlap(const double* F, float sx1, float sy1, float sx2, float sy2, float tx1, float ty1,
                           float tx2, float ty2, bool fast = false) {
    d3 p1{(double)sx1, (double)sy1, 1.0}, p2{(double)sx2, (double)sy2, 1.0};
    d3 e1 = mul33(F, p1), e2 = mul33(F, p2);
    d3 q1{(double)tx1, (double)ty1, 1.0}, q2{(double)tx2, (double)ty2, 1.0};
    d3 l2 = cross(q1, q2);
    d3 x1 = cross(l2, e1), x2 = cross(l2, e2);
    if (!(fabs(x1.z) > kEps && fabs(x2.z) > kEps)) return 0.0f;
    double px[4], py[4];
    L3D_DIV2(fast, x1.x, x1.y, x1.z, px[0], py[0]);
    L3D_DIV2(fast, x2.x, x2.y, x2.z, px[1], py[1]);
    px[2] = q1.x; px[3] = q2.x; py[2] = q1.y; py[3] = q2.y;
    return mutual_overlap(px, py, fast);
}

// Line3D::triangulationDepths, line3D.cc:1168-1193, with the per-segment invariants hoisted:
// depths of the two rays ra, rb (camera centre Ca) w.r.t. the plane (n, cn = Cplane.n).
L3D_HD void tri_depths(const double* Ca, const double* ra, const double* rb, const double* n, double cn,
                       double& d1, double& d2, bool fast = false) {
    d3 N{n[0], n[1], n[2]}, A{ra[0], ra[1], ra[2]}, B{rb[0], rb[1], rb[2]}, C1{Ca[0], Ca[1], Ca[2]};
    double da = dot(A, N), db = dot(B, N);
    if (fabs(da) < kEps || fabs(db) < kEps) { d1 = -1.0; d2 = -1.0; return; }
    double num = cn - dot(N, C1);
    d1 = L3D_DIV(fast, num, da);
    d2 = L3D_DIV(fast, num, db);
}

// depth part of the acceptance test: line3D.cc:960-980 (both triangulations, all four depths > 1e-12)
L3D_HD bool exact_depths(const SegX& sx, const SegX& tx, const double* Cs, const double* Ct, PairResult& out, bool fast = false) {
    double ds1, ds2, dt1, dt2;
    tri_depths(Cs, sx.r1, sx.r2, tx.n, tx.cn, ds1, ds2, fast);
    tri_depths(Ct, tx.r1, tx.r2, sx.n, sx.cn, dt1, dt2, fast);
    if (!(ds1 > kEps && ds2 > kEps && dt1 > kEps && dt2 > kEps)) return false;
    out.dp1 = (float)ds1; out.dp2 = (float)ds2; out.dq1 = (float)dt1; out.dq2 = (float)dt2;
    return true;
}

// ---- the DECISION of the depth test without its divisions ----------------------------------------------------------
// exact_depths accepts iff num/da > L3D_EPS (IEEE division) for the four (num, da) pairs it forms.  For |da| >= L3D_EPS
// the quotient exceeds L3D_EPS iff num lies beyond L3D_EPS * da on the side of da's sign -- except when num / da is
// within a few ulp of L3D_EPS, where the roundings of the division and of the product could disagree.  tri_positive
// decides with one multiplication per depth and reports `certain = false` in that sliver (|num - eps*da| <= 2^-48 |eps*da|,
// a hundred times the rounding of either operation), where the caller falls back to the division itself.  Same da, db
// and num, operation for operation, as tri_depths.
L3D_HD bool tri_positive(const double* Ca, const double* ra, const double* rb, const double* n, double cn, bool& certain) {
    d3 N{n[0], n[1], n[2]}, A{ra[0], ra[1], ra[2]}, B{rb[0], rb[1], rb[2]}, C1{Ca[0], Ca[1], Ca[2]};
    const double da = dot(A, N), db = dot(B, N);
    if (fabs(da) < kEps || fabs(db) < kEps) return false;          // tri_depths: (-1, -1)
    const double num = cn - dot(N, C1);
    const double pa = kEps * da, pb = kEps * db;
    const double ea = num - pa, eb = num - pb;
    if (!(fabs(ea) > 3.5527136788005009e-15 * fabs(pa)) || !(fabs(eb) > 3.5527136788005009e-15 * fabs(pb))) certain = false;
    // num / da > eps  <=>  (da > 0 ? num > eps*da : num < eps*da)
    return (da > 0.0 ? ea > 0.0 : ea < 0.0) && (db > 0.0 ? eb > 0.0 : eb < 0.0);
}
// the decision of exact_depths (all four depths > L3D_EPS) for source record s and target record t (SegX or SegD)
template <class S, class T>
L3D_HD bool depths_positive(const S& sx, const T& tx, const double* Cs, const double* Ct) {
    bool certain = true;
    const bool a = tri_positive(Cs, sx.r1, sx.r2, tx.n, tx.cn, certain);
    const bool b = tri_positive(Ct, tx.r1, tx.r2, sx.n, sx.cn, certain);
    if (certain) return a && b;
    double ds1, ds2, dt1, dt2;                                      // the sliver: the reference's own arithmetic
    tri_depths(Cs, sx.r1, sx.r2, tx.n, tx.cn, ds1, ds2);
    tri_depths(Ct, tx.r1, tx.r2, sx.n, sx.cn, dt1, dt2);
    return ds1 > kEps && ds2 > kEps && dt1 > kEps && dt2 > kEps;
}

// ---- the same DECISION in fp32, with a certificate ----------------------------------------------------------------------
// num / da > L3D_EPS for the four (num, da) pairs of exact_depths, where num = cn_t - n_t.C_s = n_t.(C_t - C_s) for the
// source rays against the target's plane and num' = cn_s - n_s.C_t = -n_s.(C_t - C_s) for the target rays against the
// source's plane (up to the double rounding of cn, 1e-16 |C|).  Rays and normals are unit vectors and B = C_t - C_s is a
// constant of the pair, so the float dot products carry ABSOLUTE errors below 3e-7 (three products of factors rounded
// to 2^-24 each, two additions) resp. 3e-7 |B|.  When all six of them are at least ten times that far from zero
// (kDepthTol32, PairDesc::tolB = kDepthTol32 |B| + what the cancellation in cn can add) every exact quantity has the
// sign of its float value, |da| > 1e-6 is nowhere near the reference's |da| < L3D_EPS guard, and |num / da| >
// 3e-6 |B| is nowhere near L3D_EPS (the host sets tolB = inf for a pair with |B| < 1e-3, which then never takes this
// path): the decision is "num and da have the same sign", four multiplications.  Otherwise `certain` comes back false
// and the caller runs depths_positive on the double records -- a few candidates in a million on the BASELINE scenes.
// NaN anywhere fails the magnitude comparisons: uncertain.  tests/cpp/depth_sign.cpp pins certain => equal to exact_depths.
constexpr float kDepthTol32 = 4.0e-6f;
L3D_HD float dot3f(const float* a, const float* b) { return __builtin_fmaf(a[0], b[0], __builtin_fmaf(a[1], b[1], a[2] * b[2])); }
L3D_HD bool depths_positive32(const SegD32& s, const SegD32& t, const float* B, float tolB, bool& certain) {
    const float da = dot3f(s.r1, t.n), db = dot3f(s.r2, t.n), num = dot3f(t.n, B);
    const float ea = dot3f(t.r1, s.n), eb = dot3f(t.r2, s.n), mun = dot3f(s.n, B);      // second numerator = -mun
    certain = __builtin_fabsf(da) > kDepthTol32 && __builtin_fabsf(db) > kDepthTol32 && __builtin_fabsf(ea) > kDepthTol32 &&
              __builtin_fabsf(eb) > kDepthTol32 && __builtin_fabsf(num) > tolB && __builtin_fabsf(mun) > tolB;
    return num * da > 0.0f && num * db > 0.0f && mun * ea < 0.0f && mun * eb < 0.0f;
}

// full acceptance test of one (src seg, tgt seg) pair: line3D.cc:931-995
L3D_HD bool exact_pair(const double* F, const float4& s, const float4& t, const SegX& sx, const SegX& tx,
                       const double* Cs, const double* Ct, float thr, PairResult& out) {
    const float ov = exact_overlap(F, s.x, s.y, s.z, s.w, t.x, t.y, t.z, t.w);
    if (!(ov > thr)) return false;
    out.overlap = ov;
    return exact_depths(sx, tx, Cs, Ct, out);
}

// Segment3D(C + r1*d1, C + r2*d2): view.cc:356-371 + segment3D.h:47-66
struct Seg3 {
    d3 P1, P2, dir;
    float length;
};
L3D_HD Seg3 unproject(const double* C, const double* r1, const double* r2, float d1, float d2) {
    const d3 c{C[0], C[1], C[2]};
    const d3 a = c + d3{r1[0], r1[1], r1[2]} * (double)d1;
    const d3 b = c + d3{r2[0], r2[1], r2[2]} * (double)d2;
    Seg3 s;
    s.length = (float)norm(a - b);
    if (s.length > kEps) {
        s.P1 = a; s.P2 = b; s.dir = normalized(b - a);
    } else {
        s.P1 = d3{0, 0, 0}; s.P2 = d3{0, 0, 0}; s.dir = d3{0, 0, 0}; s.length = 0.0f;
    }
    return s;
}

// checkMatchOrientation: line3D.cc:831-839, view.cc:466-484
// The reference decides on acos(dp) in (PI/32, 31PI/32) with dp clamped to [-1,1].  acos is monotone, so the
// same decision is dp in [t.lo, t.hi] for two doubles the host finds once by bisection WITH THE SAME libm acos
// the reference links (l3d_api.hip: orientation_thresholds) -- exact agreement and no device-side acos.
struct OrientThr { double lo, hi; };
L3D_HD bool orientation_ok(const double* C, const SegX& sx, float d1, float d2, const OrientThr t) {
    const Seg3 s = unproject(C, sx.r1, sx.r2, d1, d2);
    const double dp = fmin(fmax(dot(d3{sx.rm[0], sx.rm[1], sx.rm[2]}, s.dir), -1.0), 1.0);
    return dp >= t.lo && dp <= t.hi;
}
// The same DECISION without the two square roots and three divisions of unproject (norm, normalized): with v = P2 - P1,
// c = rm . v and n2 = v . v the reference's dp is c / sqrt(n2) up to a few ulp, and dp in [lo, hi] with lo < 0 < hi
// (cos(31 PI/32) and cos(PI/32): checked by the caller of this form) is c^2 <= (c < 0 ? lo^2 : hi^2) * n2.  Within
// 1e-11 relative of that boundary -- four orders of magnitude more than the rounding of either form -- and for segments
// whose float length is not clearly above L3D_EPS, the reference's own arithmetic decides (orientation_ok).  v is
// computed operation for operation as unproject computes it.
L3D_HD bool orientation_ok_fast(const double* C, const SegX& sx, float d1, float d2, const OrientThr t) {
    const d3 c{C[0], C[1], C[2]};
    const d3 a = c + d3{sx.r1[0], sx.r1[1], sx.r1[2]} * (double)d1;
    const d3 b = c + d3{sx.r2[0], sx.r2[1], sx.r2[2]} * (double)d2;
    const d3 v = b - a;
    const double n2 = dot(v, v);
    const double cc = dot(d3{sx.rm[0], sx.rm[1], sx.rm[2]}, v);
    const double T = cc < 0.0 ? t.lo * t.lo : t.hi * t.hi;
    const double lhs = cc * cc, rhs = T * n2;
    // (n2 > 1e-20: the float length sqrt(n2) is far above L3D_EPS = 1e-12; NaN fails every comparison -> exact path)
    if (n2 > 1e-20 && fabs(lhs - rhs) > 1e-11 * rhs) return lhs < rhs;
    return orientation_ok(C, sx, d1, d2, t);
}

// (overlap desc, tgt asc) total order used for the kNN selection of rows WITHOUT equal overlaps; rows with equal
// overlaps are replayed in the reference's priority_queue order (l3d_heap.h, k_match_tied_rows)
L3D_HD bool better(float ova, uint32_t ia, float ovb, uint32_t ib) {
    return ova > ovb || (ova == ovb && ia < ib);
}

}  // namespace l3d
