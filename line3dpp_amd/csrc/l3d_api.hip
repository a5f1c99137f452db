// l3d_api.hip -- C-ABI entry points of libl3dpp_hip.so (include/l3dpp_hip.h) and the host-side
// driver that mirrors Line3D::addImage / matchImages / computingAffinityMatrix
// (line3D.cc:112-227, 375-497, 702-778, 1749-1778, 1852-1979).  No CPU fallback exists: every
// compute step is a HIP kernel launch; without a usable device the calls fail with L3D_ERR_HIP.
#include <algorithm>
#include <fstream>
#include <sstream>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <unordered_set>

#include "l3d_host.h"
#include "l3d_recon.h"

namespace l3d {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

// ---- prototypes of the launchers in k_views.hip / k_affinity.hip -------------------------------
hipError_t launch_scan(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* tmp, uint32_t* total, hipStream_t);
hipError_t launch_collin(int pass, const ViewDev* views, uint32_t n_views, uint32_t max_M, const uint32_t* seg_base,
                         float collin_t, uint32_t* cnt, const uint32_t* coll_off, uint32_t* coll_idx, hipStream_t);
hipError_t launch_aff_coll_count(int mode, uint32_t n_items, const uint32_t* surv_tg, const float* simv, const HypRec*,
                                 const uint32_t* seg_base, const uint32_t* coll_off, uint32_t* cnt, hipStream_t);
hipError_t launch_aff_coll_sim(int mode, uint32_t n_items, const uint32_t* surv_sg, const uint32_t* surv_tg,
                               const int32_t* hyp_of_seg, const HypRec*, const ViewDev*, const uint32_t* seg_base,
                               const uint32_t* gseg_view, const uint32_t* coll_off, const uint32_t* coll_idx,
                               const uint32_t* item_off, const ViewAff*, const float* medians, const float* msdl,
                               float two_sigA_sqr, uint32_t* out_seg, float* out_sim, hipStream_t);
hipError_t launch_seam_entries(uint32_t n, const float4* m4, const float2* rt, const ViewDev*, float k, DEntry*, hipStream_t);
hipError_t launch_seam_all_present(uint32_t G, const uint32_t* off, const uint32_t* boff, uint64_t* bits, hipStream_t);
hipError_t launch_seam_scores_out(uint32_t n, const DEntry*, float* scores, hipStream_t);
hipError_t launch_fill_gseg_view(const uint32_t* seg_base, uint32_t V, uint32_t max_M, uint32_t* gseg_view, hipStream_t);
hipError_t launch_orient_pairs(const ViewDev*, const PairDesc*, uint32_t n_pairs, uint64_t max_slots,
                               const uint32_t* seg_base, Slot* slots, unsigned long long* cnt_pack, uint32_t* inv_pos,
                               double thr_lo, double thr_hi, hipStream_t);
hipError_t launch_unpack_counts(uint32_t G, const unsigned long long* cnt_pack, uint32_t* cnt_all, uint32_t* cnt_inv,
                                hipStream_t);
hipError_t launch_inv_fill(const PairDesc*, uint32_t n_pairs, uint64_t max_slots, const uint32_t* seg_base,
                           const Slot* slots, const uint32_t* inv_off, const uint32_t* inv_pos, InvRef* refs,
                           hipStream_t);
hipError_t launch_build_lists_all(uint32_t G, const ViewDev*, const PairDesc*, const uint32_t* seg_base,
                                  const uint32_t* gseg_view, const uint32_t* vout_off, const uint32_t* vout_pairs,
                                  const uint32_t* off, const uint32_t* inv_off, const InvRef*, const Slot*, DEntry*,
                                  uint32_t* eref, uint32_t uniform_K, hipStream_t);
hipError_t launch_bits_len(uint32_t G, const uint32_t* off, uint32_t* len, uint32_t* long_list, uint32_t* n_long,
                           hipStream_t);
hipError_t launch_support_long(uint32_t n_long, const uint32_t* long_list, const uint32_t* off, const uint32_t* boff,
                               const DEntry*, uint64_t* bits, const ViewDev*, const uint32_t* gseg_view, SimConst,
                               hipStream_t);
hipError_t launch_support_all(uint32_t g0, uint32_t G, const uint32_t* off, const uint32_t* boff, const DEntry*, uint64_t* bits,
                              const ViewDev*, const uint32_t* seg_base, const uint32_t* gseg_view, SimConst, hipStream_t);
hipError_t launch_presence_view(uint32_t g0, uint32_t M, const uint32_t* off, const uint32_t* boff,
                                const uint32_t* inv_off, const uint32_t* eref, uint64_t* bits, uint8_t* positive,
                                hipStream_t);
hipError_t launch_score_all(uint32_t g0, uint32_t G, const uint32_t* off, const uint32_t* boff, const uint32_t* gseg_view, DEntry*,
                            const uint64_t* bits, Slot*, uint32_t* max_score_bits, const ViewDev*, const uint32_t* seg_base,
                            SimConst, hipStream_t);
hipError_t launch_filter_all(uint32_t G, const uint32_t* off, const uint32_t* gseg_view, DEntry*,
                             const uint32_t* max_score_bits, uint32_t* surv_cnt, uint32_t* has_best,
                             uint32_t* best_pos, hipStream_t);
hipError_t launch_filter_write_all(const ViewDev*, const PairDesc*, const uint32_t* seg_base, uint32_t G,
                                   const uint32_t* gseg_view, const uint32_t* off, const DEntry*, const Slot*,
                                   const uint32_t* surv_off, const uint32_t* hyp_off, const uint32_t* best_pos,
                                   Match* surv, uint32_t* surv_tg, uint32_t* surv_sg, int32_t* hyp_of_seg, HypRec*,
                                   float* depths, hipStream_t);
hipError_t launch_median_all(uint32_t V, const float* depths, const uint32_t* hyp_off, const uint32_t* seg_base,
                             float* out, hipStream_t);
hipError_t launch_aff_sim(uint32_t N, const uint32_t* surv_sg, const uint32_t* surv_tg, const int32_t* hyp_of_seg,
                          const HypRec*, const ViewAff*, const float* medians, const float* msdl, float two_sigA_sqr,
                          float* simv, int32_t* ca, int32_t* cb, hipStream_t);
hipError_t launch_aff_flag(uint32_t N, const uint32_t* surv_off, const uint32_t* surv_sg, const uint32_t* surv_tg,
                           const float* simv, const int32_t* ca, const int32_t* cb, uint32_t* flag, hipStream_t);
hipError_t launch_fill_u32(uint32_t*, uint32_t n, uint32_t val, hipStream_t);
hipError_t launch_aff_touch(uint32_t N, const uint32_t* flag, const uint32_t* epos, const int32_t* ca,
                            const int32_t* cb, uint32_t* first_touch, hipStream_t);
hipError_t launch_aff_mark(uint32_t H, const uint32_t* first_touch, uint32_t* touch_flag, hipStream_t);
hipError_t launch_aff_emit(uint32_t N, const uint32_t* flag, const uint32_t* epos, const int32_t* ca,
                           const int32_t* cb, const float* simv, const uint32_t* first_touch,
                           const uint32_t* touch_rank, const HypRec*, void* edges, void* local2global, hipStream_t);

}  // namespace l3d

using namespace l3d;

struct l3d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::recursive_mutex mu;                                  // view_mutex_/view_reserve_mutex_ stand-in
    std::map<uint32_t, std::unique_ptr<HostView>> views;   // views_ (ascending camID)
    std::vector<HostView*> order;                   // view index -> view (ascending camID)
    std::vector<float> views_avg_depths;            // views_avg_depths_
    // params (matchImages)
    float sigma_p = 2.5f, sigma_a = 10.0f, two_sigA_sqr = 200.0f, epipolar_overlap = 0.25f;
    float const_regularization_depth = -1.0f, med_scene_depth = (float)kEps, med_scene_depth_lines = 0.0f;
    int kNN = 10, num_neighbors = 10;
    bool fixed3Dregularizer = false;
    bool brute = false;                             // test hook: disable the fp32 pre-filter
    double orient_lo = -1.0, orient_hi = 1.0;       // dp window equivalent to acos(dp) in (PI/32, 31PI/32)
    d3 translation{0, 0, 0};
    // state
    enum { IDLE, BEGUN, MATCHED } state = IDLE;
    bool affinity_done = false;
    std::vector<PairDesc> pairs;
    std::vector<uint32_t> pair_src_cam, pair_tgt_cam;
    std::vector<char> pair_done;
    uint64_t n_slots = 0, pair_tests = 0;
    uint32_t n_rows_total = 0;
    // device
    DevBuf<ViewDev> d_views;
    DevBuf<PairDesc> d_pairs;
    DevBuf<WorkItem> d_work;
    DevBuf<Slot> d_slots;
    DevBuf<uint32_t> d_slot_idx;   // compact exchange form of d_slots (N > 1 ranks): target index per slot
    // epipolar-band culling pools (l3d_kernels.h)
    std::vector<PairCull> cull;
    DevBuf<PairCull> d_cull;
    DevBuf<uint32_t> d_src_perm, d_tgt_perm;
    DevBuf<float2> d_src_band, d_chunk_band, d_tgt_band;
    DevBuf<float4> d_tgt_sf;
    bool use_cull = true;
    unsigned visibility_t = 3;                      // visibility_t_ / perform_RDD_ of the last reconstruct3Dlines
    bool perform_rdd = false;
    // A_ / local2global_ stay on the device; the host copies (edges, l2g) are fetched on first use
    uint32_t aff_n_edges = 0, aff_n_rows = 0;
    bool aff_host_valid = true;
    PinnedBuf<uint32_t> h_cnt;
    hipStream_t aux[2] = {nullptr, nullptr};        // aux[0] high priority (phase-A first half, the chain), aux[1]
    std::vector<hipEvent_t> pipe_ev;
    std::vector<uint8_t> pair_counted;   // the pair's slots carry orientation flags and are in the phase-B counters
    hipEvent_t sev[5] = {};                         // prepared, half A done, half B done, memsets done, orient A done
    DevBuf<SegX> d_gsegx;                           // SegX of every segment, global segment order
    float collinearity_t = -1.0f;                   // collinearity_t_ (reconstruct3Dlines); > 0: collinear links
    DevBuf<uint32_t> d_coll_cnt, d_coll_off, d_coll_idx, d_item_cnt, d_item_off, d_item_seg;
    DevBuf<float> d_item_sim;
    // pinned staging of the small host->device tables (reused across calls; every public call ends synchronised)
    PinnedBuf<ViewDev> h_views;
    PinnedBuf<PairDesc> h_pairs;
    PinnedBuf<PairCull> h_cull;
    PinnedBuf<WorkItem> h_work;
    PinnedBuf<uint32_t> h_vout, h_small;
    bool timing_pending = false;                    // phase-A events recorded but not read yet
    uint32_t pending_launches = 0;
    DevBuf<uint32_t> d_row_counts;
    // phase B (global over all views; G = sum of M)
    uint32_t G = 0, n_ents = 0, n_surv = 0, n_hyps = 0;
    std::vector<uint32_t> seg_base;                 // [V+1]
    DevBuf<uint32_t> d_seg_base, d_gseg_view, d_cnt, d_off, d_scan_tmp, d_scal, d_max_score;
    DevBuf<uint32_t> d_surv_cnt, d_has_best, d_best_pos, d_surv_off, d_hyp_off, d_surv_tg, d_surv_sg;
    DevBuf<InvRef> d_refs;
    DevBuf<uint64_t> d_bits;
    DevBuf<uint32_t> d_eref;
    DevBuf<uint8_t> d_positive;
    DevBuf<uint32_t> d_bits_len, d_boff, d_long_list;
    DevBuf<uint32_t> d_cnt_inv, d_inv_off, d_vout_pairs, d_vout_off, d_inv_pos;
    DevBuf<unsigned long long> d_cnt_pack;
    std::vector<uint32_t> vout_off;
    DevBuf<DEntry> d_dents;
    DevBuf<Match> d_surv;
    DevBuf<int32_t> d_hyp_of_seg;
    DevBuf<float> d_depths, d_medians;              // d_medians[V]
    DevBuf<HypRec> d_hyps;
    std::vector<uint32_t> h_surv_off, h_hyp_off;    // lazily fetched for the accessors
    bool host_offsets_valid = false;
    // affinity
    DevBuf<ViewAff> d_vaff;
    DevBuf<float> d_simv, d_msdl;
    DevBuf<int32_t> d_ca, d_cb;
    DevBuf<uint32_t> d_flag, d_epos, d_first_touch, d_touch_flag, d_touch_rank;
    DevBuf<l3d_cledge> d_edges;
    DevBuf<l3d_segment2d> d_l2g;
    std::vector<l3d_cledge> edges;
    std::vector<l3d_segment2d> l2g;
    std::vector<ReconLine> lines3D;                 // lines3D_ (original frame)
    bool lines_done = false;
    // timings
    hipEvent_t ev[10] = {};
    l3d_timings tm{};
};

namespace {

int fail(int code, const std::string& msg) { set_error(msg); return code; }

// View::View, view.cc:6-42
void init_view(HostView& v, const double K[9], const double R[9], const double t[3]) {
    std::memcpy(v.K.m, K, 72);
    std::memcpy(v.R.m, R, 72);
    v.t = d3{t[0], t[1], t[2]};
    v.pp = d3{v.K.m[2], v.K.m[5], 1.0};
    v.Kinv = m3_inv(v.K);
    v.Rt = m3_t(v.R);
    v.RtKinv = m3_mul(v.Rt, v.Kinv);
    v.C = mul33(v.Rt.m, d3{-1.0 * v.t.x, -1.0 * v.t.y, -1.0 * v.t.z});
}

// View::translate, view.cc:510-514
void translate_view(HostView& v, const d3& d) {
    v.C = v.C + d;
    const d3 rc = mul33(v.R.m, v.C);
    v.t = d3{-rc.x, -rc.y, -rc.z};
}

// Line3D::translate, line3D.cc:500-536
void translate(l3d_ctx& c) {
    double tr[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        std::vector<double> coords;
        for (auto* v : c.order) {
            const double val = i == 0 ? v->C.x : (i == 1 ? v->C.y : v->C.z);
            if (std::fabs(val) > kEps) coords.push_back(val);
        }
        if (!coords.empty()) {
            std::sort(coords.begin(), coords.end());
            tr[i] = coords[coords.size() / 2];
        }
    }
    c.translation = d3{tr[0], tr[1], tr[2]};
    for (auto* v : c.order) translate_view(*v, d3{-tr[0], -tr[1], -tr[2]});
}
void untranslate(l3d_ctx& c) {  // line3D.cc:539-545
    for (auto* v : c.order) translate_view(*v, c.translation);
}

// View::getSpecificSpatialReg, view.cc:307-314
float spatial_reg(const HostView& v, float r) {
    const d3 a = normalized(mul33(v.RtKinv.m, v.pp));
    const d3 b = normalized(mul33(v.RtKinv.m, v.pp + d3{(double)r, 0.0, 0.0}));
    const double alpha = std::acos(std::fmin(std::fmax(dot(a, b), -1.0), 1.0));
    return (float)std::sin(alpha);
}

// Line3D::getFundamentalMatrix, line3D.cc:874-892
void fundamental(const HostView& s, const HostView& t, double F[9]) {
    const M3 R = m3_mul(t.R, m3_t(s.R));
    const d3 Rt1 = mul33(R.m, s.t);
    const d3 tt = t.t - Rt1;
    const M3 T{{0.0, -tt.z, tt.y, tt.z, 0.0, -tt.x, -tt.y, tt.x, 0.0}};
    const M3 E = m3_mul(T, R);
    const M3 Fm = m3_mul(m3_mul(m3_inv(m3_t(t.K)), E), m3_inv(s.K));
    std::memcpy(F, Fm.m, 72);
}

// Epipolar-band culling set-up for one directed pair (k_match.hip / l3d_kernels.h PairCull): the epipole e in the
// target image is the left null vector of F; the transversal runs through the target image centre c,
// perpendicular to the direction from c towards e.  tau(x) is where the pencil line through x crosses it:
//     target point/direction q:  l = e x q,  tau = -(l.c)/(l.n) = (At.q)/(Bt.q),  At = e x c, Bt = n x e
//     source point p:            l = F p,    tau = -(l.c)/(l.n) = (As.p)/(Bs.p),  As = -F^T c, Bs = F^T n
// Culling is only enabled when both denominators keep one sign (with margin) over the respective image, i.e.
// no epipolar line that can occur is near-parallel to the transversal -- otherwise the pair streams unculled.
void make_cull(const double F[9], double ws, double hs, double wt, double ht, PairCull& pc) {
    pc.enabled = 0;
    auto col = [&](int j) { return d3{F[j], F[3 + j], F[6 + j]}; };
    double fn = 0;
    for (int i = 0; i < 9; ++i) fn = std::fmax(fn, std::fabs(F[i]));
    if (!(fn > 0.0) || !std::isfinite(fn)) return;
    d3 e{0, 0, 0}; double best = 0;
    const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (auto& q : pr) {
        const d3 x = cross(col(q[0]), col(q[1]));
        const double n = norm(x);
        if (n > best) { best = n; e = x; }
    }
    if (!(best > 1e-30 * fn * fn)) return;
    e = e * (1.0 / best);
    for (int j = 0; j < 3; ++j)   // rank check: e^T F == 0 (all lines F p concurrent in e)
        if (std::fabs(dot(e, col(j))) > 1e-10 * norm(col(j)) + 1e-300) return;
    const d3 c{0.5 * wt, 0.5 * ht, 1.0};
    double mx = e.x - c.x * e.z, my = e.y - c.y * e.z;          // direction centre -> epipole (up to sign)
    const double ml = std::sqrt(mx * mx + my * my);
    if (!(ml > 1e-12 * (std::fabs(e.x) + std::fabs(e.y) + std::fabs(e.z)))) return;
    mx /= ml; my /= ml;
    const d3 n{-my, mx, 0.0};
    d3 At = cross(e, c), Bt = cross(n, e);
    const double st = dot(Bt, c);
    if (!(std::fabs(st) > 1e-300)) return;
    At = At * (1.0 / st); Bt = Bt * (1.0 / st);
    // As = -F^T c, Bs = F^T n
    d3 As{-(F[0] * c.x + F[3] * c.y + F[6] * c.z), -(F[1] * c.x + F[4] * c.y + F[7] * c.z),
          -(F[2] * c.x + F[5] * c.y + F[8] * c.z)};
    d3 Bs{F[0] * n.x + F[3] * n.y, F[1] * n.x + F[4] * n.y, F[2] * n.x + F[5] * n.y};
    const d3 cs{0.5 * ws, 0.5 * hs, 1.0};
    const double ss = dot(Bs, cs);
    if (!(std::fabs(ss) > 1e-300)) return;
    As = As * (1.0 / ss); Bs = Bs * (1.0 / ss);
    auto ok = [](const d3& B, double w, double h) {
        const double mw = 0.05 * w, mh = 0.05 * h;
        const double xs[2] = {-mw, w + mw}, ys[2] = {-mh, h + mh};
        for (double x : xs) for (double y : ys) if (!(B.x * x + B.y * y + B.z >= 0.1)) return false;
        return true;
    };
    if (!ok(Bt, wt, ht) || !ok(Bs, ws, hs)) return;
    for (double v : {At.x, At.y, At.z, Bt.x, Bt.y, Bt.z, As.x, As.y, As.z, Bs.x, Bs.y, Bs.z})
        if (!std::isfinite(v)) return;
    pc.As[0] = As.x; pc.As[1] = As.y; pc.As[2] = As.z; pc.Bs[0] = Bs.x; pc.Bs[1] = Bs.y; pc.Bs[2] = Bs.z;
    pc.At[0] = At.x; pc.At[1] = At.y; pc.At[2] = At.z; pc.Bt[0] = Bt.x; pc.Bt[1] = Bt.y; pc.Bt[2] = Bt.z;
    pc.enabled = 1;
}

int upload_views(l3d_ctx& c) {
    const size_t V = c.order.size();
    L3D_HIP_CHECK(c.d_views.reserve(V));
    L3D_HIP_CHECK(c.h_views.reserve(V));
    ViewDev* hv = c.h_views.p;
    uint32_t max_M = 0;
    // the per-segment invariants of all views live in ONE array indexed by the global segment id, so that the
    // phase-B kernels reach them with one load (gsegx[g]) instead of g -> view -> pointer -> record
    size_t Gtot = 0;
    for (size_t i = 0; i < V; ++i) Gtot += c.order[i]->M;
    L3D_HIP_CHECK(c.d_gsegx.reserve(std::max<size_t>(Gtot, 1)));
    size_t gbase = 0;
    for (size_t i = 0; i < V; ++i) {
        HostView& v = *c.order[i];
        ViewDev& d = hv[i];
        d.C[0] = v.C.x; d.C[1] = v.C.y; d.C[2] = v.C.z;
        std::memcpy(d.RtKinv, v.RtKinv.m, 72);
        d.seg4 = v.d_seg4.p; d.segf = v.d_segf.p; d.segx = c.d_gsegx.p + gbase;
        gbase += v.M;
        d.M = v.M; d.cam = v.cam; d.k = v.k;
        d.cx = 0.5f * (float)v.width; d.cy = 0.5f * (float)v.height; d.pad = 0;
        max_M = std::max(max_M, v.M);
    }
    L3D_HIP_CHECK(hipMemcpyAsync(c.d_views.p, hv, V * sizeof(ViewDev), hipMemcpyHostToDevice, c.stream));
    L3D_HIP_CHECK(launch_prep_views(c.d_views.p, (uint32_t)V, max_M, c.stream));
    return L3D_OK;
}

// checkMatchOrientation keeps a match iff L3D_PI_1_32 < acos(dp) < L3D_PI_31_32 (line3D.cc:836, double compare
// against the float constants).  acos is monotone non-increasing on [-1,1]: find by bisection over the doubles,
// with this libm's acos, the largest dp with acos(dp) > PI/32 and the smallest dp with acos(dp) < 31PI/32.
void orientation_thresholds(double& lo, double& hi) {
    const double a1 = (double)kPi_1_32, a2 = (double)kPi_31_32;
    double x = -1.0, y = 1.0;                 // pred(d) = acos(d) > a1 : true at -1, false at 1
    for (int it = 0; it < 200 && std::nextafter(x, y) < y; ++it) { const double m = 0.5 * (x + y); if (std::acos(m) > a1) x = m; else y = m; }
    hi = x;
    x = -1.0; y = 1.0;                        // pred(d) = acos(d) < a2 : false at -1, true at 1
    for (int it = 0; it < 200 && std::nextafter(x, y) < y; ++it) { const double m = 0.5 * (x + y); if (std::acos(m) < a2) y = m; else x = m; }
    lo = y;
}

// Thresholds of the decision form of similarityForScoring (k_views.hip sim_decide): the smallest/largest float
// arguments for which the reference's own expressions (libm expf / acos as line3D.cc:1428,1438,1575 call them)
// exceed L3D_DEF_MIN_SIMILARITY_3D.  Every expression is monotone in its argument, so bisection over floats
// finds the exact switch points.
template <class Pred>
float first_true(float lo, float hi, Pred pred) {   // pred(lo) false, pred(hi) true, monotone; returns first true
    for (int it = 0; it < 300; ++it) {
        const float mid = lo + 0.5f * (hi - lo);
        if (!(mid > lo && mid < hi)) break;
        if (pred(mid)) hi = mid; else lo = mid;
    }
    return hi;
}
SimConst sim_thresholds(float two_sigA_sqr) {
    SimConst sc;
    sc.two_sigA_sqr = two_sigA_sqr;
    const float min_sim = 0.5f;   // L3D_DEF_MIN_SIMILARITY_3D
    auto pe = [&](float y) { return expf(y) > min_sim; };
    // y_thr = largest y that fails
    const float first = first_true(-2.0f, 0.0f, pe);
    sc.y_thr = std::nextafterf(first, -INFINITY);
    auto pa = [&](float x) {   // angleBetweenSeg3D(.., undirected) + sim_a, line3D.cc:1571-1583, 1428
        float angle = (float)(std::acos(std::fmax(std::fmin((double)x, 1.0), -1.0)) / M_PI * 180.0f);
        if (angle > 90.0f) angle = 180.0f - angle;
        return expf(-angle * angle / two_sigA_sqr) > min_sim;
    };
    if (pa(0.0f)) { sc.x_hi = 0.0f; sc.x_lo = 0.0f; return sc; }   // every direction passes
    sc.x_hi = first_true(0.0f, 1.0f, pa);
    sc.x_lo = -first_true(0.0f, 1.0f, [&](float m) { return pa(-m); });
    return sc;
}

// L3D_TRACE=1: host-side wall-clock checkpoints of matchImages on stderr (diagnostics)
struct HostTrace {
    bool on = std::getenv("L3D_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<std::pair<double, const char*>> marks;   // printed by flush(): printing inside the timeline distorts it
    void mark(const char* what) {
        if (!on) return;
        marks.emplace_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
    }
    void flush() {
        for (auto& m : marks) std::fprintf(stderr, "[l3d trace] %9.1f us  %s\n", m.first, m.second);
        marks.clear();
    }
};
static HostTrace g_trace;

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

}  // namespace

extern "C" {

const char* l3d_last_error(void) { return g_err.c_str(); }
const char* l3d_build_info(void) { return "libl3dpp_hip gfx950 hip fp-contract=off"; }

l3d_ctx* l3d_create(int device, void* stream) {
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed: no usable HIP device"); return nullptr; }
    auto* c = new l3d_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    orientation_thresholds(c->orient_lo, c->orient_hi);
    c->use_cull = std::getenv("L3D_NO_CULL") == nullptr;   // diagnostic switch: stream every pair unculled
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); delete c; return nullptr; }
    return c;
}

void l3d_destroy(l3d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->views) {
        HostView& v = *kv.second;
        v.d_seg4.release(); v.d_segf.release();
    }
    c->d_views.release(); c->d_pairs.release(); c->d_work.release(); c->d_slots.release(); c->d_slot_idx.release();
    c->h_views.release(); c->h_pairs.release(); c->h_cull.release(); c->h_work.release(); c->h_vout.release();
    c->h_small.release(); c->h_cnt.release();
    c->d_cull.release(); c->d_src_perm.release(); c->d_tgt_perm.release(); c->d_src_band.release();
    c->d_chunk_band.release(); c->d_tgt_sf.release(); c->d_tgt_band.release();
    c->d_row_counts.release();
    c->d_seg_base.release(); c->d_gseg_view.release(); c->d_cnt.release(); c->d_off.release();
    c->d_scan_tmp.release(); c->d_scal.release(); c->d_max_score.release(); c->d_surv_cnt.release();
    c->d_has_best.release(); c->d_best_pos.release(); c->d_surv_off.release(); c->d_hyp_off.release();
    c->d_surv_tg.release(); c->d_surv_sg.release(); c->d_refs.release(); c->d_bits.release(); c->d_eref.release(); c->d_positive.release(); c->d_bits_len.release(); c->d_boff.release(); c->d_cnt_inv.release(); c->d_long_list.release();
    c->d_cnt_pack.release(); c->d_inv_pos.release(); c->d_gsegx.release();
    c->d_coll_cnt.release(); c->d_coll_off.release(); c->d_coll_idx.release(); c->d_item_cnt.release();
    c->d_item_off.release(); c->d_item_seg.release(); c->d_item_sim.release();
    c->d_inv_off.release(); c->d_vout_pairs.release(); c->d_vout_off.release(); c->d_dents.release(); c->d_surv.release();
    c->d_hyp_of_seg.release(); c->d_depths.release(); c->d_medians.release(); c->d_hyps.release();
    c->d_vaff.release(); c->d_simv.release(); c->d_msdl.release(); c->d_ca.release(); c->d_cb.release();
    c->d_flag.release(); c->d_epos.release(); c->d_first_touch.release(); c->d_touch_flag.release();
    c->d_touch_rank.release(); c->d_edges.release(); c->d_l2g.release();
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->pipe_ev) (void)hipEventDestroy(e);
    for (auto& e : c->sev) if (e) (void)hipEventDestroy(e);
    for (auto& s2 : c->aux) if (s2) (void)hipStreamDestroy(s2);
    delete c;
}

int l3d_slots_exchanged(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, "l3d_match_begin must precede l3d_slots_exchanged");
    std::fill(c->pair_done.begin(), c->pair_done.end(), 1);
    return L3D_OK;
}

int l3d_set_brute_force(l3d_ctx* c, int on) {  // test hook
    if (!c) return L3D_ERR_ARG;
    c->brute = on != 0;
    return L3D_OK;
}

int l3d_add_view(l3d_ctx* c, uint32_t camID, const float* segs4, uint32_t M, const double K[9], const double R[9],
                 const double t[3], uint32_t width, uint32_t height, float median_depth, const uint32_t* neighbors,
                 uint32_t n_neighbors) {
    if (!c || !K || !R || !t) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (std::max(width, height) < 800) return fail(L3D_ERR_IMAGE_SMALL, "image is too small for reliable results");
    if (c->views.count(camID)) return fail(L3D_ERR_ID_IN_USE, "camera ID already in use");
    if (n_neighbors == 0 || !neighbors) return fail(L3D_ERR_NO_NEIGHBORS, "view has no visual neighbors");
    if (M == 0 || !segs4) return fail(L3D_ERR_NO_SEGMENTS, "no line segments");
    if (M >= (1u << 23)) return fail(L3D_ERR_LIMIT, "more than 2^23 segments per view");
    (void)hipSetDevice(c->device);
    auto v = std::make_unique<HostView>();
    v->cam = camID; v->M = M;
    v->segs.assign(segs4, segs4 + 4 * (size_t)M);
    v->width = width; v->height = height;
    v->initial_median_depth = (float)std::fmax(std::fabs(median_depth), kEps);
    init_view(*v, K, R, t);
    v->fixed_nbrs.assign(neighbors, neighbors + n_neighbors);
    L3D_HIP_CHECK(v->d_seg4.reserve(M));
    L3D_HIP_CHECK(v->d_segf.reserve(M));
    L3D_HIP_CHECK(hipMemcpy(v->d_seg4.p, v->segs.data(), (size_t)M * 16, hipMemcpyHostToDevice));
    c->views_avg_depths.push_back((float)std::fmax(median_depth, kEps));
    c->views[camID] = std::move(v);
    c->order.clear();
    for (auto& kv : c->views) { kv.second->index = (uint32_t)c->order.size(); c->order.push_back(kv.second.get()); }
    c->state = l3d_ctx::IDLE;
    c->affinity_done = false;
    return L3D_OK;
}

int l3d_match_begin(l3d_ctx* c, const l3d_match_params* p) {
    if (!c || !p) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->views.empty()) return fail(L3D_ERR_NO_VIEWS, "no images to match");
    (void)hipSetDevice(c->device);
    // parameter clamps, line3D.cc:394-413
    c->num_neighbors = std::max(int(p->num_neighbors), 2);
    c->sigma_p = p->sigma_position;
    c->sigma_a = std::fmin(std::fabs(p->sigma_angle), 90.0f);
    c->two_sigA_sqr = 2.0f * c->sigma_a * c->sigma_a;
    c->epipolar_overlap = std::fmin(std::fabs(p->epipolar_overlap), 0.99f);
    c->kNN = p->kNN;
    c->const_regularization_depth = p->const_regularization_depth;
    if (c->sigma_p < 0.0f) { c->fixed3Dregularizer = true; c->sigma_p = std::fabs(c->sigma_p); }
    else { c->fixed3Dregularizer = false; c->sigma_p = std::fmax(0.1f, c->sigma_p); }
    if (c->kNN > 4096) return fail(L3D_ERR_LIMIT, "kNN > 4096");
    c->affinity_done = false;
    c->lines_done = false;
    c->n_hyps = 0;
    // line3D.cc:426-433
    c->med_scene_depth = c->const_regularization_depth;
    if (c->const_regularization_depth < 0.0f && c->fixed3Dregularizer && !c->views_avg_depths.empty()) {
        std::sort(c->views_avg_depths.begin(), c->views_avg_depths.end());
        c->med_scene_depth = c->views_avg_depths[c->views_avg_depths.size() / 2];
    }
    L3D_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    translate(*c);
    for (auto* v : c->order) {
        if (!c->fixed3Dregularizer) v->k = spatial_reg(*v, c->sigma_p);      // computeSpatialRegularizer
        else v->k = c->sigma_p / c->med_scene_depth;                          // update_k, view.h:124-127
        v->median_depth = 0.0f;
        v->out_pairs.clear(); v->in_pairs.clear();
    }
    // fixed neighbours, line3D.cc:467-479 (sets persist across calls like visual_neighbors_)
    for (auto* v : c->order)
        if (v->visual_nbrs.empty())
            for (uint32_t n : v->fixed_nbrs)
                if (c->views.count(n)) v->visual_nbrs.insert(n);
    // directed pair list, line3D.cc:704-741
    c->pairs.clear(); c->pair_src_cam.clear(); c->pair_tgt_cam.clear(); c->cull.clear();
    uint64_t cs_off = 0, ct_off = 0; uint32_t cc_off = 0;
    std::map<uint32_t, std::set<uint32_t>> matched;
    uint64_t slot_off = 0; uint32_t row_off = 0;
    c->pair_tests = 0;
    for (auto* v : c->order)
        for (uint32_t tcam : v->visual_nbrs) {
            if (matched[v->cam].count(tcam)) continue;
            HostView* t = c->views[tcam].get();
            PairDesc pd;
            fundamental(*v, *t, pd.F);
            pd.src = v->index; pd.tgt = t->index; pd.Ms = v->M; pd.Mt = t->M;
            pd.K = c->kNN > 0 ? (uint32_t)c->kNN : 0u;
            pd.row_off = row_off; pd.slot_off = slot_off;
            slot_off += (uint64_t)pd.Ms * pd.K; row_off += pd.Ms;
            const uint32_t pi = (uint32_t)c->pairs.size();
            v->out_pairs.push_back(pi);
            if (t->index > v->index) t->in_pairs.push_back(pi);   // inverse only if tgt not yet processed (:1680)
            c->pairs.push_back(pd);
            PairCull pc{};
            if (c->use_cull && c->kNN > 0 && pd.Ms <= kCullMaxSegs && pd.Mt <= kCullMaxSegs && pd.Ms && pd.Mt)
                make_cull(pd.F, v->width, v->height, t->width, t->height, pc);
            pc.s_off = cs_off; pc.t_off = ct_off; pc.c_off = cc_off;
            if (pc.enabled) { cs_off += pd.Ms; ct_off += pd.Mt; cc_off += (pd.Mt + 63) / 64; }
            c->cull.push_back(pc);
            c->pair_src_cam.push_back(v->cam); c->pair_tgt_cam.push_back(tcam);
            c->pair_tests += (uint64_t)pd.Ms * pd.Mt;
            matched[v->cam].insert(tcam); matched[tcam].insert(v->cam);
        }
    c->n_slots = slot_off; c->n_rows_total = row_off;
    if (c->n_slots >= (1ull << 32) || c->pairs.size() >= (1u << 24))
        return fail(L3D_ERR_LIMIT, "slot buffer / pair list exceed the 32-bit slot and 24-bit pair indices of phase B");
    c->pair_done.assign(c->pairs.size(), 0);
    c->pair_counted.assign(c->pairs.size(), 0);
    int rc = upload_views(*c);
    if (rc) return rc;
    {   // packed hypothesis counters of phase B: fed by the match epilogue (bounded kNN) or by k_orient_all
        uint64_t G = 0;
        for (auto* v : c->order) G += v->M;
        L3D_HIP_CHECK(c->d_cnt_pack.reserve(G + 1));
        L3D_HIP_CHECK(hipMemsetAsync(c->d_cnt_pack.p, 0, (G + 1) * 8, c->stream));
        if (c->kNN > 0) L3D_HIP_CHECK(c->d_inv_pos.reserve(std::max<uint64_t>(c->n_slots, 1)));
    }
    L3D_HIP_CHECK(c->d_pairs.reserve(std::max<size_t>(c->pairs.size(), 1)));
    if (!c->pairs.empty()) {
        L3D_HIP_CHECK(c->h_pairs.reserve(c->pairs.size()));
        std::memcpy(c->h_pairs.p, c->pairs.data(), c->pairs.size() * sizeof(PairDesc));
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_pairs.p, c->h_pairs.p, c->pairs.size() * sizeof(PairDesc),
                                     hipMemcpyHostToDevice, c->stream));
    }
    if (c->kNN > 0) L3D_HIP_CHECK(c->d_slots.reserve(std::max<uint64_t>(c->n_slots, 1)));
    L3D_HIP_CHECK(c->d_cull.reserve(std::max<size_t>(c->cull.size(), 1)));
    if (!c->cull.empty()) {
        L3D_HIP_CHECK(c->h_cull.reserve(c->cull.size()));
        std::memcpy(c->h_cull.p, c->cull.data(), c->cull.size() * sizeof(PairCull));
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_cull.p, c->h_cull.p, c->cull.size() * sizeof(PairCull),
                                     hipMemcpyHostToDevice, c->stream));
    }
    L3D_HIP_CHECK(c->d_src_perm.reserve(std::max<uint64_t>(cs_off, 1)));
    L3D_HIP_CHECK(c->d_src_band.reserve(std::max<uint64_t>(cs_off, 1)));
    L3D_HIP_CHECK(c->d_tgt_perm.reserve(std::max<uint64_t>(ct_off, 1)));
    L3D_HIP_CHECK(c->d_tgt_sf.reserve(std::max<uint64_t>(ct_off, 1)));
    L3D_HIP_CHECK(c->d_tgt_band.reserve(std::max<uint64_t>(ct_off, 1)));
    L3D_HIP_CHECK(c->d_chunk_band.reserve(std::max<uint32_t>(cc_off, 1)));
    L3D_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    c->tm = l3d_timings{};
    c->state = l3d_ctx::BEGUN;
    return L3D_OK;
}

int l3d_num_pairs(l3d_ctx* c, uint32_t* n) {
    if (!c || !n) return fail(L3D_ERR_ARG, "null argument");
    *n = (uint32_t)c->pairs.size();
    return L3D_OK;
}

int l3d_get_pairs(l3d_ctx* c, uint32_t* s, uint32_t* t, uint64_t* off) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    for (size_t i = 0; i < c->pairs.size(); ++i) {
        if (s) s[i] = c->pair_src_cam[i];
        if (t) t[i] = c->pair_tgt_cam[i];
        if (off) off[i] = c->pairs[i].slot_off;
    }
    return L3D_OK;
}

static int ensure_aux(l3d_ctx* c) {
    if (!c->aux[0]) {   // highest priority: its kernels must not queue behind thousands of other workgroups
        int lo_p = 0, hi_p = 0;
        L3D_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
        L3D_HIP_CHECK(hipStreamCreateWithPriority(&c->aux[0], hipStreamNonBlocking, hi_p));
    }
    if (!c->aux[1]) L3D_HIP_CHECK(hipStreamCreateWithFlags(&c->aux[1], hipStreamNonBlocking));
    for (auto& e : c->sev) if (!e) L3D_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return L3D_OK;
}

// reads the phase-A events of the last run_match_kernel (the stream must have passed ev[5])
static void collect_match_timing(l3d_ctx* c) {
    if (!c->timing_pending) return;
    c->tm.match_kernel_ms += ev_ms(c->ev[4], c->ev[5]);
    c->tm.cull_prepare_ms += ev_ms(c->ev[8], c->ev[4]);
    c->tm.match_kernel_launches += c->pending_launches;
    c->timing_pending = false; c->pending_launches = 0;
}

// enqueues (no host synchronisation) the cull set-up and the pair kernel for pairs [first, first+count)
static int run_match_kernel(l3d_ctx* c, int mode, uint32_t first, uint32_t count) {
    size_t n_work = 0;
    uint32_t maxK = 0, maxM = 0, maxMt = 0;
    for (uint32_t p = first; p < first + count; ++p) n_work += (c->pairs[p].Ms + kMatchRows - 1) / kMatchRows;
    if (!n_work) return L3D_OK;
    L3D_HIP_CHECK(c->h_work.reserve(n_work));
    WorkItem* work = c->h_work.p;
    size_t w = 0;
    for (uint32_t p = first; p < first + count; ++p) {
        const PairDesc& pd = c->pairs[p];
        maxK = std::max(maxK, pd.K);
        if (c->cull[p].enabled) maxM = std::max(maxM, std::max(pd.Ms, pd.Mt));
        maxMt = std::max(maxMt, pd.Mt);
        for (uint32_t s0 = 0; s0 < pd.Ms; s0 += kMatchRows) work[w++] = WorkItem{p, s0};
    }
    if (match_lds_bytes(mode, maxK, false, match_waves_per_group(mode, c->brute, (uint32_t)n_work)) > 160 * 1024)
        return fail(L3D_ERR_LIMIT, "kNN too large for the LDS top-K table");
    L3D_HIP_CHECK(c->d_work.reserve(n_work));
    L3D_HIP_CHECK(hipMemcpyAsync(c->d_work.p, work, n_work * sizeof(WorkItem), hipMemcpyHostToDevice, c->stream));
    L3D_HIP_CHECK(hipEventRecord(c->ev[8], c->stream));
    CullPools pools{c->d_cull.p, c->d_src_perm.p, c->d_src_band.p, c->d_tgt_perm.p, c->d_tgt_sf.p, c->d_tgt_band.p,
                    c->d_chunk_band.p};
    if (mode != 0 || c->brute || !maxM) pools.cull = nullptr;
    else L3D_HIP_CHECK(launch_cull_prepare(c->d_views.p, c->d_pairs.p, first, count, maxM, pools, c->stream));
    L3D_HIP_CHECK(hipEventRecord(c->ev[4], c->stream));
    const bool ix16 = maxMt < 65536u && maxK < 65536u;   // 16-bit indices in the kernel's LDS tables
    // bounded kNN: the orientation filter and the hypothesis counters of phase B are fused into the epilogue
    const OrientFuse of{mode == 0 ? c->d_cnt_pack.p : nullptr, mode == 0 ? c->d_inv_pos.p : nullptr,
                        OrientThr{c->orient_lo, c->orient_hi}};
    L3D_HIP_CHECK(launch_match_pairs(mode, c->brute, c->d_views.p, c->d_pairs.p, c->d_work.p, (uint32_t)n_work, maxK,
                                     c->d_slots.p, c->d_row_counts.p, c->epipolar_overlap, pools, of, ix16, c->stream));
    L3D_HIP_CHECK(hipEventRecord(c->ev[5], c->stream));
    if (mode == 0)
        for (uint32_t p = first; p < first + count; ++p) c->pair_counted[p] = 1;
    if (pools.cull)
        for (uint32_t p = first; p < first + count; ++p) c->tm.culled_pairs += c->cull[p].enabled;
    c->timing_pending = true; c->pending_launches += 1;
    return L3D_OK;
}

static int match_pairs_impl(l3d_ctx* c, uint32_t first, uint32_t count, bool sync);

int l3d_match_pairs(l3d_ctx* c, uint32_t first, uint32_t count) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    return match_pairs_impl(c, first, count, true);
}

// sync = false (l3d_match_images): nothing waits for the GPU; the events are read at the end of l3d_match_finish
static int match_pairs_impl(l3d_ctx* c, uint32_t first, uint32_t count, bool sync) {
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, "l3d_match_begin must precede l3d_match_pairs");
    if ((uint64_t)first + count > c->pairs.size()) return fail(L3D_ERR_ARG, "pair range out of bounds");
    for (uint32_t p = first; p < first + count; ++p)
        if (c->pair_done[p]) return fail(L3D_ERR_STATE, "l3d_match_pairs: pair already matched since l3d_match_begin");
    (void)hipSetDevice(c->device);
    L3D_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    int rc = L3D_OK;
    if (c->kNN > 0) {
        rc = run_match_kernel(c, 0, first, count);
    } else {
        // kNN <= 0: keep every accepted match (line3D.cc:987-992): count, size the rows, fill
        if (first != 0 || count != c->pairs.size())
            return fail(L3D_ERR_LIMIT, "kNN <= 0 needs all pairs in one l3d_match_pairs call");
        L3D_HIP_CHECK(c->d_row_counts.reserve(std::max<uint32_t>(c->n_rows_total, 1)));
        rc = run_match_kernel(c, 1, first, count);
        if (rc) return rc;
        L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
        collect_match_timing(c);
        std::vector<uint32_t> counts(c->n_rows_total);
        L3D_HIP_CHECK(hipMemcpy(counts.data(), c->d_row_counts.p, counts.size() * 4, hipMemcpyDeviceToHost));
        uint64_t slot_off = 0;
        for (auto& pd : c->pairs) {
            uint32_t k = 1;
            for (uint32_t r = 0; r < pd.Ms; ++r) k = std::max(k, counts[pd.row_off + r]);
            pd.K = k; pd.slot_off = slot_off;
            slot_off += (uint64_t)pd.Ms * k;
        }
        c->n_slots = slot_off;
        if (c->n_slots >= (1ull << 32))
            return fail(L3D_ERR_LIMIT, "slot buffer exceeds the 32-bit slot indices of phase B");
        L3D_HIP_CHECK(c->d_slots.reserve(std::max<uint64_t>(c->n_slots, 1)));
        L3D_HIP_CHECK(hipMemcpy(c->d_pairs.p, c->pairs.data(), c->pairs.size() * sizeof(PairDesc),
                                hipMemcpyHostToDevice));
        rc = run_match_kernel(c, 2, first, count);
    }
    if (rc) return rc;
    L3D_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    if (sync) {
        L3D_HIP_CHECK(hipEventSynchronize(c->ev[3]));
        collect_match_timing(c);
        c->tm.match_pairs_ms += ev_ms(c->ev[2], c->ev[3]);
    }
    for (uint32_t p = first; p < first + count; ++p) c->pair_done[p] = 1;
    return L3D_OK;
}

int l3d_slot_buffer(l3d_ctx* c, void** dev_ptr, uint64_t* n_slots) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (dev_ptr) *dev_ptr = c->d_slots.p;
    if (n_slots) *n_slots = c->n_slots;
    return L3D_OK;
}

// ---- compact exchange (N > 1 ranks): 4 B per slot travel instead of 32 --------------------------------------------
int l3d_slot_index_buffer(l3d_ctx* c, void** dev_ptr, uint64_t* n_slots) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, "l3d_match_begin must precede l3d_slot_index_buffer");
    if (c->kNN <= 0) return fail(L3D_ERR_LIMIT, "the compact exchange needs kNN > 0 (fixed rows)");
    (void)hipSetDevice(c->device);
    L3D_HIP_CHECK(c->d_slot_idx.reserve(std::max<uint64_t>(c->n_slots, 1)));
    if (dev_ptr) *dev_ptr = c->d_slot_idx.p;
    if (n_slots) *n_slots = c->n_slots;
    return L3D_OK;
}

static int check_exchange_range(l3d_ctx* c, uint32_t first, uint32_t count, const char* who) {
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, std::string("l3d_match_begin must precede ") + who);
    if (c->kNN <= 0) return fail(L3D_ERR_LIMIT, "the compact exchange needs kNN > 0 (fixed rows)");
    if ((uint64_t)first + count > c->pairs.size()) return fail(L3D_ERR_ARG, "pair range out of bounds");
    return L3D_OK;
}

// target indices of the slots of pairs [first, first+count) -> index buffer; returns when they are there, so that
// a collective on another stream (RCCL runs on its own) may read them
int l3d_pack_slot_indices(l3d_ctx* c, uint32_t first, uint32_t count) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    int rc = check_exchange_range(c, first, count, "l3d_pack_slot_indices");
    if (rc) return rc;
    for (uint32_t p = first; p < first + count; ++p)
        if (!c->pair_done[p]) return fail(L3D_ERR_STATE, "l3d_pack_slot_indices: pair not matched on this rank");
    (void)hipSetDevice(c->device);
    L3D_HIP_CHECK(c->d_slot_idx.reserve(std::max<uint64_t>(c->n_slots, 1)));
    if (!count) return L3D_OK;
    const uint64_t lo = c->pairs[first].slot_off;
    const PairDesc& last = c->pairs[first + count - 1];
    const uint64_t hi = last.slot_off + (uint64_t)last.Ms * last.K;
    L3D_HIP_CHECK(launch_pack_slot_idx(c->d_slots.p, c->d_slot_idx.p, lo, hi, c->stream));
    L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
    return L3D_OK;
}

// index buffer -> full slot records for pairs [first, first+count) (received from their owning ranks); the pairs
// then count as matched.  Enqueued on the context's stream: l3d_match_finish follows in order.
int l3d_expand_slot_indices(l3d_ctx* c, uint32_t first, uint32_t count) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    int rc = check_exchange_range(c, first, count, "l3d_expand_slot_indices");
    if (rc) return rc;
    if (!c->d_slot_idx.p) return fail(L3D_ERR_STATE, "l3d_expand_slot_indices: no index buffer");
    if (!count) return L3D_OK;
    (void)hipSetDevice(c->device);
    uint32_t max_row_slots = 0;
    for (uint32_t p = first; p < first + count; ++p)
        max_row_slots = std::max<uint64_t>(max_row_slots, (uint64_t)c->pairs[p].Ms * c->pairs[p].K);
    for (uint32_t p = first; p < first + count; ++p)
        if (c->pair_done[p]) return fail(L3D_ERR_STATE, "l3d_expand_slot_indices: pair already present on this rank");
    const OrientFuse of{c->d_cnt_pack.p, c->d_inv_pos.p, OrientThr{c->orient_lo, c->orient_hi}};
    L3D_HIP_CHECK(launch_expand_slot_idx(c->d_views.p, c->d_pairs.p, first, count, max_row_slots, c->d_slot_idx.p,
                                         c->d_slots.p, of, c->stream));
    for (uint32_t p = first; p < first + count; ++p) c->pair_done[p] = c->pair_counted[p] = 1;
    return L3D_OK;
}

// phase B: line3D.cc:745-773 for every view in ascending camID order (k_views.hip)
static int match_finish_impl(l3d_ctx* c);

int l3d_match_finish(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, "l3d_match_begin must precede l3d_match_finish");
    (void)hipSetDevice(c->device);
    const int rc = match_finish_impl(c);
    if (rc != L3D_OK) {
        // leave a defined state behind: drain every stream this call may have used, restore the views
        // (matchImages translates them, line3D.cc:436/493) and require a new l3d_match_begin
        const std::string why = l3d_last_error();
        (void)hipStreamSynchronize(c->stream);
        for (auto& s2 : c->aux) if (s2) (void)hipStreamSynchronize(s2);
        untranslate(*c);
        c->timing_pending = false; c->pending_launches = 0;
        c->state = l3d_ctx::IDLE;
        set_error(why);
    }
    return rc;
}

static int match_finish_impl(l3d_ctx* c) {
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size(), P = (uint32_t)c->pairs.size();
    // global segment ids
    c->seg_base.assign(V + 1, 0);
    for (uint32_t vi = 0; vi < V; ++vi) c->seg_base[vi + 1] = c->seg_base[vi] + c->order[vi]->M;
    const uint32_t G = c->G = c->seg_base[V];
    uint64_t max_slots = 0;
    for (auto& pd : c->pairs) max_slots = std::max<uint64_t>(max_slots, (uint64_t)pd.Ms * pd.K);
    if (2 * c->n_slots >= (1ull << 32)) return fail(L3D_ERR_LIMIT, "more than 2^32 hypotheses");
    L3D_HIP_CHECK(c->d_seg_base.reserve(V + 1)); L3D_HIP_CHECK(c->d_gseg_view.reserve(G + 1));
    L3D_HIP_CHECK(c->d_cnt.reserve(G + 1)); L3D_HIP_CHECK(c->d_off.reserve(G + 1));
    L3D_HIP_CHECK(c->d_scan_tmp.reserve(G / 4096 + 1024)); L3D_HIP_CHECK(c->d_scal.reserve(16));
    L3D_HIP_CHECK(c->d_max_score.reserve(V + 1));
    L3D_HIP_CHECK(c->d_surv_cnt.reserve(G + 1)); L3D_HIP_CHECK(c->d_has_best.reserve(G + 1));
    L3D_HIP_CHECK(c->d_best_pos.reserve(G + 1)); L3D_HIP_CHECK(c->d_surv_off.reserve(G + 1));
    L3D_HIP_CHECK(c->d_hyp_off.reserve(G + 1)); L3D_HIP_CHECK(c->d_hyp_of_seg.reserve(G + 1));
    L3D_HIP_CHECK(c->d_medians.reserve(V + 1));
    L3D_HIP_CHECK(hipEventRecord(c->ev[6], st));
    L3D_HIP_CHECK(c->h_small.reserve(V + 1));
    g_trace.mark("finish: reserves done");
    std::memcpy(c->h_small.p, c->seg_base.data(), ((size_t)V + 1) * 4);
    L3D_HIP_CHECK(hipMemcpyAsync(c->d_seg_base.p, c->h_small.p, ((size_t)V + 1) * 4, hipMemcpyHostToDevice, st));
    {
        uint32_t max_M = 0;
        for (auto* v : c->order) max_M = std::max(max_M, v->M);
        L3D_HIP_CHECK(launch_fill_gseg_view(c->d_seg_base.p, V, max_M, c->d_gseg_view.p, st));
    }
    g_trace.mark("finish: seg_base + gseg enqueued");
    // outgoing pairs of every view (ascending target), for the fresh part of the lists
    // staged as [vout_off (V+1) | vout_pairs (P)] in one pinned buffer
    c->vout_off.assign(V + 1, 0);
    L3D_HIP_CHECK(c->h_vout.reserve((size_t)V + 1 + P + 1));
    {
        uint32_t n = 0;
        uint32_t* vp = c->h_vout.p + (V + 1);
        for (uint32_t vi = 0; vi < V; ++vi) {
            for (uint32_t p : c->order[vi]->out_pairs) vp[n++] = p;
            c->vout_off[vi + 1] = n;
        }
        std::memcpy(c->h_vout.p, c->vout_off.data(), ((size_t)V + 1) * 4);
        L3D_HIP_CHECK(c->d_vout_pairs.reserve((size_t)n + 1));
        L3D_HIP_CHECK(c->d_vout_off.reserve(V + 1));
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_vout_off.p, c->h_vout.p, ((size_t)V + 1) * 4, hipMemcpyHostToDevice, st));
        if (n) L3D_HIP_CHECK(hipMemcpyAsync(c->d_vout_pairs.p, vp, (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    L3D_HIP_CHECK(c->d_cnt_inv.reserve(G + 1)); L3D_HIP_CHECK(c->d_inv_off.reserve(G + 1));
    // (d_cnt_pack was zeroed by l3d_match_begin: the match epilogue already counts into it)
    L3D_HIP_CHECK(c->d_inv_pos.reserve(std::max<uint64_t>(c->n_slots, 1)));
    L3D_HIP_CHECK(hipMemsetAsync(c->d_max_score.p, 0, ((size_t)V + 1) * 4, st));
    L3D_HIP_CHECK(hipMemsetAsync(c->d_scal.p, 0, 16 * 4, st));
    g_trace.mark("finish: memsets enqueued");
    // ---- pre-pass: orientation flags, list offsets, transposed index of potential inverse matches ----
    // (bounded kNN: already done by the match epilogue / the exchange expansion; what is left are the pairs of the
    // keep-all mode and pairs whose full records arrived through l3d_slots_exchanged)
    for (uint32_t p0 = 0; p0 < P;) {
        if (c->pair_counted[p0]) { ++p0; continue; }
        uint32_t p1 = p0;
        while (p1 < P && !c->pair_counted[p1]) ++p1;
        L3D_HIP_CHECK(launch_orient_pairs(c->d_views.p, c->d_pairs.p + p0, p1 - p0, max_slots, c->d_seg_base.p,
                                          c->d_slots.p, c->d_cnt_pack.p, c->d_inv_pos.p, c->orient_lo, c->orient_hi, st));
        for (uint32_t p = p0; p < p1; ++p) c->pair_counted[p] = 1;
        p0 = p1;
    }
    L3D_HIP_CHECK(launch_unpack_counts(G, c->d_cnt_pack.p, c->d_cnt.p, c->d_cnt_inv.p, st));
    L3D_HIP_CHECK(launch_scan(c->d_cnt.p, G, c->d_off.p, c->d_scan_tmp.p, c->d_scal.p + 0, st));
    L3D_HIP_CHECK(launch_scan(c->d_cnt_inv.p, G, c->d_inv_off.p, c->d_scan_tmp.p, c->d_scal.p + 5, st));
    L3D_HIP_CHECK(c->d_bits_len.reserve(G + 1)); L3D_HIP_CHECK(c->d_boff.reserve(G + 1));
    L3D_HIP_CHECK(c->d_long_list.reserve(G + 1));
    L3D_HIP_CHECK(launch_bits_len(G, c->d_off.p, c->d_bits_len.p, c->d_long_list.p, c->d_scal.p + 7, st));
    L3D_HIP_CHECK(launch_scan(c->d_bits_len.p, G, c->d_boff.p, c->d_scan_tmp.p, c->d_scal.p + 6, st));
    uint32_t tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    L3D_HIP_CHECK(hipMemcpyAsync(tot, c->d_scal.p, 8 * 4, hipMemcpyDeviceToHost, st));
    // The list kernels only need buffers sized by bounds the host already knows (every slot yields at most one own
    // and one inverse hypothesis), so they are enqueued BEFORE the totals are read back: the host round trip that
    // sizes the support bitsets hides behind them.  (If the bound-sized buffers cannot be had, read first.)
    L3D_HIP_CHECK(c->d_positive.reserve(std::max<uint64_t>(c->n_slots, 1)));
    L3D_HIP_CHECK(hipMemsetAsync(c->d_positive.p, 0, std::max<uint64_t>(c->n_slots, 1), st));
    const uint64_t ents_bound = std::max<uint64_t>(2 * c->n_slots, 1), inv_bound = std::max<uint64_t>(c->n_slots, 1);
    bool lists_enqueued = false;
    auto enqueue_lists = [&]() -> int {
        L3D_HIP_CHECK(launch_inv_fill(c->d_pairs.p, P, max_slots, c->d_seg_base.p, c->d_slots.p, c->d_inv_off.p,
                                      c->d_inv_pos.p, c->d_refs.p, st));
        L3D_HIP_CHECK(launch_build_lists_all(G, c->d_views.p, c->d_pairs.p, c->d_seg_base.p, c->d_gseg_view.p,
                                             c->d_vout_off.p, c->d_vout_pairs.p, c->d_off.p, c->d_inv_off.p, c->d_refs.p,
                                             c->d_slots.p, c->d_dents.p, c->d_eref.p, c->kNN > 0 ? (uint32_t)c->kNN : 0u, st));
        return L3D_OK;
    };
    if (c->d_dents.reserve(ents_bound) == hipSuccess && c->d_eref.reserve(ents_bound) == hipSuccess &&
        c->d_refs.reserve(inv_bound) == hipSuccess) {
        const int rc = enqueue_lists();
        if (rc) return rc;
        lists_enqueued = true;
    } else {
        (void)hipGetLastError();
    }
    g_trace.mark("pre-pass + lists enqueued, waiting for sizes");
    L3D_HIP_CHECK(hipStreamSynchronize(st));   // first point at which the host waits for the GPU in matchImages
    g_trace.mark("sizes known");
    const uint32_t n_ents = c->n_ents = tot[0], n_inv = tot[5], n_words = tot[6];
    c->tm.list_entries = n_ents; c->tm.support_words = n_words;
    L3D_HIP_CHECK(c->d_bits.reserve(std::max<uint32_t>(n_words, 1)));
    if (!lists_enqueued) {
        L3D_HIP_CHECK(c->d_dents.reserve(std::max<uint32_t>(n_ents, 1)));
        L3D_HIP_CHECK(c->d_refs.reserve(std::max<uint32_t>(n_inv, 1)));
        L3D_HIP_CHECK(c->d_eref.reserve(std::max<uint32_t>(n_ents, 1)));
        const int rc = enqueue_lists();
        if (rc) return rc;
    }
    const SimConst simc = sim_thresholds(c->two_sigA_sqr);
    // lists too long for one wave's LDS staging: one workgroup each, before the pipeline (chain independent)
    L3D_HIP_CHECK(launch_support_long(tot[7], c->d_long_list.p, c->d_off.p, c->d_boff.p, c->d_dents.p, c->d_bits.p,
                                      c->d_views.p, c->d_gseg_view.p, simc, st));
    // Three-stage software pipeline over chunks of views on three streams:
    //   A (the context's stream)  support bitsets of chunk k          (chain independent, heavy)
    //   B                         THE CHAIN: one tiny bit-propagation launch per view, ascending camID; view v only
    //                             needs the support rows of its own chunk and the chain state of the views before it
    //   C                         scores of chunk k (needs the presence masks of chunk k)
    // so the launch-latency-bound chain (V dependent launches that keep < 5 % of the GPU busy) hides behind the
    // support / score kernels of the neighbouring chunks.
    {
        // 4 views per chunk (at most 64 chunks): measured best on C1 among 1/2/4/8/16 and ramped schedules -- the
        // chain cannot start before the first chunk's support rows exist and the last chunk's scores cannot start
        // before the chain ends, while very small chunks leave the support / score launches too small to fill the GPU
        const uint32_t n_chunks = std::min<uint32_t>(std::max<uint32_t>((V + 3) / 4, 1), 64);
        const uint32_t per = (V + n_chunks - 1) / n_chunks;
        while (c->pipe_ev.size() < 2 * (size_t)n_chunks + 2) {
            hipEvent_t e;
            L3D_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->pipe_ev.push_back(e);
        }
        { const int rc2 = ensure_aux(c); if (rc2) return rc2; }
        hipStream_t sB = c->aux[0], sC = c->aux[1];
        hipEvent_t ev_start = c->pipe_ev[2 * n_chunks], ev_done = c->pipe_ev[2 * n_chunks + 1];
        L3D_HIP_CHECK(hipEventRecord(ev_start, st));
        L3D_HIP_CHECK(hipStreamWaitEvent(sB, ev_start, 0));
        L3D_HIP_CHECK(hipStreamWaitEvent(sC, ev_start, 0));
        for (uint32_t k = 0; k < n_chunks; ++k) {
            const uint32_t v0 = std::min(V, k * per), v1 = std::min(V, v0 + per);
            const uint32_t g0 = c->seg_base[v0], g1 = c->seg_base[v1];
            L3D_HIP_CHECK(launch_support_all(g0, g1, c->d_off.p, c->d_boff.p, c->d_dents.p, c->d_bits.p, c->d_views.p,
                                             c->d_seg_base.p, c->d_gseg_view.p, simc, st));
            L3D_HIP_CHECK(hipEventRecord(c->pipe_ev[2 * k], st));
            L3D_HIP_CHECK(hipStreamWaitEvent(sB, c->pipe_ev[2 * k], 0));
            g_trace.mark("  chunk: support enqueued");
            for (uint32_t vi = v0; vi < v1; ++vi)
                L3D_HIP_CHECK(launch_presence_view(c->seg_base[vi], c->order[vi]->M, c->d_off.p, c->d_boff.p,
                                                   c->d_inv_off.p, c->d_eref.p, c->d_bits.p, c->d_positive.p, sB));
            g_trace.mark("  chunk: chain launches enqueued");
            L3D_HIP_CHECK(hipEventRecord(c->pipe_ev[2 * k + 1], sB));
            L3D_HIP_CHECK(hipStreamWaitEvent(sC, c->pipe_ev[2 * k + 1], 0));
            L3D_HIP_CHECK(launch_score_all(g0, g1, c->d_off.p, c->d_boff.p, c->d_gseg_view.p, c->d_dents.p, c->d_bits.p,
                                           c->d_slots.p, c->d_max_score.p, c->d_views.p, c->d_seg_base.p, simc, sC));
        }
        L3D_HIP_CHECK(hipEventRecord(ev_done, sC));
        L3D_HIP_CHECK(hipStreamWaitEvent(st, ev_done, 0));
    }
    // ---- post-pass: filterMatches for all views ----
    L3D_HIP_CHECK(launch_filter_all(G, c->d_off.p, c->d_gseg_view.p, c->d_dents.p, c->d_max_score.p, c->d_surv_cnt.p,
                                    c->d_has_best.p, c->d_best_pos.p, st));
    L3D_HIP_CHECK(launch_scan(c->d_surv_cnt.p, G, c->d_surv_off.p, c->d_scan_tmp.p, c->d_scal.p + 1, st));
    L3D_HIP_CHECK(launch_scan(c->d_has_best.p, G, c->d_hyp_off.p, c->d_scan_tmp.p, c->d_scal.p + 2, st));
    uint32_t nh[2] = {0, 0};
    L3D_HIP_CHECK(hipMemcpyAsync(nh, c->d_scal.p + 1, 8, hipMemcpyDeviceToHost, st));
    g_trace.mark("lists/support/chain/scores/filter enqueued, waiting for counts");
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    g_trace.mark("counts known");
    c->n_surv = nh[0]; c->n_hyps = nh[1];
    L3D_HIP_CHECK(c->d_surv.reserve(std::max<uint32_t>(c->n_surv, 1)));
    L3D_HIP_CHECK(c->d_surv_tg.reserve(std::max<uint32_t>(c->n_surv, 1)));
    L3D_HIP_CHECK(c->d_surv_sg.reserve(std::max<uint32_t>(c->n_surv, 1)));
    L3D_HIP_CHECK(c->d_hyps.reserve(std::max<uint32_t>(c->n_hyps, 1)));
    L3D_HIP_CHECK(c->d_depths.reserve(2 * (size_t)c->n_hyps + 2));
    L3D_HIP_CHECK(launch_filter_write_all(c->d_views.p, c->d_pairs.p, c->d_seg_base.p, G, c->d_gseg_view.p,
                                          c->d_off.p, c->d_dents.p, c->d_slots.p, c->d_surv_off.p, c->d_hyp_off.p,
                                          c->d_best_pos.p, c->d_surv.p, c->d_surv_tg.p, c->d_surv_sg.p,
                                          c->d_hyp_of_seg.p, c->d_hyps.p, c->d_depths.p, st));
    L3D_HIP_CHECK(launch_median_all(V, c->d_depths.p, c->d_hyp_off.p, c->d_seg_base.p, c->d_medians.p, st));
    // View::update_median_depth for every view (line3D.cc:1665-1668); in fixed-regulariser mode k is
    // re-set to the same sigma_p/med_scene_depth value, so k is unchanged either way
    std::vector<float> med(V);
    L3D_HIP_CHECK(hipMemcpyAsync(med.data(), c->d_medians.p, (size_t)V * 4, hipMemcpyDeviceToHost, st));
    L3D_HIP_CHECK(hipEventRecord(c->ev[7], st));
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    for (uint32_t vi = 0; vi < V; ++vi) c->order[vi]->median_depth = med[vi];
    c->host_offsets_valid = false;
    if (c->timing_pending) {   // phase A ran unsynchronised (l3d_match_images)
        collect_match_timing(c);
        c->tm.match_pairs_ms += ev_ms(c->ev[2], c->ev[3]);
    }
    c->tm.finish_ms = ev_ms(c->ev[6], c->ev[7]);
    c->tm.begin_ms = ev_ms(c->ev[0], c->ev[1]);
    untranslate(*c);   // line3D.cc:493
    c->state = l3d_ctx::MATCHED;
    return L3D_OK;
}

int l3d_match_images(l3d_ctx* c, const l3d_match_params* p) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    g_trace.t0 = std::chrono::steady_clock::now();
    g_trace.mark("matchImages enter");
    int rc = l3d_match_begin(c, p);
    g_trace.mark("begin enqueued");
    if (rc) return rc;
    // nothing between begin and the first sizing read-back of phase B waits for the GPU
    rc = match_pairs_impl(c, 0, (uint32_t)c->pairs.size(), false);
    g_trace.mark("phase A enqueued");
    if (rc) return rc;
    rc = l3d_match_finish(c);
    g_trace.mark("matchImages done");
    g_trace.flush();
    return rc;
}

// computingAffinityMatrix with collinearity_t_ > 0 (line3D.cc:1852-1979 incl. the links to collinear segments,
// :1904-1974).  The GPU does the arithmetic -- the per-view all-pairs collinearity tests (View::findCollinCPU)
// and the similarity of every potential link -- and hands the host three candidate streams (primary =
// surviving matches, children = collinear segments of a passing primary's target, own = collinear segments of
// the hypothesis' segment).  The bookkeeping that is sequential BY DEFINITION in the reference (used_ claims a
// pair for whoever comes first, a child is only visited when its parent was accepted, row ids in first-touch
// order) is one linear pass over those streams in the reference's single-thread order.  d_simv is ready.
static int affinity_collinear(l3d_ctx* c) {
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size(), G = c->G, N = c->n_surv, H = c->n_hyps;
    uint32_t max_M = 0;
    for (auto* v : c->order) max_M = std::max(max_M, v->M);
    // ---- per-view collinear lists (CSR over global segments) ----
    L3D_HIP_CHECK(c->d_coll_cnt.reserve(G + 1)); L3D_HIP_CHECK(c->d_coll_off.reserve(G + 1));
    L3D_HIP_CHECK(launch_collin(0, c->d_views.p, V, max_M, c->d_seg_base.p, c->collinearity_t, c->d_coll_cnt.p, nullptr,
                                nullptr, st));
    L3D_HIP_CHECK(launch_scan(c->d_coll_cnt.p, G, c->d_coll_off.p, c->d_scan_tmp.p, c->d_scal.p + 9, st));
    uint32_t n_coll = 0;
    L3D_HIP_CHECK(hipMemcpyAsync(&n_coll, c->d_scal.p + 9, 4, hipMemcpyDeviceToHost, st));
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    L3D_HIP_CHECK(c->d_coll_idx.reserve(std::max<uint32_t>(n_coll, 1)));
    L3D_HIP_CHECK(launch_collin(1, c->d_views.p, V, max_M, c->d_seg_base.p, c->collinearity_t, nullptr, c->d_coll_off.p,
                                c->d_coll_idx.p, st));
    // ---- similarities of the child and own candidates ----
    std::vector<uint32_t> off[2], seg[2];
    std::vector<float> sim[2];
    const uint32_t n_items[2] = {N, H};
    for (int mode = 0; mode < 2; ++mode) {
        const uint32_t n = n_items[mode];
        L3D_HIP_CHECK(c->d_item_cnt.reserve(n + 1)); L3D_HIP_CHECK(c->d_item_off.reserve(n + 1));
        L3D_HIP_CHECK(c->d_scan_tmp.reserve((size_t)n / 4096 + 1024));
        L3D_HIP_CHECK(launch_aff_coll_count(mode, n, c->d_surv_tg.p, c->d_simv.p, c->d_hyps.p, c->d_seg_base.p,
                                            c->d_coll_off.p, c->d_item_cnt.p, st));
        L3D_HIP_CHECK(launch_scan(c->d_item_cnt.p, n, c->d_item_off.p, c->d_scan_tmp.p, c->d_scal.p + 10, st));
        uint32_t total = 0;
        L3D_HIP_CHECK(hipMemcpyAsync(&total, c->d_scal.p + 10, 4, hipMemcpyDeviceToHost, st));
        L3D_HIP_CHECK(hipStreamSynchronize(st));
        L3D_HIP_CHECK(c->d_item_seg.reserve(std::max<uint32_t>(total, 1)));
        L3D_HIP_CHECK(c->d_item_sim.reserve(std::max<uint32_t>(total, 1)));
        L3D_HIP_CHECK(launch_aff_coll_sim(mode, n, c->d_surv_sg.p, c->d_surv_tg.p, c->d_hyp_of_seg.p, c->d_hyps.p,
                                          c->d_views.p, c->d_seg_base.p, c->d_gseg_view.p, c->d_coll_off.p,
                                          c->d_coll_idx.p, c->d_item_off.p, c->d_vaff.p, c->d_medians.p, c->d_msdl.p,
                                          c->two_sigA_sqr, c->d_item_seg.p, c->d_item_sim.p, st));
        off[mode].resize((size_t)n + 1); seg[mode].resize(total); sim[mode].resize(total);
        L3D_HIP_CHECK(hipMemcpyAsync(off[mode].data(), c->d_item_off.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
        if (total) {
            L3D_HIP_CHECK(hipMemcpyAsync(seg[mode].data(), c->d_item_seg.p, (size_t)total * 4, hipMemcpyDeviceToHost, st));
            L3D_HIP_CHECK(hipMemcpyAsync(sim[mode].data(), c->d_item_sim.p, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        }
        L3D_HIP_CHECK(hipStreamSynchronize(st));
    }
    // ---- primary stream + hypothesis -> segment map ----
    std::vector<uint32_t> surv_off((size_t)G + 1), surv_tg(N);
    std::vector<float> simv(N);
    std::vector<int32_t> hyp_of_seg(G);
    L3D_HIP_CHECK(hipMemcpyAsync(surv_off.data(), c->d_surv_off.p, ((size_t)G + 1) * 4, hipMemcpyDeviceToHost, st));
    L3D_HIP_CHECK(hipMemcpyAsync(surv_tg.data(), c->d_surv_tg.p, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    L3D_HIP_CHECK(hipMemcpyAsync(simv.data(), c->d_simv.p, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    L3D_HIP_CHECK(hipMemcpyAsync(hyp_of_seg.data(), c->d_hyp_of_seg.p, (size_t)G * 4, hipMemcpyDeviceToHost, st));
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<uint32_t> seg_of_hyp(H, kEmpty);
    for (uint32_t g = 0; g < G; ++g) if (hyp_of_seg[g] >= 0) seg_of_hyp[(uint32_t)hyp_of_seg[g]] = g;
    // ---- the sequential pass: unused() (line3D.cc:1982-2002), getLocalID() (:2005-2023) ----
    std::unordered_set<uint64_t> used;
    std::vector<int32_t> local_id(G, -1);
    std::vector<uint32_t> row_seg;
    auto unused = [&](uint32_t a, uint32_t b) {
        const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
        return used.insert(key).second;
    };
    auto get_id = [&](uint32_t g) {
        if (local_id[g] < 0) { local_id[g] = (int32_t)row_seg.size(); row_seg.push_back(g); }
        return local_id[g];
    };
    c->edges.clear();
    auto push = [&](int32_t i, int32_t j, float w) {
        c->edges.push_back(l3d_cledge{i, j, w}); c->edges.push_back(l3d_cledge{j, i, w});
    };
    for (uint32_t h = 0; h < H; ++h) {
        const uint32_t a = seg_of_hyp[h];
        if (a == kEmpty) continue;
        int32_t id1 = -1;
        bool found_aff = false;
        for (uint32_t p = surv_off[a]; p < surv_off[a + 1]; ++p) {
            const uint32_t b = surv_tg[p];
            if (simv[p] > kMinAffinity && unused(a, b)) {
                if (id1 < 0) id1 = get_id(a);
                const int32_t id2 = get_id(b);
                push(id1, id2, simv[p]);
                found_aff = true;
                for (uint32_t k = off[0][p]; k < off[0][p + 1]; ++k)
                    if (sim[0][k] > kMinAffinity && unused(a, seg[0][k])) push(id1, get_id(seg[0][k]), sim[0][k]);
            }
        }
        if (found_aff && id1 >= 0)
            for (uint32_t k = off[1][h]; k < off[1][h + 1]; ++k)
                if (sim[1][k] > kMinAffinity && unused(a, seg[1][k])) push(id1, get_id(seg[1][k]), sim[1][k]);
    }
    c->l2g.resize(row_seg.size());
    for (size_t r = 0; r < row_seg.size(); ++r) {
        const uint32_t g = row_seg[r];
        const uint32_t vi = (uint32_t)(std::upper_bound(c->seg_base.begin(), c->seg_base.end(), g) - c->seg_base.begin()) - 1;
        c->l2g[r].camID_ = c->order[vi]->cam; c->l2g[r].segID_ = g - c->seg_base[vi];
    }
    c->aff_n_edges = (uint32_t)c->edges.size(); c->aff_n_rows = (uint32_t)c->l2g.size(); c->aff_host_valid = true;
    // A_ stays device resident as well (matrix diffusion reads it there)
    L3D_HIP_CHECK(c->d_edges.reserve(std::max<size_t>(c->edges.size(), 1)));
    if (!c->edges.empty())
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_edges.p, c->edges.data(), c->edges.size() * sizeof(l3d_cledge),
                                     hipMemcpyHostToDevice, st));
    return L3D_OK;
}

// med_scene_depth_lines_ + computingAffinityMatrix (line3D.cc:1759-1778) in the CURRENT (translated) frame
static int affinity_core(l3d_ctx* c) {
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size();
    c->edges.clear(); c->l2g.clear();
    c->aff_n_edges = 0; c->aff_n_rows = 0; c->aff_host_valid = true;
    bool counts_pending = false;
    // med_scene_depth_lines_, line3D.cc:1759-1774
    std::vector<float> sd;
    for (auto* v : c->order) if (v->median_depth > kEps) sd.push_back(v->median_depth);
    if (!sd.empty()) { std::sort(sd.begin(), sd.end()); c->med_scene_depth_lines = sd[sd.size() / 2]; }
    else c->med_scene_depth_lines = 0.0f;
    std::vector<ViewAff> va(V);
    for (uint32_t vi = 0; vi < V; ++vi) { va[vi].k = c->order[vi]->k; va[vi].pad = 0; }
    const uint32_t N = c->n_surv, H = c->n_hyps;
    L3D_HIP_CHECK(hipEventRecord(c->ev[6], st));
    if (N > 0 && H > 0) {
        L3D_HIP_CHECK(c->d_vaff.reserve(V)); L3D_HIP_CHECK(c->d_msdl.reserve(1));
        L3D_HIP_CHECK(c->d_simv.reserve(N)); L3D_HIP_CHECK(c->d_ca.reserve(N)); L3D_HIP_CHECK(c->d_cb.reserve(N));
        L3D_HIP_CHECK(c->d_flag.reserve(N + 1)); L3D_HIP_CHECK(c->d_epos.reserve(N + 1));
        L3D_HIP_CHECK(c->d_first_touch.reserve(H));
        L3D_HIP_CHECK(c->d_scan_tmp.reserve(std::max<size_t>(N, 2 * (size_t)N) / 4096 + 1024));
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_vaff.p, va.data(), V * sizeof(ViewAff), hipMemcpyHostToDevice, st));
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_msdl.p, &c->med_scene_depth_lines, 4, hipMemcpyHostToDevice, st));
        L3D_HIP_CHECK(launch_aff_sim(N, c->d_surv_sg.p, c->d_surv_tg.p, c->d_hyp_of_seg.p, c->d_hyps.p, c->d_vaff.p,
                                     c->d_medians.p, c->d_msdl.p, c->two_sigA_sqr, c->d_simv.p, c->d_ca.p, c->d_cb.p,
                                     st));
        if (c->collinearity_t > (float)kEps) {
            const int rc = affinity_collinear(c);
            if (rc) return rc;
            L3D_HIP_CHECK(hipEventRecord(c->ev[7], st));
            L3D_HIP_CHECK(hipStreamSynchronize(st));
            c->tm.affinity_ms = ev_ms(c->ev[6], c->ev[7]);
            c->affinity_done = true;
            return L3D_OK;
        }
        L3D_HIP_CHECK(launch_aff_flag(N, c->d_surv_off.p, c->d_surv_sg.p, c->d_surv_tg.p, c->d_simv.p, c->d_ca.p,
                                      c->d_cb.p, c->d_flag.p, st));
        L3D_HIP_CHECK(launch_scan(c->d_flag.p, N, c->d_epos.p, c->d_scan_tmp.p, c->d_scal.p + 3, st));
        // no read-back of the edge count: everything downstream is sized by its upper bound N (flags beyond the
        // 2E touched positions stay zero), the two counts are read once at the end
        {
            L3D_HIP_CHECK(c->d_touch_flag.reserve(2 * (size_t)N + 1));
            L3D_HIP_CHECK(c->d_touch_rank.reserve(2 * (size_t)N + 1));
            L3D_HIP_CHECK(c->d_edges.reserve(2 * (size_t)N));
            L3D_HIP_CHECK(c->d_l2g.reserve(H));
            L3D_HIP_CHECK(launch_fill_u32(c->d_first_touch.p, H, kEmpty, st));
            L3D_HIP_CHECK(hipMemsetAsync(c->d_touch_flag.p, 0, (2 * (size_t)N + 1) * 4, st));
            L3D_HIP_CHECK(launch_aff_touch(N, c->d_flag.p, c->d_epos.p, c->d_ca.p, c->d_cb.p, c->d_first_touch.p, st));
            L3D_HIP_CHECK(launch_aff_mark(H, c->d_first_touch.p, c->d_touch_flag.p, st));
            L3D_HIP_CHECK(launch_scan(c->d_touch_flag.p, 2 * N, c->d_touch_rank.p, c->d_scan_tmp.p, c->d_scal.p + 4, st));
            L3D_HIP_CHECK(launch_aff_emit(N, c->d_flag.p, c->d_epos.p, c->d_ca.p, c->d_cb.p, c->d_simv.p,
                                          c->d_first_touch.p, c->d_touch_rank.p, c->d_hyps.p, c->d_edges.p,
                                          c->d_l2g.p, st));
            L3D_HIP_CHECK(c->h_cnt.reserve(4));
            L3D_HIP_CHECK(hipMemcpyAsync(c->h_cnt.p, c->d_scal.p + 3, 8, hipMemcpyDeviceToHost, st));
            counts_pending = true;
        }
    }
    L3D_HIP_CHECK(hipEventRecord(c->ev[7], st));
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    if (counts_pending) { c->aff_n_edges = 2 * c->h_cnt.p[0]; c->aff_n_rows = c->h_cnt.p[1]; c->aff_host_valid = false; }
    c->tm.affinity_ms = ev_ms(c->ev[6], c->ev[7]);
    c->affinity_done = true;
    return L3D_OK;
}

// host copies of A_ / local2global_ (fetched on first use)
static int ensure_affinity_host(l3d_ctx* c) {
    if (c->aff_host_valid) return L3D_OK;
    (void)hipSetDevice(c->device);
    c->edges.resize(c->aff_n_edges);
    c->l2g.resize(c->aff_n_rows);
    L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->aff_n_edges)
        L3D_HIP_CHECK(hipMemcpy(c->edges.data(), c->d_edges.p, (size_t)c->aff_n_edges * sizeof(l3d_cledge), hipMemcpyDeviceToHost));
    if (c->aff_n_rows)
        L3D_HIP_CHECK(hipMemcpy(c->l2g.data(), c->d_l2g.p, (size_t)c->aff_n_rows * sizeof(l3d_segment2d), hipMemcpyDeviceToHost));
    c->aff_host_valid = true;
    return L3D_OK;
}

int l3d_compute_affinity(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::MATCHED) return fail(L3D_ERR_STATE, "matchImages must precede the affinity step");
    // translate()/untranslate() (line3D.cc:1749,1820) only move camera centres, which the affinity terms never
    // read; they are applied to keep the host state identical to the reference's.
    translate(*c);
    const int rc = affinity_core(c);
    untranslate(*c);
    return rc;
}

// Line3D::reconstruct3Dlines, line3D.cc:1702-1824
int l3d_reconstruct_3d_lines(l3d_ctx* c, uint32_t visibility_t, int perform_diffusion, float collinearity_t,
                             int use_CERES, uint32_t max_iter_CERES) {
    (void)max_iter_CERES;
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::MATCHED || c->n_hyps == 0)
        return fail(L3D_ERR_STATE, "no clusterable segments! forgot to match lines?");   // line3D.cc:1712-1718
    c->collinearity_t = collinearity_t;                                                          // :1725-1726
    if (use_CERES) set_error("CERES not available, no optimization will be performed");             // :1741-1743
    const unsigned vis = std::max<unsigned>(visibility_t, 3);
    c->visibility_t = vis; c->perform_rdd = perform_diffusion != 0;
    c->lines3D.clear();
    translate(*c);
    int rc = affinity_core(c);
    // matrix diffusion (performRDD, line3D.cc:1787-1791) on the device-resident A_
    if (rc == L3D_OK && perform_diffusion && c->aff_n_edges) {
        const uint32_t nnz = c->aff_n_edges, n_rows = c->aff_n_rows;
        rc = ensure_affinity_host(c);       // l2g; the edge list is replaced below
        if (rc) { untranslate(*c); return rc; }
        const size_t wb = rdd_workspace_bytes(nnz, n_rows);
        DevBuf<char> ws; DevBuf<l3d_cledge> out;
        hipError_t e = ws.reserve(wb);
        if (e == hipSuccess) e = out.reserve(nnz);
        if (e == hipSuccess) e = launch_rdd(c->d_edges.p, nnz, n_rows, 10 /* L3D_DEF_RDD_MAX_ITER */, out.p, ws.p, wb, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->d_edges.p, out.p, (size_t)nnz * sizeof(l3d_cledge), hipMemcpyDeviceToDevice, c->stream);
        c->edges.resize(nnz);
        if (e == hipSuccess) e = hipMemcpyAsync(c->edges.data(), out.p, (size_t)nnz * sizeof(l3d_cledge), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        ws.release(); out.release();
        if (e != hipSuccess) { untranslate(*c); return fail(L3D_ERR_HIP, std::string("matrix diffusion: ") + hipGetErrorString(e)); }
    }
    if (rc == L3D_OK) rc = ensure_affinity_host(c);
    if (rc == L3D_OK) {
        ReconInput in;
        in.visibility_t = vis;
        in.edges = c->edges;
        in.l2g = c->l2g;
        in.hyps.resize(c->n_hyps);
        if (hipMemcpy(in.hyps.data(), c->d_hyps.p, in.hyps.size() * sizeof(HypRec), hipMemcpyDeviceToHost) != hipSuccess) {
            untranslate(*c);
            return fail(L3D_ERR_HIP, "copying the 3D hypotheses failed");
        }
        for (size_t i = 0; i < in.hyps.size(); ++i) in.entry_map[{in.hyps[i].m.src_cam, in.hyps[i].m.src_seg}] = i;
        for (auto* v : c->order) in.views[v->cam] = v;
        uint32_t ncl = 0, nvalid = 0;
        reconstruct_lines(in, c->lines3D, &ncl, &nvalid);
        // untranslate the lines (performTranslation(translation_), line3D.cc:559-574)
        const d3 t = c->translation;
        auto shift = [&](ReconSeg3D& s) { s.P1 = s.P1 + t; s.P2 = s.P2 + t; };
        for (auto& L : c->lines3D) { for (auto& s : L.collinear) shift(s); shift(L.cluster_seg); }
        c->lines_done = true;
    }
    untranslate(*c);
    return rc;
}

// Line3D::createOutputFilename, line3D.cc:2853-2893 (stream formatting of the float parameters as there)
static std::string output_filename(l3d_ctx* c, int max_image_width) {
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

int l3d_synchronize(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
    return L3D_OK;
}

int l3d_pair_tests(l3d_ctx* c, uint64_t* n) {
    if (!c || !n) return fail(L3D_ERR_ARG, "null argument");
    *n = c->pair_tests;
    return L3D_OK;
}

static int fetch_host_offsets(l3d_ctx* c) {
    if (c->host_offsets_valid) return L3D_OK;
    c->h_surv_off.assign((size_t)c->G + 1, 0);
    c->h_hyp_off.assign((size_t)c->G + 1, 0);
    L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
    L3D_HIP_CHECK(hipMemcpy(c->h_surv_off.data(), c->d_surv_off.p, ((size_t)c->G + 1) * 4, hipMemcpyDeviceToHost));
    L3D_HIP_CHECK(hipMemcpy(c->h_hyp_off.data(), c->d_hyp_off.p, ((size_t)c->G + 1) * 4, hipMemcpyDeviceToHost));
    c->host_offsets_valid = true;
    return L3D_OK;
}

int l3d_get_matches(l3d_ctx* c, uint32_t camID, l3d_match* out, uint64_t cap, uint32_t* seg_offsets, uint64_t* n) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (c->state != l3d_ctx::MATCHED) return fail(L3D_ERR_STATE, "no matches yet");
    auto f = c->views.find(camID);
    if (f == c->views.end()) return fail(L3D_ERR_ARG, "unknown camera ID");
    HostView& v = *f->second;
    (void)hipSetDevice(c->device);
    int rc = fetch_host_offsets(c);
    if (rc) return rc;
    const uint32_t g0 = c->seg_base[v.index], g1 = c->seg_base[v.index + 1];
    const uint32_t base = c->h_surv_off[g0], cnt = c->h_surv_off[g1] - base;
    if (n) *n = cnt;
    if (seg_offsets) for (uint32_t s = 0; s <= v.M; ++s) seg_offsets[s] = c->h_surv_off[g0 + s] - base;
    if (out) {
        const uint64_t m = std::min<uint64_t>(cap, cnt);
        if (m) L3D_HIP_CHECK(hipMemcpy(out, c->d_surv.p + base, m * sizeof(l3d_match), hipMemcpyDeviceToHost));
    }
    return L3D_OK;
}

int l3d_get_pair_slots(l3d_ctx* c, uint32_t pi, l3d_slot* out, uint64_t cap, uint32_t* Ms, uint32_t* K) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (pi >= c->pairs.size() || !c->pair_done[pi]) return fail(L3D_ERR_STATE, "pair not matched");
    const PairDesc& pd = c->pairs[pi];
    if (Ms) *Ms = pd.Ms;
    if (K) *K = pd.K;
    (void)hipSetDevice(c->device);
    if (out) {
        const uint64_t m = std::min<uint64_t>(cap, (uint64_t)pd.Ms * pd.K);
        L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (m) L3D_HIP_CHECK(hipMemcpy(out, c->d_slots.p + pd.slot_off, m * sizeof(l3d_slot), hipMemcpyDeviceToHost));
    }
    return L3D_OK;
}

int l3d_num_best(l3d_ctx* c, uint32_t* n) {
    if (!c || !n) return fail(L3D_ERR_ARG, "null argument");
    *n = c->state == l3d_ctx::MATCHED ? c->n_hyps : 0;
    return L3D_OK;
}

int l3d_get_best(l3d_ctx* c, l3d_segment2d* seg2d, l3d_segment3d* seg3d, l3d_match* best) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (c->state != l3d_ctx::MATCHED) return fail(L3D_ERR_STATE, "no matches yet");
    (void)hipSetDevice(c->device);
    std::vector<HypRec> h(c->n_hyps);
    if (c->n_hyps) L3D_HIP_CHECK(hipMemcpy(h.data(), c->d_hyps.p, h.size() * sizeof(HypRec), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) {
        if (seg2d) { seg2d[i].camID_ = h[i].m.src_cam; seg2d[i].segID_ = h[i].m.src_seg; }
        if (seg3d) {
            std::memcpy(seg3d[i].P1, h[i].P1, 24); std::memcpy(seg3d[i].P2, h[i].P2, 24);
            std::memcpy(seg3d[i].dir, h[i].dir, 24);
            seg3d[i].length_ = h[i].length; seg3d[i].valid_ = h[i].valid;
        }
        if (best) std::memcpy(&best[i], &h[i].m, sizeof(l3d_match));
    }
    return L3D_OK;
}

int l3d_view_info(l3d_ctx* c, uint32_t camID, float* k, float* median_depth) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    auto f = c->views.find(camID);
    if (f == c->views.end()) return fail(L3D_ERR_ARG, "unknown camera ID");
    if (k) *k = f->second->k;
    if (median_depth) *median_depth = f->second->median_depth;
    return L3D_OK;
}

int l3d_translation(l3d_ctx* c, double t[3]) {
    if (!c || !t) return fail(L3D_ERR_ARG, "null argument");
    t[0] = c->translation.x; t[1] = c->translation.y; t[2] = c->translation.z;
    return L3D_OK;
}

int l3d_num_affinity(l3d_ctx* c, uint32_t* n_edges, uint32_t* n_rows) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (!c->affinity_done) return fail(L3D_ERR_STATE, "l3d_compute_affinity has not run");
    if (n_edges) *n_edges = c->aff_n_edges;
    if (n_rows) *n_rows = c->aff_n_rows;
    return L3D_OK;
}

int l3d_get_affinity(l3d_ctx* c, l3d_cledge* edges, l3d_segment2d* l2g, float* msdl) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (!c->affinity_done) return fail(L3D_ERR_STATE, "l3d_compute_affinity has not run");
    { std::lock_guard<std::recursive_mutex> lk(c->mu); const int rc = ensure_affinity_host(c); if (rc) return rc; }
    if (edges && !c->edges.empty()) std::memcpy(edges, c->edges.data(), c->edges.size() * sizeof(l3d_cledge));
    if (l2g && !c->l2g.empty()) std::memcpy(l2g, c->l2g.data(), c->l2g.size() * sizeof(l3d_segment2d));
    if (msdl) *msdl = c->med_scene_depth_lines;
    return L3D_OK;
}

// SparseMatrix::SparseMatrix(entries, n, 1.0f, sort_by_row), sparsematrix.cc:8-60.  std::list::sort is
// stable, so equal keys keep A_'s order.
int l3d_get_sparse_matrix(l3d_ctx* c, int sort_by_row, l3d_float4* entries, int32_t* start_indices) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    if (!c->affinity_done) return fail(L3D_ERR_STATE, "l3d_compute_affinity has not run");
    { std::lock_guard<std::recursive_mutex> lk(c->mu); const int rc = ensure_affinity_host(c); if (rc) return rc; }
    std::vector<l3d_cledge> e = c->edges;
    if (sort_by_row)
        std::stable_sort(e.begin(), e.end(), [](const l3d_cledge& a, const l3d_cledge& b) {
            return a.i_ < b.i_ || (a.i_ == b.i_ && a.j_ < b.j_); });   // sortCLEdgesByRow, clustering.h
    else
        std::stable_sort(e.begin(), e.end(), [](const l3d_cledge& a, const l3d_cledge& b) {
            return a.j_ < b.j_ || (a.j_ == b.j_ && a.i_ < b.i_); });   // sortCLEdgesByCol
    if (start_indices) for (size_t i = 0; i < c->l2g.size(); ++i) start_indices[i] = -1;
    int cur = -1;
    for (size_t pos = 0; pos < e.size(); ++pos) {
        if (entries) entries[pos] = l3d_float4{(float)e[pos].i_, (float)e[pos].j_, e[pos].w_, 0.0f};
        const int rc = sort_by_row ? e[pos].i_ : e[pos].j_;
        if (rc != cur) { if (start_indices) start_indices[rc] = (int)pos; cur = rc; }
    }
    return L3D_OK;
}

int l3d_get_timings(l3d_ctx* c, l3d_timings* t) {
    if (!c || !t) return fail(L3D_ERR_ARG, "null argument");
    *t = c->tm;
    return L3D_OK;
}

// seam layer: match_lines_GPU replacement (cudawrapper.h:54-63) with CPU-path semantics
int l3d_diffuse_affinity(int device, const l3d_cledge* edges, uint32_t n_edges, uint32_t n_rows, uint32_t iterations,
                         l3d_cledge* out) {
    if ((!edges || !out) && n_edges) return fail(L3D_ERR_ARG, "null argument");
    if (!n_edges || !n_rows) return L3D_OK;
    for (uint32_t k = 0; k < n_edges; ++k)
        if (edges[k].i_ < 0 || edges[k].j_ < 0 || (uint32_t)edges[k].i_ >= n_rows || (uint32_t)edges[k].j_ >= n_rows)
            return fail(L3D_ERR_ARG, "edge index outside [0, n_rows)");
    if (hipSetDevice(device) != hipSuccess) return fail(L3D_ERR_HIP, "hipSetDevice failed: no usable HIP device");
    DevBuf<l3d_cledge> din, dout; DevBuf<char> ws;
    const size_t wb = rdd_workspace_bytes(n_edges, n_rows);
    hipError_t e = din.reserve(n_edges);
    if (e == hipSuccess) e = dout.reserve(n_edges);
    if (e == hipSuccess) e = ws.reserve(wb);
    if (e == hipSuccess) e = hipMemcpy(din.p, edges, (size_t)n_edges * sizeof(l3d_cledge), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_rdd(din.p, n_edges, n_rows, iterations, dout.p, ws.p, wb, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, (size_t)n_edges * sizeof(l3d_cledge), hipMemcpyDeviceToHost);
    din.release(); dout.release(); ws.release();
    if (e != hipSuccess) return fail(L3D_ERR_HIP, std::string("l3d_diffuse_affinity: ") + hipGetErrorString(e));
    return L3D_OK;
}

// Replaces the body of View::findCollinGPU (view.cc:173-209) / find_collinear_segments_GPU (cudawrapper.h:66-68) with
// the semantics of View::findCollinCPU (view.cc:213-258): for every segment the ascending list of the segments of
// the same image that are collinear to it (no overlap along the line, all four point-to-line distances < dist_t).
// CSR output: offsets[M+1]; idx receives the lists if cap >= *n (call once with idx = nullptr to size it).
int l3d_find_collinear_segments(int device, const float* lines4, uint32_t M, float dist_t, uint32_t* offsets,
                                uint32_t* idx, uint64_t cap, uint64_t* n) {
    if ((!lines4 && M) || !offsets || !n) return fail(L3D_ERR_ARG, "null argument");
    *n = 0;
    for (uint32_t i = 0; i <= M; ++i) offsets[i] = 0;
    if (!M || !(dist_t > (float)kEps)) return L3D_OK;                       // view.cc:158: nothing to do
    if (hipSetDevice(device) != hipSuccess) return fail(L3D_ERR_HIP, "hipSetDevice failed: no usable HIP device");
    DevBuf<float4> seg4; DevBuf<ViewDev> dv; DevBuf<uint32_t> base, cnt, off, tmp, tot, lists;
    auto cleanup = [&]() { seg4.release(); dv.release(); base.release(); cnt.release(); off.release(); tmp.release();
                           tot.release(); lists.release(); };
    const int rc = [&]() -> int {
        L3D_HIP_CHECK(seg4.reserve(M)); L3D_HIP_CHECK(dv.reserve(1)); L3D_HIP_CHECK(base.reserve(2));
        L3D_HIP_CHECK(cnt.reserve(M + 1)); L3D_HIP_CHECK(off.reserve(M + 1)); L3D_HIP_CHECK(tmp.reserve(M / 4096 + 1024));
        L3D_HIP_CHECK(tot.reserve(1));
        L3D_HIP_CHECK(hipMemcpy(seg4.p, lines4, (size_t)M * 16, hipMemcpyHostToDevice));
        ViewDev hv{};
        hv.seg4 = seg4.p; hv.M = M;
        const uint32_t hb[2] = {0, M};
        L3D_HIP_CHECK(hipMemcpy(dv.p, &hv, sizeof(hv), hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemcpy(base.p, hb, sizeof(hb), hipMemcpyHostToDevice));
        L3D_HIP_CHECK(launch_collin(0, dv.p, 1, M, base.p, dist_t, cnt.p, nullptr, nullptr, 0));
        L3D_HIP_CHECK(launch_scan(cnt.p, M, off.p, tmp.p, tot.p, 0));
        uint32_t total = 0;
        L3D_HIP_CHECK(hipMemcpy(&total, tot.p, 4, hipMemcpyDeviceToHost));
        L3D_HIP_CHECK(hipMemcpy(offsets, off.p, ((size_t)M + 1) * 4, hipMemcpyDeviceToHost));
        *n = total;
        if (idx && cap >= total && total) {
            L3D_HIP_CHECK(lists.reserve(total));
            L3D_HIP_CHECK(launch_collin(1, dv.p, 1, M, base.p, dist_t, nullptr, off.p, lists.p, 0));
            L3D_HIP_CHECK(hipDeviceSynchronize());
            L3D_HIP_CHECK(hipMemcpy(idx, lists.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        }
        return L3D_OK;
    }();
    cleanup();
    return rc;
}

// Replaces score_matches_GPU (cudawrapper.h:70-73; caller Line3D::scoringGPU, line3D.cc:1297-1414) with the
// semantics of Line3D::scoringCPU (line3D.cc:1208-1294): score3D of every match of ONE view.  Inputs as scoringGPU
// marshals them: lines4[M]; matches4[n] = (src segment, target camera, depth_p1, depth_p2) grouped per segment
// and, inside a segment, by target camera (sortMatches); ranges2[M] = (first, last) inclusive or (-1, -1);
// reg_tgt2[n] = the two View::regularizerFrom3Dpoint values; RtKinv / C of the view in double (translated frame).
int l3d_score_matches(int device, const float* lines4, uint32_t M, const float* matches4, const int32_t* ranges2,
                      const float* reg_tgt2, uint32_t n, const double RtKinv[9], const double C[3], float two_sigA_sqr,
                      float k, float* scores) {
    if ((!lines4 && M) || ((!matches4 || !reg_tgt2 || !scores) && n) || (!ranges2 && M) || !RtKinv || !C)
        return fail(L3D_ERR_ARG, "null argument");
    if (!M || !n) return L3D_OK;
    std::vector<uint32_t> off((size_t)M + 1);
    {
        uint32_t next = 0;
        for (uint32_t i = 0; i < M; ++i) {
            const int32_t a = ranges2[2 * i], b = ranges2[2 * i + 1];
            if (a >= 0) {
                if ((uint32_t)a != next || b < a || (uint32_t)b >= n) return fail(L3D_ERR_ARG, "ranges are not a partition of the matches");
                off[i] = (uint32_t)a; next = (uint32_t)b + 1;
            } else off[i] = next;
        }
        off[M] = next;
        if (next != n) return fail(L3D_ERR_ARG, "ranges do not cover the matches");
    }
    if (hipSetDevice(device) != hipSuccess) return fail(L3D_ERR_HIP, "hipSetDevice failed: no usable HIP device");
    DevBuf<float4> seg4, m4; DevBuf<float2> rt; DevBuf<SegF> segf; DevBuf<SegX> segx; DevBuf<ViewDev> dv;
    DevBuf<uint32_t> d_off, d_boff, d_len, d_tmp, d_scal, d_gv, d_long, d_max; DevBuf<DEntry> dents; DevBuf<uint64_t> bits;
    DevBuf<float> d_scores;
    auto cleanup = [&]() { seg4.release(); m4.release(); rt.release(); segf.release(); segx.release(); dv.release();
                           d_off.release(); d_boff.release(); d_len.release(); d_tmp.release(); d_scal.release();
                           d_gv.release(); d_long.release(); d_max.release(); dents.release(); bits.release();
                           d_scores.release(); };
    const int rc = [&]() -> int {
        L3D_HIP_CHECK(seg4.reserve(M)); L3D_HIP_CHECK(segf.reserve(M)); L3D_HIP_CHECK(segx.reserve(M));
        L3D_HIP_CHECK(m4.reserve(n)); L3D_HIP_CHECK(rt.reserve(n)); L3D_HIP_CHECK(dv.reserve(1));
        L3D_HIP_CHECK(d_off.reserve(M + 1)); L3D_HIP_CHECK(d_boff.reserve(M + 1)); L3D_HIP_CHECK(d_len.reserve(M + 1));
        L3D_HIP_CHECK(d_tmp.reserve(M / 4096 + 1024)); L3D_HIP_CHECK(d_scal.reserve(4)); L3D_HIP_CHECK(d_gv.reserve(M + 1));
        L3D_HIP_CHECK(d_long.reserve(M + 1)); L3D_HIP_CHECK(d_max.reserve(2)); L3D_HIP_CHECK(dents.reserve(n));
        L3D_HIP_CHECK(d_scores.reserve(n));
        L3D_HIP_CHECK(hipMemcpy(seg4.p, lines4, (size_t)M * 16, hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemcpy(m4.p, matches4, (size_t)n * 16, hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemcpy(rt.p, reg_tgt2, (size_t)n * 8, hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemcpy(d_off.p, off.data(), ((size_t)M + 1) * 4, hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemset(d_gv.p, 0, ((size_t)M + 1) * 4));
        L3D_HIP_CHECK(hipMemset(d_scal.p, 0, 16)); L3D_HIP_CHECK(hipMemset(d_max.p, 0, 8));
        ViewDev hv{};
        std::memcpy(hv.C, C, 24); std::memcpy(hv.RtKinv, RtKinv, 72);
        hv.seg4 = seg4.p; hv.segf = segf.p; hv.segx = segx.p; hv.M = M; hv.k = k;
        L3D_HIP_CHECK(hipMemcpy(dv.p, &hv, sizeof(hv), hipMemcpyHostToDevice));
        L3D_HIP_CHECK(launch_prep_views(dv.p, 1, M, 0));
        L3D_HIP_CHECK(launch_seam_entries(n, m4.p, rt.p, dv.p, k, dents.p, 0));
        L3D_HIP_CHECK(launch_bits_len(M, d_off.p, d_len.p, d_long.p, d_scal.p + 1, 0));
        L3D_HIP_CHECK(launch_scan(d_len.p, M, d_boff.p, d_tmp.p, d_scal.p + 0, 0));
        uint32_t tot[2] = {0, 0};
        L3D_HIP_CHECK(hipMemcpy(tot, d_scal.p, 8, hipMemcpyDeviceToHost));
        L3D_HIP_CHECK(bits.reserve(std::max<uint32_t>(tot[0], 1)));
        const SimConst simc = sim_thresholds(two_sigA_sqr);
        L3D_HIP_CHECK(launch_support_long(tot[1], d_long.p, d_off.p, d_boff.p, dents.p, bits.p, dv.p, d_gv.p, simc, 0));
        L3D_HIP_CHECK(launch_support_all(0, M, d_off.p, d_boff.p, dents.p, bits.p, dv.p, nullptr, d_gv.p, simc, 0));
        L3D_HIP_CHECK(launch_seam_all_present(M, d_off.p, d_boff.p, bits.p, 0));
        L3D_HIP_CHECK(launch_score_all(0, M, d_off.p, d_boff.p, d_gv.p, dents.p, bits.p, nullptr, d_max.p, dv.p, nullptr,
                                       simc, 0));
        L3D_HIP_CHECK(launch_seam_scores_out(n, dents.p, d_scores.p, 0));
        L3D_HIP_CHECK(hipDeviceSynchronize());
        L3D_HIP_CHECK(hipMemcpy(scores, d_scores.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        return L3D_OK;
    }();
    cleanup();
    return rc;
}

int l3d_match_lines(int device, const float* lines_src4, uint32_t Ms, const float* lines_tgt4, uint32_t Mt,
                    const double F[9], const double RtKinv_src[9], const double RtKinv_tgt[9], const double C_src[3],
                    const double C_tgt[3], uint32_t width, uint32_t height, float epi_overlap, int32_t kNN,
                    l3d_slot* out_slots, uint64_t* num_matches) {
    if (!lines_src4 || !lines_tgt4 || !F || !RtKinv_src || !RtKinv_tgt || !C_src || !C_tgt || !out_slots)
        return fail(L3D_ERR_ARG, "null argument");
    if (kNN <= 0) return fail(L3D_ERR_ARG, "l3d_match_lines needs kNN > 0");
    if (Ms == 0 || Mt == 0) { if (num_matches) *num_matches = 0; return L3D_OK; }
    if (hipSetDevice(device) != hipSuccess) return fail(L3D_ERR_HIP, "hipSetDevice failed: no usable HIP device");
    const uint32_t M[2] = {Ms, Mt};
    const float* lines[2] = {lines_src4, lines_tgt4};
    const double* A[2] = {RtKinv_src, RtKinv_tgt};
    const double* Cc[2] = {C_src, C_tgt};
    DevBuf<float4> seg4[2]; DevBuf<SegF> segf[2];
    DevBuf<SegX> segx;   // one array for both views (source first), like the context's global array: the match kernel
                         // derives global segment ids from it for the phase-B counters it feeds
    DevBuf<unsigned long long> cnt_pack; DevBuf<uint32_t> inv_pos;
    DevBuf<double> consts; DevBuf<ViewDev> dv; DevBuf<PairDesc> dp; DevBuf<WorkItem> dw; DevBuf<Slot> ds;
    DevBuf<PairCull> dc; DevBuf<uint32_t> sperm, tperm; DevBuf<float2> sband, tband, cband; DevBuf<float4> tsf;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) { seg4[i].release(); segf[i].release(); }
        segx.release(); cnt_pack.release(); inv_pos.release();
        consts.release(); dv.release(); dp.release(); dw.release(); ds.release();
        dc.release(); sperm.release(); tperm.release(); sband.release(); tband.release(); cband.release(); tsf.release();
    };
    int rc = [&]() -> int {
        ViewDev hv[2];
        double hc[24];
        L3D_HIP_CHECK(consts.reserve(24));
        L3D_HIP_CHECK(segx.reserve((size_t)Ms + Mt));
        for (int i = 0; i < 2; ++i) {
            L3D_HIP_CHECK(seg4[i].reserve(M[i])); L3D_HIP_CHECK(segf[i].reserve(M[i]));
            L3D_HIP_CHECK(hipMemcpy(seg4[i].p, lines[i], (size_t)M[i] * 16, hipMemcpyHostToDevice));
            std::memcpy(hc + 12 * i, A[i], 72); std::memcpy(hc + 12 * i + 9, Cc[i], 24);
            std::memcpy(hv[i].C, Cc[i], 24); std::memcpy(hv[i].RtKinv, A[i], 72);
            hv[i].seg4 = seg4[i].p; hv[i].segf = segf[i].p; hv[i].segx = segx.p + (i ? Ms : 0u);
            hv[i].M = M[i]; hv[i].cam = (uint32_t)i; hv[i].k = 0;
            hv[i].cx = 0.5f * (float)width; hv[i].cy = 0.5f * (float)height; hv[i].pad = 0;
        }
        PairDesc pd;
        std::memcpy(pd.F, F, 72);
        pd.src = 0; pd.tgt = 1; pd.Ms = Ms; pd.Mt = Mt; pd.K = (uint32_t)kNN; pd.row_off = 0; pd.slot_off = 0;
        std::vector<WorkItem> work;
        for (uint32_t s0 = 0; s0 < Ms; s0 += kMatchRows) work.push_back(WorkItem{0, s0});
        if (match_lds_bytes(0, pd.K, false, match_waves_per_group(0, false, (uint32_t)work.size())) > 160 * 1024)
            return fail(L3D_ERR_LIMIT, "kNN too large for the LDS top-K table");
        L3D_HIP_CHECK(dv.reserve(2)); L3D_HIP_CHECK(dp.reserve(1)); L3D_HIP_CHECK(dw.reserve(work.size()));
        L3D_HIP_CHECK(ds.reserve((size_t)Ms * pd.K));
        L3D_HIP_CHECK(hipMemcpy(dv.p, hv, sizeof(hv), hipMemcpyHostToDevice));
        L3D_HIP_CHECK(launch_prep_views(dv.p, 2, std::max(Ms, Mt), 0));
        L3D_HIP_CHECK(hipMemcpy(dp.p, &pd, sizeof(pd), hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemcpy(dw.p, work.data(), work.size() * sizeof(WorkItem), hipMemcpyHostToDevice));
        const float thr = std::fmin(std::fabs(epi_overlap), 0.99f);
        // epipolar-band culling when F is a proper fundamental matrix and the epipoles are well outside the images
        PairCull pc{};
        if (Ms <= kCullMaxSegs && Mt <= kCullMaxSegs && std::getenv("L3D_NO_CULL") == nullptr)
            make_cull(pd.F, width, height, width, height, pc);
        CullPools pools{};
        if (pc.enabled) {
            L3D_HIP_CHECK(dc.reserve(1)); L3D_HIP_CHECK(sperm.reserve(Ms)); L3D_HIP_CHECK(sband.reserve(Ms));
            L3D_HIP_CHECK(tperm.reserve(Mt)); L3D_HIP_CHECK(tsf.reserve(Mt)); L3D_HIP_CHECK(tband.reserve(Mt));
            L3D_HIP_CHECK(cband.reserve((Mt + 63) / 64));
            L3D_HIP_CHECK(hipMemcpy(dc.p, &pc, sizeof(pc), hipMemcpyHostToDevice));
            pools = CullPools{dc.p, sperm.p, sband.p, tperm.p, tsf.p, tband.p, cband.p};
            L3D_HIP_CHECK(launch_cull_prepare(dv.p, dp.p, 0, 1, std::max(Ms, Mt), pools, 0));
        }
        // the kernel also applies the orientation filter (slot flags) and feeds the phase-B counters: scratch here
        L3D_HIP_CHECK(cnt_pack.reserve((size_t)Ms + Mt + 1)); L3D_HIP_CHECK(inv_pos.reserve((size_t)Ms * pd.K));
        L3D_HIP_CHECK(hipMemset(cnt_pack.p, 0, ((size_t)Ms + Mt + 1) * 8));
        OrientFuse of{cnt_pack.p, inv_pos.p, OrientThr{-1.0, 1.0}};
        orientation_thresholds(of.thr.lo, of.thr.hi);
        L3D_HIP_CHECK(launch_match_pairs(0, false, dv.p, dp.p, dw.p, (uint32_t)work.size(), pd.K, ds.p, nullptr, thr,
                                         pools, of, Mt < 65536u && pd.K < 65536u, 0));
        L3D_HIP_CHECK(hipDeviceSynchronize());
        L3D_HIP_CHECK(hipMemcpy(out_slots, ds.p, (size_t)Ms * pd.K * sizeof(Slot), hipMemcpyDeviceToHost));
        return L3D_OK;
    }();
    cleanup();
    if (rc == L3D_OK && num_matches) {
        uint64_t n = 0;
        for (uint64_t i = 0; i < (uint64_t)Ms * (uint32_t)kNN; ++i) n += out_slots[i].tgt_seg != kEmpty;
        *num_matches = n;
    }
    return rc;
}

}  // extern "C"
