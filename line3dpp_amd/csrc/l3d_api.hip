// l3d_api.hip -- context layer of libl3dpp_hip.so (include/l3dpp_hip.h): the host-side driver that mirrors
// Line3D::addImage / matchImages (line3D.cc:112-227, 375-497, 702-778).  No CPU fallback exists: every compute step
// is a HIP kernel launch; without a usable device the calls fail with L3D_ERR_HIP.
#include <thread>

#include "l3d_ctx.h"

namespace l3d {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
const char* last_error_cstr() { return g_err.c_str(); }

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

// Line3D::translate, line3D.cc:500-536: the median of the camera centres' coordinates
d3 scene_translation(const std::vector<HostView*>& order) {
    double tr[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        std::vector<double> coords;
        for (auto* v : order) {
            const double val = i == 0 ? v->C.x : (i == 1 ? v->C.y : v->C.z);
            if (std::fabs(val) > kEps) coords.push_back(val);
        }
        if (!coords.empty()) {
            std::sort(coords.begin(), coords.end());
            tr[i] = coords[coords.size() / 2];
        }
    }
    return d3{tr[0], tr[1], tr[2]};
}
void translate(l3d_ctx& c) {
    c.translation = scene_translation(c.order);
    for (auto* v : c.order) translate_view(*v, d3{-c.translation.x, -c.translation.y, -c.translation.z});
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

// PairDesc::B / tolB (l3d_dev.h: depths_positive32): the baseline in float and the magnitude above which the float value
// of n.B has the sign of the exact numerator of the depth test -- ten times the float error of the dot product
// (kDepthTol32 |B|) plus what the double rounding of cn = C.n can add (1e-14 (|C_s| + |C_t|)); infinite -- the float
// decision is never trusted -- for a baseline below 1e-3, beyond 1e30 or not finite.  L3D_NO_DEPTH32=1: always infinite.
void pair_baseline(const d3& Cs, const d3& Ct, PairDesc& pd) {
    static const bool off = std::getenv("L3D_NO_DEPTH32") != nullptr;
    const d3 B = Ct - Cs;
    const double nb = norm(B), nc = std::fabs(Cs.x) + std::fabs(Cs.y) + std::fabs(Cs.z) + std::fabs(Ct.x) + std::fabs(Ct.y) + std::fabs(Ct.z);
    pd.B[0] = (float)B.x; pd.B[1] = (float)B.y; pd.B[2] = (float)B.z;
    const double tol = (double)kDepthTol32 * nb * (1.0 + 1e-6) + 1e-14 * nc;
    pd.tolB = (!off && std::isfinite(nb) && std::isfinite(nc) && nb >= 1e-3 && nb <= 1e30 && nc <= 1e30)
                  ? std::nextafterf((float)tol, INFINITY) : INFINITY;
}

int upload_views(l3d_ctx& c) {
    const size_t V = c.order.size();
    L3D_HIP_CHECK(c.d_views.reserve(V));
    std::vector<ViewDev> table(V);
    ViewDev* hv = table.data();
    uint32_t max_M = 0;
    // the per-segment invariants of all views live in ONE array indexed by the global segment id, so that the
    // phase-B kernels reach them with one load (gsegx[g]) instead of g -> view -> pointer -> record
    size_t Gtot = 0;
    for (size_t i = 0; i < V; ++i) Gtot += c.order[i]->M;
    L3D_HIP_CHECK(c.d_gsegx.reserve(std::max<size_t>(Gtot, 1)));
    L3D_HIP_CHECK(c.d_gsegd32.reserve(std::max<size_t>(Gtot, 1)));
    std::memset((void*)hv, 0, V * sizeof(ViewDev));   // (padding bytes take part in the comparison of upload_table)
    size_t gbase = 0;
    for (size_t i = 0; i < V; ++i) {
        HostView& v = *c.order[i];
        ViewDev& d = hv[i];
        d.C[0] = v.C.x; d.C[1] = v.C.y; d.C[2] = v.C.z;
        std::memcpy(d.RtKinv, v.RtKinv.m, 72);
        d.seg4 = v.d_seg4.p; d.segf = v.d_segf.p; d.segx = c.d_gsegx.p + gbase; d.segd32 = c.d_gsegd32.p + gbase;
        gbase += v.M;
        d.M = v.M; d.cam = v.cam; d.k = v.k;
        d.cx = 0.5f * (float)v.width; d.cy = 0.5f * (float)v.height; d.pad = 0;
        max_M = std::max(max_M, v.M);
    }
    L3D_HIP_CHECK(upload_table(c.d_views, c.h_views, hv, V * sizeof(ViewDev), c.up_views, c.stream));
    // per-segment invariants (k_prep_views)
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

HostTrace g_trace;   // (l3d_ctx.h)

// ---- process-wide cache of released blocks (l3d_host.h) ----
namespace {
struct CachedBlock { void* p; size_t bytes; int device; bool pinned; };
std::mutex g_cache_mu;
std::vector<CachedBlock> g_cache;
size_t g_cache_bytes[2] = {0, 0};
constexpr size_t kCacheMaxBlocks = 512;
constexpr size_t kCacheMaxPinned = (size_t)1 << 30;
thread_local int g_release_synced = 0;   // > 0: inside a ReleaseSynced scope
// device blocks: a quarter of the device's memory at most (the rest stays with the runtime: torch / RCCL live in the
// same process), whatever the device: L3D_CACHE_MAX_MB overrides
size_t cache_max_device_bytes() {
    static const size_t cap = [] {
        if (const char* e = std::getenv("L3D_CACHE_MAX_MB")) return (size_t)std::strtoull(e, nullptr, 10) << 20;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || !tot) { (void)hipGetLastError(); return (size_t)16 << 30; }
        return tot / 4;
    }();
    return cap;
}
}  // namespace
ReleaseSynced::ReleaseSynced() { ++g_release_synced; }
ReleaseSynced::~ReleaseSynced() { --g_release_synced; }
void* block_cache_take(bool pinned, size_t bytes, size_t* got_bytes) {
    if (!bytes) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t best = g_cache.size();
    for (size_t i = 0; i < g_cache.size(); ++i) {   // smallest block that fits, at most twice (small: 64 KiB more than) the request
        const CachedBlock& b = g_cache[i];
        if (b.pinned != pinned || (!pinned && b.device != dev) || b.bytes < bytes || b.bytes > 2 * bytes + (64u << 10)) continue;
        if (best == g_cache.size() || b.bytes < g_cache[best].bytes) best = i;
    }
    if (best == g_cache.size()) return nullptr;
    const CachedBlock b = g_cache[best];
    g_cache[best] = g_cache.back(); g_cache.pop_back();
    g_cache_bytes[pinned] -= b.bytes;
    *got_bytes = b.bytes;
    return b.p;
}
bool block_cache_give(bool pinned, void* p, size_t bytes) {
    static const bool off = std::getenv("L3D_NO_BLOCK_CACHE") != nullptr;   // diagnostic switch
    if (off || !p || !bytes) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t cap = pinned ? kCacheMaxPinned : cache_max_device_bytes();
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (g_cache.size() >= kCacheMaxBlocks || g_cache_bytes[pinned] + bytes > cap) return false;
    }
    // Kernels or copies of the releasing context may still be in flight on the block (a buffer that grows between two
    // launches): the next owner -- another context, another stream, the host writing a pinned block -- must not meet
    // them.  Same wait as the hipFree / hipHostFree this replaces.
    if (!g_release_synced && hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (g_cache.size() >= kCacheMaxBlocks || g_cache_bytes[pinned] + bytes > cap) return false;
    g_cache.push_back(CachedBlock{p, bytes, dev, pinned});
    g_cache_bytes[pinned] += bytes;
    return true;
}
size_t block_cache_trim(int kind) {
    std::vector<CachedBlock> out;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size();) {
            if (kind < 0 || (int)g_cache[i].pinned == kind) {
                out.push_back(g_cache[i]); g_cache_bytes[g_cache[i].pinned] -= g_cache[i].bytes;
                g_cache[i] = g_cache.back(); g_cache.pop_back();
            } else ++i;
        }
    }
    size_t freed = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const CachedBlock& b : out) {
        if (b.pinned) (void)hipHostFree(b.p);
        else { (void)hipSetDevice(b.device); (void)hipFree(b.p); }
        freed += b.bytes;
    }
    if (!out.empty()) (void)hipSetDevice(cur);
    return freed;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}
std::atomic<uint64_t> g_keep_all_repeats{0};    // test hook: count passes of the keep-all mode repeated with a larger row scratch
std::atomic<uint64_t> g_knn_replay_calls{0};   // test hook (l3d_debug_counter): calls whose kNN exceeded the LDS tables of k_match_pairs

float ev_ms(const ::l3d_ctx* c, int a, int b) { return c->ev_on(a) && c->ev_on(b) ? ev_ms(c->ev[a], c->ev[b]) : 0.0f; }

}  // namespace l3d

using namespace l3d;

extern "C" {

const char* l3d_last_error(void) { return g_err.c_str(); }
#ifndef L3D_BUILD_ID
#define L3D_BUILD_ID "unknown"
#endif
const char* l3d_build_info(void) { return "libl3dpp_hip gfx950 hip fp-contract=off build=" L3D_BUILD_ID; }

uint64_t l3d_trim_cache(void) { return (uint64_t)block_cache_trim(-1); }

l3d_ctx* l3d_create(int device, void* stream) {
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed: no usable HIP device"); return nullptr; }
    auto* c = new l3d_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    orientation_thresholds(c->orient_lo, c->orient_hi);
    c->use_cull = std::getenv("L3D_NO_CULL") == nullptr;   // diagnostic switch: stream every pair unculled
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); delete c; return nullptr; }
    // What the runtime sets up lazily is set up HERE, beside its own start-up (~0.3 s), not inside the first
    // matchImages of the process: the code object of every translation unit with kernels (loaded at the first launch
    // of one of its kernels, ~0.6 ms each) and the copy paths of both directions (the first device-to-host
    // hipMemcpyAsync of a process took 7.3 ms -- of an 11 ms first call on a scene whose later calls take 0.3 ms).
    // L3D_NO_WARMUP=1 leaves it to the first call (diagnostic).  Once per process and device: later contexts find the
    // code objects loaded and the copy paths set up.
    static std::mutex warm_mu;
    static std::set<int> warmed;
    bool need_warm = !std::getenv("L3D_NO_WARMUP");
    // (`warmed` holds a device only once its warm-up has SUCCEEDED -- ADVICE round 4: after a transient failure every later
    // context would have skipped it; two contexts created concurrently may both warm up, which is harmless)
    if (need_warm) { std::lock_guard<std::mutex> lk(warm_mu); need_warm = warmed.count(device) == 0; }
    if (need_warm) {
        hipStream_t st = c->stream;
        DevBuf<uint32_t> d; PinnedBuf<uint32_t> h;
        // (the runtime takes a different copy path per size class and sets each one up at its first use: measured on the
        // first matchImages of a process, profiles/r03_first_call_of_a_process.txt -- 64 KiB + 64 B covered C1's 9 KiB view
        // table but not C0's 3.7 KiB one (7.8 ms inside l3d_match_begin) nor C3's 147 KiB; one copy of 64 B, 4 KiB,
        // 64 KiB and 1 MiB in each direction moves all of it here)
        constexpr size_t kWarmWords = (1u << 20) / 4;
        bool ok = d.reserve(kWarmWords) == hipSuccess && h.reserve(kWarmWords) == hipSuccess;
        ok = ok && hipMemsetAsync(d.p, 0, kWarmWords * 4, st) == hipSuccess;
        // (round 4, profiles/r04_first_call.txt: the runtime sets its copy paths up lazily PER SIZE CLASS, DIRECTION AND
        // STATE OF THE STREAM -- 8 ms at the first host-to-device copy of 16 KiB or more (tools/copy_probe.py); with
        // start-up copies that all met an idle stream the first matchImages was fast and the SECOND paid 7.4 ms in a
        // device-to-host read-back queued behind its kernels; with start-up copies that all met a busy stream the first
        // upload of the first l3d_match_begin -- idle stream -- paid 8 ms again.  So both: every size class in both
        // directions once on an idle stream and once right behind a fill that keeps it busy.)
        const size_t sizes[] = {64, 1024, 4096, 16384, 65536, 262144, (size_t)1 << 20};
        for (size_t bytes : sizes) {
            ok = ok && hipMemcpyAsync(d.p, h.p, bytes, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            ok = ok && hipMemcpyAsync(h.p, d.p, bytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        }
        for (int rep = 0; rep < 2; ++rep)
            for (size_t bytes : sizes) {
                ok = ok && hipMemsetAsync(d.p, 0, kWarmWords * 4, st) == hipSuccess;
                ok = ok && hipMemcpyAsync(h.p, d.p, bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
                ok = ok && hipMemsetAsync(d.p, 0, kWarmWords * 4, st) == hipSuccess;
                ok = ok && hipMemcpyAsync(d.p, h.p, bytes, hipMemcpyHostToDevice, st) == hipSuccess;
            }
        ok = ok && warm_match(st) == hipSuccess && warm_lists(st) == hipSuccess && warm_scan(st) == hipSuccess &&
             warm_views(st) == hipSuccess && warm_affinity(st) == hipSuccess && warm_rdd(st) == hipSuccess;
        // (waited for UNCONDITIONALLY: on a failure the short-circuit above would have skipped it, and the blocks below are
        // handed to the cache as "drained")
        const bool drained_ok = hipStreamSynchronize(st) == hipSuccess;
        ok = ok && drained_ok;
        { const ReleaseSynced drained; d.release(); h.release(); }
        if (!ok) { set_error("start-up launches failed: no usable HIP device"); l3d_destroy(c); return nullptr; }
        { std::lock_guard<std::mutex> lk(warm_mu); warmed.insert(device); }
    }
    return c;
}

void l3d_destroy(l3d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    const ReleaseSynced drained;   // everything this context enqueued has been waited for: its blocks may change hands
    for (auto& kv : c->views) {
        HostView& v = *kv.second;
        v.d_seg4.release(); v.d_segf.release();
    }
    c->d_views.release(); c->d_pairs.release(); c->d_work.release(); c->d_slots.release(); c->d_slot_idx.release();
    c->h_views.release(); c->h_pairs.release(); c->h_cull.release(); c->h_work.release();
    c->h_segb.release(); c->h_cnt.release(); c->h_fin.release();
    c->d_poff.release(); c->d_csr_dummy.release(); c->d_pair_present.release(); c->d_cnt64.release(); c->d_off64s.release(); c->d_scan_ws.release();
    c->d_huge_u64.release(); c->d_inv_refs.release(); c->d_lzero.release(); c->d_list2.release(); c->d_list4.release(); c->d_listH.release();
    c->d_seg_of_g.release(); c->d_huge_u32.release(); c->d_huge_f32.release(); c->d_ledges.release(); c->d_lhyps.release();
    c->d_lsegs.release(); c->d_lcands.release(); c->d_lchdrs.release(); c->d_ltab.release(); c->h_ltab.release();
    c->d_cull.release(); c->d_src_perm.release(); c->d_tgt_perm.release(); c->d_src_band.release();
    c->d_chunk_band.release(); c->d_cull_keys.release(); c->d_tgt_sf.release(); c->d_tgt_band.release(); c->d_tgt_s4.release(); c->d_tgt_sd.release();
    c->d_row_counts.release(); c->d_keep_rec.release(); c->d_row_start.release(); c->d_slot_row.release(); c->d_keep_info.release(); c->d_row_pair.release(); c->d_blk_row.release(); c->h_keep_info.release();
    c->d_seg_base.release(); c->d_gseg_view.release();
    c->d_scal.release();
    c->d_surv_off.release(); c->d_hyp_off.release();
    c->d_surv_tg.release(); c->d_surv_sg.release();
    c->d_inv_tgt.release(); c->d_hyp_p.release(); c->d_hyp_q.release(); c->d_gsegx.release(); c->d_gsegd32.release();
    c->d_tie_count.release(); c->d_tie_list.release(); c->d_tie_heap.release(); c->d_item_bucket.release(); c->d_item_order.release(); c->d_order_done.release();
    c->d_coll_cnt.release(); c->d_coll_off.release(); c->d_coll_idx.release(); c->d_item_cnt.release();
    c->d_item_off.release(); c->d_item_seg.release(); c->d_item_sim.release();
    c->d_surv.release();
    c->d_hyp_of_seg.release(); c->d_depths.release(); c->d_med = nullptr; c->d_hyps.release();
    c->d_vaff.release(); c->d_simv.release(); c->h_vaff.release(); c->d_ca.release(); c->d_cb.release();
    c->d_flag.release(); c->d_epos.release(); c->d_first_touch.release(); c->d_touch_flag.release();
    c->d_touch_rank.release(); c->d_edges.release(); c->d_l2g.release();
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
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

static int add_view(l3d_ctx* c, uint32_t camID, const float* segs4, uint32_t M, const double K[9], const double R[9],
                    const double t[3], uint32_t width, uint32_t height, float median_depth, const uint32_t* neighbors,
                    uint32_t n_neighbors, bool by_worldpoints);

int l3d_add_view(l3d_ctx* c, uint32_t camID, const float* segs4, uint32_t M, const double K[9], const double R[9],
                 const double t[3], uint32_t width, uint32_t height, float median_depth, const uint32_t* neighbors,
                 uint32_t n_neighbors) {
    return add_view(c, camID, segs4, M, K, R, t, width, height, median_depth, neighbors, n_neighbors, false);
}
int l3d_add_view_worldpoints(l3d_ctx* c, uint32_t camID, const float* segs4, uint32_t M, const double K[9],
                             const double R[9], const double t[3], uint32_t width, uint32_t height, float median_depth,
                             const uint32_t* worldpoints, uint32_t n_worldpoints) {
    return add_view(c, camID, segs4, M, K, R, t, width, height, median_depth, worldpoints, n_worldpoints, true);
}
int l3d_get_visual_neighbors(l3d_ctx* c, uint32_t camID, uint32_t* out, uint32_t cap, uint32_t* n) {
    if (!c || !n) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    auto it = c->views.find(camID);
    if (it == c->views.end()) return fail(L3D_ERR_ARG, "unknown camera ID");
    *n = (uint32_t)it->second->visual_nbrs.size();
    uint32_t i = 0;
    if (out) for (uint32_t o : it->second->visual_nbrs) { if (i < cap) out[i] = o; ++i; }
    return L3D_OK;
}

static int add_view(l3d_ctx* c, uint32_t camID, const float* segs4, uint32_t M, const double K[9], const double R[9],
                    const double t[3], uint32_t width, uint32_t height, float median_depth, const uint32_t* neighbors,
                    uint32_t n_neighbors, bool by_worldpoints) {
    if (!c || !K || !R || !t) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (std::max(width, height) < 800) return fail(L3D_ERR_IMAGE_SMALL, "image is too small for reliable results");
    if (c->views.count(camID)) return fail(L3D_ERR_ID_IN_USE, "camera ID already in use");
    if (n_neighbors == 0 || !neighbors)
        return fail(L3D_ERR_NO_NEIGHBORS, by_worldpoints ? "view has no worldpoints" : "view has no visual neighbors");
    if (M == 0 || !segs4) return fail(L3D_ERR_NO_SEGMENTS, "no line segments");
    if (M >= (1u << 23)) return fail(L3D_ERR_LIMIT, "more than 2^23 segments per view");
    (void)hipSetDevice(c->device);
    auto v = std::make_unique<HostView>();
    v->cam = camID; v->M = M;
    v->segs.assign(segs4, segs4 + 4 * (size_t)M);
    for (float x : v->segs) v->coord_max = std::isfinite(x) ? std::fmax(v->coord_max, (double)std::fabs(x)) : INFINITY;
    v->width = width; v->height = height;
    v->initial_median_depth = (float)std::fmax(std::fabs(median_depth), kEps);
    init_view(*v, K, R, t);
    if (by_worldpoints) { v->by_worldpoints = true; v->worldpoints.assign(neighbors, neighbors + n_neighbors); }   // processWPlist
    else v->fixed_nbrs.assign(neighbors, neighbors + n_neighbors);                                                  // setVisualNeighbors
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

// leaves the BEGUN state without results: everything queued is drained, the views are moved back
// (matchImages translates them, line3D.cc:436/493) and a new l3d_match_begin is required
void abort_match(l3d_ctx* c) {   // (declared in l3d_ctx.h: l3d_phase_b.hip closes failed calls with it)
    (void)hipStreamSynchronize(c->stream);
    untranslate(*c);
    c->timing_pending = false; c->pending_launches = 0;
    c->state = l3d_ctx::IDLE;
    c->shard_world = 0; c->lists_ready = false; c->lists_prepared = false;
}

int l3d_match_abort(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    (void)hipSetDevice(c->device);
    if (c->state == l3d_ctx::BEGUN) abort_match(c);
    return L3D_OK;
}

static int match_begin_body(l3d_ctx* c);

// Failure behaviour: the views are translated for the duration BEGUN only.  Every exit of this function that is not
// L3D_OK -- limits, HIP errors -- leaves them untranslated and the context IDLE; a call while a previous begin is
// still open closes that one first (its views are moved back before the new translation is computed).
int l3d_match_begin(l3d_ctx* c, const l3d_match_params* p) {
    if (!c || !p) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->views.empty()) return fail(L3D_ERR_NO_VIEWS, "no images to match");
    (void)hipSetDevice(c->device);
    if (c->state == l3d_ctx::BEGUN) abort_match(c);
    if (p->kNN > 4096) return fail(L3D_ERR_LIMIT, "kNN > 4096");
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
    c->affinity_done = false;
    c->lines_done = false;
    c->n_hyps = 0;
    // line3D.cc:426-433
    c->med_scene_depth = c->const_regularization_depth;
    if (c->const_regularization_depth < 0.0f && c->fixed3Dregularizer && !c->views_avg_depths.empty()) {
        std::sort(c->views_avg_depths.begin(), c->views_avg_depths.end());
        c->med_scene_depth = c->views_avg_depths[c->views_avg_depths.size() / 2];
    }
    c->state = l3d_ctx::IDLE;
    // (a sharded affinity fill left open -- begin without finish -- holds the views translated: closed here)
    if (c->aff_shard_open) { untranslate(*c); c->aff_shard_open = false; }
    c->aff_parts_world = 0;
    translate(*c);
    const int rc = match_begin_body(c);
    if (rc != L3D_OK) {
        const std::string why = l3d_last_error();
        abort_match(c);
        set_error(why);
    }
    return rc;
}

// work items of a pair with Ms source rows in the bounded-kNN launch of this context (row form: 64 rows each; tile form: R
// positions of the padded class layout each)
static uint32_t match_items(const l3d_ctx* c, uint32_t Ms) {
    return c->layout_rows ? tile_src_cap(Ms, c->layout_rows) / c->layout_rows : (Ms + kMatchRows - 1) / kMatchRows;
}

static int match_begin_body(l3d_ctx* c) {
    g_trace.mark("begin: enter (views translated)");
    if (c->ev_on(0)) L3D_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    for (auto* v : c->order) {
        if (!c->fixed3Dregularizer) v->k = spatial_reg(*v, c->sigma_p);      // computeSpatialRegularizer
        else v->k = c->sigma_p / c->med_scene_depth;                          // update_k, view.h:124-127
        v->median_depth = 0.0f;
    }
    g_trace.mark("begin: regularisers");
    // fixed neighbours, line3D.cc:467-479 (sets persist across calls like visual_neighbors_)
    // (camIDs of the context, ascending: c->order is ordered by camID -- searched instead of the std::map of views)
    std::vector<uint32_t> cams(c->order.size());
    for (size_t i = 0; i < c->order.size(); ++i) cams[i] = c->order[i]->cam;
    for (auto* v : c->order)
        if (!v->by_worldpoints && v->visual_nbrs.empty())
            for (uint32_t n : v->fixed_nbrs)
                if (std::binary_search(cams.begin(), cams.end(), n)) v->visual_nbrs.insert(n);
    g_trace.mark("begin: fixed neighbours");
    // neighbours from the worldpoint overlap, line3D.cc:480-484 (every call anew, on the translated views)
    {
        bool any = false;
        for (auto& kv : c->views) any |= kv.second->by_worldpoints;
        if (any) {
            std::map<uint32_t, HostView*> vm;
            for (auto& kv : c->views) vm[kv.first] = kv.second.get();
            neighbors_from_worldpoints(vm, (uint32_t)c->num_neighbors);
        }
    }
    // The pair list, the fundamental matrices and the culling set-up are a function of the (translated) views, their
    // neighbour sets and kNN alone: when those are byte for byte what the previous call saw, the lists it built are
    // kept (C1: 0.1 ms of host work before the first kernel of the call can be enqueued).
    std::vector<unsigned char> sig;
    {
        auto put = [&](const void* q, size_t n) { const unsigned char* b = (const unsigned char*)q; sig.insert(sig.end(), b, b + n); };
        // tile form of the bounded-kNN kernel (k_match.hip): R rows per work item, the source pools laid out for it
        // (the launch the row form would make: every unordered neighbour pair is matched once, ceil(M / 64) items per pair)
        uint64_t est_items = 0;
        for (auto* v : c->order) est_items += (uint64_t)v->visual_nbrs.size() * ((v->M + kMatchRows - 1) / kMatchRows);
        est_items = (est_items + 1) / 2;
        c->tile_rows = c->kNN > 0 ? match_tile_rows(0, c->brute, est_items) : 0u;
        c->layout_rows = c->kNN > 0 ? match_layout_rows(0, c->brute, c->tile_rows) : 0u;   // padded class layout of the source rows (k_cull_prepare)
        const int32_t head[5] = {c->kNN, c->use_cull ? 1 : 0, c->brute ? 1 : 0, (int32_t)c->tile_rows, (int32_t)c->layout_rows};
        put(head, sizeof(head));
        for (auto* v : c->order) {
            const uint32_t iv[5] = {v->cam, v->index, v->M, (uint32_t)v->width, (uint32_t)v->height};
            put(iv, sizeof(iv));
            put(v->K.m, 72); put(v->R.m, 72); put(&v->t, sizeof(v->t)); put(&v->C, sizeof(v->C));
            const uint32_t nn = (uint32_t)v->visual_nbrs.size();
            put(&nn, 4);
            for (uint32_t n : v->visual_nbrs) put(&n, 4);
        }
    }
    g_trace.mark("begin: regularisers, neighbours, scene signature");
    const bool same_scene = c->begin_sig_valid && sig == c->begin_sig;
    uint64_t cs_off = c->cull_tot[0], ct_off = c->cull_tot[1], ck_off = c->cull_tot[2]; uint32_t cc_off = (uint32_t)c->cull_tot[3];
    if (!same_scene) {
    c->begin_sig_valid = false;
    for (auto* v : c->order) { v->out_pairs.clear(); v->in_pairs.clear(); }
    cs_off = ct_off = ck_off = 0; cc_off = 0;
    // directed pair list, line3D.cc:704-741.  Round 6: in two steps -- the ENUMERATION (which pairs, in which order, where their
    // slots / rows / culling pools start: sequential by definition, no arithmetic) and the per-pair ARITHMETIC (fundamental
    // matrix, baseline, culling descriptor: independent per pair, spread over a few host threads from 1 024 pairs on).  The
    // first matchImages of a scene of 1 024 views spent 2-5 ms here before its first kernel could be enqueued
    // (profiles/r06_first_call.txt).
    c->pairs.clear(); c->pair_src_cam.clear(); c->pair_tgt_cam.clear(); c->cull.clear();
    uint32_t w_item = 0;
    uint64_t slot_off = 0; uint32_t row_off = 0;
    c->pair_tests = 0;
    {
        size_t n_guess = 0;
        for (auto* v : c->order) n_guess += v->visual_nbrs.size();
        c->pairs.reserve(n_guess); c->cull.reserve(n_guess); c->pair_src_cam.reserve(n_guess); c->pair_tgt_cam.reserve(n_guess);
    }
    std::vector<std::pair<HostView*, HostView*>> ends;
    // camID -> view: c->order is ascending in camID
    auto view_of = [&](uint32_t cam) -> HostView* {
        auto it = std::lower_bound(cams.begin(), cams.end(), cam);
        return it != cams.end() && *it == cam ? c->order[(size_t)(it - cams.begin())] : nullptr;
    };
    for (auto* v : c->order)
        for (uint32_t tcam : v->visual_nbrs) {
            HostView* t = view_of(tcam);
            if (!t) t = c->views[tcam].get();   // (never: a neighbour set only holds views of the context)
            // `matched_[src]` of the reference holds tcam already iff the pair was created from the other side: tcam comes
            // earlier in the order and lists this view among its neighbours (a set holds a neighbour once)
            if (t->index < v->index && t->visual_nbrs.count(v->cam)) continue;
            PairDesc pd{};
            pd.src = v->index; pd.tgt = t->index; pd.Ms = v->M; pd.Mt = t->M;
            pd.K = c->kNN > 0 ? (uint32_t)c->kNN : 0u;
            pd.row_off = row_off; pd.slot_off = slot_off;
            slot_off += (uint64_t)pd.Ms * pd.K; row_off += pd.Ms;
            const uint32_t pi = (uint32_t)c->pairs.size();
            v->out_pairs.push_back(pi);
            if (t->index > v->index) t->in_pairs.push_back(pi);   // inverse only if tgt not yet processed (:1680)
            c->pairs.push_back(pd);
            PairCull pc{};
            // (the pools of a pair are reserved whether or not its geometry turns out to suit culling: the offsets must not
            // depend on the arithmetic below)
            const bool may_cull = c->use_cull && pd.Ms <= kCullMaxSegs && pd.Mt <= kCullMaxSegs && pd.Ms && pd.Mt;
            pc.s_off = cs_off; pc.t_off = ct_off; pc.c_off = cc_off; pc.k_off = ~0ull;
            pc.w_item0 = w_item; pc.sorted_copy = pd.Mt >= kSortedCopyMinSegs ? 1u : 0u; w_item += match_items(c, pd.Ms);
            if (may_cull && std::max(pd.Ms, pd.Mt) > kCullLdsSegs) {   // sort keys of this pair in global scratch
                uint32_t a = 64, b = 64;
                while (a < pd.Ms) a <<= 1;
                while (b < pd.Mt) b <<= 1;
                pc.k_off = ck_off; ck_off += (uint64_t)a + b;
            }
            // (tile form: the rows of a pair take tile_src_cap positions -- every width class padded to a multiple of R)
            if (may_cull) { cs_off += c->layout_rows ? tile_src_cap(pd.Ms, c->layout_rows) : pd.Ms; ct_off += pd.Mt; cc_off += (pd.Mt + 63) / 64; }
            c->cull.push_back(pc);
            ends.emplace_back(v, t);
            c->pair_src_cam.push_back(v->cam); c->pair_tgt_cam.push_back(tcam);
            c->pair_tests += (uint64_t)pd.Ms * pd.Mt;
        }
    g_trace.mark("begin: pair list enumerated");
    {
        static const bool no_fast = std::getenv("L3D_NO_FASTMATH") != nullptr;   // diagnostic switch: the compiler's own expansions
        const bool use_cull = c->use_cull;
        auto fill = [&](size_t p0, size_t p1) {
            for (size_t p = p0; p < p1; ++p) {
                const HostView* v = ends[p].first; const HostView* t = ends[p].second;
                PairDesc& pd = c->pairs[p];
                fundamental(*v, *t, pd.F);
                // |C_src - C_tgt|, rounded up (k_lists.hip: window_sq)
                const double d = norm(v->C - t->C);
                pd.cc_dist = std::nextafterf((float)(d * (1.0 + 1e-6)), INFINITY);
                // kPairFastMath (l3d_dev.h): everything that enters the pair's exact arithmetic is finite and far from the
                // ends of the double range, so that IEEE division / sqrt need no operand scaling
                bool fm = v->coord_max <= 1e7 && t->coord_max <= 1e7;
                for (int k = 0; k < 9 && fm; ++k) { const double a = std::fabs(pd.F[k]); fm = std::isfinite(a) && (a == 0.0 || (a >= 1e-30 && a <= 1e30)); }
                for (double cval : {v->C.x, v->C.y, v->C.z, t->C.x, t->C.y, t->C.z}) fm = fm && std::isfinite(cval) && std::fabs(cval) <= 1e30;
                pd.flags = (fm && !no_fast) ? kPairFastMath : 0u;
                pair_baseline(v->C, t->C, pd);
                PairCull& pc = c->cull[p];
                if (use_cull && pd.Ms <= kCullMaxSegs && pd.Mt <= kCullMaxSegs && pd.Ms && pd.Mt)
                    make_cull(pd.F, v->width, v->height, t->width, t->height, pc);
            }
        };
        const size_t P = c->pairs.size();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t n_thr = P >= 1024 ? std::min<size_t>({8, hw / 2 ? hw / 2 : 1, P / 256}) : 1;
        if (n_thr > 1) {
            std::vector<std::thread> th;
            for (size_t k = 1; k < n_thr; ++k) th.emplace_back(fill, P * k / n_thr, P * (k + 1) / n_thr);
            fill(0, P / n_thr);
            for (auto& x : th) x.join();
        } else fill(0, P);
    }
    c->n_slots = slot_off; c->n_rows_total = row_off;
    if (c->n_slots >= (1ull << 32) || c->pairs.size() >= (1u << 24))
        return fail(L3D_ERR_LIMIT, "slot buffer / pair list exceed the 32-bit slot and 24-bit pair indices of phase B");
    if (c->kNN > 0 && !c->pairs.empty()) {
        // bounded kNN keeps the per-row top-K tables of 64 rows in LDS (k_match.hip): the real limit of this build
        size_t n_work = 0; uint32_t maxMt = 0;
        for (auto& pd : c->pairs) { n_work += match_items(c, pd.Ms); maxMt = std::max(maxMt, pd.Mt); }
        const uint32_t wpg = match_waves_per_group(0, c->brute, (uint32_t)std::min<size_t>(n_work, 0xFFFFFFFFu));
        auto fits = [&](uint32_t K) { return match_lds_bytes(0, K, maxMt < 65536u && K < 32768u && !c->brute, wpg, c->brute, c->tile_rows,
                                                           match_row_cache(0, c->brute, (uint32_t)std::min<size_t>(n_work, 0xFFFFFFFFu), c->tile_rows)) <= 160 * 1024; };
        // beyond that, every row takes the exact replay path (k_match_tied_rows): slower per row, any kNN <= 4096
        c->knn_replay = !fits((uint32_t)c->kNN);
        if (c->knn_replay) l3d::g_knn_replay_calls.fetch_add(1, std::memory_order_relaxed);
    } else c->knn_replay = false;
    c->cull_tot[0] = cs_off; c->cull_tot[1] = ct_off; c->cull_tot[2] = ck_off; c->cull_tot[3] = cc_off;
    c->begin_sig.swap(sig); c->begin_sig_valid = true;
    }
    g_trace.mark(same_scene ? "begin: pair list kept" : "begin: pair list, fundamental matrices, cull descriptors built");
    c->pair_done.assign(c->pairs.size(), 0);
    c->pair_counted.assign(c->pairs.size(), 0);
    c->ragged = false;
    c->shard_world = 0; c->lists_ready = false; c->lists_prepared = false;
    int rc = upload_views(*c);
    if (rc) return rc;
    g_trace.mark("begin: views uploaded / k_prep_views enqueued");
    // the inverse-target stream of phase B: written by the match epilogue (bounded kNN) or by k_orient_all; 16-bit entries
    // when every view has fewer than 65 535 segments (L3D_INV_TGT32=1: always 32-bit, A/B switch)
    {
        uint32_t mx = 0;
        for (auto* v : c->order) mx = std::max(mx, v->M);
        c->tgt16 = (mx < 65535u && !std::getenv("L3D_INV_TGT32")) ? 1u : 0u;
    }
    if (c->kNN > 0) {
        L3D_HIP_CHECK(c->d_inv_tgt.reserve(std::max<uint64_t>(c->n_slots, 1)));
        L3D_HIP_CHECK(c->d_hyp_p.reserve(std::max<uint64_t>(c->n_slots, 1)));
        L3D_HIP_CHECK(c->d_hyp_q.reserve(std::max<uint64_t>(c->n_slots, 1)));
    }
    L3D_HIP_CHECK(c->d_pairs.reserve(std::max<size_t>(c->pairs.size(), 1)));
    {
        bool sent = false;
        L3D_HIP_CHECK(upload_table(c->d_pairs, c->h_pairs, c->pairs.data(), c->pairs.size() * sizeof(PairDesc), c->up_pairs,
                                   c->stream, &sent));
        if (sent || c->pairs.empty()) ++c->pairs_version;
    }
    if (c->kNN > 0) L3D_HIP_CHECK(c->d_slots.reserve(std::max<uint64_t>(c->n_slots, 1)));
    L3D_HIP_CHECK(c->d_cull.reserve(std::max<size_t>(c->cull.size(), 1)));
    L3D_HIP_CHECK(upload_table(c->d_cull, c->h_cull, c->cull.data(), c->cull.size() * sizeof(PairCull), c->up_cull, c->stream));
    L3D_HIP_CHECK(c->d_src_perm.reserve(std::max<uint64_t>(cs_off, 1)));
    L3D_HIP_CHECK(c->d_src_band.reserve(std::max<uint64_t>(cs_off, 1)));
    L3D_HIP_CHECK(c->d_tgt_perm.reserve(std::max<uint64_t>(ct_off, 1)));
    L3D_HIP_CHECK(c->d_tgt_sf.reserve(std::max<uint64_t>(ct_off, 1)));
    {   // walk-order copies of the targets' exact-test records: only when some pair keeps them (large views)
        bool any_sorted = false;
        for (const PairCull& pc : c->cull) any_sorted |= pc.enabled && pc.sorted_copy;
        L3D_HIP_CHECK(c->d_tgt_s4.reserve(any_sorted ? std::max<uint64_t>(ct_off, 1) : 1));
        L3D_HIP_CHECK(c->d_tgt_sd.reserve(any_sorted ? std::max<uint64_t>(ct_off, 1) : 1));   // (48-byte float records since round 4)
    }
    L3D_HIP_CHECK(c->d_tgt_band.reserve(std::max<uint64_t>(ct_off, 1)));
    L3D_HIP_CHECK(c->d_chunk_band.reserve(std::max<uint32_t>(cc_off, 1)));
    L3D_HIP_CHECK(c->d_cull_keys.reserve(std::max<uint64_t>(ck_off, 1)));
    if (c->ev_on(1)) L3D_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    g_trace.mark("begin: tables uploaded, pools reserved");
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

// reads the phase-A events of the last run_match_kernel (the stream must have passed ev[5])
void collect_match_timing(l3d_ctx* c) {   // (declared in l3d_ctx.h: l3d_phase_b.hip reads the events at the end of the call)
    if (!c->timing_pending) return;
    c->tm.match_kernel_ms += ev_ms(c, 4, 5);
    c->tm.cull_prepare_ms += ev_ms(c, 8, 4);
    c->tm.match_kernel_launches += c->pending_launches;
    c->timing_pending = false; c->pending_launches = 0;
}

// enqueues (no host synchronisation) the cull set-up and the pair kernel for pairs [first, first+count)
static int run_match_kernel(l3d_ctx* c, int mode, uint32_t first, uint32_t count) {
    size_t n_work = 0;
    uint32_t maxK = 0, maxM = 0, maxMt = 0;
    const bool hot = mode == 0 && !c->brute && !c->knn_replay;
    const uint32_t tile = hot ? c->tile_rows : 0u;          // kernel form (keep-all passes, brute force: the row form)
    const uint32_t layout = hot ? c->layout_rows : 0u;      // padded class layout of the source rows (0: Ms rows per pair)
    const uint32_t rows_per_item = layout ? layout : (uint32_t)kMatchRows;
    auto items_of = [&](uint32_t Ms) { return layout ? tile_src_cap(Ms, layout) / layout : (Ms + kMatchRows - 1) / kMatchRows; };
    for (uint32_t p = first; p < first + count; ++p) n_work += items_of(c->pairs[p].Ms);
    if (!n_work) return L3D_OK;
    for (uint32_t p = first; p < first + count; ++p) {
        const PairDesc& pd = c->pairs[p];
        maxK = std::max(maxK, pd.K);
        if (c->cull[p].enabled) maxM = std::max(maxM, std::max(pd.Ms, pd.Mt));
        maxMt = std::max(maxMt, pd.Mt);
    }
    const bool replay_all = mode == 0 && c->knn_replay;   // kNN beyond the LDS tables: every row through k_match_tied_rows
    if (!replay_all && match_lds_bytes(mode, maxK, false, match_waves_per_group(mode, c->brute, (uint32_t)n_work), c->brute, tile,
                                       match_row_cache(mode, c->brute, (uint32_t)n_work, tile)) > 160 * 1024)
        return fail(L3D_ERR_LIMIT, "kNN too large for the LDS top-K table");
    L3D_HIP_CHECK(c->d_work.reserve(n_work));
    // the work list of these pairs is on the device already when the pair list has not changed since it was sent
    if (!(c->work_key.version == c->pairs_version && c->work_key.first == first && c->work_key.count == count &&
          c->work_key.dev == (const void*)c->d_work.p && c->work_key.rows == rows_per_item && c->work_key.items == n_work)) {
        L3D_HIP_CHECK(c->h_work.reserve(n_work));
        WorkItem* work = c->h_work.p;
        size_t w = 0;
        for (uint32_t p = first; p < first + count; ++p)
            for (uint32_t k = 0, n = items_of(c->pairs[p].Ms); k < n; ++k) work[w++] = WorkItem{p, k * rows_per_item};
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_work.p, work, n_work * sizeof(WorkItem), hipMemcpyHostToDevice, c->stream));
        c->work_key.version = c->pairs_version; c->work_key.first = first; c->work_key.count = count;
        c->work_key.dev = c->d_work.p; c->work_key.rows = rows_per_item; c->work_key.items = n_work;
    }
    if (c->ev_on(8)) L3D_HIP_CHECK(hipEventRecord(c->ev[8], c->stream));
    CullPools pools{c->d_cull.p, c->d_src_perm.p, c->d_src_band.p, c->d_tgt_perm.p, c->d_tgt_sf.p, c->d_tgt_band.p,
                    c->d_chunk_band.p, c->d_cull_keys.p};
    pools.tgt_s4 = c->d_tgt_s4.p; pools.tgt_sd = c->d_tgt_sd.p; pools.padded_rows = layout;
    const bool keep_all_unculled = std::getenv("L3D_KEEPALL_NO_CULL") != nullptr;   // diagnostic switch (A/B; read per call)
    // (keep-all mode: the count pass takes the culled walk, the fill pass streams -- k_match.hip)
    if (c->brute || !maxM || mode == 2 || (mode == 1 && keep_all_unculled)) pools.cull = nullptr;
    else {
        static const int no_order = [] { const char* e = std::getenv("L3D_MATCH_ORDER"); return e && std::atoi(e) == 0; }();
        // longest-first launch order (k_order_items) where the launch has one to three items per wave slot (fewer: all
        // start at once; more: the tail is short against the whole) -- C1: kernel 1.08 -> 1.03 ms; C2 / C4: +6 % with it
        // (tile form: four to eight times as many, shorter items, the widest class of every pair first: list order)
        // (not for the exact-replay path of a kNN beyond the LDS tables: its launch is unpadded while PairCull::w_item0 counts
        // the padded class layout -- k_order_items would file items past the end of d_item_bucket)
        if (!tile && !no_order && !replay_all && n_work > kMatchOrderMinItems && n_work <= kMatchOrderMaxItems) {
            L3D_HIP_CHECK(c->d_item_bucket.reserve(n_work)); L3D_HIP_CHECK(c->d_item_order.reserve(n_work));
            if (!c->d_order_done.p) {            // zeroed once: the kernel re-arms it
                L3D_HIP_CHECK(c->d_order_done.reserve(1));
                L3D_HIP_CHECK(hipMemsetAsync(c->d_order_done.p, 0, 4, c->stream));
            }
            pools.item_bucket = c->d_item_bucket.p; pools.item_order = c->d_item_order.p;
            pools.order_done = c->d_order_done.p;
            pools.w_base = c->cull[first].w_item0; pools.cost_max = maxMt;
        }
        // (the source pools are laid out for the form this launch takes: pools.padded_rows)
        L3D_HIP_CHECK(launch_cull_prepare(c->d_views.p, c->d_pairs.p, first, count, maxM, pools, layout, c->stream));
        L3D_HIP_CHECK(launch_order_items(c->d_pairs.p, first, count, maxMt, pools, (uint32_t)n_work, c->stream));
    }
    if (c->ev_on(4)) L3D_HIP_CHECK(hipEventRecord(c->ev[4], c->stream));
    const bool ix16 = maxMt < 65536u && maxK < 32768u;   // 16-bit indices in the kernel's LDS tables (top bit of a row's minpos: tie flag)
    // bounded kNN: the orientation filter of phase B is fused into the epilogue
    OrientFuse of{mode == 0 ? c->d_inv_tgt.p : nullptr, c->tgt16, OrientThr{c->orient_lo, c->orient_hi}, nullptr, nullptr, 0u, nullptr, nullptr,
                  mode == 0 ? c->d_hyp_p.p : nullptr, mode == 0 ? c->d_hyp_q.p : nullptr};
    if (mode == 1 && c->ragged) { of.keep_rec = c->d_keep_rec.p; of.keep_cap = c->keep_cap; }   // (the single pass of the keep-all mode)
    uint32_t tie_stride = 0;
    if (mode == 0) {   // rows with equal overlaps are collected here and replayed in the reference's heap order
        uint32_t mt = 0;
        for (auto& pd : c->pairs) mt = std::max(mt, pd.Mt);
        tie_stride = std::max(mt, 1u);
        if (!c->d_tie_count.p) {   // zeroed once: the replay kernel zeroes the next launch's counter
            L3D_HIP_CHECK(c->d_tie_count.reserve(4));
            L3D_HIP_CHECK(hipMemsetAsync(c->d_tie_count.p, 0, 16, c->stream));
        }
        L3D_HIP_CHECK(c->d_tie_list.reserve(std::max<uint32_t>(c->n_rows_total, 1)));
        L3D_HIP_CHECK(c->d_tie_heap.reserve(2 * (size_t)match_tied_grid(tie_stride) * tie_stride));
        // two alternating queue counters ([0], [1]); [2] = rows replayed so far
        of.tie_count = c->d_tie_count.p + (c->tie_seq & 1u); of.tie_next = c->d_tie_count.p + ((c->tie_seq + 1) & 1u);
        of.tie_total = c->d_tie_count.p + 2; ++c->tie_seq;
        of.tie_list = c->d_tie_list.p; of.tie_cap = c->n_rows_total;
    }
    if (replay_all) {
        uint32_t maxMs = 0;
        for (uint32_t p = first; p < first + count; ++p) maxMs = std::max(maxMs, c->pairs[p].Ms);
        L3D_HIP_CHECK(launch_queue_all_rows(c->d_pairs.p, first, count, maxMs, c->pairs[first].row_off, of, c->stream));
    } else
    L3D_HIP_CHECK(launch_match_pairs(mode, c->brute, c->d_views.p, c->d_pairs.p, c->d_work.p, (uint32_t)n_work, maxK,
                                     c->d_slots.p, c->d_row_counts.p, c->epipolar_overlap, pools, of, ix16, tile, c->stream));
    if (c->ev_on(5)) L3D_HIP_CHECK(hipEventRecord(c->ev[5], c->stream));
    if (mode == 0) {
        L3D_HIP_CHECK(launch_match_tied_rows(c->d_views.p, c->d_pairs.p, c->d_slots.p, maxK, c->epipolar_overlap, of, pools,
                                             c->d_tie_heap.p, tie_stride, c->stream));
        for (uint32_t p = first; p < first + count; ++p) c->pair_counted[p] = 1;
    }
    if (pools.cull)
        for (uint32_t p = first; p < first + count; ++p) c->tm.culled_pairs += c->cull[p].enabled;
    c->timing_pending = true; c->pending_launches += 1;
    return L3D_OK;
}

static int match_pairs_impl(l3d_ctx* c, uint32_t first, uint32_t count, bool sync);

// ---- keep-all mode (kNN <= 0): one culled pass + assembly of ragged rows ----------------------------------------------
// The count pass keeps every accepted match at (row, arrival index) in a scratch of keep_cap records per row; a scan of the
// counts places the rows, the host learns {first slot, longest row, slots} of every pair (ONE small read-back: it has to size
// the slot buffer), k_keep_assemble writes slots, flags and the streams of phase B.  A row longer than the scratch: the pass
// is repeated with a larger one (the counts are exact either way); a scratch beyond kKeepScratchMax: the two-pass form.
static constexpr int kKeepAllTwoPass = -1000;
static constexpr uint64_t kKeepScratchMax = 48ull << 30;
static int keep_all_single_pass(l3d_ctx* c, uint32_t first, uint32_t count) {
    const uint32_t P = (uint32_t)c->pairs.size(), R = c->n_rows_total;
    if (!P || !R) return kKeepAllTwoPass;
    hipStream_t st = c->stream;
    if (!c->keep_cap) {   // (L3D_KEEPALL_CAP: the first scratch size of a context -- a test hook for the repeat with a larger one)
        const char* e = std::getenv("L3D_KEEPALL_CAP");
        c->keep_cap = e && std::atoi(e) > 0 ? (uint32_t)std::atoi(e) : 96u;
    }
    L3D_HIP_CHECK(c->d_row_start.reserve((size_t)R + 3));   // (row starts [R + 1] | the scan's total | the longest row)
    L3D_HIP_CHECK(c->d_keep_info.reserve(P)); L3D_HIP_CHECK(c->d_row_pair.reserve(R)); L3D_HIP_CHECK(c->h_keep_info.reserve(P));
    L3D_HIP_CHECK(c->d_scan_ws.reserve_zeroed(scan_ws_words((size_t)R + 1, 4), st));
    const uint4* info = c->h_keep_info.p;
    c->ragged = true;
    auto outputs = [&](uint64_t ns) -> hipError_t {   // the arrays k_keep_assemble writes; returns through `cap` what they hold
        hipError_t e = c->d_slots.reserve(ns);
        if (e == hipSuccess) e = c->d_slot_row.reserve(ns);
        if (e == hipSuccess) e = c->d_inv_tgt.reserve(ns);
        if (e == hipSuccess) e = c->d_hyp_p.reserve(ns);
        if (e == hipSuccess) e = c->d_hyp_q.reserve(ns);
        return e;
    };
    auto assemble = [&](uint64_t cap) -> hipError_t {
        OrientFuse of{c->d_inv_tgt.p, c->tgt16, OrientThr{c->orient_lo, c->orient_hi}, nullptr, nullptr, 0u, nullptr, nullptr,
                      c->d_hyp_p.p, c->d_hyp_q.p};
        of.keep_rec = c->d_keep_rec.p; of.keep_cap = c->keep_cap; of.slot_row = c->d_slot_row.p;
        return launch_keep_assemble(c->d_views.p, c->d_pairs.p, R, cap, c->d_row_start.p, c->d_row_pair.p, c->d_blk_row.p, c->d_row_start.p + R + 2, c->d_slots.p, of, st);
    };
    // The outputs are sized BEFORE the pass -- from the previous call of the context, else 16 slots per row -- and the assembly
    // is enqueued right behind the scan: the host reads the pair table back once, at the end, and only launches the assembly
    // again when the guess was too small (the kernel writes nothing then).
    const uint64_t guess = std::min<uint64_t>(std::max<uint64_t>(c->keep_last_total + c->keep_last_total / 8 + (1u << 16), (uint64_t)R * 16), 0xFFFFFFFFull);
    L3D_HIP_CHECK(outputs(guess));
    uint64_t cap = std::min<uint64_t>({c->d_slots.cap, c->d_slot_row.cap, c->d_hyp_p.cap, c->d_hyp_q.cap,
                                       c->d_inv_tgt.cap * (c->tgt16 ? 2u : 1u), (size_t)0xFFFFFFFFull});
    cap = std::min<uint64_t>(cap, guess + guess / 2);    // (a block cache may hand out far larger arrays: the grid covers `cap`)
    for (int attempt = 0;; ++attempt) {
        if ((uint64_t)R * c->keep_cap * sizeof(Slot) > kKeepScratchMax || attempt > 2) { c->ragged = false; return kKeepAllTwoPass; }
        L3D_HIP_CHECK(c->d_keep_rec.reserve((size_t)R * c->keep_cap));
        // (a block mark per kKeepBlock slots of at most min(rows x cap, 2^32) slots)
        const uint32_t n_blk = (uint32_t)(std::min<uint64_t>((uint64_t)R * c->keep_cap, 1ull << 32) / kKeepBlock) + 2;
        L3D_HIP_CHECK(c->d_blk_row.reserve(n_blk));
        int rc = run_match_kernel(c, 1, first, count);
        if (rc) { c->ragged = false; return rc; }
        L3D_HIP_CHECK(launch_scan(c->d_row_counts.p, R, c->d_row_start.p, c->d_scan_ws.p, c->d_row_start.p + R + 1, st));
        L3D_HIP_CHECK(launch_keep_pair_info(c->d_pairs.p, P, c->d_row_counts.p, c->d_row_start.p, c->d_keep_info.p, c->d_row_pair.p, c->d_blk_row.p, n_blk, c->d_row_start.p + R + 2, st));
        L3D_HIP_CHECK(assemble(cap));
        if (c->ev_on(5)) L3D_HIP_CHECK(hipEventRecord(c->ev[5], st));   // (the kernel time of the call: pass + scan + assembly)
        L3D_HIP_CHECK(hipMemcpyAsync(c->h_keep_info.p, c->d_keep_info.p, (size_t)P * sizeof(uint4), hipMemcpyDeviceToHost, st));
        L3D_HIP_CHECK(hipStreamSynchronize(st));
        collect_match_timing(c);
        uint32_t longest = 0;
        for (uint32_t p = 0; p < P; ++p) longest = std::max(longest, info[p].y);
        if (longest <= c->keep_cap) break;
        c->keep_cap = (longest + longest / 8 + 31u) & ~31u;   // (kept for the calls to come)
        l3d::g_keep_all_repeats.fetch_add(1, std::memory_order_relaxed);
    }
    uint64_t total = 0;
    c->pair_nslots.assign(P, 0);
    for (uint32_t p = 0; p < P; ++p) {
        const uint64_t n = (uint64_t)info[p].z | ((uint64_t)info[p].w << 32);
        c->pair_nslots[p] = n; total += n;
    }
    if (total >= (1ull << 32)) { c->ragged = false; return fail(L3D_ERR_LIMIT, "slot buffer exceeds the 32-bit slot indices of phase B"); }
    for (uint32_t p = 0; p < P; ++p) {           // (K: the longest row, what l3d_get_pair_slots pads the rows to)
        c->pairs[p].slot_off = info[p].x; c->pairs[p].K = std::max(info[p].y, 1u);
    }
    c->n_slots = total; c->keep_last_total = total;
    if (total > cap) {                           // the guess was too small: nothing has been written
        L3D_HIP_CHECK(outputs(total));
        if (c->ev_on(8)) L3D_HIP_CHECK(hipEventRecord(c->ev[8], st));
        if (c->ev_on(4)) L3D_HIP_CHECK(hipEventRecord(c->ev[4], st));
        L3D_HIP_CHECK(assemble(total));
        if (c->ev_on(5)) L3D_HIP_CHECK(hipEventRecord(c->ev[5], st));
        c->timing_pending = true;
        l3d::g_keep_all_repeats.fetch_add(1u << 16, std::memory_order_relaxed);   // (high half: assemblies launched twice)
    } else L3D_HIP_CHECK(outputs(std::max<uint64_t>(total, 1)));   // (no-op: the arrays hold `cap` >= total)
    // (K and slot_off of the pairs: through the pinned staging copy, which stays the record of what the device holds)
    L3D_HIP_CHECK(upload_table(c->d_pairs, c->h_pairs, c->pairs.data(), (size_t)P * sizeof(PairDesc), c->up_pairs, st));
    for (uint32_t p = 0; p < P; ++p) c->pair_counted[p] = 1;   // (flags and streams are written: no k_orient_all in phase B)
    return L3D_OK;
}

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
    if (c->ev_on(2)) L3D_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    int rc = L3D_OK;
    if (c->kNN > 0) {
        rc = run_match_kernel(c, 0, first, count);
    } else {
        // kNN <= 0: keep every accepted match (line3D.cc:987-992)
        if (first != 0 || count != c->pairs.size())
            return fail(L3D_ERR_LIMIT, "kNN <= 0 needs all pairs in one l3d_match_pairs call");
        L3D_HIP_CHECK(c->d_row_counts.reserve((size_t)c->n_rows_total + 1));
        // ONE culled pass that keeps what it accepts + an assembly of RAGGED rows (round 6, k_match.hip: k_keep_assemble).
        // L3D_KEEPALL_TWO_PASS=1 / L3D_KEEPALL_NO_CULL=1 (A/B, read per call): the form of rounds 3-5 below -- count pass
        // (culled / streamed), rows sized by the host from the counts (K of a pair = its longest row), fill pass streamed.
        const bool two_pass = std::getenv("L3D_KEEPALL_TWO_PASS") != nullptr || std::getenv("L3D_KEEPALL_NO_CULL") != nullptr;
        if (!two_pass) {
            rc = keep_all_single_pass(c, first, count);
            if (rc != kKeepAllTwoPass) {
                if (rc) return rc;
                if (c->ev_on(3)) L3D_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
                if (sync) {
                    L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
                    collect_match_timing(c);
                    c->tm.match_pairs_ms += ev_ms(c, 2, 3);
                }
                for (uint32_t p = first; p < first + count; ++p) c->pair_done[p] = 1;
                return L3D_OK;
            }
        }
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
    if (c->ev_on(3)) L3D_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    if (sync) {
        if (c->ev_on(3)) L3D_HIP_CHECK(hipEventSynchronize(c->ev[3]));
        else L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
        collect_match_timing(c);
        c->tm.match_pairs_ms += ev_ms(c, 2, 3);
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
    const OrientFuse of{c->d_inv_tgt.p, c->tgt16, OrientThr{c->orient_lo, c->orient_hi}, nullptr, nullptr, 0u, nullptr, nullptr,
                        c->d_hyp_p.p, c->d_hyp_q.p};
    L3D_HIP_CHECK(launch_expand_slot_idx(c->d_views.p, c->d_pairs.p, first, count, max_row_slots, c->d_slot_idx.p,
                                         c->d_slots.p, of, c->stream));
    for (uint32_t p = first; p < first + count; ++p) c->pair_done[p] = c->pair_counted[p] = 1;
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
    if (rc) {
        const std::string why = l3d_last_error();
        abort_match(c);
        set_error(why);
        return rc;
    }
    rc = l3d_match_finish(c);
    g_trace.mark("matchImages done");
    g_trace.flush();
    return rc;
}

}  // extern "C"
