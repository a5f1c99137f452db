// l3d_api.hip -- context layer of libl3dpp_hip.so (include/l3dpp_hip.h): the host-side driver that mirrors
// Line3D::addImage / matchImages (line3D.cc:112-227, 375-497, 702-778).  No CPU fallback exists: every compute step
// is a HIP kernel launch; without a usable device the calls fail with L3D_ERR_HIP.
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
    if (need_warm) { std::lock_guard<std::mutex> lk(warm_mu); need_warm = warmed.insert(device).second; }
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
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        { const ReleaseSynced drained; d.release(); h.release(); }
        if (!ok) { set_error("start-up launches failed: no usable HIP device"); l3d_destroy(c); return nullptr; }
    }
    return c;
}

void l3d_destroy(l3d_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& s2 : c->aux) if (s2) (void)hipStreamSynchronize(s2);
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
    c->d_row_counts.release();
    c->d_seg_base.release(); c->d_gseg_view.release();
    c->d_scal.release();
    c->d_surv_off.release(); c->d_hyp_off.release();
    c->d_surv_tg.release(); c->d_surv_sg.release();
    c->d_inv_tgt.release(); c->d_gsegx.release(); c->d_gsegd32.release();
    c->d_tie_count.release(); c->d_tie_list.release(); c->d_tie_heap.release(); c->d_item_bucket.release(); c->d_item_order.release(); c->d_order_done.release();
    c->d_coll_cnt.release(); c->d_coll_off.release(); c->d_coll_idx.release(); c->d_item_cnt.release();
    c->d_item_off.release(); c->d_item_seg.release(); c->d_item_sim.release();
    c->d_surv.release();
    c->d_hyp_of_seg.release(); c->d_depths.release(); c->d_med = nullptr; c->d_hyps.release();
    c->d_vaff.release(); c->d_simv.release(); c->h_vaff.release(); c->d_ca.release(); c->d_cb.release();
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
static void abort_match(l3d_ctx* c) {
    (void)hipStreamSynchronize(c->stream);
    for (auto& s2 : c->aux) if (s2) (void)hipStreamSynchronize(s2);
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
    translate(*c);
    const int rc = match_begin_body(c);
    if (rc != L3D_OK) {
        const std::string why = l3d_last_error();
        abort_match(c);
        set_error(why);
    }
    return rc;
}

static int match_begin_body(l3d_ctx* c) {
    if (c->ev_on(0)) L3D_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    for (auto* v : c->order) {
        if (!c->fixed3Dregularizer) v->k = spatial_reg(*v, c->sigma_p);      // computeSpatialRegularizer
        else v->k = c->sigma_p / c->med_scene_depth;                          // update_k, view.h:124-127
        v->median_depth = 0.0f;
    }
    // fixed neighbours, line3D.cc:467-479 (sets persist across calls like visual_neighbors_)
    for (auto* v : c->order)
        if (!v->by_worldpoints && v->visual_nbrs.empty())
            for (uint32_t n : v->fixed_nbrs)
                if (c->views.count(n)) v->visual_nbrs.insert(n);
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
        const int32_t head[3] = {c->kNN, c->use_cull ? 1 : 0, c->brute ? 1 : 0};
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
    const bool same_scene = c->begin_sig_valid && sig == c->begin_sig;
    uint64_t cs_off = c->cull_tot[0], ct_off = c->cull_tot[1], ck_off = c->cull_tot[2]; uint32_t cc_off = (uint32_t)c->cull_tot[3];
    if (!same_scene) {
    c->begin_sig_valid = false;
    for (auto* v : c->order) { v->out_pairs.clear(); v->in_pairs.clear(); }
    cs_off = ct_off = ck_off = 0; cc_off = 0;
    // directed pair list, line3D.cc:704-741
    c->pairs.clear(); c->pair_src_cam.clear(); c->pair_tgt_cam.clear(); c->cull.clear();
    uint32_t w_item = 0;
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
            {   // |C_src - C_tgt|, rounded up (k_lists.hip: window_sq)
                const double d = norm(v->C - t->C);
                pd.cc_dist = std::nextafterf((float)(d * (1.0 + 1e-6)), INFINITY);
                // kPairFastMath (l3d_dev.h): everything that enters the pair's exact arithmetic is finite and far from the
                // ends of the double range, so that IEEE division / sqrt need no operand scaling
                bool fm = v->coord_max <= 1e7 && t->coord_max <= 1e7;
                for (int k = 0; k < 9 && fm; ++k) { const double a = std::fabs(pd.F[k]); fm = std::isfinite(a) && (a == 0.0 || (a >= 1e-30 && a <= 1e30)); }
                for (double cval : {v->C.x, v->C.y, v->C.z, t->C.x, t->C.y, t->C.z}) fm = fm && std::isfinite(cval) && std::fabs(cval) <= 1e30;
                static const bool no_fast = std::getenv("L3D_NO_FASTMATH") != nullptr;   // diagnostic switch: the compiler's own expansions
                pd.flags = (fm && !no_fast) ? kPairFastMath : 0u;
                pair_baseline(v->C, t->C, pd);
            }
            slot_off += (uint64_t)pd.Ms * pd.K; row_off += pd.Ms;
            const uint32_t pi = (uint32_t)c->pairs.size();
            v->out_pairs.push_back(pi);
            if (t->index > v->index) t->in_pairs.push_back(pi);   // inverse only if tgt not yet processed (:1680)
            c->pairs.push_back(pd);
            PairCull pc{};
            if (c->use_cull && pd.Ms <= kCullMaxSegs && pd.Mt <= kCullMaxSegs && pd.Ms && pd.Mt)
                make_cull(pd.F, v->width, v->height, t->width, t->height, pc);
            pc.s_off = cs_off; pc.t_off = ct_off; pc.c_off = cc_off; pc.k_off = ~0ull;
            pc.w_item0 = w_item; pc.sorted_copy = pd.Mt >= kSortedCopyMinSegs ? 1u : 0u; w_item += (pd.Ms + kMatchRows - 1) / kMatchRows;
            if (pc.enabled && std::max(pd.Ms, pd.Mt) > kCullLdsSegs) {   // sort keys of this pair in global scratch
                uint32_t a = 64, b = 64;
                while (a < pd.Ms) a <<= 1;
                while (b < pd.Mt) b <<= 1;
                pc.k_off = ck_off; ck_off += (uint64_t)a + b;
            }
            if (pc.enabled) { cs_off += pd.Ms; ct_off += pd.Mt; cc_off += (pd.Mt + 63) / 64; }
            c->cull.push_back(pc);
            c->pair_src_cam.push_back(v->cam); c->pair_tgt_cam.push_back(tcam);
            c->pair_tests += (uint64_t)pd.Ms * pd.Mt;
            matched[v->cam].insert(tcam); matched[tcam].insert(v->cam);
        }
    c->n_slots = slot_off; c->n_rows_total = row_off;
    if (c->n_slots >= (1ull << 32) || c->pairs.size() >= (1u << 24))
        return fail(L3D_ERR_LIMIT, "slot buffer / pair list exceed the 32-bit slot and 24-bit pair indices of phase B");
    if (c->kNN > 0 && !c->pairs.empty()) {
        // bounded kNN keeps the per-row top-K tables of 64 rows in LDS (k_match.hip): the real limit of this build
        size_t n_work = 0; uint32_t maxMt = 0;
        for (auto& pd : c->pairs) { n_work += (pd.Ms + kMatchRows - 1) / kMatchRows; maxMt = std::max(maxMt, pd.Mt); }
        const uint32_t wpg = match_waves_per_group(0, c->brute, (uint32_t)std::min<size_t>(n_work, 0xFFFFFFFFu));
        auto fits = [&](uint32_t K) { return match_lds_bytes(0, K, maxMt < 65536u && K < 32768u && !c->brute, wpg, c->brute) <= 160 * 1024; };
        // beyond that, every row takes the exact replay path (k_match_tied_rows): slower per row, any kNN <= 4096
        c->knn_replay = !fits((uint32_t)c->kNN);
    } else c->knn_replay = false;
    c->cull_tot[0] = cs_off; c->cull_tot[1] = ct_off; c->cull_tot[2] = ck_off; c->cull_tot[3] = cc_off;
    c->begin_sig.swap(sig); c->begin_sig_valid = true;
    }
    c->pair_done.assign(c->pairs.size(), 0);
    c->pair_counted.assign(c->pairs.size(), 0);
    c->shard_world = 0; c->lists_ready = false; c->lists_prepared = false;
    int rc = upload_views(*c);
    if (rc) return rc;
    // the inverse-target stream of phase B: written by the match epilogue (bounded kNN) or by k_orient_all; 16-bit entries
    // when every view has fewer than 65 535 segments (L3D_INV_TGT32=1: always 32-bit, A/B switch)
    {
        uint32_t mx = 0;
        for (auto* v : c->order) mx = std::max(mx, v->M);
        c->tgt16 = (mx < 65535u && !std::getenv("L3D_INV_TGT32")) ? 1u : 0u;
    }
    if (c->kNN > 0) L3D_HIP_CHECK(c->d_inv_tgt.reserve(std::max<uint64_t>(c->n_slots, 1)));
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
    c->tm.match_kernel_ms += ev_ms(c, 4, 5);
    c->tm.cull_prepare_ms += ev_ms(c, 8, 4);
    c->tm.match_kernel_launches += c->pending_launches;
    c->timing_pending = false; c->pending_launches = 0;
}

// enqueues (no host synchronisation) the cull set-up and the pair kernel for pairs [first, first+count)
static int run_match_kernel(l3d_ctx* c, int mode, uint32_t first, uint32_t count) {
    size_t n_work = 0;
    uint32_t maxK = 0, maxM = 0, maxMt = 0;
    for (uint32_t p = first; p < first + count; ++p) n_work += (c->pairs[p].Ms + kMatchRows - 1) / kMatchRows;
    if (!n_work) return L3D_OK;
    for (uint32_t p = first; p < first + count; ++p) {
        const PairDesc& pd = c->pairs[p];
        maxK = std::max(maxK, pd.K);
        if (c->cull[p].enabled) maxM = std::max(maxM, std::max(pd.Ms, pd.Mt));
        maxMt = std::max(maxMt, pd.Mt);
    }
    const bool replay_all = mode == 0 && c->knn_replay;   // kNN beyond the LDS tables: every row through k_match_tied_rows
    if (!replay_all && match_lds_bytes(mode, maxK, false, match_waves_per_group(mode, c->brute, (uint32_t)n_work), c->brute) > 160 * 1024)
        return fail(L3D_ERR_LIMIT, "kNN too large for the LDS top-K table");
    L3D_HIP_CHECK(c->d_work.reserve(n_work));
    // the work list of these pairs is on the device already when the pair list has not changed since it was sent
    if (!(c->work_key.version == c->pairs_version && c->work_key.first == first && c->work_key.count == count &&
          c->work_key.dev == (const void*)c->d_work.p)) {
        L3D_HIP_CHECK(c->h_work.reserve(n_work));
        WorkItem* work = c->h_work.p;
        size_t w = 0;
        for (uint32_t p = first; p < first + count; ++p)
            for (uint32_t s0 = 0; s0 < c->pairs[p].Ms; s0 += kMatchRows) work[w++] = WorkItem{p, s0};
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_work.p, work, n_work * sizeof(WorkItem), hipMemcpyHostToDevice, c->stream));
        c->work_key.version = c->pairs_version; c->work_key.first = first; c->work_key.count = count;
        c->work_key.dev = c->d_work.p;
    }
    if (c->ev_on(8)) L3D_HIP_CHECK(hipEventRecord(c->ev[8], c->stream));
    CullPools pools{c->d_cull.p, c->d_src_perm.p, c->d_src_band.p, c->d_tgt_perm.p, c->d_tgt_sf.p, c->d_tgt_band.p,
                    c->d_chunk_band.p, c->d_cull_keys.p};
    pools.tgt_s4 = c->d_tgt_s4.p; pools.tgt_sd = c->d_tgt_sd.p;
    const bool keep_all_unculled = std::getenv("L3D_KEEPALL_NO_CULL") != nullptr;   // diagnostic switch (A/B; read per call)
    // (keep-all mode: the count pass takes the culled walk, the fill pass streams -- k_match.hip)
    if (c->brute || !maxM || mode == 2 || (mode == 1 && keep_all_unculled)) pools.cull = nullptr;
    else {
        static const int no_order = [] { const char* e = std::getenv("L3D_MATCH_ORDER"); return e && std::atoi(e) == 0; }();
        // longest-first launch order (k_order_items) where the launch has one to three items per wave slot (fewer: all
        // start at once; more: the tail is short against the whole) -- C1: kernel 1.08 -> 1.03 ms; C2 / C4: +6 % with it
        if (!no_order && n_work > kMatchOrderMinItems && n_work <= kMatchOrderMaxItems) {
            L3D_HIP_CHECK(c->d_item_bucket.reserve(n_work)); L3D_HIP_CHECK(c->d_item_order.reserve(n_work));
            if (!c->d_order_done.p) {            // zeroed once: the kernel re-arms it
                L3D_HIP_CHECK(c->d_order_done.reserve(1));
                L3D_HIP_CHECK(hipMemsetAsync(c->d_order_done.p, 0, 4, c->stream));
            }
            pools.item_bucket = c->d_item_bucket.p; pools.item_order = c->d_item_order.p;
            pools.order_done = c->d_order_done.p;
            pools.w_base = c->cull[first].w_item0; pools.cost_max = maxMt;
        }
        L3D_HIP_CHECK(launch_cull_prepare(c->d_views.p, c->d_pairs.p, first, count, maxM, pools, c->stream));
        L3D_HIP_CHECK(launch_order_items(c->d_pairs.p, first, count, maxMt, pools, (uint32_t)n_work, c->stream));
    }
    if (c->ev_on(4)) L3D_HIP_CHECK(hipEventRecord(c->ev[4], c->stream));
    const bool ix16 = maxMt < 65536u && maxK < 32768u;   // 16-bit indices in the kernel's LDS tables (top bit of a row's minpos: tie flag)
    // bounded kNN: the orientation filter of phase B is fused into the epilogue
    OrientFuse of{mode == 0 ? c->d_inv_tgt.p : nullptr, c->tgt16, OrientThr{c->orient_lo, c->orient_hi}, nullptr, nullptr, 0u, nullptr, nullptr};
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
                                     c->d_slots.p, c->d_row_counts.p, c->epipolar_overlap, pools, of, ix16, c->stream));
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
    const OrientFuse of{c->d_inv_tgt.p, c->tgt16, OrientThr{c->orient_lo, c->orient_hi}, nullptr, nullptr, 0u, nullptr, nullptr};
    L3D_HIP_CHECK(launch_expand_slot_idx(c->d_views.p, c->d_pairs.p, first, count, max_row_slots, c->d_slot_idx.p,
                                         c->d_slots.p, of, c->stream));
    for (uint32_t p = first; p < first + count; ++p) c->pair_done[p] = c->pair_counted[p] = 1;
    return L3D_OK;
}

// phase B: line3D.cc:745-773 for every view in ascending camID order (k_lists.hip)
static int match_finish_impl(l3d_ctx* c);
static int lists_prepare(l3d_ctx* c, int caps_mode);
static int lists_reserve(l3d_ctx* c);
static int lists_run(l3d_ctx* c, uint32_t v0, uint32_t nv, uint32_t pool0, uint32_t npools);

// The list pass of phase B for this rank's share of the views (include/l3dpp_hip.h)
static int lists_shard_impl(l3d_ctx* c, uint32_t rank, uint32_t world, int64_t view0, int64_t view1, void* slab_ptr[4],
                            uint64_t slab_bytes[4], void* full_ptr[4]) {
    if (!c || !slab_ptr || !slab_bytes || !full_ptr) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, "l3d_match_begin must precede l3d_lists_shard");
    (void)hipSetDevice(c->device);
    // every exit that is not L3D_OK closes the open call (views untranslated, context idle), as the header promises
    const int rc = [&]() -> int {
        if (world == 0 || world > kListPools || rank >= world) return fail(L3D_ERR_ARG, "rank / world out of range");
        const uint32_t V = (uint32_t)c->order.size();
        uint32_t v0, v1;
        if (view0 >= 0) {
            if (view1 < view0 || (uint64_t)view1 > V) return fail(L3D_ERR_ARG, "view range out of bounds");
            v0 = (uint32_t)view0; v1 = (uint32_t)view1;
        } else {
            // contiguous view ranges of (nearly) equal segment count, the same partition on every rank
            uint64_t G = 0;
            std::vector<uint64_t> base(V + 1, 0);
            for (uint32_t vi = 0; vi < V; ++vi) base[vi + 1] = base[vi] + c->order[vi]->M;
            G = base[V];
            auto bound = [&](uint32_t r) -> uint32_t {
                if (r >= world) return V;
                const uint64_t target = G * r / world;
                uint32_t v = 0;
                while (v < V && base[v] < target) ++v;
                return v;
            };
            v0 = bound(rank); v1 = bound(rank + 1);
        }
        // what the pass reads: the fresh slots of the views' outgoing pairs and the inverse records of their incoming
        // ones -- the pairs that touch [v0, v1) must be present (own pairs, or received: l3d_expand_slot_indices)
        for (size_t p = 0; p < c->pairs.size(); ++p) {
            const PairDesc& pd = c->pairs[p];
            const bool touches = (pd.src >= v0 && pd.src < v1) || (pd.tgt >= v0 && pd.tgt < v1);
            if (touches && !c->pair_done[p])
                return fail(L3D_ERR_STATE, "l3d_lists_shard: the slots of a pair that touches this rank's views are not present");
        }
        int r2;
        if (!c->lists_prepared) {
            r2 = lists_prepare(c, 1);
            if (r2) return r2;
            c->lists_prepared = true; c->lp_attempts = 0;
        }
        r2 = lists_reserve(c);
        if (r2) return r2;
        const uint32_t ppr = kListPools / world, pool0 = rank * ppr;
        r2 = lists_run(c, v0, v1 - v0, pool0, ppr);
        if (r2) return r2;
        c->shard_rank = rank; c->shard_v0 = v0; c->shard_v1 = v1; c->shard_pool0 = pool0; c->shard_ppr = ppr;
        c->tail_counted = false; c->tail_written = false;
        L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->shard_world = world; c->lists_ready = true;
        slab_ptr[0] = c->d_ledges.p + (size_t)pool0 * c->lp_ecap; slab_bytes[0] = (uint64_t)ppr * c->lp_ecap * sizeof(EdgeRec); full_ptr[0] = c->d_ledges.p;
        slab_ptr[1] = c->d_lhyps.p + (size_t)pool0 * c->lp_hcap; slab_bytes[1] = (uint64_t)ppr * c->lp_hcap * sizeof(HypHdr); full_ptr[1] = c->d_lhyps.p;
        slab_ptr[2] = c->d_lsegs.p + (size_t)pool0 * c->lp_scap; slab_bytes[2] = (uint64_t)ppr * c->lp_scap * sizeof(SegHdr); full_ptr[2] = c->d_lsegs.p;
        slab_ptr[3] = c->d_lzero.p + (size_t)pool0 * 16; slab_bytes[3] = (uint64_t)ppr * 16 * 4; full_ptr[3] = c->d_lzero.p;
        return L3D_OK;
    }();
    if (rc != L3D_OK) {
        const std::string why = l3d_last_error();
        abort_match(c);
        set_error(why);
    }
    return rc;
}

int l3d_lists_shard(l3d_ctx* c, uint32_t rank, uint32_t world, void* slab_ptr[4], uint64_t slab_bytes[4], void* full_ptr[4]) {
    return lists_shard_impl(c, rank, world, -1, -1, slab_ptr, slab_bytes, full_ptr);
}
int l3d_lists_shard_views(l3d_ctx* c, uint32_t rank, uint32_t world, uint32_t view0, uint32_t view1, void* slab_ptr[4],
                          uint64_t slab_bytes[4], void* full_ptr[4]) {
    return lists_shard_impl(c, rank, world, (int64_t)view0, (int64_t)view1, slab_ptr, slab_bytes, full_ptr);
}

// Partition of a matchImages call over `world` ranks (host only, no context: the plan is a function of the pair list):
// contiguous ranges of views (ascending camID order) whose OUTGOING pairs carry equal shares of the matching cost.
// Rank r matches the pairs whose source view it owns -- a contiguous range of the pair list, which is ordered by
// source view -- and runs phase B's list pass for its views.  view_bounds / pair_bounds: world + 1 entries each.
int l3d_plan_shards(uint32_t n_views, uint32_t n_pairs, const uint32_t* pair_src_view, const uint64_t* pair_cost,
                    uint32_t world, uint32_t* view_bounds, uint32_t* pair_bounds) {
    if (!world || !view_bounds || !pair_bounds || (n_pairs && (!pair_src_view || !pair_cost))) return fail(L3D_ERR_ARG, "null argument");
    std::vector<double> vcost(n_views + 1, 0.0);
    for (uint32_t p = 0; p < n_pairs; ++p) {
        if (pair_src_view[p] >= n_views || (p && pair_src_view[p] < pair_src_view[p - 1]))
            return fail(L3D_ERR_ARG, "pair list is not ordered by source view");
        vcost[pair_src_view[p]] += (double)pair_cost[p];
    }
    double total = 0.0;
    for (uint32_t v = 0; v < n_views; ++v) total += vcost[v];
    view_bounds[0] = 0;
    uint32_t v = 0; double acc = 0.0;
    for (uint32_t r = 1; r < world; ++r) {
        const double target = total * r / world;
        // the boundary whose cumulative cost is closest to the target, never behind the previous one
        while (v < n_views && std::fabs(acc + vcost[v] - target) <= std::fabs(acc - target)) { acc += vcost[v]; ++v; }
        view_bounds[r] = v;
    }
    view_bounds[world] = n_views;
    uint32_t p = 0;
    for (uint32_t r = 0; r <= world; ++r) {
        while (p < n_pairs && pair_src_view[p] < view_bounds[r]) ++p;
        pair_bounds[r] = p;
    }
    pair_bounds[world] = n_pairs;
    return L3D_OK;
}

int l3d_match_finish(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN) return fail(L3D_ERR_STATE, "l3d_match_begin must precede l3d_match_finish");
    (void)hipSetDevice(c->device);
    const int rc = match_finish_impl(c);
    if (rc == L3D_ERR_RETRY) return rc;   // sharded list pass with enlarged pools: the call stays open (l3d_lists_shard again)
    if (rc != L3D_OK) {
        // leave a defined state behind: drain every stream this call may have used, restore the views
        // (matchImages translates them, line3D.cc:436/493) and require a new l3d_match_begin
        const std::string why = l3d_last_error();
        abort_match(c);
        set_error(why);
    }
    return rc;
}

// ---- phase B (sparse form, k_lists.hip) in three host stages -------------------------------------------------------
//   lists_prepare  tables (views, outgoing / incoming pairs, CSR bases): once per matchImages
//   lists_run      the list pass for a range of views into a range of pools: zero the work arrays, inverse records of
//                  the pairs that hand matches to those views, sorted by target segment (k_pair_csr), candidates
//                  (k_lists), edges + headers (k_edges)
//   tail_run       chain sweeps, scores, filterMatches, outputs, view medians + the read-backs of the pass
// One GPU runs prepare, lists_run(all views, all pools), tail_run.  With the list pass sharded over ranks
// (l3d_lists_shard) every rank runs lists_run for ITS views into ITS pools, the pool slabs are all-gathered by the
// caller, and every rank runs tail_run on the complete records.  Everything is enqueued without host synchronisation;
// sizes are optimistic (pools sized from the slot count or from what an earlier call needed): a pass that outgrows them
// says so and is repeated with larger ones, so the one host synchronisation of matchImages is the one at its end.
static constexpr uint32_t kChainSweeps = 16;   // chain launches enqueued blindly (each one is a no-op once nothing changes;
                                               // a launch follows a dependency chain for several links, k_chain_sweep)

static ListPools list_pools(l3d_ctx* c, uint32_t pool0 = 0, uint32_t npools = kListPools) {
    ListPools lp;
    lp.cnt = c->d_lzero.p; lp.edges = c->d_ledges.p; lp.hyps = c->d_lhyps.p; lp.segs = c->d_lsegs.p;
    lp.cands = c->d_lcands.p; lp.chdrs = c->d_lchdrs.p;
    lp.ecap = c->lp_ecap; lp.hcap = c->lp_hcap; lp.scap = c->lp_scap; lp.ccap = c->lp_ccap;
    lp.flags = c->d_lzero.p + kListPools * 16;
    lp.list2 = c->d_list2.p; lp.list4 = c->d_list4.p; lp.listH = c->d_listH.p;
    lp.pool0 = pool0; lp.npools = npools;
    static const bool no_stat = std::getenv("L3D_NO_LIST_STAT") != nullptr;   // diagnostic switch (A/B of the per-list counter)
    lp.count_entries = no_stat ? 0u : 1u;
    return lp;
}

// d_med = [tot64 x 4 ([0] unused; [1] survivors | hypotheses) | median depth of each view], fin_med(V) words
static unsigned long long* tot64_of(l3d_ctx* c) { return (unsigned long long*)c->d_med; }
static float* medians_of(l3d_ctx* c) { return c->d_med + 8; }
static size_t fin_med(uint32_t V) { return ((size_t)8 + V + 3) & ~(size_t)3; }
static constexpr size_t kFinHead = (size_t)kListPools * 16 + 96;   // pool counters | flags (32) | changed (64)

// layout of the zero block d_lzero (one memset per pass): pool counters | flags (32) | changed (64) | totals and
// medians (fin_med(V): written at the end of the tail; the call's read-back is ONE copy of the block's first
// kFinHead + fin_med(V) words -- two copies cost a second ~12 us bubble on the stream) | max_score (V+1) | kept_cnt (G)
// | best_pack (G x u64, 8-byte aligned)
struct ZeroLayout { size_t flags, changed, med, max_score, kept, best, words; };
static ZeroLayout zero_layout(uint32_t V, uint32_t G);
// positive[slot]: the hypothesis of that slot has a positive score (k_chain_sweep); lives behind the zero block
static uint8_t* positive_of(l3d_ctx* c) {
    return (uint8_t*)(c->d_lzero.p + zero_layout((uint32_t)c->order.size(), c->G).words);
}
static ZeroLayout zero_layout(uint32_t V, uint32_t G) {
    ZeroLayout z;
    z.flags = (size_t)kListPools * 16; z.changed = z.flags + 32; z.med = z.changed + 64; z.max_score = z.med + fin_med(V);
    z.kept = z.max_score + ((size_t)V + 1) * 16;   // 16 replicas per view (k_lists.hip: kMaxReplicas)
    z.best = (z.kept + G + 1) & ~(size_t)1;
    z.words = z.best + 2 * (size_t)G + 2;
    return z;
}

static int lists_prepare(l3d_ctx* c, int caps_mode) {
    if (caps_mode != c->caps_mode) {   // the capacities of the other kind of call (l3d_ctx.h: caps_saved)
        c->caps_saved[c->caps_mode] = l3d_ctx::PoolCaps{c->lp_ecap, c->lp_hcap, c->lp_scap, c->lp_ccap, c->huge_cap, c->huge_skip};
        const l3d_ctx::PoolCaps& pc = c->caps_saved[caps_mode];
        c->lp_ecap = pc.e; c->lp_hcap = pc.h; c->lp_scap = pc.s; c->lp_ccap = pc.c; c->huge_cap = pc.huge; c->huge_skip = pc.huge_skip;
        c->caps_mode = caps_mode;
    }
    // a sharded list pass always runs k_lists_huge: whether a rank's views hold a list for it is not known to the other
    // ranks before the pass, and a rank that had to repeat the pass alone would leave the collectives of the others
    if (caps_mode == 1) c->huge_skip = false;
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
    L3D_HIP_CHECK(c->d_surv_off.reserve(G + 2)); L3D_HIP_CHECK(c->d_hyp_off.reserve(G + 2));
    L3D_HIP_CHECK(c->d_hyp_of_seg.reserve(G + 1));
    L3D_HIP_CHECK(c->d_cnt64.reserve(G + 2)); L3D_HIP_CHECK(c->d_off64s.reserve(G + 2));
    L3D_HIP_CHECK(c->d_scan_ws.reserve_zeroed(scan_ws_words(G, 8), st));
    L3D_HIP_CHECK(c->d_seg_of_g.reserve(G + 1)); L3D_HIP_CHECK(c->d_list2.reserve(G + 1)); L3D_HIP_CHECK(c->d_list4.reserve(G + 1)); L3D_HIP_CHECK(c->d_listH.reserve(G + 1));
    L3D_HIP_CHECK(c->d_hyps.reserve(std::max<uint32_t>(G, 1))); L3D_HIP_CHECK(c->d_depths.reserve(2 * (size_t)G + 2));
    L3D_HIP_CHECK(c->d_inv_refs.reserve(std::max<uint64_t>(c->n_slots, 1)));
    L3D_HIP_CHECK(c->d_inv_tgt.reserve(std::max<uint64_t>(c->n_slots, 1)));
    L3D_HIP_CHECK(c->h_fin.reserve(kFinHead + fin_med(V)));
    if (c->ev_on(6)) L3D_HIP_CHECK(hipEventRecord(c->ev[6], st));
    g_trace.mark("finish: reserves done");
    {   // segment -> view table: a function of the view sizes alone, kept while they (and the array) are the same
        bool sent = false;
        L3D_HIP_CHECK(upload_table(c->d_seg_base, c->h_segb, c->seg_base.data(), ((size_t)V + 1) * 4, c->up_seg_base, st, &sent));
        if (sent || c->gseg_view_for != (const void*)c->d_gseg_view.p) {
            uint32_t max_M = 0;
            for (auto* v : c->order) max_M = std::max(max_M, v->M);
            L3D_HIP_CHECK(launch_fill_gseg_view(c->d_seg_base.p, V, max_M, c->d_gseg_view.p, st));
            c->gseg_view_for = c->d_gseg_view.p;
        }
    }
    // per-view / per-pair tables of the list pass (l3d_lists.h), staged in one pinned buffer:
    // [ListView x V | OutPair x P | InPair x P_in | PairCsr x P]: outgoing pairs of a view in ascending target order,
    // incoming pairs (those that hand inverse matches over: src < tgt, line3D.cc:1680) in ascending pair index =
    // ascending source view; the per-pair CSR offsets of a view's incoming pairs are one transposed table per view in
    // d_poff (ListView::pbase, (M + 1) rows of ni columns), PairCsr tells k_pair_csr which column a pair fills
    {
        static_assert(sizeof(ListView) == 32 && sizeof(OutPair) == 32 && sizeof(InPair) == 16 && sizeof(PairCsr) == 16, "table layout");
        uint32_t n_in = 0;
        for (uint32_t vi = 0; vi < V; ++vi) n_in += (uint32_t)c->order[vi]->in_pairs.size();
        const size_t o_in = ((size_t)V + P) * 32, o_pp = o_in + (size_t)n_in * 16;
        std::vector<char> table(o_pp + ((size_t)P + 1) * 16 + 32, 0);
        ListView* hv = (ListView*)table.data();
        OutPair* hp = (OutPair*)(table.data() + (size_t)V * 32);
        InPair* hi = (InPair*)(table.data() + o_in);
        PairCsr* hpp = (PairCsr*)(table.data() + o_pp);
        for (uint32_t p = 0; p < P; ++p) hpp[p] = PairCsr{kEmpty, 0u, 0u, 0u};
        uint32_t n = 0, ni = 0;
        uint64_t poff_total = 0;
        for (uint32_t vi = 0; vi < V; ++vi) {
            ListView& lv = hv[vi];
            lv = ListView{};
            lv.seg_base = c->seg_base[vi]; lv.M = c->order[vi]->M; lv.q0 = n; lv.k = c->order[vi]->k;
            for (uint32_t p : c->order[vi]->out_pairs) {
                const PairDesc& pd = c->pairs[p];
                OutPair op{};
                op.slot_off = pd.slot_off; op.tgt = pd.tgt; op.pair = p; op.K = pd.K;
                hp[n++] = op;
            }
            lv.nq = n - lv.q0;
            lv.i0 = ni; lv.ni = (uint32_t)c->order[vi]->in_pairs.size(); lv.pbase = (uint32_t)poff_total;
            uint32_t q = 0;
            for (uint32_t p : c->order[vi]->in_pairs) {   // (built in pair order: ascending)
                const PairDesc& pd = c->pairs[p];
                InPair ip{};
                ip.rec_base = (uint32_t)pd.slot_off; ip.src = pd.src; ip.pair = p;
                hpp[p] = PairCsr{(uint32_t)poff_total, lv.ni, q++, 0u};
                hi[ni++] = ip;
            }
            poff_total += ((uint64_t)lv.M + 1) * lv.ni;
            if (poff_total >= (1ull << 32)) return fail(L3D_ERR_LIMIT, "more than 2^32 per-pair CSR offsets in phase B");
        }
        c->poff_total = (uint32_t)poff_total; c->n_in_pairs = ni;
        L3D_HIP_CHECK(c->d_poff.reserve(std::max<uint64_t>(poff_total, 1)));
        L3D_HIP_CHECK(c->d_ltab.reserve(table.size()));
        L3D_HIP_CHECK(upload_table(c->d_ltab, c->h_ltab, table.data(), o_pp + (size_t)P * 16, c->up_ltab, st));
    }
    // ---- pre-pass: orientation flags and inverse-target stream of the pairs that do not carry them yet ----
    // (bounded kNN: done by the match epilogue / the exchange expansion; what is left are the pairs of the keep-all
    // mode and pairs whose full records arrived through l3d_slots_exchanged)
    // (pairs that are not present on this rank -- a multi-GPU run keeps the pairs that touch the rank's views only --
    // are left alone: their slots are not valid)
    for (uint32_t p0 = 0; p0 < P;) {
        if (c->pair_counted[p0] || !c->pair_done[p0]) { ++p0; continue; }
        uint32_t p1 = p0;
        while (p1 < P && !c->pair_counted[p1] && c->pair_done[p1]) ++p1;
        L3D_HIP_CHECK(launch_orient_pairs(c->d_views.p, c->d_pairs.p + p0, p1 - p0, max_slots, c->d_slots.p, c->d_inv_tgt.p,
                                          c->tgt16, c->orient_lo, c->orient_hi, st));
        for (uint32_t p = p0; p < p1; ++p) c->pair_counted[p] = 1;
        p0 = p1;
    }
    if (!c->lp_ecap) {
        // L3D_POOL_SCALE (diagnostic): scales the initial record pools; a small value makes the first list passes
        // overflow, so that the regrow path (check_pass) can be exercised at any scene size
        const double scale = [] { const char* e = std::getenv("L3D_POOL_SCALE"); const double v = e ? std::atof(e) : 1.0; return v > 0.0 ? v : 1.0; }();
        const double ns = scale * (double)c->n_slots;
        // (first call of a scene: generous -- the strides are FITTED to what the scene needed afterwards, and a pass that
        // outgrows its pools is repeated: the bundled testdata, C0, with its many supporters per slot, used to repeat its
        // first list pass twice.  Four times the round-3 estimate while that stays below ~4 GiB of pools in all.)
        const double bytes1 = ns * (sizeof(EdgeRec) / 4.0 + sizeof(HypHdr) / 8.0 + sizeof(CandRec) / 2.0);
        const double gen = scale < 1.0 ? 1.0 : std::min(4.0, std::max(1.0, 4.0e9 / std::max(bytes1, 1.0)));
        c->lp_ecap = (uint32_t)std::max<double>(gen * ns / 4 / kListPools, scale < 1.0 ? 16 : 512);
        c->lp_hcap = (uint32_t)std::max<double>(gen * ns / 8 / kListPools, scale < 1.0 ? 16 : 256);
        c->lp_ccap = (uint32_t)std::max<double>(gen * ns / 2 / kListPools, scale < 1.0 ? 32 : 1024);
    }
    c->lp_scap = std::max<uint32_t>(c->lp_scap, G / kListPools + 64);   // 30-50 % of the segments have candidates; grows on demand
    if (!c->huge_cap) c->huge_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c->n_slots / 16, 1u << 20), 1u << 30);
    return L3D_OK;
}

static int lists_reserve(l3d_ctx* c) {
    g_trace.mark("lists_reserve enter");
    const uint32_t V = (uint32_t)c->order.size();
    const ZeroLayout z = zero_layout(V, c->G);
    L3D_HIP_CHECK(c->d_lzero.reserve(z.words + 2 + (c->n_slots + 3) / 4));   // zero block | positive[] (one byte per slot)
    c->d_med = (float*)(c->d_lzero.p + z.med);
    L3D_HIP_CHECK(c->d_ledges.reserve((size_t)kListPools * c->lp_ecap));
    L3D_HIP_CHECK(c->d_lhyps.reserve((size_t)kListPools * c->lp_hcap));
    L3D_HIP_CHECK(c->d_lsegs.reserve((size_t)kListPools * c->lp_scap));
    L3D_HIP_CHECK(c->d_lchdrs.reserve((size_t)kListPools * c->lp_scap));
    L3D_HIP_CHECK(c->d_lcands.reserve((size_t)kListPools * c->lp_ccap));
    L3D_HIP_CHECK(c->d_huge_f32.reserve(2 * (size_t)c->huge_cap)); L3D_HIP_CHECK(c->d_huge_u32.reserve(3 * (size_t)c->huge_cap));
    L3D_HIP_CHECK(c->d_huge_u64.reserve(c->huge_cap));
    // n_surv <= number of headers: the outputs are sized by that bound
    const size_t surv_cap = (size_t)kListPools * c->lp_hcap;
    L3D_HIP_CHECK(c->d_surv.reserve(surv_cap)); L3D_HIP_CHECK(c->d_surv_tg.reserve(surv_cap));
    L3D_HIP_CHECK(c->d_surv_sg.reserve(surv_cap));
    return L3D_OK;
}

// the list pass of the views [v0, v0 + nv) into the pools [pool0, pool0 + npools)
static int lists_run(l3d_ctx* c, uint32_t v0, uint32_t nv, uint32_t pool0, uint32_t npools) {
    g_trace.mark("lists_run enter");
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size(), P = (uint32_t)c->pairs.size(), G = c->G;
    const ZeroLayout z = zero_layout(V, G);
    const ListPools lp = list_pools(c, pool0, npools);
    const SimConst simc = sim_thresholds(c->two_sigA_sqr);
    uint64_t max_slots = 0;
    for (auto& pd : c->pairs) max_slots = std::max<uint64_t>(max_slots, (uint64_t)pd.Ms * pd.K);
    // one memset per pass: the zero block and, behind it, positive[] (seg_of_g needs none: it is only read for
    // segments with surviving hypotheses, whose header this very pass has written)
    L3D_HIP_CHECK(hipMemsetAsync(c->d_lzero.p, 0, z.words * 4 + std::max<uint64_t>(c->n_slots, 1), st));
    g_trace.mark("zero block memset enqueued");
    const ListView* lviews = (const ListView*)c->d_ltab.p;
    const OutPair* opairs = (const OutPair*)(c->d_ltab.p + (size_t)V * 32);
    const InPair* ipairs = (const InPair*)(c->d_ltab.p + ((size_t)V + P) * 32);
    const PairCsr* pair_poff = (const PairCsr*)(c->d_ltab.p + ((size_t)V + P) * 32 + (size_t)c->n_in_pairs * 16);
    {   // the inverse hypotheses of the pairs that hand matches to these views, sorted by target segment
        uint32_t max_Mt = 0;
        for (auto& pd : c->pairs) if (pd.tgt > pd.src && pd.tgt >= v0 && pd.tgt < v0 + nv) max_Mt = std::max(max_Mt, pd.Mt);
        if (max_Mt) {
            // (views beyond the LDS capacity of k_pair_csr keep their cursors in global memory: 64 dummy words per pair)
            if (max_Mt > 32768 || std::getenv("L3D_CSR_GLOBAL")) L3D_HIP_CHECK(c->d_csr_dummy.reserve((size_t)P * 64));
            L3D_HIP_CHECK(launch_pair_csr(c->d_pairs.p, P, max_Mt, pair_poff, c->d_inv_tgt.p, c->tgt16, c->d_poff.p, c->d_inv_refs.p,
                                          c->d_csr_dummy.p, v0, v0 + nv, st));
        }
    }
    g_trace.mark("pair CSRs enqueued");
    // mean list length: every alive slot is a hypothesis of its source segment and, towards a later view, of its target
    // segment too (~0.8 of the slots are alive, ~half of the pairs hand inverse matches over); exact after the first call
    const uint32_t mean_list = c->n_ents ? (uint32_t)(c->n_ents / std::max<uint32_t>(G, 1))
                                         : (uint32_t)(1.5 * (double)c->n_slots / std::max<uint32_t>(G, 1));
    const HugeScratchArgs hsa{c->d_huge_f32.p, c->d_huge_u32.p, (uint64_t*)c->d_huge_u64.p, c->huge_cap, mean_list,
                              c->huge_skip ? 0u : 1u};
    c->huge_ran = !c->huge_skip;
    uint32_t max_M = 0;
    for (uint32_t vi = v0; vi < v0 + nv; ++vi) max_M = std::max(max_M, c->order[vi]->M);
    L3D_HIP_CHECK(launch_lists(v0, nv, max_M, c->d_views.p, c->d_pairs.p, lviews, opairs, ipairs, c->d_gseg_view.p,
                               c->d_poff.p, c->d_inv_refs.p, c->d_slots.p, c->kNN > 0 ? (uint32_t)c->kNN : 0u, simc, lp,
                               c->d_seg_of_g.p, hsa, st));
    if (c->ev_on(9)) L3D_HIP_CHECK(hipEventRecord(c->ev[9], st));
    g_trace.mark("list pass enqueued");
    return L3D_OK;
}

// The tail works on a SHARD of the scene: all of it on one GPU and in the replicated tail of a multi-GPU run; the views
// [v0, v1) = segments [g0, g1) = pools [pool0, pool0 + npools) of this rank when the tail is sharded (l3d_tail_shard_*).
struct TailShard { uint32_t v0, v1, g0, g1, pool0, npools; };
static TailShard whole_tail(const l3d_ctx* c) { return TailShard{0u, (uint32_t)c->order.size(), 0u, c->G, 0u, kListPools}; }

// phase 1: the chain on ALL records (a global fixed point), then scores, filterMatches and the per-segment counts of the
// shard, scanned over its segments (tot64[1] = its surviving matches | its best hypotheses)
static int tail_count_run(l3d_ctx* c, bool fresh, const TailShard& ts) {
    g_trace.mark("tail_run enter");
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size(), G = c->G;
    const ZeroLayout z = zero_layout(V, G);
    const ListPools lp = list_pools(c), lps = list_pools(c, ts.pool0, ts.npools);
    uint32_t* changed = c->d_lzero.p + z.changed;
    uint32_t* max_score = c->d_lzero.p + z.max_score;
    uint32_t* kept = c->d_lzero.p + z.kept;
    unsigned long long* best = (unsigned long long*)(c->d_lzero.p + z.best);
    if (!fresh) L3D_HIP_CHECK(hipMemsetAsync(changed, 0, (z.words - z.changed) * 4, st));
    // as many launches as the last call needed + 1 (a launch is a no-op once nothing changes; the last one enqueued
    // must report "no change", else the host keeps sweeping)
    const uint32_t n_sweeps = std::min(kChainSweeps, std::max(2u, c->chain_need + 1));
    c->chain_enqueued = n_sweeps;
    for (uint32_t s2 = 0; s2 < n_sweeps; ++s2)
        L3D_HIP_CHECK(launch_chain_sweep(lp, positive_of(c), changed, s2, st));
    g_trace.mark("chain sweeps enqueued");
    // which pairs' slots this rank holds (sharded calls only: on one GPU every pair is present)
    const uint8_t* present = nullptr;
    if (c->shard_world > 1) {
        L3D_HIP_CHECK(c->d_pair_present.reserve(std::max<size_t>(c->pair_done.size(), 1)));
        L3D_HIP_CHECK(hipMemcpyAsync(c->d_pair_present.p, c->pair_done.data(), c->pair_done.size(), hipMemcpyHostToDevice, st));
        present = c->d_pair_present.p;
    }
    L3D_HIP_CHECK(launch_hyp_scores(lps, positive_of(c), c->d_gseg_view.p, c->d_slots.p, present, max_score, st));
    g_trace.mark("hyp_scores enqueued");
    L3D_HIP_CHECK(launch_hyp_filter(lps, ts.g0, ts.g1, c->d_gseg_view.p, max_score, kept, best, c->d_cnt64.p, st));
    g_trace.mark("hyp_filter enqueued");
    L3D_HIP_CHECK(launch_scan64(c->d_cnt64.p + ts.g0, ts.g1 - ts.g0, c->d_off64s.p + ts.g0, c->d_scan_ws.p, tot64_of(c) + 1, st));
    g_trace.mark("scan enqueued");
    return L3D_OK;
}

// phase 2: the outputs of the shard's segments at their places in the full arrays (base64: what the shards before it
// hold), the medians of its views; `publish`: the call's read-back, written by the last workgroup of k_median_all
static int tail_write_run(l3d_ctx* c, const TailShard& ts, unsigned long long base64, bool publish) {
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size(), G = c->G;
    const ZeroLayout z = zero_layout(V, G);
    const ListPools lps = list_pools(c, ts.pool0, ts.npools);
    unsigned long long* best = (unsigned long long*)(c->d_lzero.p + z.best);
    L3D_HIP_CHECK(launch_seg_write(ts.g0, ts.g1, base64, c->d_views.p, c->d_pairs.p, c->d_seg_base.p, c->d_gseg_view.p,
                                   c->d_off64s.p, best, c->d_seg_of_g.p, lps, c->d_slots.p, c->d_surv_off.p, c->d_hyp_off.p,
                                   c->d_surv.p, c->d_surv_tg.p, c->d_surv_sg.p, c->d_hyp_of_seg.p, c->d_hyps.p, c->d_depths.p, st));
    g_trace.mark("seg_write enqueued");
    // the read-back: the head of the zero block -- pool counters, flags (32), changed (64) -- and behind it [0..7] the
    // 64-bit totals, [8..8+V) the medians, written into the pinned host buffer by the last workgroup of k_median_all
    // (flags[16] counts its workgroups): no copy command on the stream
    void* h_dev = nullptr;
    L3D_HIP_CHECK(hipHostGetDevicePointer(&h_dev, c->h_fin.p, 0));
    L3D_HIP_CHECK(launch_median_all(ts.v1 - ts.v0, c->d_depths.p, c->d_hyp_off.p, c->d_seg_base.p,
                                    c->d_tie_count.p ? c->d_tie_count.p + 2 : nullptr, (uint32_t*)tot64_of(c), medians_of(c),
                                    c->d_lzero.p, publish ? (uint32_t*)h_dev : nullptr, (uint32_t)(kFinHead + fin_med(V)),
                                    c->d_lzero.p + z.flags + 16, st, ts.v0));
    g_trace.mark("median + read-back enqueued");
    return L3D_OK;
}

// chain, scores, filterMatches, outputs, medians on the complete records (`fresh`: first tail after a list pass;
// otherwise the chain continues from what earlier sweeps found and only the later stages start over)
static int tail_run(l3d_ctx* c, bool fresh) {
    const TailShard ts = whole_tail(c);
    int rc = tail_count_run(c, fresh, ts);
    if (rc) return rc;
    rc = tail_write_run(c, ts, 0ull, true);
    if (rc) return rc;
    if (c->ev_on(7)) L3D_HIP_CHECK(hipEventRecord(c->ev[7], c->stream));
    g_trace.mark("tail enqueued");
    return L3D_OK;
}

// flags of the finished pass (own, and in sharded mode those every rank published in its first pool's counters):
// L3D_OK, an error, or kRetry after the pools were enlarged
static constexpr int kRetry = 1;
static int check_pass(l3d_ctx* c) {
    const uint32_t* h = c->h_fin.p;                                       // h[...]: pool counters
    const uint32_t* hf = h + kListPools * 16;                             // flags
    uint32_t fl[8];
    for (int k = 0; k < 8; ++k) fl[k] = hf[k];
    if (c->shard_world > 1) {
        const uint32_t ppr = kListPools / c->shard_world;
        for (uint32_t r = 0; r < c->shard_world; ++r)
            for (int k = 0; k < 4; ++k) fl[k] |= h[(size_t)r * ppr * 16 + 8 + k];
        fl[6] = 0;
        for (uint32_t r = 0; r < c->shard_world; ++r) fl[6] = std::max(fl[6], h[(size_t)r * ppr * 16 + 12]);
    }
    if (fl[1]) return fail(L3D_ERR_LIMIT, "a 2D segment has more than 65535 match hypotheses");
    // lists for the global-memory kernel although its launch was left out: repeat with it (and keep it from now on)
    const bool huge_missed = fl[5] && !c->huge_ran;
    c->huge_skip = fl[5] == 0 && c->shard_world <= 1;
    if (huge_missed) { ++c->tm.pool_retries; return kRetry; }
    if (fl[3]) return fail(L3D_ERR_HIP, "internal error: hypothesis counters and slot flags disagree");
    if (fl[0] || fl[2]) {
        ++c->tm.pool_retries;
        if (++c->lp_attempts > 6) return fail(L3D_ERR_LIMIT, "phase-B pools keep overflowing");
        if (fl[0]) {   // size from what this pass asked for, with head room
            // (a pass that ran out of candidate space never reached the edges: those pools double)
            uint32_t me = 0, mh = 0, ms = 0, mc = 0;
            for (uint32_t q = 0; q < kListPools; ++q) {
                me = std::max(me, h[q * 16]); mh = std::max(mh, h[q * 16 + 1]);
                ms = std::max(ms, std::max(h[q * 16 + 2], h[q * 16 + 4])); mc = std::max(mc, h[q * 16 + 3]);
            }
            const bool cands_over = mc > c->lp_ccap || ms > c->lp_scap;
            c->lp_ccap = std::max(c->lp_ccap, mc + mc / 2 + 64); c->lp_scap = std::max(c->lp_scap, ms + ms / 2 + 64);
            c->lp_ecap = std::max(cands_over ? 2 * c->lp_ecap : c->lp_ecap, me + me / 2 + 64);
            c->lp_hcap = std::max(cands_over ? 2 * c->lp_hcap : c->lp_hcap, mh + mh / 2 + 64);
        }
        if (fl[2]) c->huge_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * c->huge_cap, fl[6] + 1024ull), 1u << 31);
        return kRetry;
    }
    return L3D_OK;
}

// results of the converged pass -> context; matchImages' epilogue (line3D.cc:493)
static int finish_commit(l3d_ctx* c) {
    const uint32_t V = (uint32_t)c->order.size();
    const uint32_t* h = c->h_fin.p;                                       // h[...]: pool counters
    const uint32_t* h0 = h + kFinHead;                                    // totals and medians
    const uint32_t* changed = h + kListPools * 16 + 32;
    c->n_surv = h0[2]; c->n_hyps = h0[3];
    c->tm.tied_rows = h0[0];   // rows replayed in the reference's priority_queue order, cumulative (k_median_all hands it over)
    {   // total length of the hypothesis lists: counted by the list pass per pool (k_lists.hip: cnt[pool * 16 + 5])
        uint64_t ents = 0;
        for (uint32_t q = 0; q < kListPools; ++q) ents += h[q * 16 + 5];
        c->n_ents = (uint32_t)std::min<uint64_t>(ents, 0xFFFFFFFFu);
        c->tm.list_entries = c->n_ents;
    }
    for (uint32_t s2 = 0; s2 < c->chain_enqueued; ++s2) c->tm.chain_sweeps += changed[s2] ? 1u : 0u;   // of the last round
    c->chain_need = c->tm.chain_extra_rounds ? kChainSweeps : c->tm.chain_sweeps;
    {
        uint64_t ne = 0;
        uint32_t me = 0, mh = 0, ms = 0, mc = 0;
        for (uint32_t q = 0; q < kListPools; ++q) {
            ne += h[q * 16];
            me = std::max(me, h[q * 16]); mh = std::max(mh, h[q * 16 + 1]);
            ms = std::max(ms, std::max(h[q * 16 + 2], h[q * 16 + 4])); mc = std::max(mc, h[q * 16 + 3]);
        }
        c->tm.support_words = (uint32_t)ne;    // supporting (hypothesis, supporter) pairs
        // Pool strides fitted to what the scene needs (the allocations stay): the record slabs a multi-GPU run
        // all-gathers are pools x stride, so a stride several times the fullest pool's count is traffic for nothing.
        // Every rank sees every counter, so every rank takes the same decision.  (A later call that needs more grows
        // them again through the retry path.)
        auto fit = [](uint32_t cap, uint32_t need, uint32_t lowest) {
            const uint32_t want = std::max(need + need / 4 + 64, lowest);
            return cap > want + want / 2 ? want : cap;
        };
        c->lp_ecap = fit(c->lp_ecap, me, 256); c->lp_hcap = fit(c->lp_hcap, mh, 128);
        c->lp_scap = fit(c->lp_scap, ms, 64); c->lp_ccap = fit(c->lp_ccap, mc, 512);
    }
    // View::update_median_depth for every view (line3D.cc:1665-1668); in fixed-regulariser mode k is
    // re-set to the same sigma_p/med_scene_depth value, so k is unchanged either way
    for (uint32_t vi = 0; vi < V; ++vi) c->order[vi]->median_depth = ((const float*)h0)[8 + vi];
    c->host_offsets_valid = false;
    if (c->timing_pending) {   // phase A ran unsynchronised (l3d_match_images)
        collect_match_timing(c);
        c->tm.match_pairs_ms += ev_ms(c, 2, 3);
    }
    c->tm.finish_ms = ev_ms(c, 6, 7);
    c->tm.lists_ms = ev_ms(c, 6, 9);
    c->tm.record_kbytes = (uint32_t)(((uint64_t)kListPools * ((uint64_t)c->lp_ecap * sizeof(EdgeRec) + (uint64_t)c->lp_hcap * sizeof(HypHdr) +
                                                              (uint64_t)c->lp_scap * sizeof(SegHdr) + 64)) >> 10);
    c->tm.begin_ms = ev_ms(c, 0, 1);
    untranslate(*c);   // line3D.cc:493
    c->state = l3d_ctx::MATCHED;
    c->shard_world = 0; c->lists_ready = false; c->lists_prepared = false;
    return L3D_OK;
}

// the tail until the chain has converged (the blind sweeps normally suffice; any chain depth is handled)
static int tail_until_converged(l3d_ctx* c) {
    hipStream_t st = c->stream;
    int rc = tail_run(c, true);
    if (rc) return rc;
    g_trace.mark("phase B enqueued, waiting");
    L3D_HIP_CHECK(hipStreamSynchronize(st));   // the one point at which matchImages waits for the GPU
    g_trace.mark("phase B done");
    rc = check_pass(c);
    if (rc) return rc;
    while (c->h_fin.p[kListPools * 16 + 32 + c->chain_enqueued - 1]) {
        rc = tail_run(c, false);
        if (rc) return rc;
        L3D_HIP_CHECK(hipStreamSynchronize(st));
        ++c->tm.chain_extra_rounds;
    }
    return L3D_OK;
}

static int match_finish_impl(l3d_ctx* c) {
    int rc;
    if (c->lists_ready) {
        // the list pass ran sharded (l3d_lists_shard) and the caller has all-gathered the pool slabs: index the segment
        // headers of all ranks, then the tail on the complete records
        const ListPools lp = list_pools(c);
        L3D_HIP_CHECK(launch_seg_index(lp, c->d_seg_of_g.p, c->G, c->shard_world, c->stream));
        rc = tail_until_converged(c);
        if (rc == kRetry) { c->lists_ready = false; return fail(L3D_ERR_RETRY, "phase-B pools enlarged: repeat l3d_lists_shard and the exchange"); }
        if (rc) return rc;
        return finish_commit(c);
    }
    rc = lists_prepare(c, 0);
    if (rc) return rc;
    c->lp_attempts = 0;
    const uint32_t V = (uint32_t)c->order.size();
    for (;;) {
        rc = lists_reserve(c);
        if (rc) return rc;
        rc = lists_run(c, 0, V, 0, kListPools);
        if (rc) return rc;
        rc = tail_until_converged(c);
        if (rc == kRetry) continue;
        if (rc) return rc;
        break;
    }
    return finish_commit(c);
}

// ---- the tail of phase B sharded by views (N > 1 ranks) ---------------------------------------------------------------
// After l3d_lists_shard* and the exchange of the record slabs every rank holds ALL records.  The chain is a global fixed
// point over them and is run by every rank; scores, filterMatches, the outputs and the medians are per view and are
// computed by the rank that owns the view:
//   l3d_tail_shard_count   chain + scores + filterMatches + counts of this rank's views  -> its two counts
//   (the caller all-gathers the counts)
//   l3d_tail_shard_layout  this rank's outputs, written at their places in the full arrays; where every rank's parts are
//   (the caller exchanges the parts, in place)
//   l3d_tail_shard_commit  medians of all views to the host, totals: the call is closed like l3d_match_finish closes it
static int tail_count_until_converged(l3d_ctx* c, const TailShard& ts) {
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size();
    auto run = [&](bool fresh) -> int {
        const int rc = tail_count_run(c, fresh, ts);
        if (rc) return rc;
        L3D_HIP_CHECK(hipMemcpyAsync(c->h_fin.p, c->d_lzero.p, (kFinHead + fin_med(V)) * 4, hipMemcpyDeviceToHost, st));
        L3D_HIP_CHECK(hipStreamSynchronize(st));
        return L3D_OK;
    };
    int rc = run(true);
    if (rc) return rc;
    rc = check_pass(c);
    if (rc) return rc;
    while (c->h_fin.p[kListPools * 16 + 32 + c->chain_enqueued - 1]) {
        rc = run(false);
        if (rc) return rc;
        ++c->tm.chain_extra_rounds;
    }
    return L3D_OK;
}

static int close_failed_call(l3d_ctx* c, int rc) {   // as l3d_match_finish: a defined state, the error text kept
    if (rc != L3D_OK && rc != L3D_ERR_RETRY) {
        const std::string why = l3d_last_error();
        abort_match(c);
        set_error(why);
    }
    return rc;
}

int l3d_tail_shard_count(l3d_ctx* c, uint32_t counts[2]) {
    if (!c || !counts) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN || !c->lists_ready || c->shard_world < 2)
        return fail(L3D_ERR_STATE, "l3d_tail_shard_count follows l3d_lists_shard* and the exchange of its slabs (world > 1)");
    (void)hipSetDevice(c->device);
    const int rc = [&]() -> int {
        const ListPools lp = list_pools(c);
        L3D_HIP_CHECK(launch_seg_index(lp, c->d_seg_of_g.p, c->G, c->shard_world, c->stream));
        const TailShard ts{c->shard_v0, c->shard_v1, c->seg_base[c->shard_v0], c->seg_base[c->shard_v1], c->shard_pool0, c->shard_ppr};
        const int r = tail_count_until_converged(c, ts);
        if (r == kRetry) { c->lists_ready = false; return fail(L3D_ERR_RETRY, "phase-B pools enlarged: repeat l3d_lists_shard and the exchange"); }
        if (r) return r;
        const uint32_t* med = c->h_fin.p + kFinHead;
        counts[0] = med[2]; counts[1] = med[3];
        c->tail_counted = true; c->tail_written = false;
        return L3D_OK;
    }();
    return close_failed_call(c, rc);
}

int l3d_tail_shard_layout(l3d_ctx* c, uint32_t world, const uint32_t* counts_all, const uint32_t* view_bounds, void* base_ptr[9],
                          uint64_t elt_bytes[9], uint64_t* first, uint64_t* count) {
    if (!c || !counts_all || !view_bounds || !base_ptr || !elt_bytes || !first || !count) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN || !c->tail_counted || world != c->shard_world)
        return fail(L3D_ERR_STATE, "l3d_tail_shard_layout follows l3d_tail_shard_count (same world size)");
    (void)hipSetDevice(c->device);
    const int rc = [&]() -> int {
        const uint32_t V = (uint32_t)c->order.size();
        if (view_bounds[0] != 0 || view_bounds[world] != V || view_bounds[c->shard_rank] != c->shard_v0 ||
            view_bounds[c->shard_rank + 1] != c->shard_v1)
            return fail(L3D_ERR_ARG, "view bounds do not match the list pass of this rank");
        c->tail_base_n.assign(world + 1, 0); c->tail_base_h.assign(world + 1, 0);
        uint64_t n = 0, h = 0;
        for (uint32_t r = 0; r < world; ++r) {
            if (view_bounds[r + 1] < view_bounds[r]) return fail(L3D_ERR_ARG, "view bounds are not ascending");
            c->tail_base_n[r] = (uint32_t)n; c->tail_base_h[r] = (uint32_t)h;
            n += counts_all[2 * r]; h += counts_all[2 * r + 1];
        }
        if (n >= (1ull << 32) || n > (uint64_t)kListPools * c->lp_hcap || h > c->G)
            return fail(L3D_ERR_LIMIT, "the ranks' counts exceed the output arrays");
        c->tail_base_n[world] = (uint32_t)n; c->tail_base_h[world] = (uint32_t)h;
        const uint32_t me = c->shard_rank;
        const TailShard ts{c->shard_v0, c->shard_v1, c->seg_base[c->shard_v0], c->seg_base[c->shard_v1], c->shard_pool0, c->shard_ppr};
        const int r2 = tail_write_run(c, ts, (unsigned long long)c->tail_base_n[me] | ((unsigned long long)c->tail_base_h[me] << 32), false);
        if (r2) return r2;
        void* bp[9] = {c->d_surv.p, c->d_surv_tg.p, c->d_surv_sg.p, c->d_hyps.p, c->d_depths.p, c->d_surv_off.p, c->d_hyp_off.p,
                       c->d_hyp_of_seg.p, medians_of(c)};
        const uint64_t eb[9] = {sizeof(Match), 4, 4, sizeof(HypRec), 8, 4, 4, 4, 4};
        for (int k = 0; k < 9; ++k) { base_ptr[k] = bp[k]; elt_bytes[k] = eb[k]; }
        for (uint32_t r = 0; r < world; ++r) {
            const uint64_t g0 = c->seg_base[view_bounds[r]], g1 = c->seg_base[view_bounds[r + 1]], end = r + 1 == world ? 1 : 0;
            const uint64_t f[9] = {c->tail_base_n[r], c->tail_base_n[r], c->tail_base_n[r], c->tail_base_h[r], c->tail_base_h[r],
                                   g0, g0, g0, view_bounds[r]};
            const uint64_t m[9] = {counts_all[2 * r], counts_all[2 * r], counts_all[2 * r], counts_all[2 * r + 1], counts_all[2 * r + 1],
                                   g1 - g0 + end, g1 - g0 + end, g1 - g0, (uint64_t)view_bounds[r + 1] - view_bounds[r]};
            for (int k = 0; k < 9; ++k) { first[9 * r + k] = f[k]; count[9 * r + k] = m[k]; }
        }
        c->tail_written = true;
        return L3D_OK;
    }();
    return close_failed_call(c, rc);
}

int l3d_tail_shard_commit(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN || !c->tail_written)
        return fail(L3D_ERR_STATE, "l3d_tail_shard_commit follows l3d_tail_shard_layout and the exchange of the parts");
    (void)hipSetDevice(c->device);
    const int rc = [&]() -> int {
        const uint32_t V = (uint32_t)c->order.size(), world = c->shard_world;
        uint32_t* med = c->h_fin.p + kFinHead;
        if (c->ev_on(7)) L3D_HIP_CHECK(hipEventRecord(c->ev[7], c->stream));
        // totals' words and the medians of ALL views (the other ranks' have arrived with the exchange)
        L3D_HIP_CHECK(hipMemcpyAsync(med, c->d_med, (8 + (size_t)V) * 4, hipMemcpyDeviceToHost, c->stream));
        L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
        med[2] = c->tail_base_n[world]; med[3] = c->tail_base_h[world];
        c->tail_counted = false; c->tail_written = false;
        return finish_commit(c);
    }();
    return close_failed_call(c, rc);
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
