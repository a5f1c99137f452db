// k_views.hip -- what is left of the dense form of phase B (round 1), and small per-view passes.
//
// matchImages runs phase B in its sparse form (k_lists.hip).  The dense kernels that remain here serve the seam entry
// l3d_score_matches (the replacement of score_matches_GPU, cudawrapper.h:70-73: one view, the caller's match list, every
// match present) and the paths that do not come out of the match epilogue:
//
//   k_orient_all      checkMatchOrientation (line3D.cc:811-858) + packed hypothesis counters for slots that did not get
//                     them from the match kernel: keep-all mode, full records received from another rank
//   k_seam_entries / k_bits_len / k_support<WPL> / k_seam_all_present / k_score_all / k_seam_scores_out
//                     one view's hypothesis lists with the spatial regularisers scoring needs (scoringCPU :1233-1248),
//                     support bitsets from the similarityForScoring decisions inside each hypothesis' depth window,
//                     scores with the reference's per-camera replace/subtract accumulation (:1255-1274)
//   k_median_all      per-view median depth (View::update_median_depth, view.h:108-121): radix select
//   k_fill_gseg_view  global segment id -> view
#include "l3d_dev.h"
#include "l3d_kernels.h"

namespace l3d {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

}  // namespace

// diagnostics build only (-DL3D_STATS): 0 lists scored, 1 hypotheses, 2 present hypotheses, 3 present (hypothesis,
// supporter) pairs, 4 hypotheses with at least one present supporter, 5 longest pair sequence of a list, 6 set bits of
// all support rows
#ifdef L3D_STATS
__device__ unsigned long long g_bstats[8];
#define L3D_BSTAT(i, n) atomicAdd(&g_bstats[i], (unsigned long long)(n))
#define L3D_BSTAT_MAX(i, n) atomicMax(&g_bstats[i], (unsigned long long)(n))
#else
#define L3D_BSTAT(i, n) ((void)0)
#define L3D_BSTAT_MAX(i, n) ((void)0)
#endif

// ---- pre-pass ---------------------------------------------------------------------------------------
// grid = (slot blocks, pairs).  One thread per slot: orientation flags (checkMatchOrientation, line3D.cc:811-858) of
// slots that do not carry them yet -- the keep-all mode and full records that arrived from another rank; the bounded-kNN
// match epilogue does the same on the way -- and the 4-byte stream k_pair_csr sorts:
//   inv_tgt[slot] target segment of a slot that hands an inverse match to its target view (kEmpty: none)
__global__ void k_orient_all(const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                             Slot* __restrict__ slots, uint32_t* __restrict__ inv_tgt, uint32_t tgt16,
                             float2* __restrict__ hyp_p, float2* __restrict__ hyp_q, OrientThr othr) {
    const PairDesc& pd = pairs[blockIdx.y];
    const uint64_t n = (uint64_t)pd.Ms * pd.K;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t row = (uint32_t)(i / pd.K);
    uint32_t itgt = kEmpty;
    Slot* sp = slots + pd.slot_off + i;
    const Slot s = *sp;
    uint32_t flags = 0;
    if (s.tgt_seg != kEmpty) {
        const ViewDev& vs = views[pd.src];
        if (orientation_ok_fast(vs.C, vs.segx[row], s.dp1, s.dp2, othr)) {
            flags = kSlotAlive;
            // inverse copy: only towards a view that is processed later (line3D.cc:1680)
            if (pd.tgt > pd.src) {
                const ViewDev& vt = views[pd.tgt];
                if (orientation_ok_fast(vt.C, vt.segx[s.tgt_seg], s.dq1, s.dq2, othr)) { flags |= kSlotInvAlive; itgt = s.tgt_seg; }
            }
        }
        sp->flags = flags;
        sp->score3D = 0.0f;
    }
    if (tgt16) ((uint16_t*)inv_tgt)[pd.slot_off + i] = (uint16_t)(itgt == kEmpty ? 0xFFFFu : itgt);
    else inv_tgt[pd.slot_off + i] = itgt;
    // the hypothesis streams the list pass reads (l3d_kernels.h: OrientFuse)
    const float nan = __builtin_nanf("");
    hyp_p[pd.slot_off + i] = (flags & kSlotAlive) ? make_float2(s.dp1, s.dp2) : make_float2(nan, nan);
    hyp_q[pd.slot_off + i] = make_float2(s.dq1, s.dq2);
}



// scoringCPU line3D.cc:1233-1248: unprojection + spatial regularisers of one hypothesis
__device__ __forceinline__ DEntry make_dentry(const ViewDev& v, const ViewDev& vt, const SegX& sx, float dp1,
                                              float dp2, uint32_t ref, uint32_t tgt_view, uint32_t pair,
                                              bool inverse) {
    const Seg3 s3 = unproject(v.C, sx.r1, sx.r2, dp1, dp2);
    const float k = v.k;
    const float sig1 = dp1 * k, sig2 = dp2 * k;
    float reg1 = 2.0f * sig1 * sig1, reg2 = 2.0f * sig2 * sig2;
    const d3 ct{vt.C[0], vt.C[1], vt.C[2]};
    const float sig1_t = (float)(norm(s3.P1 - ct) * (double)vt.k);   // View::regularizerFrom3Dpoint
    const float sig2_t = (float)(norm(s3.P2 - ct) * (double)vt.k);
    reg1 = 0.5f * (reg1 + 2.0f * sig1_t * sig1_t);
    reg2 = 0.5f * (reg2 + 2.0f * sig2_t * sig2_t);
    DEntry d;
    d.ref = ref;
    d.dp1 = dp1; d.dp2 = dp2; d.reg1 = reg1; d.reg2 = reg2;
    d.score3D = 0.0f;
    d.tgt_view = tgt_view;
    d.flags = (inverse ? kDInverse : 0u) | (s3.length < kEps ? kDZeroLen : 0u);
    d.pair = pair;
    return d;
}

// Segment3D::dir_ of a hypothesis of the segment with rays sx (view centre C): recomputed instead of stored
__device__ __forceinline__ d3 entry_dir(const double* C, const SegX& sx, float dp1, float dp2) {
    return unproject(C, sx.r1, sx.r2, dp1, dp2).dir;
}

// ---- similarity of two hypotheses of one 2D segment ------------------------------------------------
// similarityForScoring (line3D.cc:1417-1446) for hypotheses a (the scored one) and b of the same 2D
// segment.  Decisions are taken on the float quantities the reference compares; acos/exp are
// evaluated in double and rounded to float (glibc's expf/acos differ from that by < 1 float ulp).
// Decision form (k_support): "similarityForScoring > L3D_DEF_MIN_SIMILARITY_3D" without transcendentals.
// sim = fmin(sim_a, fmin(e1, e2)) > 0.5 <=> every non-NaN component is > 0.5 (fmin skips NaNs; all NaN -> false).
// expf and acos are monotone, so each component test is a comparison of its float argument with a threshold
// the host found by bisection with the libm the reference itself would use (sim_thresholds, l3d_api.hip):
//     expf(y) > 0.5f                      <=>  y > y_thr
//     expf(-angle(x)^2 / two_sigA_sqr) > 0.5f  <=>  x >= x_hi || x <= x_lo     (x = clamped float dot product)
__device__ __forceinline__ bool sim_decide(const d3 dira, bool zeroa, float adp1, float adp2, float reg1, float reg2,
                                           const d3 dirb, bool zerob, float bdp1, float bdp2, const SimConst sc) {
    if (zeroa || zerob) return false;
    const float d1 = adp1 - bdp1, d2 = adp2 - bdp2;
    // division-free early-out with a safety margin: d*d > 0.72*reg => y < -0.70 < y_thr even after rounding;
    // NaN/inf fall through to the exact comparison
    if (d1 * d1 > 0.72f * reg1 || d2 * d2 > 0.72f * reg2) return false;
    const float y1 = -d1 * d1 / reg1, y2 = -d2 * d2 / reg2;
    // NaN components drop out of the fmin chain; the angular component is never NaN (x is clamped, sigma_a > 0)
    if (y1 == y1 && !(y1 > sc.y_thr)) return false;
    if (y2 == y2 && !(y2 > sc.y_thr)) return false;
    const float dot_p = (float)dot(dira, dirb);
    const float x = fmaxf(fminf(dot_p, 1.0f), -1.0f);
    return x >= sc.x_hi || x <= sc.x_lo;
}

// Value form (k_score_all), for pairs sim_decide accepted: similarityForScoring (line3D.cc:1417-1446) without the
// final threshold.  acos/exp are evaluated in double and rounded to float (glibc's expf/acos differ from that by
// < 1 float ulp).
__device__ __forceinline__ float sim_value(const d3 dira, float adp1, float adp2, float reg1, float reg2,
                                           const d3 dirb, float bdp1, float bdp2, const SimConst sc) {
    const float d1 = adp1 - bdp1, d2 = adp2 - bdp2;
    const float y1 = -d1 * d1 / reg1, y2 = -d2 * d2 / reg2;
    const float dot_p = (float)dot(dira, dirb);
    float angle = (float)(acos((double)fmaxf(fminf(dot_p, 1.0f), -1.0f)) / M_PI * 180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    const float ya = -angle * angle / sc.two_sigA_sqr;
    // fmin(expf(ya), fmin(expf(y1), expf(y2))) == expf(fmin(ya, fmin(y1, y2))): expf is monotone and fmin skips
    // NaNs on both sides alike -- one exponential instead of three
    return (float)exp((double)fminf(ya, fminf(y1, y2)));
}

// ---- support bitsets (batched), presence propagation (the chain), scores (batched) ---------------------
// For hypothesis i of a segment with L potential hypotheses, S_i = { j : cam(j) != cam(i) and
// similarityForScoring(i, j) > 0 } is a row of W = ceil(L/64) 64-bit words.  S_i does not depend on the chain.
// Layout per global segment g: bits[boff[g] + i*W + w], i in [0, L]; row L is the presence mask P.



// The similarity decisions of the hypotheses of one 2D segment (any view) -> support bitset rows.
// similarityForScoring(i, j) can only exceed 0.5 if |dp1_i - dp1_j| <= sqrt(0.72 * reg1_i) (first early-out of
// sim_decide), so the hypotheses are rank-sorted by dp1 in LDS and hypothesis i evaluates only the contiguous
// window of candidates inside that radius (found by binary search) instead of all L: O(L^2) cheap compares for
// the sort + O(L * window) evaluations.  Everything the decision reads is staged in LDS IN SORTED ORDER (the
// window walk reads consecutive positions: no dependent index load), directions as fp32 -- the angular test is
// decided in fp32 when it is at least kDirSlack away from both thresholds and re-done in fp64 (directions
// recomputed from the depths) otherwise, so the result is the fp64 one.
//   WPL = 1: one wave per list, lists up to kStageCap (the common case; a 4-wave workgroup handles 4 lists)
//   WPL = 4: one 4-wave workgroup per list for the long lists (compacted by k_bits_len): the same code with a
//            4x LDS pool (staged up to 4*kStageCap, sort-only beyond that, all-pairs for the rest) and
//            __syncthreads instead of wave barriers
constexpr uint32_t kStageCap = 192;
constexpr uint32_t kPoolFloats = 1632;                    // per-wave LDS pool: 8.5 floats per staged hypothesis
constexpr float kDirSlack = 2e-6f;

// words per segment: (L+1) * ceil(L/64); lists longer than kStageCap are appended to long_list
__global__ void k_bits_len(uint32_t G, const uint32_t* __restrict__ off, uint32_t* __restrict__ len,
                           uint32_t* __restrict__ long_list, uint32_t* __restrict__ n_long) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint32_t L = off[g + 1] - off[g];
    len[g] = L ? (L + 1) * ((L + 63) / 64) : 0u;
    if (L > kStageCap) long_list[atomicAdd(n_long, 1u)] = g;
}

template <int WPL>
__device__ __forceinline__ void group_barrier() {
    if (WPL == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

template <int WPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
void k_support(uint32_t G, const uint32_t* __restrict__ off, const uint32_t* __restrict__ boff,
               const DEntry* __restrict__ dents, uint64_t* __restrict__ bits, const ViewDev* __restrict__ views,
               const uint32_t* __restrict__ gseg_view, SimConst sc, uint32_t g0,
               const uint32_t* __restrict__ long_list) {
    __shared__ __attribute__((aligned(16))) float s_pool[4 * kPoolFloats];
    constexpr uint32_t GS = 64 * WPL;                                   // threads per list
    constexpr uint32_t kStage = kStageCap * WPL;                        // staged capacity of the group's pool
    constexpr uint32_t kSort = (kPoolFloats * WPL * 2 / 7) & ~1u;       // sort-only capacity (3.5 floats per entry)
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t t = WPL == 1 ? lane_id() : threadIdx.x;              // thread index within the group
    uint32_t g;
    if (WPL == 1) { g = g0 + blockIdx.x * 4 + wave; if (g >= G) return; }
    else g = long_list[blockIdx.x];
    const uint32_t b = off[g], L = off[g + 1] - b;
    if (L == 0) return;
    if (WPL == 1 && L > kStageCap) return;                              // long list: k_support<4>
    const uint32_t W = (L + 63) / 64;
    uint64_t* rows = bits + boff[g];
    const ViewDev& v = views[gseg_view[g]];
    const SegX sx = views[0].segx[g];   // global array (upload_views): views[0].segx is its base
    if (L <= kSort) {
        const bool staged = L <= kStage;
        const uint32_t cap = staged ? kStage : kSort;
        float* pool = s_pool + (WPL == 1 ? wave * kPoolFloats : 0);
        // sort key of hypothesis i: order-preserving bits of dp1 in the high word, i in the low word -- one 64-bit
        // compare per pair gives the rank with the (dp1, index) tie-break
        uint64_t* s_key = (uint64_t*)pool;                 // canonical order
        float* s_sorted = (float*)(s_key + cap);           // dp1, sorted
        uint16_t* s_sidx = (uint16_t*)(s_sorted + cap);    // canonical index of a sorted position
        float* s_dp2 = (float*)(s_sidx + cap);             // staged only, sorted order
        uint32_t* s_tvf = (uint32_t*)(s_dp2 + cap);        // tgt_view | zero-length flag << 31
        float* s_dir = (float*)(s_tvf + cap);              // 3 floats per hypothesis
        for (uint32_t i = t; i < L; i += GS) {
            const uint32_t u = __float_as_uint(dents[b + i].dp1);
            const uint32_t o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            s_key[i] = ((uint64_t)o << 32) | i;
        }
        group_barrier<WPL>();
        for (uint32_t i = t; i < L; i += GS) {
            const uint64_t k = s_key[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < L; ++j) rank += (s_key[j] < k) ? 1u : 0u;
            const uint32_t ko = (uint32_t)(k >> 32);
            s_sorted[rank] = __uint_as_float((ko & 0x80000000u) ? (ko & 0x7FFFFFFFu) : ~ko);
            s_sidx[rank] = (uint16_t)i;
            if (staged) {
                const DEntry e = dents[b + i];
                const d3 ed = entry_dir(v.C, sx, e.dp1, e.dp2);
                s_dp2[rank] = e.dp2;
                s_tvf[rank] = e.tgt_view | ((e.flags & kDZeroLen) ? 0x80000000u : 0u);
                s_dir[3 * rank] = (float)ed.x; s_dir[3 * rank + 1] = (float)ed.y; s_dir[3 * rank + 2] = (float)ed.z;
            }
        }
        group_barrier<WPL>();
        constexpr uint32_t kRowRegs = (kStageCap + 63) / 64;
        if (staged) {
            // each thread takes a SORTED position p0: everything about its hypothesis except the two regularisers is
            // already in LDS, and its candidates are its neighbours in the sorted order -- walk outwards from p0 in
            // both directions while |dp1 - dp1_p0| <= r (no binary search, no second pass over the entries)
            for (uint32_t p0 = t; p0 < L; p0 += GS) {
                const uint32_t i = s_sidx[p0];
                const float a_reg1 = dents[b + i].reg1, a_reg2 = dents[b + i].reg2;
                const float a_dp1 = s_sorted[p0], a_dp2 = s_dp2[p0];
                const uint32_t a_tvf = s_tvf[p0];
                const float adx = s_dir[3 * p0], ady = s_dir[3 * p0 + 1], adz = s_dir[3 * p0 + 2];
                float r = sqrtf(0.72f * a_reg1) * 1.0001f + 1e-30f;   // padded against float rounding
                if (!(r < 1e30f)) r = __builtin_inff();               // NaN / huge: the whole list is the window
                const float kl = a_dp1 - r, kh = a_dp1 + r;
                uint64_t rw[kRowRegs] = {};
                const bool in_regs = WPL == 1;
                if (!in_regs) for (uint32_t w = 0; w < W; ++w) rows[(size_t)i * W + w] = 0ull;
                auto visit = [&](uint32_t p) {
                    const uint32_t tvf = s_tvf[p];
                    if (((tvf ^ a_tvf) & 0x7FFFFFFFu) == 0 || ((tvf | a_tvf) >> 31)) return;   // same camera / zero length
                    const float odp1 = s_sorted[p], odp2 = s_dp2[p];
                    const float d1 = a_dp1 - odp1, d2 = a_dp2 - odp2;
                    if (d1 * d1 > 0.72f * a_reg1 || d2 * d2 > 0.72f * a_reg2) return;   // see sim_decide
                    const float y1 = -d1 * d1 / a_reg1, y2 = -d2 * d2 / a_reg2;
                    if (y1 == y1 && !(y1 > sc.y_thr)) return;
                    if (y2 == y2 && !(y2 > sc.y_thr)) return;
                    // angular part: fp32 dot product, exact fp64 redo when it is near a threshold
                    const float xf = fmaxf(fminf(adx * s_dir[3 * p] + ady * s_dir[3 * p + 1] + adz * s_dir[3 * p + 2], 1.0f), -1.0f);
                    bool ok;
                    if (fabsf(xf - sc.x_hi) > kDirSlack && fabsf(xf - sc.x_lo) > kDirSlack) {
                        ok = xf >= sc.x_hi || xf <= sc.x_lo;
                    } else {
                        const float dot_p = (float)dot(entry_dir(v.C, sx, a_dp1, a_dp2), entry_dir(v.C, sx, odp1, odp2));
                        const float x = fmaxf(fminf(dot_p, 1.0f), -1.0f);
                        ok = x >= sc.x_hi || x <= sc.x_lo;
                    }
                    if (!ok) return;
                    const uint32_t j = s_sidx[p];
                    if (in_regs) {
#pragma unroll
                        for (uint32_t w = 0; w < kRowRegs; ++w) rw[w] |= (w == (j >> 6)) ? (1ull << (j & 63)) : 0ull;
                    } else {
                        rows[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63);
                    }
                };
                for (uint32_t p = p0; p-- > 0;) { if (!(s_sorted[p] >= kl)) break; visit(p); }
                for (uint32_t p = p0 + 1; p < L; ++p) { if (!(s_sorted[p] <= kh)) break; visit(p); }
                if (in_regs) {
#pragma unroll
                    for (uint32_t w = 0; w < kRowRegs; ++w) if (w < W) rows[(size_t)i * W + w] = rw[w];
                }
            }
            return;
        }
        // sort-only lists: candidates are read from global memory
        for (uint32_t i = t; i < L; i += GS) {
            const DEntry a = dents[b + i];
            const d3 ad = entry_dir(v.C, sx, a.dp1, a.dp2);
            const bool azero = (a.flags & kDZeroLen) != 0;
            float r = sqrtf(0.72f * a.reg1) * 1.0001f + 1e-30f;
            uint32_t lo = 0, hi = L;
            if (r < 1e30f) {   // also false for NaN: then the whole list is the window
                const float kl = a.dp1 - r, kh = a.dp1 + r;
                uint32_t x = 0, y = L;          // first position with key >= kl
                while (x < y) { const uint32_t m = (x + y) >> 1; if (s_sorted[m] < kl) x = m + 1; else y = m; }
                lo = x;
                y = L;                          // first position with key > kh
                while (x < y) { const uint32_t m = (x + y) >> 1; if (s_sorted[m] <= kh) x = m + 1; else y = m; }
                hi = x;
            }
            for (uint32_t w = 0; w < W; ++w) rows[(size_t)i * W + w] = 0ull;
            for (uint32_t p = lo; p < hi; ++p) {
                const uint32_t j = s_sidx[p];
                const DEntry& o = dents[b + j];
                if (o.tgt_view == a.tgt_view) continue;
                if (sim_decide(ad, azero, a.dp1, a.dp2, a.reg1, a.reg2, entry_dir(v.C, sx, o.dp1, o.dp2),
                               (o.flags & kDZeroLen) != 0, o.dp1, o.dp2, sc))
                    rows[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63);
            }
        }
        return;
    }
    // ---- all-pairs loop for lists beyond the sort capacity ----
    for (uint32_t i = t; i < L; i += GS) {
        const DEntry a = dents[b + i];
        const d3 ad = entry_dir(v.C, sx, a.dp1, a.dp2);
        for (uint32_t w = 0; w < W; ++w) {
            uint64_t word = 0;
            const uint32_t jn = min(64u, L - w * 64);
            for (uint32_t jj = 0; jj < jn; ++jj) {
                const DEntry& o = dents[b + w * 64 + jj];
                if (o.tgt_view == a.tgt_view) continue;
                word |= (uint64_t)sim_decide(ad, (a.flags & kDZeroLen) != 0, a.dp1, a.dp2, a.reg1, a.reg2,
                                             entry_dir(v.C, sx, o.dp1, o.dp2), (o.flags & kDZeroLen) != 0, o.dp1,
                                             o.dp2, sc) << jj;
            }
            rows[(size_t)i * W + w] = word;
        }
    }
}

// THE CHAIN: one launch per view in ascending camID order, one wave per 2D segment of the view.
//   presence: a fresh hypothesis always exists; an inverse one exists iff the source view found its match
//             supported (score3D > 0  <=>  some existing hypothesis of another camera has similarity > 0.5),
//             which that view's launch recorded in positive[slot] (line3D.cc:1680)
//   support:  fresh hypothesis i is positive iff S_i intersects the presence mask

constexpr uint32_t kScoreCap = 192;   // per-wave LDS staging of a list: 50 B per hypothesis
// scores of all views (batched): for every existing hypothesis i walk the existing supporters (S_i & P) in
// canonical order with the reference's per-camera replace/subtract accumulation (line3D.cc:1255-1274); a zero
// similarity never changes that accumulation, so visiting only the supporters gives the same float result.
//
// The similarity VALUES (fp64 acos + exp) are the expensive part and the supporters are spread very unevenly over the
// hypotheses of a list (the few hypotheses of a true 3D line support each other, clutter has none).  So a staged list
// is processed as a flat sequence of (hypothesis, supporter) pairs, 64 per round with one pair per lane (full lanes
// instead of "every lane walks its own supporters and the wave waits for the longest walk"); the per-camera
// accumulation, which is sequential per hypothesis by definition, then only adds up the values of the round.
__global__ __launch_bounds__(256) void k_score_all(uint32_t G, const uint32_t* __restrict__ off,
                                                   const uint32_t* __restrict__ boff,
                                                   const uint32_t* __restrict__ gseg_view,
                                                   DEntry* __restrict__ dents, const uint64_t* __restrict__ bits,
                                                   Slot* __restrict__ slots, uint32_t* __restrict__ max_score_bits,
                                                   const ViewDev* __restrict__ views,
                                                   const uint32_t* __restrict__ seg_base, SimConst sc,
                                                   uint32_t g0) {
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t g = g0 + blockIdx.x * 4 + wave;
    if (g >= G) return;
    const uint32_t b = off[g], L = off[g + 1] - b;
    if (L == 0) return;
    const uint32_t W = (L + 63) / 64;
    const uint64_t* rows = bits + boff[g];
    const uint64_t* P = rows + (size_t)L * W;
    const uint32_t vi = gseg_view[g];
    const ViewDev& v = views[vi];
    const SegX sx = views[0].segx[g];   // global array (upload_views): views[0].segx is its base
    __shared__ float s_d1[4][kScoreCap], s_d2[4][kScoreCap];
    __shared__ uint32_t s_tv[4][kScoreCap];
    __shared__ double s_dir[4][kScoreCap][3];
    __shared__ uint16_t s_start[4][kScoreCap + 2];              // first pair of every hypothesis (exclusive scan)
    __shared__ float s_acc[4][kScoreCap], s_cur[4][kScoreCap];  // running score / current camera's best (:1255-1274)
    __shared__ uint32_t s_cam[4][kScoreCap];
    __shared__ float r_sim[4][64];                              // the round's similarities and supporter cameras
    __shared__ uint32_t r_tv[4][64];
    const bool staged = L <= kScoreCap;
    float vmax = 0.0f;
    if (staged) {
        // ---- stage: depths, cameras, ONE unprojection (fp64 sqrt + 3 divisions) per hypothesis; pair counts ----
        for (uint32_t m0 = 0; m0 < L; m0 += 64) {
            const uint32_t i = m0 + lane;
            uint32_t cnt = 0;
            if (i < L) {
                const DEntry& e = dents[b + i];
                s_d1[wave][i] = e.dp1; s_d2[wave][i] = e.dp2; s_tv[wave][i] = e.tgt_view;
                const d3 dir = entry_dir(v.C, sx, e.dp1, e.dp2);
                s_dir[wave][i][0] = dir.x; s_dir[wave][i][1] = dir.y; s_dir[wave][i][2] = dir.z;
                s_acc[wave][i] = 0.0f; s_cur[wave][i] = 0.0f; s_cam[wave][i] = kEmpty;
                if ((P[i >> 6] >> (i & 63)) & 1ull)
                    for (uint32_t w = 0; w < W; ++w) cnt += (uint32_t)__popcll(rows[(size_t)i * W + w] & P[w]);
            }
            // exclusive scan of the counts over the list (wave scan per 64 + carry)
            uint32_t x = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
            const uint32_t carry = m0 ? (uint32_t)s_start[wave][m0] : 0u;
            if (i < L) s_start[wave][i + 1] = (uint16_t)(carry + x);      // <= 192 * 191 < 65536
            if (m0 == 0 && lane == 0) s_start[wave][0] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const uint32_t T = s_start[wave][L];
#ifdef L3D_STATS
        if (lane == 0) { L3D_BSTAT(0, 1); L3D_BSTAT(1, L); L3D_BSTAT(3, T); L3D_BSTAT_MAX(5, T); }
        for (uint32_t i = lane; i < L; i += 64) {
            const bool pr = (P[i >> 6] >> (i & 63)) & 1ull;
            uint32_t all = 0;
            for (uint32_t w = 0; w < W; ++w) all += (uint32_t)__popcll(rows[(size_t)i * W + w]);
            L3D_BSTAT(6, all);
            if (pr) { L3D_BSTAT(2, 1); if (s_start[wave][i + 1] > s_start[wave][i]) L3D_BSTAT(4, 1); }
        }
#endif
        // ---- rounds of 64 (hypothesis, supporter) pairs in canonical order ----
        for (uint32_t t0 = 0; t0 < T; t0 += 64) {
            const uint32_t t = t0 + lane;
            uint32_t i = 0;
            if (t < T) {
                uint32_t lo = 0, hi = L;                       // last i with start[i] <= t
                while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (s_start[wave][m] <= t) lo = m; else hi = m; }
                i = lo;
                uint32_t k = t - s_start[wave][i];             // the k-th supporter of i
                uint32_t j = 0;
                for (uint32_t w = 0; w < W; ++w) {
                    uint64_t m = rows[(size_t)i * W + w] & P[w];
                    const uint32_t c = (uint32_t)__popcll(m);
                    if (k < c) {
                        // k-th set bit of m: binary search on the popcount of the low part
                        uint32_t pos = 0;
#pragma unroll
                        for (int sh = 32; sh > 0; sh >>= 1) {
                            const uint32_t cl = (uint32_t)__popcll(m & ((1ull << sh) - 1ull));
                            if (k >= cl) { k -= cl; m >>= sh; pos += sh; }
                        }
                        j = w * 64 + pos;
                        break;
                    }
                    k -= c;
                }
                const DEntry& a = dents[b + i];
                const d3 ad{s_dir[wave][i][0], s_dir[wave][i][1], s_dir[wave][i][2]};
                const d3 od{s_dir[wave][j][0], s_dir[wave][j][1], s_dir[wave][j][2]};
                r_sim[wave][lane] = sim_value(ad, s_d1[wave][i], s_d2[wave][i], a.reg1, a.reg2, od, s_d1[wave][j],
                                              s_d2[wave][j], sc);
                r_tv[wave][lane] = s_tv[wave][j];
            }
            const uint32_t i_prev = __shfl_up(i, 1);
            const bool head = t < T && (lane == 0 || i != i_prev);   // first pair of a hypothesis within this round
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // the pairs of one hypothesis are consecutive lanes: its first lane adds them up, in order
            {
                const uint32_t e = i;
                if (head) {
                    const uint32_t p0 = lane;
                    const uint32_t p1 = min((uint32_t)s_start[wave][e + 1], t0 + 64) - t0;
                    float score3D = s_acc[wave][e], cur = s_cur[wave][e];
                    uint32_t cur_cam = s_cam[wave][e];
                    for (uint32_t p = p0; p < p1; ++p) {
                        const float sim = r_sim[wave][p];
                        const uint32_t otv = r_tv[wave][p];
                        if (otv == cur_cam) {
                            if (sim > cur) { score3D -= cur; score3D += sim; cur = sim; }
                        } else {
                            score3D += sim; cur = sim; cur_cam = otv;
                        }
                    }
                    s_acc[wave][e] = score3D; s_cur[wave][e] = cur; s_cam[wave][e] = cur_cam;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        for (uint32_t m0 = 0; m0 < L; m0 += 64) {
            const uint32_t i = m0 + lane;
            if (i < L) {
                const bool present = (P[i >> 6] >> (i & 63)) & 1ull;
                const float score3D = present ? s_acc[wave][i] : 0.0f;
                const uint32_t fl = dents[b + i].flags;
                dents[b + i].score3D = score3D;
                dents[b + i].flags = present ? (fl | kDPresent) : (fl & ~kDPresent);
                if (present) {
                    if (!(fl & kDInverse)) slots[dents[b + i].ref].score3D = score3D;
                    vmax = fmaxf(vmax, score3D);
                }
            }
        }
    } else {
        // lists beyond the staging capacity: every lane walks the supporters of its own hypotheses
        for (uint32_t m0 = 0; m0 < L; m0 += 64) {
            const uint32_t i = m0 + lane;
            if (i < L) {
                const bool present = (P[i >> 6] >> (i & 63)) & 1ull;
                DEntry a = dents[b + i];
                const d3 ad = entry_dir(v.C, sx, a.dp1, a.dp2);
                float score3D = 0.0f, cur = 0.0f;
                uint32_t cur_cam = kEmpty;
                if (present) {
                    for (uint32_t w = 0; w < W; ++w) {
                        uint64_t m = rows[(size_t)i * W + w] & P[w];
                        while (m) {
                            const uint32_t j = w * 64 + (uint32_t)__ffsll((long long)m) - 1u;
                            m &= m - 1;
                            const DEntry& o = dents[b + j];
                            const float odp1 = o.dp1, odp2 = o.dp2; const uint32_t otv = o.tgt_view;
                            const d3 od = entry_dir(v.C, sx, odp1, odp2);
                            const float sim = sim_value(ad, a.dp1, a.dp2, a.reg1, a.reg2, od, odp1, odp2, sc);
                            if (otv == cur_cam) {
                                if (sim > cur) { score3D -= cur; score3D += sim; cur = sim; }
                            } else {
                                score3D += sim; cur = sim; cur_cam = otv;
                            }
                        }
                    }
                }
                dents[b + i].score3D = score3D;
                dents[b + i].flags = present ? (a.flags | kDPresent) : (a.flags & ~kDPresent);
                if (present) {
                    if (!(a.flags & kDInverse)) slots[a.ref].score3D = score3D;
                    vmax = fmaxf(vmax, score3D);
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
    if (lane == 0 && vmax > 0.0f) atomicMax(&max_score_bits[vi], __float_as_uint(vmax));
}




// median = sorted(depths)[n/2] by 4-pass radix select on the (positive) float bit patterns;
// n == 0 -> L3D_EPS (line3D.cc:1658).  One workgroup per view.
// This is the last kernel of a call's tail, and it hands the call's results to the host itself: the workgroup that
// finishes last copies the head of the zero block -- pool counters, flags, chain state, totals, medians (l3d_api.hip:
// zero_layout) -- into the pinned host buffer.  A copy command behind the kernel cost a 12 us bubble + 1.4 us on the
// stream of a 1.5 ms call.  (Block 0 adds the cumulative count of replayed rows of k_match_tied_rows first.)
__global__ __launch_bounds__(1024) void k_median_all(const float* __restrict__ depths,
                                                     const uint32_t* __restrict__ hyp_off,
                                                     const uint32_t* __restrict__ seg_base,
                                                     const uint32_t* __restrict__ tie_total, uint32_t* __restrict__ tie_out,
                                                     float* __restrict__ out_median, const uint32_t* rb_src,
                                                     uint32_t* __restrict__ rb_host, uint32_t rb_words, uint32_t* rb_count,
                                                     uint32_t v0) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_rank, s_last;
    const uint32_t v = v0 + blockIdx.x;             // (a rank with a sharded tail serves the views [v0, v0 + gridDim.x))
    if (blockIdx.x == 0 && threadIdx.x == 0 && tie_total) *tie_out = *tie_total;
    const uint32_t h0 = hyp_off[seg_base[v]], h1 = hyp_off[seg_base[v + 1]];
    const uint32_t n = 2u * (h1 - h0);
    const float* dv = depths + 2u * h0;
    if (threadIdx.x == 0) { s_prefix = __float_as_uint((float)kEps); s_rank = n / 2; }
    if (n != 0) {
        if (threadIdx.x == 0) s_prefix = 0;
        __syncthreads();
        for (int pass = 3; pass >= 0; --pass) {
            const uint32_t shift = 8u * pass;
            if (threadIdx.x < 256) hist[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            const uint32_t himask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t bits = __float_as_uint(dv[i]);
                if ((bits & himask) == prefix) atomicAdd(&hist[(bits >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) {   // one wave: bin d with cum(d-1) <= rank < cum(d)  (4 bins per lane, wave prefix)
                const uint32_t l = threadIdx.x;
                const uint32_t c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
                uint32_t x = c0 + c1 + c2 + c3;
                const uint32_t mine = x;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (l >= (uint32_t)d) x += y; }
                const uint32_t before = x - mine, r = s_rank;
                if (r >= before && r < x) {
                    uint32_t acc = before, d = 4 * l;
                    if (acc + c0 <= r) { acc += c0; ++d; if (acc + c1 <= r) { acc += c1; ++d; if (acc + c2 <= r) { acc += c2; ++d; } } }
                    s_rank = r - acc;
                    s_prefix = prefix | (d << shift);
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        out_median[v] = __uint_as_float(s_prefix);
        s_last = 0;
        if (rb_host) {
            // release: this workgroup's median (workgroup 0: the replay count too) before its tick; every launch adds
            // gridDim.x ticks to a counter zeroed once per list pass, so "last" is a multiple of gridDim.x
            __threadfence();
            s_last = (atomicAdd(rb_count, 1u) + 1u) % gridDim.x == 0 ? 1u : 0u;
        }
    }
    __syncthreads();
    if (s_last) {
        __threadfence();   // acquire: the other workgroups' medians (written on other XCDs)
        for (uint32_t i = threadIdx.x; i < rb_words; i += blockDim.x)
            rb_host[i] = __hip_atomic_load(&rb_src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- launchers --------------------------------------------------------------------------------------
// gseg_view[g] = view of global segment g: grid = (segment blocks, views)
__global__ void k_fill_gseg_view(const uint32_t* __restrict__ seg_base, uint32_t* __restrict__ gseg_view) {
    const uint32_t v = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = seg_base[v];
    if (b + i < seg_base[v + 1]) gseg_view[b + i] = v;
}
hipError_t launch_fill_gseg_view(const uint32_t* seg_base, uint32_t V, uint32_t max_M, uint32_t* gseg_view,
                                 hipStream_t st) {
    if (!V || !max_M) return hipSuccess;
    hipLaunchKernelGGL(k_fill_gseg_view, dim3((max_M + 255) / 256, V), dim3(256), 0, st, seg_base, gseg_view);
    return hipGetLastError();
}
hipError_t launch_orient_pairs(const ViewDev* views, const PairDesc* pairs, uint32_t n_pairs, uint64_t max_slots,
                               Slot* slots, uint32_t* inv_tgt, uint32_t tgt16, float2* hyp_p, float2* hyp_q, double thr_lo,
                               double thr_hi, hipStream_t st) {
    if (!n_pairs || !max_slots) return hipSuccess;
    hipLaunchKernelGGL(k_orient_all, dim3((uint32_t)((max_slots + 255) / 256), n_pairs), dim3(256), 0, st, views,
                       pairs, slots, inv_tgt, tgt16, hyp_p, hyp_q, OrientThr{thr_lo, thr_hi});
    return hipGetLastError();
}
hipError_t launch_bits_len(uint32_t G, const uint32_t* off, uint32_t* len, uint32_t* long_list, uint32_t* n_long,
                           hipStream_t st) {
    if (!G) return hipSuccess;
    hipLaunchKernelGGL(k_bits_len, dim3((G + 255) / 256), dim3(256), 0, st, G, off, len, long_list, n_long);
    return hipGetLastError();
}
hipError_t launch_support_all(uint32_t g0, uint32_t G, const uint32_t* off, const uint32_t* boff, const DEntry* dents,
                              uint64_t* bits, const ViewDev* views, const uint32_t* seg_base,
                              const uint32_t* gseg_view, SimConst sc, hipStream_t st) {
    (void)seg_base;
    if (G <= g0) return hipSuccess;
    hipLaunchKernelGGL((k_support<1>), dim3((G - g0 + 3) / 4), dim3(256), 0, st, G, off, boff, dents, bits, views,
                       gseg_view, sc, g0, (const uint32_t*)nullptr);
    return hipGetLastError();
}
// the lists longer than one wave's staging capacity (long_list from k_bits_len), one workgroup each
hipError_t launch_support_long(uint32_t n_long, const uint32_t* long_list, const uint32_t* off, const uint32_t* boff,
                               const DEntry* dents, uint64_t* bits, const ViewDev* views, const uint32_t* gseg_view,
                               SimConst sc, hipStream_t st) {
    if (!n_long) return hipSuccess;
    hipLaunchKernelGGL((k_support<4>), dim3(n_long), dim3(256), 0, st, 0u, off, boff, dents, bits, views, gseg_view,
                       sc, 0u, long_list);
    return hipGetLastError();
}
// ---- seam-level scoring (l3d_score_matches): one view, every match present -----------------------------
// list entries from the marshalled arrays of Line3D::scoringGPU (line3D.cc:1330-1352): matches4 = (src segment,
// target camera, depth_p1, depth_p2), reg_tgt2 = View::regularizerFrom3Dpoint of the two 3D end points in the
// target view; regularisers as scoringCPU combines them (line3D.cc:1233-1248)
__global__ void k_seam_entries(uint32_t n, const float4* __restrict__ m4, const float2* __restrict__ rt,
                               const ViewDev* __restrict__ views, float k, DEntry* __restrict__ dents) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 m = m4[i];
    const float2 r = rt[i];
    const uint32_t seg = (uint32_t)m.x;
    const ViewDev& v = views[0];
    const Seg3 s3 = unproject(v.C, v.segx[seg].r1, v.segx[seg].r2, m.z, m.w);
    const float sig1 = m.z * k, sig2 = m.w * k;
    DEntry d;
    d.ref = i; d.dp1 = m.z; d.dp2 = m.w;
    d.reg1 = 0.5f * (2.0f * sig1 * sig1 + 2.0f * r.x * r.x);
    d.reg2 = 0.5f * (2.0f * sig2 * sig2 + 2.0f * r.y * r.y);
    d.score3D = 0.0f;
    d.tgt_view = (uint32_t)m.y;
    d.flags = kDInverse | (s3.length < kEps ? kDZeroLen : 0u);   // kDInverse: k_score_all must not touch a slot buffer
    d.pair = 0;
    dents[i] = d;
}
// presence mask = every hypothesis of the list
__global__ void k_seam_all_present(uint32_t G, const uint32_t* __restrict__ off, const uint32_t* __restrict__ boff,
                                   uint64_t* __restrict__ bits) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint32_t L = off[g + 1] - off[g];
    if (!L) return;
    const uint32_t W = (L + 63) / 64;
    uint64_t* P = bits + boff[g] + (size_t)L * W;
    for (uint32_t w = 0; w < W; ++w) P[w] = (w + 1 < W || (L & 63u) == 0) ? ~0ull : ((1ull << (L & 63u)) - 1ull);
}
__global__ void k_seam_scores_out(uint32_t n, const DEntry* __restrict__ dents, float* __restrict__ scores) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scores[i] = dents[i].score3D;
}
hipError_t launch_seam_entries(uint32_t n, const float4* m4, const float2* rt, const ViewDev* views, float k,
                               DEntry* dents, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_seam_entries, dim3((n + 255) / 256), dim3(256), 0, st, n, m4, rt, views, k, dents);
    return hipGetLastError();
}
hipError_t launch_seam_all_present(uint32_t G, const uint32_t* off, const uint32_t* boff, uint64_t* bits, hipStream_t st) {
    if (!G) return hipSuccess;
    hipLaunchKernelGGL(k_seam_all_present, dim3((G + 255) / 256), dim3(256), 0, st, G, off, boff, bits);
    return hipGetLastError();
}
hipError_t launch_seam_scores_out(uint32_t n, const DEntry* dents, float* scores, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_seam_scores_out, dim3((n + 255) / 256), dim3(256), 0, st, n, dents, scores);
    return hipGetLastError();
}
hipError_t launch_score_all(uint32_t g0, uint32_t G, const uint32_t* off, const uint32_t* boff, const uint32_t* gseg_view,
                            DEntry* dents, const uint64_t* bits, Slot* slots, uint32_t* max_score_bits,
                            const ViewDev* views, const uint32_t* seg_base, SimConst sc, hipStream_t st) {
    if (G <= g0) return hipSuccess;
    hipLaunchKernelGGL(k_score_all, dim3((G - g0 + 3) / 4), dim3(256), 0, st, G, off, boff, gseg_view, dents, bits,
                       slots, max_score_bits, views, seg_base, sc, g0);
    return hipGetLastError();
}
hipError_t launch_median_all(uint32_t V, const float* depths, const uint32_t* hyp_off, const uint32_t* seg_base,
                             const uint32_t* tie_total, uint32_t* tie_out, float* out, const uint32_t* rb_src,
                             uint32_t* rb_host, uint32_t rb_words, uint32_t* rb_count, hipStream_t st, uint32_t v0) {
    if (!V) return hipSuccess;
    hipLaunchKernelGGL(k_median_all, dim3(V), dim3(1024), 0, st, depths, hyp_off, seg_base, tie_total, tie_out, out, rb_src,
                       rb_host, rb_words, rb_count, v0);
    return hipGetLastError();
}

}  // namespace l3d

#ifdef L3D_STATS
extern "C" void l3d_debug_bstats(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(l3d::g_bstats), sizeof(l3d::g_bstats));
    if (reset) { unsigned long long z[8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(l3d::g_bstats), z, sizeof(z)); }
}
#endif


// ---- start-up (l3d_create): the runtime loads a translation unit's code object at the first launch of one of its
// kernels (~0.6 ms each, measured on the first matchImages of a process); an empty launch pays that at context creation
namespace l3d {
namespace { __global__ void k_warm_views() {} }
hipError_t warm_views(hipStream_t st) {
    hipLaunchKernelGGL(k_warm_views, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
}  // namespace l3d
