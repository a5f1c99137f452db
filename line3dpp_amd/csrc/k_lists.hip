// k_lists.hip -- phase B, sparse form: the per-view part of Line3D::computeMatches (line3D.cc:745-773) =
// scoringCPU (:1208-1294) + storeInverseMatches (:1672-1699) + filterMatches (:1586-1669) for ALL views.
//
//   k_pair_csr      the potential inverse hypotheses of each directed pair, counting-sorted by target segment in LDS
//                   (slot indices + per-pair CSR offsets over the target view's segments; no device atomics)
//   k_lists<WPL>    THE dense pass, one wave (long lists: one 4-wave workgroup) per 2D segment: gathers the segment's
//                   hypotheses in canonical (= reference single-thread) order, sorts them by depth, finds the pairs
//                   (i, j) with similarityForScoring(i, j) > 0.5 inside each hypothesis' depth window (conservative
//                   window from slot data alone) and leaves them as CANDIDATES.  98 % of the hypotheses have none.
//   k_cand_exact    the reference's exact decision and similarity for every candidate, one per thread, full waves
//   k_edges         one wave per segment with candidates: the accepted ones as EDGES in (i, j) order + one HEADER per
//                   supported hypothesis (l3d_lists.h)
//   k_lists_huge    the same for lists beyond the LDS capacity of k_lists<4> (global-memory staging, all pairs)
//   k_chain_sweep   the reference's chain (view v sees the inverse matches of views u < v whose score was > 0,
//                   :1680) as a monotone fixed point over the headers: positive[slot] only ever gains bits and the
//                   dependency graph is acyclic (u < v), so sweeping until nothing changes gives the sequential
//                   result; a sweep is one pass over ~10^5 headers
//   k_hyp_scores    score3D of every header from its edges (per-camera replace/subtract accumulation, :1255-1274)
//   k_hyp_filter / k_seg_filter / k_seg_write   filterMatches: 10 % of the view's best score, first strict maximum,
//                   0.75 gate; surviving matches_ lists, estimated_position3D_ entries, depths for the view medians
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "l3d_dev.h"
#include "l3d_kernels.h"
#include "l3d_lists.h"

// grids of the multi-wave list tiers (every alternative measured is slower: profiles/r05_ab_phase_b.txt)
constexpr uint32_t kLists4Grid = 512;      // fixed grid of the four-wave tier

namespace l3d {

namespace {

#define L3D_LDS __attribute__((address_space(3)))

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

template <int WPL>
__device__ __forceinline__ void group_barrier() {
    if (WPL == 1) {
        // one wave: the LDS executes a wave's instructions in order, so only the compiler has to be held back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// exclusive prefix count of a flag over the group's threads and the group total
template <int WPL>
__device__ __forceinline__ uint32_t group_scan(uint32_t v, uint32_t& total, uint32_t* red);
template <int WPL>
__device__ __forceinline__ uint32_t group_count(bool f, uint32_t& total, uint32_t* red) {
    if (WPL == 1) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(f);
        total = (uint32_t)__popcll(m);
        return (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
    }
    return group_scan<WPL>(f ? 1u : 0u, total, red);
}

// exclusive prefix sum of v over the group's threads (thread order) and the group total.  red: >= 8 words of LDS
template <int WPL>
__device__ __forceinline__ uint32_t group_scan(uint32_t v, uint32_t& total, uint32_t* red) {
    const uint32_t lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
    if (WPL == 1) {
        total = __shfl(x, 63);
        return x - v;
    }
    const uint32_t wave = threadIdx.x >> 6;
    __syncthreads();                       // red may still be read from a previous call
    if (lane == 63) red[wave] = x;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WPL; ++w) { const uint32_t c = red[w]; before += w < wave ? c : 0u; tot += c; }
    total = tot;
    return before + x - v;
}

template <int WPL>
__device__ __forceinline__ uint32_t group_max(uint32_t v, uint32_t* red) {
    uint32_t x = v;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x = max(x, (uint32_t)__shfl_xor(x, d));
    if (WPL == 1) return x;
    const uint32_t wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 0) red[wave] = x;
    __syncthreads();
    uint32_t m = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WPL; ++w) m = max(m, red[w]);
    return m;
}

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// ---- similarityForScoring (line3D.cc:1417-1446) for hypotheses a (the scored one) and b of the same 2D segment ----
// Decision form: "similarityForScoring > L3D_DEF_MIN_SIMILARITY_3D" without transcendentals.
// sim = fmin(sim_a, fmin(e1, e2)) > 0.5 <=> every non-NaN component is > 0.5 (fmin skips NaNs; all NaN -> false).
// expf and acos are monotone, so each component test is a comparison of its float argument with a threshold the host
// found by bisection with the libm the reference itself would use (sim_thresholds, l3d_api.hip):
//     expf(y) > 0.5f                           <=>  y > y_thr
//     expf(-angle(x)^2 / two_sigA_sqr) > 0.5f  <=>  x >= x_hi || x <= x_lo     (x = clamped float dot product)
__device__ __forceinline__ bool sim_decide(const d3 dira, bool zeroa, float adp1, float adp2, float reg1, float reg2,
                                           const d3 dirb, bool zerob, float bdp1, float bdp2, const SimConst sc) {
    if (zeroa || zerob) return false;
    const float d1 = adp1 - bdp1, d2 = adp2 - bdp2;
    const float y1 = -d1 * d1 / reg1, y2 = -d2 * d2 / reg2;
    // NaN components drop out of the fmin chain; the angular component is never NaN (x is clamped, sigma_a > 0)
    if (y1 == y1 && !(y1 > sc.y_thr)) return false;
    if (y2 == y2 && !(y2 > sc.y_thr)) return false;
    const float dot_p = (float)dot(dira, dirb);
    const float x = fmaxf(fminf(dot_p, 1.0f), -1.0f);
    return x >= sc.x_hi || x <= sc.x_lo;
}

// Value form, for pairs sim_decide accepted.  acos/exp are evaluated in double and rounded to float (glibc's
// expf/acos differ from that by < 1 float ulp).  fmin(expf(ya), fmin(expf(y1), expf(y2))) == expf(fmin(ya, fmin(y1,
// y2))): expf is monotone and fmin skips NaNs on both sides alike -- one exponential instead of three.
__device__ __forceinline__ float sim_value(const d3 dira, float adp1, float adp2, float reg1, float reg2,
                                           const d3 dirb, float bdp1, float bdp2, const SimConst sc) {
    const float d1 = adp1 - bdp1, d2 = adp2 - bdp2;
    const float y1 = -d1 * d1 / reg1, y2 = -d2 * d2 / reg2;
    const float dot_p = (float)dot(dira, dirb);
    float angle = (float)(acos((double)fmaxf(fminf(dot_p, 1.0f), -1.0f)) / M_PI * 180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    const float ya = -angle * angle / sc.two_sigA_sqr;
    return (float)exp((double)fminf(ya, fminf(y1, y2)));
}

// the exact test of one ordered pair (i scored, j supporter) of the segment with invariants sx in view v:
// scoringCPU's unprojection + spatial regularisers of i (line3D.cc:1233-1248) and similarityForScoring(i, j)
__device__ __forceinline__ bool exact_support(const ViewDev& v, const ViewDev* __restrict__ views, const SegX& sx,
                                              float a1, float a2, uint32_t tvi, float b1, float b2, const SimConst sc,
                                              float& sim) {
    const Seg3 si = unproject(v.C, sx.r1, sx.r2, a1, a2);
    const Seg3 sj = unproject(v.C, sx.r1, sx.r2, b1, b2);
    const ViewDev& vt = views[tvi];
    const float k = v.k;
    const float sig1 = a1 * k, sig2 = a2 * k;
    float reg1 = 2.0f * sig1 * sig1, reg2 = 2.0f * sig2 * sig2;
    const d3 ct{vt.C[0], vt.C[1], vt.C[2]};
    const float sig1_t = (float)(norm(si.P1 - ct) * (double)vt.k);   // View::regularizerFrom3Dpoint
    const float sig2_t = (float)(norm(si.P2 - ct) * (double)vt.k);
    reg1 = 0.5f * (reg1 + 2.0f * sig1_t * sig1_t);
    reg2 = 0.5f * (reg2 + 2.0f * sig2_t * sig2_t);
    if (!sim_decide(si.dir, si.length < kEps, a1, a2, reg1, reg2, sj.dir, sj.length < kEps, b1, b2, sc)) return false;
    sim = sim_value(si.dir, a1, a2, reg1, reg2, sj.dir, b1, b2, sc);
    return true;
}

// Conservative squared depth windows of hypothesis i, from slot data alone: similarityForScoring(i, .) > 0.5 needs
// d^2 < 0.6932 * reg (expf(-d^2/reg) > 0.5), and reg = (dp k)^2 + (|P - C_t| k_t)^2 with |P - C_t| <= dp + |C - C_t|
// (P = C + ray * dp, unit ray): no unprojection is needed to bound it.  0.72 * 1.001 covers every float rounding.
__device__ __forceinline__ float window_sq(float dp, float k, float kt, float cc_dist) {
    const float a = dp * k, b = (dp + cc_dist) * kt;
    const float r = 0.72f * 1.001f * (a * a + b * b) + 1e-37f;
    return (r < 1e30f) ? r : __builtin_inff();     // NaN / huge: the whole list is the window
}

// ---- compare-exchange steps of the sorting networks with SCALAR direction logic ---------------------------------------
// In a bitonic network, which lanes keep the smaller key of a pair is a fixed pattern of the lane index per stage.
// Written as `keep_min ? min : max` it costs two comparisons and selects per key and stage; here the pattern
// is a 64-bit constant, the comparison result is taken as a lane mask (ballot), the two are combined on the scalar unit
// and the key is selected by that mask: one VALU comparison + one v_cndmask per 32-bit key and stage (the sort was ~45 % of
// the list pass's VALU instructions, and the pass is VALU-bound: 1 030 instructions x 128 k lists = its 0.2 ms).
// bit t of the result: (t & d) == 0 for a power of two d < 64; all lanes for d >= 64 of a one-wave index
__device__ __forceinline__ uint64_t lanes_bit_clear(uint32_t d) {
    return d == 1 ? 0x5555555555555555ull : d == 2 ? 0x3333333333333333ull : d == 4 ? 0x0F0F0F0F0F0F0F0Full :
           d == 8 ? 0x00FF00FF00FF00FFull : d == 16 ? 0x0000FFFF0000FFFFull : d == 32 ? 0x00000000FFFFFFFFull : ~0ull;
}
// Round 6: 32-BIT SORT KEYS.  A key is the order-preserving image of the first depth with its low bits replaced by the
// canonical index (kKeyIdxBits: as many as the tier's capacity needs), so that a compare-exchange is ONE lane exchange, one
// comparison and one select where the 64-bit key (depth << 32 | index) took two, a 64-bit comparison and two -- the sort is
// 40 % of the list pass's VALU instructions on long lists (C2) and the pass is bound by them (valu_busy 0.6-0.9,
// profiles/r06_pmc_lists_C*.json).  The order is that of the TRUNCATED depth: the walk along the sorted keys allows for the
// truncation (process_list: R1w) and decides on the exact depths, so the candidate set is what the 64-bit keys gave.
__device__ __forceinline__ uint32_t select_lanes32(uint64_t mask_in, uint32_t a, uint32_t b) {
    const uint64_t mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(mask_in >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)mask_in);
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t cmp_exchange(uint32_t v, uint32_t o, uint64_t keep_min) {
    const uint64_t own_less = __builtin_amdgcn_ballot_w64(v < o);
    return select_lanes32(~(own_less ^ keep_min), v, o);
}
__device__ __forceinline__ uint32_t make_key(float d1, uint32_t idx, uint32_t idx_mask) { return (f2ord(d1) & ~idx_mask) | idx; }

// Bitonic sort of the list's depth keys with the keys in REGISTERS (thread t owns the KPT consecutive elements
// t*KPT ..): partner distance below KPT = register exchange, below 64 threads = lane exchange, and only the partners in
// another wave of the group go through LDS with barriers -- for the usual list of 65..128 hypotheses handled by one wave
// none at all instead of 28 LDS round trips, for 256 keys on two waves one barrier stage instead of 36.  The keys are
// distinct (the canonical index is part of them).  Leaves the sorted keys in keys[0 .. 64*WPL*KPT).
template <int WPL, int KPT>
__device__ __forceinline__ void list_sort_regs(L3D_LDS uint32_t* keys, L3D_LDS const float* e_d1, uint32_t L, uint32_t t, uint32_t idx_mask) {
    constexpr uint32_t GS = 64 * WPL, N = GS * KPT;
    uint32_t v[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
        const uint32_t x = t * KPT + r;
        v[r] = x < L ? make_key(e_d1[x], x, idx_mask) : ~0u;
    }
    const uint32_t wave_t = __builtin_amdgcn_readfirstlane(t) & ~63u;   // first thread index of this wave (uniform)
    // lanes whose element t*KPT + r has bit k clear ("ascending" half of the k-merge)
    auto up_mask = [&](uint32_t k, int r) -> uint64_t {
        if (k < (uint32_t)KPT) return ((uint32_t)r & k) == 0 ? ~0ull : 0ull;
        const uint32_t kk = k / KPT;
        return kk < 64 ? lanes_bit_clear(kk) : ((wave_t & kk) == 0 ? ~0ull : 0ull);
    };
#pragma unroll
    for (uint32_t k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j >= 64u * KPT; j >>= 1) {        // partner in another wave (WPL > 1 only)
            group_barrier<WPL>();
#pragma unroll
            for (int r = 0; r < KPT; ++r) keys[t * KPT + r] = v[r];
            group_barrier<WPL>();
            const uint32_t pt = t ^ (j / KPT);
            const uint64_t lower = (wave_t & (j / KPT)) == 0 ? ~0ull : 0ull;
#pragma unroll
            for (int r = 0; r < KPT; ++r) v[r] = cmp_exchange(v[r], keys[pt * KPT + r], ~(lower ^ up_mask(k, r)));
        }
#pragma unroll
        for (uint32_t j = (k >> 1) < 32u * KPT ? (k >> 1) : 32u * KPT; j >= (uint32_t)KPT; j >>= 1) {   // partner in this wave
            const uint32_t d = j / KPT;
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                const uint32_t o = __shfl_xor(v[r], (int)d);
                v[r] = cmp_exchange(v[r], o, ~(lanes_bit_clear(d) ^ up_mask(k, r)));
            }
        }
#pragma unroll
        for (int jj = KPT / 2; jj > 0; jj >>= 1) {                   // partner in this thread
            if ((uint32_t)jj < k) {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if ((r & jj) == 0) {
                        const uint32_t a = v[r], b = v[r | jj];
                        // swap where (a > b) == ascending
                        const uint64_t sw = ~(__builtin_amdgcn_ballot_w64(a > b) ^ up_mask(k, r));
                        v[r] = select_lanes32(sw, b, a); v[r | jj] = select_lanes32(sw, a, b);
                    }
                }
            }
        }
    }
    group_barrier<WPL>();
#pragma unroll
    for (int r = 0; r < KPT; ++r) keys[t * KPT + r] = v[r];
}

// BASE = hypotheses one wave stages (128 for scenes with short lists: 8 waves per SIMD; 256 for long lists, where the
// multi-wave tiers would otherwise carry most of the work)
template <int WPL, int BASE>
struct ListCfg {
    static constexpr uint32_t GS = 64 * WPL;      // threads per list
    static constexpr uint32_t CAP = BASE * WPL;   // hypotheses staged in LDS
    static constexpr uint32_t NKEY = BASE * WPL;  // sort keys (power of two >= CAP)
    static constexpr uint32_t BYTES = CAP * 20 + NKEY * 8 + CAP * 2 + 64;   // LDS per list
};

// One list: its candidate pairs.  Returns 0 (done / nothing to do) or the list's length L > CAP (needs a larger kernel:
// nothing was written).  The length is not known beforehand (round 4: no per-segment counters are kept): it is the number
// of inverse records in the segment's rows of its incoming pairs' CSRs plus the alive fresh slots the pass finds.
// Group-uniform control flow.
template <int WPL, int BASE>
__device__ __forceinline__ uint32_t process_list(uint32_t vi, uint32_t seg, uint32_t pool, L3D_LDS char* lds,
                                                 const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                                                 const ListView* __restrict__ lviews, const OutPair* __restrict__ opairs,
                                                 const InPair* __restrict__ ipairs, const uint32_t* __restrict__ poff,
                                                 const uint32_t* __restrict__ inv, const float2* __restrict__ hyp_p,
                                                 const float2* __restrict__ hyp_q, uint32_t uniform_K, const ListPools lp) {
    typedef ListCfg<WPL, BASE> Cfg;
    constexpr uint32_t GS = Cfg::GS, CAP = Cfg::CAP, NKEY = Cfg::NKEY;
    const uint32_t t = WPL == 1 ? lane_id() : threadIdx.x;
    const ListView lv = lviews[vi];
    const uint32_t g = lv.seg_base + seg;
    L3D_LDS float* e_d1 = (L3D_LDS float*)lds;
    L3D_LDS float* e_d2 = e_d1 + CAP;
    L3D_LDS uint32_t* e_tv = (L3D_LDS uint32_t*)(e_d2 + CAP);
    L3D_LDS uint32_t* e_ref = e_tv + CAP;
    L3D_LDS uint32_t* e_pf = e_ref + CAP;
    // (32-bit sort keys in the first half of the 8-byte-per-key area, whose full size the staging of the inverse hypotheses
    // below still uses as scratch)
    L3D_LDS uint32_t* keys = (L3D_LDS uint32_t*)(e_pf + CAP);
    L3D_LDS uint16_t* pos_of = (L3D_LDS uint16_t*)(keys + 2 * NKEY);    // sorted position of hypothesis i
    constexpr uint32_t kIdxMask = NKEY - 1;                             // low bits of a key: the canonical index
    static_assert((NKEY & (NKEY - 1)) == 0 && CAP <= NKEY, "the index bits of a sort key cover the tier's capacity");
    uint32_t* red = (uint32_t*)(pos_of + CAP);                          // 16 words (generic pointer into LDS)
    const float vk = lv.k;
    const uint32_t q0 = lv.q0, nq = lv.nq;
    const uint32_t T = uniform_K ? nq * uniform_K : 0u;
    // (measured twice and not adopted, profiles/r04_ab_list_pass.txt: requesting the fresh slots BEFORE the inverse chain
    // and holding them in registers so that the two load chains run side by side -- 8 % slower with eight registers per
    // slot (70 VGPRs), 2-3 % slower in a lean form with five (56 VGPRs): the tier is bound by issue, not by the chains)
    // ---- inverse hypotheses: the segment's row of every incoming pair's CSR, pairs in ascending index (= ascending
    // source view), the entries of a row in ascending slot index -- that IS the canonical order.  The rows' starts and
    // their prefix within the list are staged in LDS (the keys array is not needed yet); an entry finds its row by a scan
    // over the handful of incoming pairs and its place inside the row by counting the row's smaller slot indices (a row
    // holds the few source segments of ONE view that matched this segment; until round 3 every record was ranked against
    // all inverse records of the list) ----
    uint32_t n_inv = 0;
    {
        // rows (incoming pairs) per round: a view has a handful.  Per row: [5q] first entry, [5q + 1] its place in the
        // list, [5q + 2] its end, [5q + 3] source view, [5q + 4] pair; behind the table the slot indices of the round
        constexpr uint32_t kRows = (2 * NKEY - CAP) / 5 >= 32 ? 32 : 16;
        L3D_LDS uint32_t* k32 = (L3D_LDS uint32_t*)keys;
        L3D_LDS uint32_t* kref = k32 + 5 * kRows;
        static_assert(5 * kRows + CAP <= 2 * NKEY, "LDS staging of the inverse hypotheses");
        constexpr uint32_t kInvPer = (CAP + GS - 1) / GS;               // entries per thread and round
        for (uint32_t qb = 0; qb < lv.ni; qb += kRows) {
            const uint32_t nq_here = min(kRows, lv.ni - qb), q = qb + t;
            uint32_t c = 0;
            if (t < nq_here) {   // (three independent loads: the transposed offsets need nothing of the pair's table entry)
                const uint32_t e0 = poff[lv.pbase + seg * lv.ni + q], e1 = poff[lv.pbase + (seg + 1) * lv.ni + q];
                const InPair ip = ipairs[lv.i0 + q];
                c = e1 - e0;
                k32[5 * t] = ip.rec_base + e0; k32[5 * t + 3] = ip.src; k32[5 * t + 4] = ip.pair;
            }
            uint32_t total;
            const uint32_t ex = group_scan<WPL>(c, total, red);
            if (t < nq_here) { k32[5 * t + 1] = ex; k32[5 * t + 2] = ex + c; }
            group_barrier<WPL>();
            if (n_inv + total > CAP) { n_inv += total; group_barrier<WPL>(); continue; }   // too long for this tier: only the length matters
            uint32_t ref[kInvPer], row[kInvPer];
            float2 dq[kInvPer];
#pragma unroll
            for (uint32_t cc = 0; cc < kInvPer; ++cc) {
                const uint32_t x = cc * GS + t;
                if (x < total) {
                    uint32_t r = 0;
                    {   // (four table words per LDS round trip: the trip count is a run-time value and the loop was one
                        // dependent read after the other)
                        uint32_t j = 1;
                        for (; j + 4 <= nq_here; j += 4) {
                            const uint32_t s0 = k32[5 * j + 1], s1 = k32[5 * j + 6], s2 = k32[5 * j + 11], s3 = k32[5 * j + 16];
                            r = s0 <= x ? j : r; r = s1 <= x ? j + 1 : r; r = s2 <= x ? j + 2 : r; r = s3 <= x ? j + 3 : r;
                        }
                        for (; j < nq_here; ++j) r = k32[5 * j + 1] <= x ? j : r;
                    }
                    row[cc] = r;
                    ref[cc] = inv[k32[5 * r] + (x - k32[5 * r + 1])];
                    kref[x] = ref[cc];
                    dq[cc] = hyp_q[ref[cc]];                            // the depths of the target's end points = this hypothesis' own
                }
            }
            group_barrier<WPL>();
#pragma unroll
            for (uint32_t cc = 0; cc < kInvPer; ++cc) {
                const uint32_t x = cc * GS + t;
                if (x < total) {
                    const uint32_t r = row[cc], r0 = k32[5 * r + 1], r1 = k32[5 * r + 2];
                    uint32_t rank = r0;                                  // place among the entries of its own row (a handful)
                    {
                        uint32_t y = r0;
                        for (; y + 4 <= r1; y += 4) {
                            const uint32_t k0 = kref[y], k1 = kref[y + 1], k2 = kref[y + 2], k3 = kref[y + 3];
                            rank += (k0 < ref[cc] ? 1u : 0u) + (k1 < ref[cc] ? 1u : 0u) + (k2 < ref[cc] ? 1u : 0u) + (k3 < ref[cc] ? 1u : 0u);
                        }
                        for (; y < r1; ++y) rank += (kref[y] < ref[cc]) ? 1u : 0u;
                    }
                    const uint32_t at = n_inv + rank;
                    e_d1[at] = dq[cc].x; e_d2[at] = dq[cc].y; e_tv[at] = k32[5 * r + 3]; e_ref[at] = ref[cc]; e_pf[at] = k32[5 * r + 4] | kHypInv;
                }
            }
            n_inv += total;
            group_barrier<WPL>();   // the table is rebuilt / the keys array is overwritten (sort keys, below)
        }
    }
    // ---- fresh hypotheses: the alive slots of the view's outgoing pairs, ascending (target view, slot) ----
    uint32_t pos = n_inv;
    {
        if (uniform_K) {
            for (uint32_t t0 = 0; t0 < T; t0 += GS) {
                const uint32_t x = t0 + t;
                bool alive = false;
                float2 s = make_float2(0.0f, 0.0f); uint32_t ref = 0, pi = 0, tv = 0;
                if (x < T) {
                    const OutPair op = opairs[q0 + x / uniform_K];
                    pi = op.pair; tv = op.tgt;
                    ref = (uint32_t)(op.slot_off + (uint64_t)seg * uniform_K + x % uniform_K);
                    s = hyp_p[ref];              // 8 bytes of the stream beside the slots: NaN = empty slot or filtered by orientation
                    alive = s.x == s.x;
                }
                uint32_t total;
                const uint32_t at = pos + group_count<WPL>(alive, total, red);
                if (alive && at < CAP) { e_d1[at] = s.x; e_d2[at] = s.y; e_tv[at] = tv; e_ref[at] = ref; e_pf[at] = pi; }
                pos += total;
            }
        } else {
            for (uint32_t q = q0; q < q0 + nq; ++q) {
                const OutPair op = opairs[q];
                uint64_t row0; uint32_t Kr;                               // (ragged rows in the keep-all mode: l3d_lists.h)
                out_row_slots(op, seg, lp, row0, Kr);
                for (uint32_t j0 = 0; j0 < Kr; j0 += GS) {
                    bool alive = false;
                    float2 s = make_float2(0.0f, 0.0f);
                    if (j0 + t < Kr) {
                        s = hyp_p[row0 + j0 + t];
                        alive = s.x == s.x;
                    }
                    uint32_t total;
                    const uint32_t at = pos + group_count<WPL>(alive, total, red);
                    if (alive && at < CAP) { e_d1[at] = s.x; e_d2[at] = s.y; e_tv[at] = op.tgt; e_ref[at] = (uint32_t)(row0 + j0 + t); e_pf[at] = op.pair; }
                    pos += total;
                }
            }
        }
    }
    group_barrier<WPL>();
    const uint32_t L = pos;
    if (L > CAP) return L;                       // nothing written beyond the staging arrays: a larger tier takes the list
    // (total list length of the pass = the reference's number of hypotheses: statistics, and the mean list length that
    // picks the staging width of the next call)
    if (t == 0 && L && lp.count_entries) { atomicAdd(&lp.cnt[pool * 16 + 5], L); if (n_inv) atomicAdd(&lp.cnt[pool * 16 + 6], n_inv); }
    if (L < 2) return 0;
    // ---- sort by the first depth: key = (order-preserving bits of dp1, canonical index) ----
    uint32_t N = 2;
    while (N < L) N <<= 1;
    if (WPL == 1 && N <= 64) {
        // one key per lane: the bitonic network runs on lane exchanges, no LDS round trip per stage
        uint32_t key = t < L ? make_key(e_d1[t], t, kIdxMask) : ~0u;
#pragma unroll
        for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const uint32_t other = __shfl_xor(key, (int)j);
                // lanes that keep the smaller key: ((t & j) == 0) == ((t & k) == 0)
                key = cmp_exchange(key, other, ~(lanes_bit_clear(j) ^ lanes_bit_clear(k)));
            }
        keys[t] = key;
    } else if (N == 2 * GS) {
        list_sort_regs<WPL, 2>(keys, e_d1, L, t, kIdxMask);
    } else if (CAP >= 4 * GS && N == 4 * GS) {
        list_sort_regs<WPL, (CAP >= 4 * GS ? 4 : 2)>(keys, e_d1, L, t, kIdxMask);
    } else {
    for (uint32_t x = t; x < N; x += GS) keys[x] = x < L ? make_key(e_d1[x], x, kIdxMask) : ~0u;
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            group_barrier<WPL>();
            for (uint32_t px = t; px < N / 2; px += GS) {
                const uint32_t lo = ((px & ~(j - 1)) << 1) | (px & (j - 1)), hi = lo | j;
                const uint32_t a = keys[lo], b = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    group_barrier<WPL>();
    for (uint32_t p = t; p < L; p += GS) pos_of[keys[p] & kIdxMask] = (uint16_t)p;
    group_barrier<WPL>();
    // ---- candidate pairs: every hypothesis walks outwards from its sorted position while the first depths differ by
    // no more than its window; a candidate also has another camera and its second depth inside the second window.
    // Threads take the hypotheses in CANONICAL order, so the records come out grouped by i in canonical order.
    // Two passes: count (-> one reservation for the whole list), then write the records ----
    constexpr uint32_t kPer = (CAP + GS - 1) / GS;                      // hypotheses per thread
    uint32_t my_cnt[kPer], my_total = 0;
    float my_R1[kPer], my_R2[kPer], my_R1w[kPer];                        // squared windows of my hypotheses; R1w: the walk's
    // The sorted keys carry the first depth TRUNCATED to its upper bits (t_j <= d_j < t_j (1 + kKeySlack)): the walk ends where
    // (a1 - t_j)^2 exceeds R1w = (r1 + (a1 + r1) kKeySlack)^2, r1 = sqrt(R1) -- beyond that no entry's true depth is within r1
    // of a1 in either direction --, and an entry inside is a candidate on its EXACT depths, as with the 64-bit keys.
    constexpr float kKeySlack = (float)NKEY * 1.1920929e-7f * 1.001f;   // 2^kKeyIdxBits ulps, relative
    auto walk = [&](uint32_t i, float R1, float R1w, float R2, auto&& emit) {
        const uint32_t p = pos_of[i];
        const float a1 = e_d1[i], a2 = e_d2[i];
        const uint32_t tvi = e_tv[i];
        // (98 % of the hypotheses have nobody inside their window: both first neighbours are requested together, so that the
        // usual walk is one LDS round trip instead of two)
        const uint32_t kd0 = p > 0 ? keys[p - 1] : 0u, ku0 = p + 1 < L ? keys[p + 1] : 0u;
        auto test = [&](uint32_t kq) -> bool {            // false: outside the window -- the walk in this direction ends
            const float d = a1 - ord2f(kq & ~kIdxMask);
            if (!(d * d <= R1w)) return false;
            const uint32_t j = kq & kIdxMask;
            const float d1 = a1 - e_d1[j], d2 = a2 - e_d2[j];
            if (d1 * d1 <= R1 && e_tv[j] != tvi && d2 * d2 <= R2) emit(j);
            return true;
        };
        if (p > 0 && test(kd0)) for (uint32_t q = p - 1; q-- > 0;) if (!test(keys[q])) break;
        if (p + 1 < L && test(ku0)) for (uint32_t q = p + 2; q < L; ++q) if (!test(keys[q])) break;
    };
    // hypothesis i = c * GS + t: canonical order = ascending (c, t), so the offsets are one group scan per c
    uint32_t H = 0;
#pragma unroll
    for (uint32_t c = 0; c < kPer; ++c) {
        const uint32_t i = c * GS + t;
        uint32_t cnt = 0;
        my_R1[c] = my_R2[c] = my_R1w[c] = -1.0f;
        if (c * GS < L) {
            if (i < L) {
                const float kt = views[e_tv[i]].k, D = pairs[e_pf[i] & 0x7FFFFFFFu].cc_dist;
                my_R1[c] = window_sq(e_d1[i], vk, kt, D); my_R2[c] = window_sq(e_d2[i], vk, kt, D);
                {   // (inf stays inf: the whole list is the window; the factor covers the rounding of the root and the square)
                    const float r1 = __builtin_sqrtf(my_R1[c]);
                    const float rw = r1 + (__builtin_fabsf(e_d1[i]) + r1) * kKeySlack;
                    my_R1w[c] = rw * rw * 1.00001f;
                }
                walk(i, my_R1[c], my_R1w[c], my_R2[c], [&](uint32_t) { ++cnt; });
            }
            uint32_t total;
            my_cnt[c] = H + group_scan<WPL>(cnt, total, red);          // offset of my records within the list
            H += total;
        }
        my_total += cnt;
    }
    if (H == 0) return 0;
    if (t == 0) {
        red[8] = atomicAdd(&lp.cnt[pool * 16 + 3], H);
        red[9] = atomicAdd(&lp.cnt[pool * 16 + 4], 1u);
    }
    group_barrier<WPL>();
    const uint32_t cb = red[8], hb = red[9];
    if (cb + H > lp.ccap || hb + 1 > lp.scap) {
        if (t == 0) atomicOr(&lp.flags[0], 1u);
        return 0;
    }
    const uint32_t c0 = pool * lp.ccap + cb;
    if (my_total) {
#pragma unroll
        for (uint32_t c = 0; c < kPer; ++c) {
            const uint32_t i = c * GS + t;
            if (i < L) {
                uint32_t w = c0 + my_cnt[c];
                walk(i, my_R1[c], my_R1w[c], my_R2[c], [&](uint32_t j) {
                    CandRec r;
                    r.ij = (i << 16) | j; r.ref_i = e_ref[i]; r.ref_j = e_ref[j]; r.pf_i = e_pf[i];
                    r.tvj = e_tv[j] | (e_pf[j] & kHypInv);
                    r.a1 = e_d1[i]; r.a2 = e_d2[i]; r.b1 = e_d1[j]; r.b2 = e_d2[j]; r.sim = __uint_as_float(g);
                    lp.cands[w++] = r;
                });
            }
        }
    }
    if (t == 0) lp.chdrs[pool * lp.scap + hb] = CandHdr{g, c0, H, 0u};
    return 0;
}

}  // namespace

// ---- the potential inverse hypotheses of a pair, sorted by target segment ------------------------------------------------
// storeInverseMatches (line3D.cc:1672-1699) hands every scored match of view u to the view of its target segment.  Until
// round 3 a slot took its place among the inverse references of its target with a device-scope atomic in the match
// epilogue (5.5 M of them on C1, each with a returned value the wave waited for), a scan turned the counters into
// offsets and k_inv_records scattered 16-byte records to them.  Now the match epilogue only writes the 4-byte stream
// inv_tgt (target segment of a slot that hands an inverse match over, kEmpty otherwise) and ONE WORKGROUP PER PAIR
// counting-sorts its slots by target segment: histogram over the target view's segments in LDS, scan, then every slot
// takes the next place of its target's run from an LDS cursor and writes its index there.  A run holds the handful of
// source rows that matched this target, in the order the LDS served the atomics; the list pass puts them into ascending
// slot order (the canonical order) on the way.  Traffic: the 4-byte stream twice (the second time from the L2) and
// 4 bytes per inverse hypothesis out -- a fifth of what the scatter of records moved.  Views beyond the LDS capacity (LDSCNT = false) keep the
// cursors in the pair's own offset array.
//   (first record of a pair: its slot_off -- a pair never has more records than slots)
constexpr uint32_t kCsrBlock = 1024;
constexpr uint32_t kCsrLdsSegs = 32768;        // up to 128 KiB of LDS cursors (+ the 64 dummy ones): gfx950 has 160 KiB per workgroup
constexpr uint32_t kCsrChunk = 8;              // slots per thread whose memory operations are in flight together
// The offsets are stored TRANSPOSED per target view: offT[pbase_v + t * ni_v + q] = first record of target segment t in
// the q-th incoming pair of view v (t = 0 .. M_v; ni_v incoming pairs).  A list of segment t then reads the offsets of all
// its incoming pairs with two coalesced loads whose addresses follow from the view table alone.
// The loops are written WITHOUT data-dependent branches: a workgroup is the only one on its CU most of the time (a
// handful of waves per SIMD), so what hides the latency of a load or of an LDS atomic is the next independent one of the
// same wave -- slots that hand nothing over count on one of 64 dummy cursors instead of skipping the atomic (a branch
// around each access serialised them: 0.11 ms on C1 where the batched form takes a fraction of that).
// STAGED (round 6, LDS cursors and 16-bit targets): the sorted order is assembled in LDS and leaves as COALESCED stores.  What
// the direct form waits for is its scattered stores (a scattered 4-byte store dirties a 32-byte sector: 48 bytes of counted
// write traffic per entry; halving the LDS atomics -- ranks kept in registers -- changed nothing, profiles/r06_ab_pair_csr.txt).
// The targets are cut into ranges whose entries fit the stage: per range one pass over the pair's 2-byte target stream
// (L2-resident), a slot of the range takes its place with the cursor atomic and leaves its index in the stage; then the stage
// goes out back to back.  C2: 0.68 -> 0.45 ms.
__host__ __device__ constexpr uint32_t lds_segs_round(uint32_t Mt) { return (Mt + 3u) & ~3u; }   // (the stage behind the cursors stays 16-byte aligned)
template <bool LDSCNT, bool TGT16, bool STAGED>
__global__ __launch_bounds__(kCsrBlock) void k_pair_csr(const PairDesc* __restrict__ pairs, uint32_t first_pair,
                                                        const PairCsr* __restrict__ pair_csr,
                                                        const uint32_t* __restrict__ inv_tgt,
                                                        uint32_t* __restrict__ poff, uint32_t* __restrict__ refs,
                                                        uint32_t* __restrict__ dummy, uint32_t tgt_v0, uint32_t tgt_v1,
                                                        uint32_t lds_segs, uint32_t stage_cap,
                                                        const uint32_t* __restrict__ row_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t part[kCsrBlock];
    const uint32_t pi = first_pair + blockIdx.x;
    const PairCsr pcs = pair_csr[pi];
    if (pcs.base == kEmpty) return;                                 // the pair hands nothing over (tgt < src)
    const PairDesc& pd = pairs[pi];
    if (pd.tgt < tgt_v0 || pd.tgt >= tgt_v1) return;                // only the target views of this pass
    const uint32_t Mt = pd.Mt, tid = threadIdx.x;
    if ((Mt <= lds_segs) != LDSCNT) return;                         // the other instantiation takes this pair
    // slots of the pair (< 2^32: l3d_match_begin limits the slot buffer); ragged rows (keep-all mode): from the rows' starts
    const uint32_t S = row_start ? row_start[pd.row_off + pd.Ms] - row_start[pd.row_off] : pd.Ms * pd.K;
    if (!S) {                                                       // (a ragged pair without a match: all runs are empty)
        for (uint32_t t = threadIdx.x; t <= pd.Mt; t += kCsrBlock) poff[pcs.base + pcs.q + (size_t)t * pcs.ni] = 0;
        return;
    }
    typedef typename std::conditional<TGT16, uint16_t, uint32_t>::type tgt_t;     // (0xFFFF / 0xFFFFFFFF: none -- both >= Mt)
    const tgt_t* __restrict__ it = (const tgt_t*)inv_tgt + pd.slot_off;
    // end of run t = start of run t + 1 lives at row t + 1 of the transposed table, column q of this pair
    uint32_t* __restrict__ offq = poff + pcs.base + pcs.q;
    const uint32_t ni = pcs.ni;
    // cursors: cur[t] runs from the start of run t to its end; [Mt, Mt + 64): dummies (global form: 64 words per workgroup)
    L3D_LDS uint32_t* lcur = (L3D_LDS uint32_t*)smem;
    uint32_t* gdum = dummy + (size_t)blockIdx.x * 64;
    auto gcur = [&](uint32_t t) -> uint32_t* { return t < Mt ? offq + (size_t)(t + 1) * ni : gdum + (t - Mt); };
    auto cur_get = [&](uint32_t t) -> uint32_t { return LDSCNT ? lcur[t] : *gcur(t); };
    auto cur_set = [&](uint32_t t, uint32_t v) { if (LDSCNT) lcur[t] = v; else *gcur(t) = v; };
    for (uint32_t t = tid; t < Mt + 64; t += kCsrBlock) cur_set(t, 0u);
    __threadfence_block();
    __syncthreads();
    const uint32_t dum = Mt + (tid & 63u);
    // ---- histogram ----
    for (uint32_t i0 = 0; i0 < S; i0 += kCsrChunk * kCsrBlock) {
        uint32_t tv[kCsrChunk];
#pragma unroll
        for (uint32_t k = 0; k < kCsrChunk; ++k) { const uint32_t i = i0 + k * kCsrBlock + tid; tv[k] = it[min(i, S - 1)]; tv[k] = (i < S && tv[k] < Mt) ? tv[k] : dum; }
#pragma unroll
        for (uint32_t k = 0; k < kCsrChunk; ++k) {
            if (LDSCNT) (void)__hip_atomic_fetch_add(&lcur[tv[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(gcur(tv[k]), 1u);
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- exclusive scan: thread x owns the segments [x * per, (x + 1) * per) ----
    const uint32_t per = (Mt + kCsrBlock - 1) / kCsrBlock;
    const uint32_t t0 = min(tid * per, Mt), t1 = min(t0 + per, Mt);
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += cur_get(t);
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < kCsrBlock; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t t = t0; t < t1; ++t) { const uint32_t c = cur_get(t); cur_set(t, run); run += c; }
    if (tid == 0) offq[0] = 0;
    __threadfence_block();
    __syncthreads();
    // ---- scatter: a slot takes the next place of its target's run (cursor) and its INDEX goes there -- 4 bytes; the
    // list pass gathers the two depths it needs by that index (round 6: from the 8-byte stream hyp_q; 16-byte entries that
    // carry the depths through the sort were measured twice and lose: in round 3 reading them from the 32-byte slots, in
    // round 6 from hyp_q -- coalesced in slot order or gathered in sorted order, profiles/r06_ab_pair_csr.txt).  Inside a run
    // the order is the order in which the atomics were served: the list pass ranks the handful of entries of a run (the
    // canonical order) ----
    uint32_t* __restrict__ out = refs + pd.slot_off;
    if (STAGED) {
        L3D_LDS uint32_t* stage = lcur + (lds_segs_round(Mt) + 64);       // [stage_cap] slot indices (within the pair) in sorted order
        const uint32_t total = part[kCsrBlock - 1];
        const uint32_t window = stage_cap - min(stage_cap / 8, 1024u);     // records a range aims at (the rest of the stage: its last run's overhang)
        uint32_t t_lo = 0, base = 0;
        while (base < total) {                                              // (uniform)
            // first target whose run starts at or beyond base + window: every thread searches the pristine starts [t_lo, Mt)
            uint32_t lo = t_lo + 1, hi = Mt;                                // (at least one target per range)
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lcur[mid] < base + window) lo = mid + 1; else hi = mid; }
            const uint32_t t_hi = lo, end = t_hi < Mt ? lcur[t_hi] : total;
            __syncthreads();                                                // (all have read the starts before a cursor moves)
            for (uint32_t i0 = 0; i0 < S; i0 += kCsrChunk * kCsrBlock) {
                uint32_t tv[kCsrChunk], at[kCsrChunk];
#pragma unroll
                for (uint32_t k = 0; k < kCsrChunk; ++k) { const uint32_t i = i0 + k * kCsrBlock + tid; tv[k] = it[min(i, S - 1)]; tv[k] = (i < S && tv[k] >= t_lo && tv[k] < t_hi) ? tv[k] : kEmpty; }
#pragma unroll
                for (uint32_t k = 0; k < kCsrChunk; ++k)
                    at[k] = tv[k] != kEmpty ? __hip_atomic_fetch_add(&lcur[tv[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
#pragma unroll
                for (uint32_t k = 0; k < kCsrChunk; ++k) {
                    if (tv[k] == kEmpty) continue;
                    const uint32_t i = i0 + k * kCsrBlock + tid, x = at[k] - base;
                    if (x < stage_cap) stage[x] = i;
                    else out[at[k]] = (uint32_t)pd.slot_off + i;       // (a run beyond the stage: rare)
                }
            }
            __threadfence_block();
            __syncthreads();
            const uint32_t n_here = min(end - base, stage_cap);
            for (uint32_t x = tid; x < n_here; x += kCsrBlock) out[base + x] = (uint32_t)pd.slot_off + stage[x];
            __syncthreads();                                                // (the stage is reused)
            t_lo = t_hi; base = end;
        }
        for (uint32_t t = tid; t < Mt; t += kCsrBlock) *gcur(t) = lcur[t];   // (the cursors have reached the ends of their runs)
        return;
    }
    for (uint32_t i0 = 0; i0 < S; i0 += kCsrChunk * kCsrBlock) {
        uint32_t tv[kCsrChunk], at[kCsrChunk];
#pragma unroll
        for (uint32_t k = 0; k < kCsrChunk; ++k) { const uint32_t i = i0 + k * kCsrBlock + tid; tv[k] = it[min(i, S - 1)]; tv[k] = (i < S && tv[k] < Mt) ? tv[k] : dum; }
#pragma unroll
        for (uint32_t k = 0; k < kCsrChunk; ++k)
            at[k] = LDSCNT ? __hip_atomic_fetch_add(&lcur[tv[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                           : atomicAdd(gcur(tv[k]), 1u);
#pragma unroll
        for (uint32_t k = 0; k < kCsrChunk; ++k)
            if (tv[k] < Mt) out[at[k]] = (uint32_t)pd.slot_off + i0 + k * kCsrBlock + tid;
    }
    if (LDSCNT) {                                                   // (LDSCNT = false: the cursors ARE the table entries)
        __syncthreads();
        for (uint32_t t = tid; t < Mt; t += kCsrBlock) *gcur(t) = lcur[t];
    }
}

// WPL = 1: grid (segment blocks, views), one wave per segment; WPL = 4: fixed grid over the segments handed on
// (one list per workgroup also for WPL = 1: with four lists per workgroup the LDS of a workgroup stayed allocated until
// its longest list was done -- 3.7 resident waves per SIMD on average where 6 fit)
template <int WPL, int BASE>
__global__ __launch_bounds__(64 * WPL) void k_lists(const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                                               const ListView* __restrict__ lviews, const OutPair* __restrict__ opairs,
                                               const InPair* __restrict__ ipairs, const uint32_t* __restrict__ gseg_view,
                                               const uint32_t* __restrict__ poff,
                                               const uint32_t* __restrict__ inv, const float2* __restrict__ hyp_p,
                                               const float2* __restrict__ hyp_q, uint32_t uniform_K, const ListPools lp, uint32_t view0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (WPL == 1) {
        const uint32_t vi = view0 + blockIdx.y, seg = blockIdx.x;
        if (seg >= lviews[vi].M) return;
        const uint32_t pool = lp.pool0 + ((blockIdx.y * gridDim.x + blockIdx.x) >> 2) % lp.npools;
        const uint32_t L = process_list<1, BASE>(vi, seg, pool, (L3D_LDS char*)smem, views, pairs, lviews, opairs,
                                                 ipairs, poff, inv, hyp_p, hyp_q, uniform_K, lp);
        if (L && lane_id() == 0) {              // longer than one wave stages: handed to a larger tier
            const uint32_t g = lviews[vi].seg_base + seg;
            if (L > 65535u) atomicOr(&lp.flags[1], 1u);
            else if (L > ListCfg<4, BASE>::CAP) lp.listH[atomicAdd(&lp.flags[5], 1u)] = g;
            else if (L > ListCfg<2, BASE>::CAP) lp.list4[atomicAdd(&lp.flags[4], 1u)] = g;
            else lp.list2[atomicAdd(&lp.flags[7], 1u)] = g;
        }
    } else {
        const uint32_t n = WPL == 2 ? lp.flags[7] : lp.flags[4];
        const uint32_t* list = WPL == 2 ? lp.list2 : lp.list4;
        for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
            const uint32_t g = list[idx];
            const uint32_t vi = gseg_view[g];
            (void)process_list<WPL, BASE>(vi, g - lviews[vi].seg_base, lp.pool0 + blockIdx.x % lp.npools, (L3D_LDS char*)smem, views, pairs,
                                          lviews, opairs, ipairs, poff, inv, hyp_p, hyp_q, uniform_K, lp);
            __syncthreads();
        }
    }
}

// ---- lists beyond the LDS capacity: one workgroup per list, hypotheses staged in global scratch, every ordered pair
// through the conservative window test (no sort), two passes (count, then write the candidate records) -------------
struct HugeScratch {
    float* d1; float* d2; uint32_t* tv; uint32_t* ref; uint32_t* pf; uint64_t* key;   // [cap] each
    uint32_t cap;
};
__global__ __launch_bounds__(256) void k_lists_huge(const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                                                    const ListView* __restrict__ lviews,
                                                    const OutPair* __restrict__ opairs, const InPair* __restrict__ ipairs,
                                                    const uint32_t* __restrict__ gseg_view,
                                                    const uint32_t* __restrict__ poff,
                                                    const uint32_t* __restrict__ inv, const float2* __restrict__ hyp_p,
                                                    const float2* __restrict__ hyp_q, const ListPools lp, const HugeScratch hs) {
    __shared__ uint32_t red[16];
    __shared__ uint32_t s_base;
    constexpr uint32_t kRows = 256;                 // incoming pairs staged per round
    __shared__ uint32_t s_row[4 * kRows];           // [4q] first entry of row q, [4q + 1] its place in the list, [4q + 2] source view, [4q + 3] pair
    const uint32_t t = threadIdx.x;
    const uint32_t nH = lp.flags[5];
    for (uint32_t idx = blockIdx.x; idx < nH; idx += gridDim.x) {
        const uint32_t g = lp.listH[idx];
        const uint32_t vi = gseg_view[g];
        const ListView lv = lviews[vi];
        const uint32_t seg = g - lv.seg_base;
        const ViewDev& v = views[vi];
        // ---- the list's length: inverse records (the segment's rows of the incoming pairs' CSRs) + alive fresh slots ----
        uint32_t mine_n = 0;
        for (uint32_t q = t; q < lv.ni; q += 256)
            mine_n += poff[lv.pbase + (seg + 1) * lv.ni + q] - poff[lv.pbase + seg * lv.ni + q];
        uint32_t n_inv;
        (void)group_scan<4>(mine_n, n_inv, red);
        mine_n = 0;
        for (uint32_t q = lv.q0; q < lv.q0 + lv.nq; ++q) {
            const OutPair op = opairs[q];
            uint64_t row0; uint32_t Kr;
            out_row_slots(op, seg, lp, row0, Kr);
            for (uint32_t j = t; j < Kr; j += 256) {
                const float2 s = hyp_p[row0 + j];
                mine_n += s.x == s.x ? 1u : 0u;
            }
        }
        uint32_t n_fresh;
        (void)group_scan<4>(mine_n, n_fresh, red);
        const uint32_t L = n_inv + n_fresh;
        __syncthreads();
        if (t == 0) {
            s_base = atomicAdd(&lp.flags[6], L);
            atomicAdd(&lp.cnt[(lp.pool0 + blockIdx.x % lp.npools) * 16 + 5], L); atomicAdd(&lp.cnt[(lp.pool0 + blockIdx.x % lp.npools) * 16 + 6], n_inv);
        }
        __syncthreads();
        const uint32_t base = s_base;
        if ((uint64_t)base + L > hs.cap) { if (t == 0) atomicOr(&lp.flags[2], 1u); continue; }
        float* e_d1 = hs.d1 + base; float* e_d2 = hs.d2 + base;
        uint32_t* e_tv = hs.tv + base; uint32_t* e_ref = hs.ref + base; uint32_t* e_pf = hs.pf + base;
        uint64_t* keys = hs.key + base;
        // ---- inverse records in canonical order: incoming pairs ascending, a pair's row as k_pair_csr left it ----
        uint32_t pos = 0;
        for (uint32_t qb = 0; qb < lv.ni; qb += kRows) {
            const uint32_t q = qb + t, nq_here = min(kRows, lv.ni - qb);
            uint32_t c = 0, a = 0;
            if (q < lv.ni) {
                const uint32_t e0 = poff[lv.pbase + seg * lv.ni + q], e1 = poff[lv.pbase + (seg + 1) * lv.ni + q];
                const InPair ip = ipairs[lv.i0 + q];
                a = ip.rec_base + e0; c = e1 - e0;
                s_row[4 * t + 2] = ip.src; s_row[4 * t + 3] = ip.pair;
            }
            uint32_t total;
            const uint32_t ex = group_scan<4>(c, total, red);
            if (q < lv.ni) { s_row[4 * t] = a; s_row[4 * t + 1] = ex; }
            __syncthreads();
            for (uint32_t x = t; x < total; x += 256) {
                uint32_t lo = 0, hi = nq_here;                      // last row whose prefix is <= x
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (s_row[4 * mid + 1] <= x) lo = mid; else hi = mid; }
                const uint32_t* rowp = inv + s_row[4 * lo];
                const uint32_t r0 = s_row[4 * lo + 1], r1 = lo + 1 < nq_here ? s_row[4 * lo + 5] : total;
                const uint32_t ref = rowp[x - r0];
                uint32_t rank = r0;                                 // ascending slot index inside the row (k_pair_csr leaves a row unordered)
                for (uint32_t y = 0; y < r1 - r0; ++y) rank += rowp[y] < ref ? 1u : 0u;
                const float2 dq = hyp_q[ref];
                const uint32_t at = pos + rank;
                e_d1[at] = dq.x; e_d2[at] = dq.y; e_tv[at] = s_row[4 * lo + 2]; e_ref[at] = ref; e_pf[at] = s_row[4 * lo + 3] | kHypInv;
            }
            pos += total;
            __syncthreads();
        }
        for (uint32_t q = lv.q0; q < lv.q0 + lv.nq; ++q) {
            const OutPair op = opairs[q];
            uint64_t row0; uint32_t Kr;
            out_row_slots(op, seg, lp, row0, Kr);
            for (uint32_t j0 = 0; j0 < Kr; j0 += 256) {
                bool alive = false;
                float2 s = make_float2(0.0f, 0.0f);
                if (j0 + t < Kr) { s = hyp_p[row0 + j0 + t]; alive = s.x == s.x; }
                uint32_t total;
                const uint32_t at = pos + group_scan<4>(alive ? 1u : 0u, total, red);
                if (alive && at < L) { e_d1[at] = s.x; e_d2[at] = s.y; e_tv[at] = op.tgt; e_ref[at] = (uint32_t)(row0 + j0 + t); e_pf[at] = op.pair; }
                pos += total;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (pos != L) { if (t == 0) atomicOr(&lp.flags[3], 1u); continue; }   // (the two passes over the same data disagree: internal error)
        auto for_candidates = [&](uint32_t i, auto&& emit) {
            const float a1 = e_d1[i], a2 = e_d2[i];
            const uint32_t tvi = e_tv[i];
            const float kt = views[tvi].k, D = pairs[e_pf[i] & 0x7FFFFFFFu].cc_dist;
            const float R1 = window_sq(a1, v.k, kt, D), R2 = window_sq(a2, v.k, kt, D);
            for (uint32_t j = 0; j < L; ++j) {
                if (j == i || e_tv[j] == tvi) continue;
                const float d1 = a1 - e_d1[j], d2 = a2 - e_d2[j];
                if ((d1 * d1 <= R1) && (d2 * d2 <= R2)) emit(j);
            }
        };
        uint32_t mine = 0;
        for (uint32_t i = t; i < L; i += 256) {
            uint32_t c = 0;
            for_candidates(i, [&](uint32_t) { ++c; });
            keys[i] = c;
            mine += c;
        }
        uint32_t H;
        (void)group_scan<4>(mine, H, red);
        __threadfence_block();
        __syncthreads();
        if (H == 0) continue;
        const uint32_t pool = lp.pool0 + blockIdx.x % lp.npools;
        if (t == 0) {
            red[8] = atomicAdd(&lp.cnt[pool * 16 + 3], H);
            red[9] = atomicAdd(&lp.cnt[pool * 16 + 4], 1u);
        }
        __syncthreads();
        const uint32_t cb = red[8], hb = red[9];
        if (cb + H > lp.ccap || hb + 1 > lp.scap) { if (t == 0) atomicOr(&lp.flags[0], 1u); continue; }
        const uint32_t c0 = pool * lp.ccap + cb;
        for (uint32_t i = t; i < L; i += 256) {
            if (!keys[i]) continue;
            uint32_t w = c0;
            for (uint32_t y = 0; y < i; ++y) w += (uint32_t)keys[y];
            for_candidates(i, [&](uint32_t j) {
                CandRec r;
                r.ij = (i << 16) | j; r.ref_i = e_ref[i]; r.ref_j = e_ref[j]; r.pf_i = e_pf[i];
                r.tvj = e_tv[j] | (e_pf[j] & kHypInv);
                r.a1 = e_d1[i]; r.a2 = e_d2[i]; r.b1 = e_d1[j]; r.b2 = e_d2[j]; r.sim = __uint_as_float(g);
                lp.cands[w++] = r;
            });
        }
        if (t == 0) lp.chdrs[pool * lp.scap + hb] = CandHdr{g, c0, H, 0u};
    }
}

// ---- the reference's decision and value for every candidate: ONE CANDIDATE PER THREAD over the pools ---------------
// This is where the fp64 unprojections, acos and exp of phase B live.  Until round 4 k_edges did it with one wave per
// segment -- a segment has ~17 candidates on C1, so three of four lanes idled through the most expensive code of the
// pass (PMC: 4.2 active lanes per issue cycle against 10-11 in k_lists / k_match_pairs).  The candidates of a pool are
// contiguous, so a flat launch over (pool, index) runs the same arithmetic with full waves; what was wave-uniform there
// (the segment's view, its rays) is gathered per lane here -- neighbouring lanes mostly share it.  The list pass leaves
// the segment in the record's `sim` word; this kernel replaces it by the similarity, or by -1.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_cand_exact(
        const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ gseg_view,
        const SimConst sc, const ListPools lp) {
    if (lp.flags[0] | lp.flags[2]) return;   // the candidate pools overflowed: the pass is discarded
    const uint32_t pool = lp.pool0 + blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= min(lp.cnt[pool * 16 + 3], lp.ccap)) return;
    CandRec* cr = lp.cands + (size_t)pool * lp.ccap + x;
    const CandRec r = *cr;
    const uint32_t g = __float_as_uint(r.sim);
    const PairDesc& pd = pairs[r.pf_i & 0x7FFFFFFFu];
    const uint32_t tvi = (r.pf_i & kHypInv) ? pd.src : pd.tgt;
    float sim = -1.0f;
    const bool ok = exact_support(views[gseg_view[g]], views, views[0].segx[g], r.a1, r.a2, tvi, r.b1, r.b2, sc, sim);
    cr->sim = ok ? sim : -1.0f;
}

// ---- candidates -> edges: one wave per segment with candidates.  The accepted ones (k_cand_exact) become EDGES in
// (i, j) order with one HEADER per hypothesis i that has any.  The candidates arrive grouped by i in canonical order
// (one contiguous run per i, j in walk order), so the order only has to be fixed inside a run.
constexpr uint32_t kEdgeLds = 512;
// the candidates of one segment, by one wave
__device__ __forceinline__ void edges_of_segment(const PairDesc* __restrict__ pairs, const Slot* __restrict__ slots,
                                                 const ListPools& lp, uint32_t* __restrict__ seg_of_g, const CandHdr ch,
                                                 uint32_t opool, L3D_LDS uint32_t* s_ij_w, L3D_LDS float* s_sim_w,
                                                 uint32_t lane) {
    const uint32_t g = ch.g, n = ch.cnt;
    CandRec* cand = lp.cands + ch.begin;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // (ij, sim) of the list's candidates live in LDS for the order fixing below (lists beyond kEdgeLds entries walk
    // global memory instead: correct, slow, rare)
    const bool in_lds = n <= kEdgeLds;
    // pass 1: how many passed the exact test
    uint32_t n_acc = 0;
    for (uint32_t x0 = 0; x0 < n; x0 += 64) {
        const uint32_t x = x0 + lane;
        bool ok = false;
        if (x < n) {
            const float sim = cand[x].sim;
            ok = sim >= 0.0f;
            if (in_lds) { s_ij_w[x] = cand[x].ij; s_sim_w[x] = sim; }
        }
        n_acc += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ok));
    }
    if (n_acc == 0) return;
    if (in_lds) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }   // (beyond the LDS capacity the passes below read the records k_cand_exact left: nothing of this wave's own)
    auto ij_of = [&](uint32_t y) -> uint32_t { return in_lds ? s_ij_w[y] : cand[y].ij; };
    auto sim_of = [&](uint32_t y) -> float { return in_lds ? s_sim_w[y] : cand[y].sim; };
    // inside its run: rank of an accepted pair by j, accepted pairs before it (0: it writes the header), run total
    auto in_run = [&](uint32_t x, uint32_t& rank, uint32_t& earlier, uint32_t& mine) {
        const uint32_t ij = ij_of(x), i = ij >> 16;
        rank = 0; earlier = 0; mine = 1;
        for (uint32_t y = x; y-- > 0;) {
            const uint32_t o = ij_of(y);
            if ((o >> 16) != i) break;
            if (sim_of(y) >= 0.0f) { ++earlier; ++mine; rank += o < ij ? 1u : 0u; }
        }
        for (uint32_t y = x + 1; y < n; ++y) {
            const uint32_t o = ij_of(y);
            if ((o >> 16) != i) break;
            if (sim_of(y) >= 0.0f) { ++mine; rank += o < ij ? 1u : 0u; }
        }
    };
    // pass 2: headers = runs with an accepted pair
    uint32_t n_h = 0;
    for (uint32_t x0 = 0; x0 < n; x0 += 64) {
        const uint32_t x = x0 + lane;
        bool first = false;
        if (x < n && sim_of(x) >= 0.0f) {
            const uint32_t i = ij_of(x) >> 16;
            first = true;
            for (uint32_t y = x; y-- > 0;) {
                if ((ij_of(y) >> 16) != i) break;
                if (sim_of(y) >= 0.0f) { first = false; break; }
            }
        }
        n_h += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(first));
    }
    uint32_t eb = 0, hb = 0, sb = 0;
    if (lane == 0) {
        eb = atomicAdd(&lp.cnt[opool * 16 + 0], n_acc);
        hb = atomicAdd(&lp.cnt[opool * 16 + 1], n_h);
        sb = atomicAdd(&lp.cnt[opool * 16 + 2], 1u);
    }
    eb = __shfl(eb, 0); hb = __shfl(hb, 0); sb = __shfl(sb, 0);
    if (eb + n_acc > lp.ecap || hb + n_h > lp.hcap || sb + 1 > lp.scap) {
        if (lane == 0) atomicOr(&lp.flags[0], 1u);
        return;
    }
    const uint32_t e0 = opool * lp.ecap + eb, h0 = opool * lp.hcap + hb;
    // pass 3: write.  accepted pairs before mine - earlier ones of my run = first edge of my run
    uint32_t acc_before = 0, first_before = 0;
    for (uint32_t x0 = 0; x0 < n; x0 += 64) {
        const uint32_t x = x0 + lane;
        bool acc = false, first = false;
        uint32_t rank = 0, earlier = 0, mine = 0;
        float sim = -1.0f;
        if (x < n) {
            sim = sim_of(x);
            acc = sim >= 0.0f;
            if (acc) { in_run(x, rank, earlier, mine); first = earlier == 0; }
        }
        const uint64_t ma = __builtin_amdgcn_ballot_w64(acc), mf = __builtin_amdgcn_ballot_w64(first);
        if (acc) {
            const CandRec r = cand[x];
            const uint32_t run0 = acc_before + (uint32_t)__popcll(ma & lt_mask) - earlier;
            EdgeRec e;
            e.ref_j = r.ref_j; e.sim = sim;
            e.j_cam = (r.ij & 0xFFFFu) | ((r.tvj & kHypInv) ? kEdgeInv : 0u);
            e.tv_j = r.tvj & 0x7FFFFFFFu;
            lp.edges[e0 + run0 + rank] = e;
            if (first) {
                HypHdr hdr;
                hdr.g = g; hdr.ref = r.ref_i; hdr.pair_flags = r.pf_i; hdr.canon = r.ij >> 16;
                hdr.dp1 = r.a1; hdr.dp2 = r.a2;
                hdr.edge_begin = e0 + run0; hdr.edge_cnt = mine; hdr.score3D = 0.0f; hdr.state = 0;
                {   // the hypothesis' own slot, once per header (line3D.cc:1682-1692 for the role swap of an inverse one)
                    const Slot hs = slots[r.ref_i];
                    const bool hinv = (r.pf_i & kHypInv) != 0;
                    const PairDesc& hp = pairs[r.pf_i & 0x7FFFFFFFu];
                    hdr.tgt_seg = !hinv ? hs.tgt_seg : lp.slot_row ? lp.slot_row[r.ref_i] : (uint32_t)((r.ref_i - hp.slot_off) / hp.K);
                    hdr.overlap = hs.overlap;
                    hdr.oq1 = hinv ? hs.dp1 : hs.dq1; hdr.oq2 = hinv ? hs.dp2 : hs.dq2;
                }
                hdr.pad[0] = hdr.pad[1] = 0;
                lp.hyps[h0 + first_before + (uint32_t)__popcll(mf & lt_mask)] = hdr;
            }
        }
        acc_before += (uint32_t)__popcll(ma); first_before += (uint32_t)__popcll(mf);
    }
    if (lane == 0) {
        const uint32_t si = opool * lp.scap + sb;
        lp.segs[si] = SegHdr{g, h0, n_h, 0u};
        seg_of_g[g] = si;
    }
}

// grid (ceil(scap / kEdgesSplit), pools): a wave takes every gridDim.x-th segment header of its pool.  One wave per
// header over the pools' CAPACITY launched 144 k waves on C1 of which 100 k found nothing to do.
constexpr uint32_t kEdgesSplit = 4;
__global__ __launch_bounds__(64) void k_edges(const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                                               const uint32_t* __restrict__ gseg_view, const Slot* __restrict__ slots,
                                               const ListPools lp, uint32_t* __restrict__ seg_of_g) {
    const uint32_t pool = lp.pool0 + blockIdx.y, lane = lane_id();
    if (lp.flags[0] | lp.flags[2]) return;   // the candidate pools overflowed: headers beyond the last complete list are not valid
    __shared__ uint32_t s_ij[kEdgeLds];
    __shared__ float s_sim[kEdgeLds];
    const uint32_t n_hdr = min(lp.cnt[pool * 16 + 4], lp.scap);
    for (uint32_t k = blockIdx.x; k < n_hdr; k += gridDim.x) {
        // output pool: spread over ALL pools of the pass whatever the fill of the input pools (on small scenes
        // `(k >> 2) % npools` left most pools empty and overflowed the rest -- the one pool retry of a first C1 call)
        const uint32_t opool = lp.pool0 + (k + 61u * blockIdx.y) % lp.npools;
        edges_of_segment(pairs, slots, lp, seg_of_g, lp.chdrs[pool * lp.scap + k], opool, (L3D_LDS uint32_t*)s_ij, (L3D_LDS float*)s_sim, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the next segment reuses the LDS arrays
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the chain as a monotone fixed point ------------------------------------------------------------------------
// grid (header blocks, pools).  Launch s runs only if launch s-1 changed something (changed[] is zeroed by the host).
// ONE look per launch: letting an undecided header look again inside a launch (all headers are resident, a bit set by another
// thread can travel down a chain within the launch) was measured in round 5 and does not pay -- `finish` of C1 0.571 / 0.580 /
// 0.641 / 0.717 ms with 1 / 2 / 6 / 16 looks (profiles/r05_ab_phase_b.txt); fetching four edges per round trip loses the exit at
// the first satisfied edge.  The launches enqueued: the largest need of the last four calls + 3 (l3d_phase_b.hip).
// Round 6: the FIRST sweep decides most headers (a fresh supporter always exists: every header with one is positive at once)
// and files the rest -- fresh hypotheses whose supporters are all inverse and not yet positive -- in a compact list; the
// following sweeps walk that list instead of all headers (C2: 3.5 M headers of 64 bytes per sweep, 60-125 us each, nine
// launches per call; the list holds a few per cent of them).  One list per pool (its region of `undecided`, an output array
// the tail writes later; its length in the pool's counter word 7): ONE counter for all waves made the first sweep five
// times as long as the round-5 sweep it replaced -- 55 000 returned atomics on one address.
__global__ void k_chain_sweep(const ListPools lp, uint8_t* __restrict__ positive, uint32_t* __restrict__ changed,
                              uint32_t* __restrict__ undecided) {
    if (lp.flags[0] | lp.flags[2]) return;                    // an overflowed pass is discarded: its records are incomplete
    const uint32_t pool = lp.pool0 + blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;   // (grid.y = lp.npools)
    bool later = false;
    if (k < min(lp.cnt[pool * 16 + 1], lp.hcap)) {
        const HypHdr& h = lp.hyps[pool * lp.hcap + k];
        // (an inverse hypothesis exists iff its SOURCE is positive: it has no bit of its own)
        if (!(h.pair_flags & kHypInv) && !__hip_atomic_load(&positive[h.ref], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const uint32_t n = h.edge_cnt, e0 = h.edge_begin;
            bool done = false, any_inverse = false;
            for (uint32_t e = 0; e < n && !done; ++e) {
                const EdgeRec& ed = lp.edges[e0 + e];
                const bool inv = (ed.j_cam & kEdgeInv) != 0;
                any_inverse |= inv;
                // a fresh supporter always exists (line3D.cc:1680)
                if (!inv || __hip_atomic_load(&positive[ed.ref_j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&positive[h.ref], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    changed[0] = 1;
                    done = true;
                }
            }
            later = !done && any_inverse;                     // (no inverse supporter: can never become positive)
        }
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(later);
    if (m) {
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&lp.cnt[pool * 16 + 7], (uint32_t)__popcll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (later) undecided[pool * lp.hcap + base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull))] = pool * lp.hcap + k;
    }
}
constexpr uint32_t kChainListBlocks = 4;                     // workgroups per pool that walk its list
__global__ void k_chain_sweep_list(const ListPools lp, uint8_t* __restrict__ positive, uint32_t* __restrict__ changed,
                                   uint32_t sweep, const uint32_t* __restrict__ undecided) {
    if (lp.flags[0] | lp.flags[2]) return;
    if (!changed[sweep - 1]) return;                          // the previous sweep found nothing new: the fixed point is reached
    const uint32_t pool = lp.pool0 + blockIdx.y;
    const uint32_t n_u = min(lp.cnt[pool * 16 + 7], lp.hcap);
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_u; idx += gridDim.x * blockDim.x) {
        const HypHdr& h = lp.hyps[undecided[pool * lp.hcap + idx]];
        if (__hip_atomic_load(&positive[h.ref], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
        const uint32_t n = h.edge_cnt, e0 = h.edge_begin;
        for (uint32_t e = 0; e < n; ++e) {
            const EdgeRec& ed = lp.edges[e0 + e];
            if ((ed.j_cam & kEdgeInv) && __hip_atomic_load(&positive[ed.ref_j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&positive[h.ref], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                changed[sweep] = 1;
                break;
            }
        }
    }
}

constexpr uint32_t kMaxReplicas = 16;   // the view maxima are kept in 16 replicas (by pool) against atomic contention
// score3D of every header: the existing supporters in canonical order with the reference's per-camera
// replace/subtract accumulation (line3D.cc:1255-1274); the view's maximum for filterMatches
__global__ void k_hyp_scores(const ListPools lp, const uint8_t* __restrict__ positive,
                             const uint32_t* __restrict__ gseg_view, Slot* __restrict__ slots,
                             const uint8_t* __restrict__ pair_present, uint32_t* __restrict__ max_score_bits) {
    const uint32_t pool = lp.pool0 + blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;   // (grid.y = lp.npools)
    const bool active = !(lp.flags[0] | lp.flags[2]) && k < min(lp.cnt[pool * 16 + 1], lp.hcap);
    float score3D = 0.0f;
    uint32_t view = kEmpty;
    if (active) {
        HypHdr& h = lp.hyps[pool * lp.hcap + k];
        const bool inv = (h.pair_flags & kHypInv) != 0;
        const bool exists = !inv || positive[h.ref] != 0;
        float cur = 0.0f;
        uint32_t cur_cam = kEmpty;
        if (exists) {
            // (four edge records, then their existence bytes, per global round trip; accumulated in the list's order)
            const uint32_t n = h.edge_cnt, e0 = h.edge_begin;
            for (uint32_t e = 0; e < n; e += 4) {
                EdgeRec ed[4]; bool ok[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) ed[i] = lp.edges[e0 + min(e + i, n - 1)];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) ok[i] = e + i < n && (!(ed[i].j_cam & kEdgeInv) || positive[ed[i].ref_j] != 0);
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                    if (!ok[i]) continue;
                    if (ed[i].tv_j == cur_cam) {
                        if (ed[i].sim > cur) { score3D -= cur; score3D += ed[i].sim; cur = ed[i].sim; }
                    } else {
                        score3D += ed[i].sim; cur = ed[i].sim; cur_cam = ed[i].tv_j;
                    }
                }
            }
        }
        h.score3D = score3D;
        h.state = exists ? kHypExists : 0u;
        // (the slot's own copy of the score, for l3d_get_pair_slots.  A rank of a multi-GPU run holds the slots of the
        // pairs that touch its views only: the headers of all ranks are here, the slot regions of the other pairs are not
        // valid memory contents and are left alone -- pair_present, null on one GPU)
        if (exists && !inv && (!pair_present || pair_present[h.pair_flags & 0x7FFFFFFFu])) slots[h.ref].score3D = score3D;
        view = gseg_view[h.g];
    }
    // the view's maximum: neighbouring headers mostly belong to one view -- one atomic per (wave, view), and none
    // when the view's maximum already exceeds the wave's
    uint64_t todo = __builtin_amdgcn_ballot_w64(score3D > 0.0f);
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t lv = __builtin_amdgcn_readlane(view, leader);
        const bool mine = score3D > 0.0f && view == lv;
        float mx = mine ? score3D : 0.0f;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
        if (lane_id() == leader) {
            uint32_t* p = &max_score_bits[lv * kMaxReplicas + (pool & (kMaxReplicas - 1))];   // replicas: less contention
            if (__float_as_uint(mx) > *(volatile uint32_t*)p) atomicMax(p, __float_as_uint(mx));
        }
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
    }
}

// filterMatches, line3D.cc:1602-1653, per header: kept iff score3D > 0 and > 10 % of the view's best; the segment's
// best = first strict maximum among the kept ones = largest (score, -canonical index)
__global__ void k_hyp_filter(const ListPools lp, const uint32_t* __restrict__ gseg_view,
                             const uint32_t* __restrict__ max_score_bits, uint32_t* __restrict__ kept_cnt,
                             unsigned long long* __restrict__ best_pack) {
    const uint32_t pool = lp.pool0 + blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;   // (grid.y = lp.npools)
    if ((lp.flags[0] | lp.flags[2]) || k >= min(lp.cnt[pool * 16 + 1], lp.hcap)) return;
    HypHdr& h = lp.hyps[pool * lp.hcap + k];
    uint32_t mx = 0;
    {
        const uint32_t* p = max_score_bits + (size_t)gseg_view[h.g] * kMaxReplicas;
#pragma unroll
        for (uint32_t r = 0; r < kMaxReplicas; ++r) mx = max(mx, p[r]);   // positive floats order like their bit patterns
    }
    const float lim = kMinBestScorePerc * __uint_as_float(mx);
    const float s = h.score3D;
    if ((h.state & kHypExists) && s > 0.0f && s > lim) {
        h.state |= kHypKeep;
        atomicAdd(&kept_cnt[h.g], 1u);
        atomicMax(&best_pack[h.g], ((unsigned long long)__float_as_uint(s) << 32) | (0xFFFFFFFFu - h.canon));
    }
}

// per segment: the 0.75 gate; cnt64[g] = surviving matches (low word) | has a best hypothesis (high word)
__global__ void k_seg_filter(uint32_t g0, uint32_t g1, const uint32_t* __restrict__ kept_cnt,
                             const unsigned long long* __restrict__ best_pack, unsigned long long* __restrict__ cnt64) {
    const uint32_t g = g0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= g1) return;
    const bool ok = __uint_as_float((uint32_t)(best_pack[g] >> 32)) > kMinBestScore3D;
    cnt64[g] = ok ? ((1ull << 32) | kept_cnt[g]) : 0ull;
}

__device__ __forceinline__ void make_match(const ViewDev* views, const PairDesc* pairs, uint32_t view, uint32_t seg,
                                           const HypHdr& h, Match& m, uint32_t& tgt_view, uint32_t& tgt_seg) {
    // (a fresh hypothesis reads its slot as is, an inverse one with the roles swapped, line3D.cc:1682-1692: k_edges has
    // put both forms into the header)
    const bool inv = (h.pair_flags & kHypInv) != 0;
    const PairDesc& pd = pairs[h.pair_flags & 0x7FFFFFFFu];
    tgt_view = inv ? pd.src : pd.tgt;
    tgt_seg = h.tgt_seg;
    m.src_cam = views[view].cam; m.src_seg = seg;
    m.tgt_cam = views[tgt_view].cam; m.tgt_seg = tgt_seg;
    m.overlap = h.overlap; m.score3D = h.score3D;
    m.dp1 = h.dp1; m.dp2 = h.dp2;
    m.dq1 = h.oq1; m.dq2 = h.oq2;
}

// offsets of all segments + the outputs of the segments that keep a best hypothesis: surviving matches (reference
// Match layout, canonical order), the estimated_position3D_ entry, the two depths for the view's median.
// Thread t serves segment t (offsets, hyp_of_seg) AND header slot t (pool t / hcap, index t % hcap): a kept header finds
// its place among the surviving matches of its segment by counting the kept headers before it (a segment's headers are
// contiguous and in canonical order).  One thread per SEGMENT walking its headers, as until round 4, ran ~950
// instructions per wave on 6 of 64 lanes (profiles/r04_all_kernels_pmc_C1.txt).
// A rank of a multi-GPU run with a SHARDED tail serves the segments [g0, g1) of its views and the headers of its pools
// (lp.pool0, lp.npools); off64s holds the scan over ITS segments and base64 what the ranks before it hold (surviving
// matches | best hypotheses << 32), so that everything is written at its place in the full arrays.  One GPU: g0 = 0,
// g1 = G, all pools, base64 = 0.
__global__ void k_seg_write(uint32_t g0, uint32_t g1, unsigned long long base64, const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                            const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ gseg_view,
                            const unsigned long long* __restrict__ off64s, const unsigned long long* __restrict__ best_pack,
                            const uint32_t* __restrict__ seg_of_g, const ListPools lp, const Slot* __restrict__ slots,
                            uint32_t* __restrict__ surv_off, uint32_t* __restrict__ hyp_off, Match* __restrict__ surv,
                            uint32_t* __restrict__ surv_tg, uint32_t* __restrict__ surv_sg,
                            int32_t* __restrict__ hyp_of_seg, HypRec* __restrict__ hyps, float* __restrict__ depths) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool pass_ok = !(lp.flags[0] | lp.flags[2]);
    if (t <= g1 - g0) {   // ---- the segment (and the end of the range) ----
        const uint32_t gs = g0 + t;
        const unsigned long long o = off64s[gs] + base64;
        surv_off[gs] = (uint32_t)o; hyp_off[gs] = (uint32_t)(o >> 32);
        if (gs < g1) hyp_of_seg[gs] = ((off64s[gs + 1] >> 32) != (off64s[gs] >> 32) && pass_ok) ? (int32_t)(uint32_t)(o >> 32) : -1;
    }
    // ---- the header ----
    if (!pass_ok) return;
    const uint32_t pq = t / lp.hcap, k = t - pq * lp.hcap, pool = lp.pool0 + pq;
    if (pq >= lp.npools || k >= min(lp.cnt[pool * 16 + 1], lp.hcap)) return;
    const uint32_t hi = pool * lp.hcap + k;
    const HypHdr& h = lp.hyps[hi];
    if (!(h.state & kHypKeep)) return;
    const uint32_t g = h.g;
    if ((off64s[g + 1] >> 32) == (off64s[g] >> 32)) return;  // the segment keeps no best hypothesis: nothing of it is written
    const unsigned long long o = off64s[g] + base64;
    const uint32_t view = gseg_view[g], seg = g - seg_base[view];
    const SegHdr sh = lp.segs[seg_of_g[g]];
    uint32_t w = (uint32_t)o;
    for (uint32_t y = sh.hyp_begin; y < hi; ++y) w += (lp.hyps[y].state & kHypKeep) ? 1u : 0u;
    Match m; uint32_t tv, tseg;
    make_match(views, pairs, view, seg, h, m, tv, tseg);
    surv_tg[w] = seg_base[tv] + tseg;
    surv_sg[w] = g;
    surv[w] = m;
    if (h.canon == 0xFFFFFFFFu - (uint32_t)best_pack[g]) {
        const uint32_t hx = (uint32_t)(o >> 32);
        const ViewDev& v = views[view];
        HypRec r;
        const SegX& sx = v.segx[seg];
        const Seg3 s3 = unproject(v.C, sx.r1, sx.r2, h.dp1, h.dp2);   // unprojectMatch(best,true), :1638
        r.P1[0] = s3.P1.x; r.P1[1] = s3.P1.y; r.P1[2] = s3.P1.z;
        r.P2[0] = s3.P2.x; r.P2[1] = s3.P2.y; r.P2[2] = s3.P2.z;
        r.dir[0] = s3.dir.x; r.dir[1] = s3.dir.y; r.dir[2] = s3.dir.z;
        r.length = s3.length;
        r.valid = s3.length > 0.0f ? 1u : 0u;
        r.m = m;
        r.view = view; r.pad = 0;
        hyps[hx] = r;
        depths[2 * hx] = h.dp1;
        depths[2 * hx + 1] = h.dp2;
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------
std::atomic<uint64_t> g_csr_global_launches{0};   // test hook (l3d_debug_counter): launches of the global-cursor k_pair_csr<false>

hipError_t launch_pair_csr(const PairDesc* pairs, uint32_t n_pairs, uint32_t max_Mt, const PairCsr* pair_poff,
                           const uint32_t* inv_tgt, uint32_t tgt16, uint32_t* poff, uint32_t* refs,
                           uint32_t* dummy, uint32_t tgt_v0, uint32_t tgt_v1, uint64_t max_pair_slots,
                           const uint32_t* row_start, hipStream_t st) {
    if (!n_pairs) return hipSuccess;
    // (L3D_CSR_GLOBAL=1, test hook: every pair takes the global-memory form that views beyond the LDS capacity need.  Read
    // PER CALL, as lists_run reads it when it sizes `dummy`: a process-wide static here was latched by whichever test ran
    // first, and the hook then tested nothing -- ADVICE round 4.  g_csr_global_launches lets the test see that the form ran.)
    const bool force_global = std::getenv("L3D_CSR_GLOBAL") != nullptr;
    const uint32_t lds_segs = force_global ? 0u : kCsrLdsSegs;
    const size_t lds = ((size_t)std::min(max_Mt, lds_segs) + 64) * 4;
    // the dynamic-LDS attribute is process-global state of the function: set ONCE to the largest size any launch asks for
    // (contexts on different threads would otherwise race with different sizes)
    static const hipError_t attr_rc = [] {
        const int mx = (int)(((size_t)kCsrLdsSegs + 64) * 4);
        hipError_t e = hipFuncSetAttribute((const void*)k_pair_csr<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_pair_csr<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        return e;
    }();
    if (attr_rc != hipSuccess) return attr_rc;
    // The staged form (LDS cursors, 16-bit targets).  Stage size: room for the entries of the largest pair in ONE range (about
    // 0.6 of its slots hand an inverse match over) -- beside the cursors inside 78 KiB if that fits (two workgroups per CU: C1,
    // C3), else inside one workgroup's whole LDS (C2: 0.26 against 0.35 ms with two ranges at 78 KiB); pairs beyond that take
    // several ranges (C4: 0.64 ms with 4 or with 27 of them).  L3D_CSR_STAGED=0: the direct form; L3D_CSR_STAGE_KB: the budget (A/B).
    const char* st_env = std::getenv("L3D_CSR_STAGED");
    if (tgt16 && !force_global && max_Mt <= lds_segs && !(st_env && std::atoi(st_env) == 0)) {
        const size_t cur_bytes = ((size_t)lds_segs_round(max_Mt) + 64) * 4;
        const char* kb_env = std::getenv("L3D_CSR_STAGE_KB");
        const size_t two = 78 * 1024, one = 156 * 1024 - 4096;
        const size_t want = (size_t)(0.6 * (double)max_pair_slots) * 4 + 4096;
        size_t budget = cur_bytes + want <= two ? two : one;
        if (kb_env) budget = std::min<size_t>(std::max<size_t>((size_t)std::atoi(kb_env) * 1024, cur_bytes + 4096), one);
        size_t cap = (budget - cur_bytes) / 4;
        cap = std::min<size_t>(cap, (size_t)std::max<uint64_t>(max_pair_slots, 1024));   // (never more entries than slots)
        static const hipError_t attr2 = hipFuncSetAttribute((const void*)k_pair_csr<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        if (attr2 != hipSuccess) return attr2;
        hipLaunchKernelGGL((k_pair_csr<true, true, true>), dim3(n_pairs), dim3(kCsrBlock), cur_bytes + cap * 4, st, pairs, 0u, pair_poff, inv_tgt,
                           poff, refs, dummy, tgt_v0, tgt_v1, lds_segs, (uint32_t)cap, row_start);
        return hipGetLastError();
    }
#define L3D_CSR(T16)                                                                                                      \
    do {                                                                                                                  \
        hipLaunchKernelGGL((k_pair_csr<true, T16, false>), dim3(n_pairs), dim3(kCsrBlock), lds, st, pairs, 0u, pair_poff, inv_tgt, \
                           poff, refs, dummy, tgt_v0, tgt_v1, lds_segs, 0u, row_start);                                   \
        if (max_Mt > lds_segs) { /* views beyond the LDS capacity: cursors in global memory (dummy: 64 words per pair) */ \
            hipLaunchKernelGGL((k_pair_csr<false, T16, false>), dim3(n_pairs), dim3(kCsrBlock), 0, st, pairs, 0u, pair_poff, inv_tgt, \
                               poff, refs, dummy, tgt_v0, tgt_v1, lds_segs, 0u, row_start);                               \
            g_csr_global_launches.fetch_add(1, std::memory_order_relaxed);                                                \
        }                                                                                                                 \
    } while (0)
    if (tgt16) L3D_CSR(true); else L3D_CSR(false);
#undef L3D_CSR
    return hipGetLastError();
}

// per-rank status next to the counters of the pass' first pool (words 8..12: the four flags and the scratch cursor):
// travels with the counter slab when the list pass is sharded
__global__ void k_publish_flags(const ListPools lp) {
    if (threadIdx.x < 4) lp.cnt[lp.pool0 * 16 + 8 + threadIdx.x] = lp.flags[threadIdx.x];
    if (threadIdx.x == 4) lp.cnt[lp.pool0 * 16 + 12] = lp.flags[6];
}

// flags of all ranks (published next to the counters of each rank's first pool) -> this rank's flags: the tail
// kernels discard a pass in which ANY rank overflowed
__global__ void k_merge_flags(const ListPools lp, uint32_t world) {
    const uint32_t ppr = kListPools / world;
    uint32_t f0 = 0, f2 = 0;
    for (uint32_t r = 0; r < world; ++r) { f0 |= lp.cnt[(size_t)r * ppr * 16 + 8]; f2 |= lp.cnt[(size_t)r * ppr * 16 + 10]; }
    if (f0) lp.flags[0] = 1;
    if (f2) lp.flags[2] = 1;
}

// seg_of_g of every segment header present in the pools of lp (after the slabs have arrived).  Round 6: lp.pool0 / npools are
// the pools whose RECORDS this rank holds (l3d_shard_options) -- the counters of all ranks arrive, the records of the ranks it
// does not depend on do not, and a counter without its records must not be followed
__global__ void k_seg_index(const ListPools lp, uint32_t* __restrict__ seg_of_g) {
    if (lp.flags[0] | lp.flags[2]) return;
    const uint32_t pool = lp.pool0 + blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= min(lp.cnt[pool * 16 + 2], lp.scap)) return;
    seg_of_g[lp.segs[pool * lp.scap + k].g] = pool * lp.scap + k;
}
hipError_t launch_seg_index(ListPools lp, uint32_t* seg_of_g, uint32_t G, uint32_t world, hipStream_t st) {
    hipLaunchKernelGGL(k_merge_flags, dim3(1), dim3(1), 0, st, lp, world ? world : 1u);
    hipLaunchKernelGGL(k_seg_index, dim3((lp.scap + 255) / 256, lp.npools), dim3(256), 0, st, lp, seg_of_g);
    return hipGetLastError();
}

constexpr uint32_t kLists2Grid = 16384;    // fixed grid of the two-wave tier (its workgroups stride over the hand-over list)
hipError_t launch_lists(uint32_t v0, uint32_t nv, uint32_t max_M, const ViewDev* views, const PairDesc* pairs,
                        const ListView* lviews, const OutPair* opairs, const InPair* ipairs, const uint32_t* gseg_view,
                        const uint32_t* poff, const uint32_t* inv, const float2* hyp_p, const float2* hyp_q, const Slot* slots, uint32_t uniform_K,
                        SimConst sc, ListPools lp, uint32_t* seg_of_g, HugeScratchArgs hsa, hipStream_t st) {
    if (!nv || !max_M) return hipSuccess;
    // the one-wave tier stages 128 hypotheses (8 waves per SIMD) unless the scene's lists are long on average
    // (L3D_LISTS_WIDE=0|1: A/B switch of the staging width, read per launch)
    const char* wide_env = std::getenv("L3D_LISTS_WIDE");
    const bool wide = wide_env ? std::atoi(wide_env) != 0 : hsa.mean_list > 96;
#define L3D_LISTS(B)                                                                                                       \
    do {                                                                                                                   \
        const size_t lds1 = ListCfg<1, B>::BYTES, lds2 = ListCfg<2, B>::BYTES, lds4 = ListCfg<4, B>::BYTES;                \
        hipError_t e = hipFuncSetAttribute((const void*)k_lists<1, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1); \
        if (e != hipSuccess) return e;                                                                                     \
        e = hipFuncSetAttribute((const void*)k_lists<2, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);        \
        if (e != hipSuccess) return e;                                                                                     \
        e = hipFuncSetAttribute((const void*)k_lists<4, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);        \
        if (e != hipSuccess) return e;                                                                                     \
        for (uint32_t a = 0; a < nv; a += 65535u) { /* grid.y limit */                                                     \
            const uint32_t n = nv - a < 65535u ? nv - a : 65535u;                                                          \
            hipLaunchKernelGGL((k_lists<1, B>), dim3(max_M, n), dim3(64), lds1, st, views, pairs, lviews, opairs, ipairs,  \
                               gseg_view, poff, inv, hyp_p, hyp_q, uniform_K, lp, v0 + a);                                        \
        }                                                                                                                  \
        /* lists beyond one wave's capacity: two waves up to 2x, four up to 4x (fixed grids over the hand-over lists) */   \
        hipLaunchKernelGGL((k_lists<2, B>), dim3(kLists2Grid), dim3(128), lds2, st, views, pairs, lviews, opairs, ipairs, \
                           gseg_view, poff, inv, hyp_p, hyp_q, uniform_K, lp, 0u);                                                \
        /* (the four-wave tier is left out while the passes hand it no list -- C1: an empty grid of 16 us --, like      */    \
        /* k_lists_huge below: a pass that then does hand one over is repeated with it, flags[4], check_pass)           */    \
        if (hsa.run_tier4)                                                                                                 \
        hipLaunchKernelGGL((k_lists<4, B>), dim3(kLists4Grid), dim3(256), lds4, st, views, pairs, lviews, opairs, ipairs, \
                           gseg_view, poff, inv, hyp_p, hyp_q, uniform_K, lp, 0u);                                                \
    } while (0)
    if (wide) L3D_LISTS(256); else L3D_LISTS(128);
#undef L3D_LISTS
    HugeScratch hs;
    hs.d1 = hsa.f32; hs.d2 = hsa.f32 + hsa.cap;
    hs.tv = hsa.u32; hs.ref = hsa.u32 + hsa.cap; hs.pf = hsa.u32 + 2 * (size_t)hsa.cap;
    hs.key = hsa.u64; hs.cap = hsa.cap;
    // lists beyond the four-wave tier: rare; when the previous pass over the same kind of scene handed none over the
    // launch is left out (a pass that then does hand one over is repeated with it: flags[5], l3d_api.hip check_pass)
    if (hsa.run_huge)
        hipLaunchKernelGGL(k_lists_huge, dim3(256), dim3(256), 0, st, views, pairs, lviews, opairs, ipairs, gseg_view, poff,
                           inv, hyp_p, hyp_q, lp, hs);
    hipLaunchKernelGGL(k_cand_exact, dim3((lp.ccap + 255) / 256, lp.npools), dim3(256), 0, st, views, pairs, gseg_view, sc, lp);
    hipLaunchKernelGGL(k_edges, dim3((lp.scap + kEdgesSplit - 1) / kEdgesSplit, lp.npools), dim3(64), 0, st, views, pairs,
                       gseg_view, slots, lp, seg_of_g);
    if (lp.npools < kListPools) hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(64), 0, st, lp);   // sharded pass
    return hipGetLastError();
}

static dim3 hyp_grid(const ListPools& lp) { return dim3((lp.hcap + 255) / 256, lp.npools); }   // the pools of lp: pool0 + blockIdx.y

hipError_t launch_chain_sweep(ListPools lp, uint8_t* positive, uint32_t* changed, uint32_t sweep, uint32_t* undecided, hipStream_t st) {
    if (sweep == 0) hipLaunchKernelGGL(k_chain_sweep, hyp_grid(lp), dim3(256), 0, st, lp, positive, changed, undecided);
    else hipLaunchKernelGGL(k_chain_sweep_list, dim3(kChainListBlocks, lp.npools), dim3(256), 0, st, lp, positive, changed, sweep, undecided);
    return hipGetLastError();
}
hipError_t launch_hyp_scores(ListPools lp, const uint8_t* positive, const uint32_t* gseg_view, Slot* slots,
                             const uint8_t* pair_present, uint32_t* max_score_bits, hipStream_t st) {
    hipLaunchKernelGGL(k_hyp_scores, hyp_grid(lp), dim3(256), 0, st, lp, positive, gseg_view, slots, pair_present, max_score_bits);
    return hipGetLastError();
}
hipError_t launch_hyp_filter(ListPools lp, uint32_t g0, uint32_t g1, const uint32_t* gseg_view, const uint32_t* max_score_bits,
                             uint32_t* kept_cnt, unsigned long long* best_pack, unsigned long long* cnt64,
                             hipStream_t st) {
    hipLaunchKernelGGL(k_hyp_filter, hyp_grid(lp), dim3(256), 0, st, lp, gseg_view, max_score_bits, kept_cnt, best_pack);
    if (g1 > g0) hipLaunchKernelGGL(k_seg_filter, dim3((g1 - g0 + 255) / 256), dim3(256), 0, st, g0, g1, kept_cnt, best_pack, cnt64);
    return hipGetLastError();
}
hipError_t launch_seg_write(uint32_t g0, uint32_t g1, unsigned long long base64, const ViewDev* views, const PairDesc* pairs, const uint32_t* seg_base,
                            const uint32_t* gseg_view, const unsigned long long* off64s,
                            const unsigned long long* best_pack, const uint32_t* seg_of_g, ListPools lp,
                            const Slot* slots, uint32_t* surv_off, uint32_t* hyp_off, Match* surv, uint32_t* surv_tg,
                            uint32_t* surv_sg, int32_t* hyp_of_seg, HypRec* hyps, float* depths, hipStream_t st) {
    const uint64_t n_thr = std::max<uint64_t>((uint64_t)(g1 - g0) + 1, (uint64_t)lp.npools * lp.hcap);
    hipLaunchKernelGGL(k_seg_write, dim3((uint32_t)((n_thr + 255) / 256)), dim3(256), 0, st, g0, g1, base64, views, pairs, seg_base, gseg_view,
                       off64s, best_pack, seg_of_g, lp, slots, surv_off, hyp_off, surv, surv_tg, surv_sg, hyp_of_seg,
                       hyps, depths);
    return hipGetLastError();
}

}  // namespace l3d


// ---- start-up (l3d_create): the runtime loads a translation unit's code object at the first launch of one of its
// kernels (~0.6 ms each, measured on the first matchImages of a process); an empty launch pays that at context creation
namespace l3d {
namespace { __global__ void k_warm_lists() {} }
hipError_t warm_lists(hipStream_t st) {
    hipLaunchKernelGGL(k_warm_lists, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
}  // namespace l3d
