// k_views.hip -- phase B: the per-view chain of Line3D::computeMatches (line3D.cc:745-773), run for
// one source view at a time in ascending camID order (the order dependence is real: view v's
// hypothesis lists contain inverse matches whose presence depends on the scores of earlier views).
//
//   k_list_count / k_list_fill : gather the view's hypotheses (fresh slots of its outgoing pairs +
//        role-swapped slots of incoming pairs whose source view scored them > 0,
//        storeInverseMatches line3D.cc:1672-1699) and apply checkMatchOrientation (:811-858)
//   k_entry_prep  : canonical (= reference single-thread) list order by rank-sort on a 64-bit key,
//                   unprojection + spatial regularisers of every hypothesis (scoringCPU :1233-1248)
//   k_score       : one wave per 2D segment, O(L^2) similarityForScoring (:1417-1446) with the
//                   reference's per-camera replace/subtract accumulation (:1255-1274)
//   k_filter_*    : filterMatches (:1586-1669): 10 % of the view's best score, first strict maximum,
//                   0.75 gate, surviving lists + best 3D hypothesis per segment
//   k_median_depth: View::update_median_depth input (sorted[n/2], :1657-1668)
#include "l3d_dev.h"
#include "l3d_kernels.h"

namespace l3d {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Segment3D(C + r1*d1, C + r2*d2): view.cc:356-371 + segment3D.h:47-66
struct Seg3 {
    d3 P1, P2, dir;
    float length;
};
__device__ __forceinline__ Seg3 unproject(const double* C, const double* r1, const double* r2, float d1, float d2) {
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
__device__ __forceinline__ bool orientation_ok(const double* C, const SegX& sx, float d1, float d2) {
    const Seg3 s = unproject(C, sx.r1, sx.r2, d1, d2);
    const double dp = dot(d3{sx.rm[0], sx.rm[1], sx.rm[2]}, s.dir);
    const double ang = acos(fmin(fmax(dp, -1.0), 1.0));
    return ang > (double)kPi_1_32 && ang < (double)kPi_31_32;
}

}  // namespace

// One thread per slot of one pair touching view v.  `outgoing`: v is the pair's source (fresh
// matches, p-depths); otherwise v is the target and the slot is an inverse match (q-depths) that
// exists only if the source view kept it (kSlotAlive) and scored it > 0 (line3D.cc:1680).
__global__ void k_list_count(const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs, uint32_t pair,
                             int outgoing, Slot* __restrict__ slots, uint32_t* __restrict__ cnt) {
    const PairDesc& pd = pairs[pair];
    const uint64_t n = (uint64_t)pd.Ms * pd.K;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Slot* sp = slots + pd.slot_off + i;
    const Slot s = *sp;
    if (s.tgt_seg == kEmpty) return;
    if (outgoing) {
        const ViewDev& v = views[pd.src];
        const uint32_t seg = (uint32_t)(i / pd.K);
        const bool ok = orientation_ok(v.C, v.segx[seg], s.dp1, s.dp2);
        sp->flags = ok ? kSlotAlive : 0u;
        sp->score3D = 0.0f;
        if (ok) atomicAdd(&cnt[seg], 1u);
    } else {
        if (!(s.flags & kSlotAlive) || !(s.score3D > 0.0f)) return;
        const ViewDev& v = views[pd.tgt];
        const uint32_t seg = s.tgt_seg;
        const bool ok = orientation_ok(v.C, v.segx[seg], s.dq1, s.dq2);
        if (ok) {
            sp->flags = s.flags | kSlotInvAlive;
            atomicAdd(&cnt[seg], 1u);
        }
    }
}

__global__ void k_list_fill(const PairDesc* __restrict__ pairs, uint32_t pair, int outgoing,
                            const Slot* __restrict__ slots, const uint32_t* __restrict__ off,
                            uint32_t* __restrict__ cur, Entry* __restrict__ ents) {
    const PairDesc& pd = pairs[pair];
    const uint64_t n = (uint64_t)pd.Ms * pd.K;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Slot s = slots[pd.slot_off + i];
    if (s.tgt_seg == kEmpty) return;
    const uint32_t row = (uint32_t)(i / pd.K), j = (uint32_t)(i % pd.K);
    Entry e;
    if (outgoing) {
        if (!(s.flags & kSlotAlive)) return;
        e.seg = row; e.tgt_view = pd.tgt; e.tgt_seg = s.tgt_seg;
        e.dp1 = s.dp1; e.dp2 = s.dp2; e.dq1 = s.dq1; e.dq2 = s.dq2;
        e.key = (1ull << 52) | ((uint64_t)pd.tgt << 32) | j;
        e.origin = pd.slot_off + i;
    } else {
        if (!(s.flags & kSlotInvAlive)) return;
        e.seg = s.tgt_seg; e.tgt_view = pd.src; e.tgt_seg = row;
        e.dp1 = s.dq1; e.dp2 = s.dq2; e.dq1 = s.dp1; e.dq2 = s.dp2;
        e.key = ((uint64_t)pd.src << 32) | row;
        e.origin = ~0ull;
    }
    e.overlap = s.overlap;
    e.score3D = 0.0f;
    const uint32_t pos = off[e.seg] + atomicAdd(&cur[e.seg], 1u);
    ents[pos] = e;
}

// clear the kSlotInvAlive marks of an incoming pair after the fill (so a second matchImages call
// starts clean); folded into fill's successor for simplicity
__global__ void k_zero_u32(uint32_t* p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// single-workgroup exclusive scan of cnt[0..n) -> off[0..n], off[n] = total (also to *total)
__global__ __launch_bounds__(1024) void k_scan(const uint32_t* __restrict__ cnt, uint32_t n,
                                               uint32_t* __restrict__ off, uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? cnt[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t c = carry;
        if (i < n) off[i] = c + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (tid == 0) { off[n] = carry; if (total) *total = carry; }
}

// One thread per hypothesis of the view: rank inside its segment's list (canonical order) and the
// derived quantities scoring needs.  Reads ents (fill order), writes dents (sorted).
__global__ void k_entry_prep(const ViewDev* __restrict__ views, uint32_t vi, const Entry* __restrict__ ents,
                             const uint32_t* __restrict__ off, uint32_t n, DEntry* __restrict__ dents) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Entry e = ents[i];
    const uint32_t b = off[e.seg], en = off[e.seg + 1];
    uint32_t rank = 0;
    for (uint32_t j = b; j < en; ++j) rank += (ents[j].key < e.key) ? 1u : 0u;
    const ViewDev& v = views[vi];
    const ViewDev& vt = views[e.tgt_view];
    const SegX& sx = v.segx[e.seg];
    const Seg3 s3 = unproject(v.C, sx.r1, sx.r2, e.dp1, e.dp2);
    // scoringCPU line3D.cc:1236-1248
    const float k = v.k;
    const float sig1 = e.dp1 * k, sig2 = e.dp2 * k;
    float reg1 = 2.0f * sig1 * sig1, reg2 = 2.0f * sig2 * sig2;
    const d3 ct{vt.C[0], vt.C[1], vt.C[2]};
    const float sig1_t = (float)(norm(s3.P1 - ct) * (double)vt.k);   // View::regularizerFrom3Dpoint
    const float sig2_t = (float)(norm(s3.P2 - ct) * (double)vt.k);
    reg1 = 0.5f * (reg1 + 2.0f * sig1_t * sig1_t);
    reg2 = 0.5f * (reg2 + 2.0f * sig2_t * sig2_t);
    DEntry d;
    d.dir[0] = s3.dir.x; d.dir[1] = s3.dir.y; d.dir[2] = s3.dir.z;
    d.length = s3.length;
    d.dp1 = e.dp1; d.dp2 = e.dp2; d.dq1 = e.dq1; d.dq2 = e.dq2;
    d.reg1 = reg1; d.reg2 = reg2;
    d.tgt_view = e.tgt_view; d.tgt_seg = e.tgt_seg;
    d.overlap = e.overlap; d.score3D = 0.0f;
    d.origin = e.origin;
    d.seg = e.seg; d.keep = 0;
    dents[b + rank] = d;
}

// similarityForScoring (line3D.cc:1417-1446) for hypotheses a (the scored one) and b of the same
// 2D segment.  Decisions are taken on the float quantities the reference compares; acos/exp are
// evaluated in double and rounded to float (glibc's expf/acos differ from that by < 1 float ulp).
struct SimConst {
    float two_sigA_sqr;
    float min_sim;     // L3D_DEF_MIN_SIMILARITY_3D
};
__device__ __forceinline__ float sim_scoring(const double* dira, float lena, float adp1, float adp2, float reg1,
                                             float reg2, const double* dirb, float lenb, float bdp1, float bdp2,
                                             const SimConst sc) {
    if (lena < kEps || lenb < kEps) return 0.0f;
    const float dot_p = (float)dot(d3{dira[0], dira[1], dira[2]}, d3{dirb[0], dirb[1], dirb[2]});
    // cheap exact rejections first: |dot| small => angle far beyond any sigma; positional terms
    const float d1 = adp1 - bdp1, d2 = adp2 - bdp2;
    const float y1 = -d1 * d1 / reg1, y2 = -d2 * d2 / reg2;
    // expf(y) > 0.5 needs y > -0.6932 (ln 0.5 = -0.693147); -0.70 is a safe early-out bound
    if (sc.min_sim >= 0.5f && (y1 < -0.70f || y2 < -0.70f)) return 0.0f;
    float angle = (float)(acos((double)fmaxf(fminf(dot_p, 1.0f), -1.0f)) / M_PI * 180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    const float ya = -angle * angle / sc.two_sigA_sqr;
    if (sc.min_sim >= 0.5f && ya < -0.70f) return 0.0f;
    const float sim_a = (float)exp((double)ya);
    const float sim_p = fminf((float)exp((double)y1), (float)exp((double)y2));
    const float sim = fminf(sim_a, sim_p);
    return sim > sc.min_sim ? sim : 0.0f;
}

// One wave per 2D segment.  Lanes own hypotheses M (strided by 64); the inner loop walks all
// hypotheses M2 of the segment in canonical order from LDS.
constexpr int kScoreChunk = 64;
__global__ __launch_bounds__(256) void k_score(const uint32_t* __restrict__ off, uint32_t M,
                                               DEntry* __restrict__ dents, Slot* __restrict__ slots,
                                               uint32_t* __restrict__ max_score_bits, SimConst sc) {
    __shared__ double s_dir[4][kScoreChunk][3];
    __shared__ float s_len[4][kScoreChunk], s_dp1[4][kScoreChunk], s_dp2[4][kScoreChunk];
    __shared__ uint32_t s_cam[4][kScoreChunk];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t seg = blockIdx.x * 4 + wave;
    if (seg >= M) return;
    const uint32_t b = off[seg], L = off[seg + 1] - b;
    if (L == 0) return;
    float vmax = 0.0f;
    for (uint32_t m0 = 0; m0 < L; m0 += 64) {
        const uint32_t mi = m0 + lane;
        const bool act = mi < L;
        DEntry a;
        if (act) a = dents[b + mi];
        float score3D = 0.0f, cur = 0.0f;
        uint32_t cur_cam = kEmpty;
        for (uint32_t c0 = 0; c0 < L; c0 += kScoreChunk) {
            const uint32_t cn = min((uint32_t)kScoreChunk, L - c0);
            __builtin_amdgcn_wave_barrier();
            if (lane < cn) {
                const DEntry& o = dents[b + c0 + lane];
                s_dir[wave][lane][0] = o.dir[0]; s_dir[wave][lane][1] = o.dir[1]; s_dir[wave][lane][2] = o.dir[2];
                s_len[wave][lane] = o.length; s_dp1[wave][lane] = o.dp1; s_dp2[wave][lane] = o.dp2;
                s_cam[wave][lane] = o.tgt_view;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (act) {
                for (uint32_t j = 0; j < cn; ++j) {
                    const uint32_t cam2 = s_cam[wave][j];
                    if (cam2 == a.tgt_view) continue;
                    const float sim = sim_scoring(a.dir, a.length, a.dp1, a.dp2, a.reg1, a.reg2, s_dir[wave][j],
                                                  s_len[wave][j], s_dp1[wave][j], s_dp2[wave][j], sc);
                    // per-camera maximum with the reference's replace/subtract pattern; hypotheses
                    // of one target camera are contiguous in canonical order
                    if (cam2 == cur_cam) {
                        if (sim > cur) { score3D -= cur; score3D += sim; cur = sim; }
                    } else {
                        score3D += sim; cur = sim; cur_cam = cam2;
                    }
                }
            }
        }
        if (act) {
            dents[b + mi].score3D = score3D;
            if (a.origin != ~0ull) slots[a.origin].score3D = score3D;
            vmax = fmaxf(vmax, score3D);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
    if (lane == 0 && vmax > 0.0f) atomicMax(max_score_bits, __float_as_uint(vmax));
}

// filterMatches, line3D.cc:1602-1653: one thread per segment walks its list in canonical order.
__global__ void k_filter(const uint32_t* __restrict__ off, uint32_t M, DEntry* __restrict__ dents,
                         const uint32_t* __restrict__ max_score_bits, uint32_t* __restrict__ surv_cnt,
                         uint32_t* __restrict__ has_best, uint32_t* __restrict__ best_pos) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= M) return;
    const float max_score = __uint_as_float(*max_score_bits);
    const float lim = kMinBestScorePerc * max_score;
    const uint32_t b = off[seg], e = off[seg + 1];
    float best = 0.0f;
    uint32_t bpos = kEmpty, kept = 0;
    for (uint32_t i = b; i < e; ++i) {
        const float s = dents[i].score3D;
        if (s > 0.0f && s > lim) {
            ++kept;
            if (s > best) { best = s; bpos = i; }
        }
    }
    const bool ok = best > kMinBestScore3D;
    if (ok)
        for (uint32_t i = b; i < e; ++i) {
            const float s = dents[i].score3D;
            dents[i].keep = (s > 0.0f && s > lim) ? 1u : 0u;
        }
    surv_cnt[seg] = ok ? kept : 0u;
    has_best[seg] = ok ? 1u : 0u;
    best_pos[seg] = ok ? bpos : kEmpty;
}

// write the surviving matches (reference Match layout) and the best hypothesis of every segment
__global__ void k_filter_write(const ViewDev* __restrict__ views, uint32_t vi, const uint32_t* __restrict__ off,
                               uint32_t M, const DEntry* __restrict__ dents, const uint32_t* __restrict__ surv_off,
                               const uint32_t* __restrict__ hyp_off, const uint32_t* __restrict__ best_pos,
                               Match* __restrict__ surv, uint32_t* __restrict__ surv_tv,
                               int32_t* __restrict__ hyp_index, uint32_t hyp_base,
                               HypRec* __restrict__ hyps, float* __restrict__ depths) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= M) return;
    const ViewDev& v = views[vi];
    const uint32_t bp = best_pos[seg];
    if (bp == kEmpty) { hyp_index[seg] = -1; return; }
    uint32_t w = surv_off[seg];
    for (uint32_t i = off[seg]; i < off[seg + 1]; ++i) {
        const DEntry& d = dents[i];
        if (!d.keep) continue;
        Match m;
        m.src_cam = v.cam; m.src_seg = seg; m.tgt_cam = views[d.tgt_view].cam; m.tgt_seg = d.tgt_seg;
        m.overlap = d.overlap; m.score3D = d.score3D;
        m.dp1 = d.dp1; m.dp2 = d.dp2; m.dq1 = d.dq1; m.dq2 = d.dq2;
        surv_tv[w] = d.tgt_view;
        surv[w++] = m;
    }
    const DEntry& d = dents[bp];
    const uint32_t h = hyp_off[seg];
    HypRec r;
    const SegX& sx = v.segx[seg];
    const Seg3 s3 = unproject(v.C, sx.r1, sx.r2, d.dp1, d.dp2);   // unprojectMatch(best,true), :1638
    r.P1[0] = s3.P1.x; r.P1[1] = s3.P1.y; r.P1[2] = s3.P1.z;
    r.P2[0] = s3.P2.x; r.P2[1] = s3.P2.y; r.P2[2] = s3.P2.z;
    r.dir[0] = s3.dir.x; r.dir[1] = s3.dir.y; r.dir[2] = s3.dir.z;
    r.length = s3.length;
    r.valid = s3.length > 0.0f ? 1u : 0u;
    r.m.src_cam = v.cam; r.m.src_seg = seg; r.m.tgt_cam = views[d.tgt_view].cam; r.m.tgt_seg = d.tgt_seg;
    r.m.overlap = d.overlap; r.m.score3D = d.score3D;
    r.m.dp1 = d.dp1; r.m.dp2 = d.dp2; r.m.dq1 = d.dq1; r.m.dq2 = d.dq2;
    r.view = vi; r.pad = 0;
    hyps[hyp_base + h] = r;
    hyp_index[seg] = (int32_t)(hyp_base + h);
    depths[2 * h] = d.dp1;
    depths[2 * h + 1] = d.dp2;
}

// median = sorted(depths)[n/2] by 4-pass radix select on the (positive) float bit patterns;
// n == 0 -> L3D_EPS (line3D.cc:1658).  Single workgroup; n is at most 2*M.
__global__ __launch_bounds__(1024) void k_median_depth(const float* __restrict__ depths,
                                                       const uint32_t* __restrict__ n_hyp_ptr,
                                                       float* __restrict__ out_median) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_rank;
    const uint32_t n = 2u * (*n_hyp_ptr);
    if (n == 0) {
        if (threadIdx.x == 0) *out_median = (float)kEps;
        return;
    }
    if (threadIdx.x == 0) { s_prefix = 0; s_rank = n / 2; }
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) {
        const uint32_t shift = 8u * pass;
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t himask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t bits = __float_as_uint(depths[i]);
            if ((bits & himask) == prefix) atomicAdd(&hist[(bits >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t r = s_rank, acc = 0, d = 0;
            for (; d < 256; ++d) {
                if (acc + hist[d] > r) break;
                acc += hist[d];
            }
            s_rank = r - acc;
            s_prefix = prefix | (d << shift);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_median = __uint_as_float(s_prefix);
}

// ---- launchers --------------------------------------------------------------------------------
hipError_t launch_list_count(const ViewDev* views, const PairDesc* pairs, uint32_t pair, uint64_t nslots,
                             int outgoing, Slot* slots, uint32_t* cnt, hipStream_t st) {
    if (!nslots) return hipSuccess;
    hipLaunchKernelGGL(k_list_count, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, st, views, pairs, pair,
                       outgoing, slots, cnt);
    return hipGetLastError();
}
hipError_t launch_list_fill(const PairDesc* pairs, uint32_t pair, uint64_t nslots, int outgoing, const Slot* slots,
                            const uint32_t* off, uint32_t* cur, Entry* ents, hipStream_t st) {
    if (!nslots) return hipSuccess;
    hipLaunchKernelGGL(k_list_fill, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, st, pairs, pair, outgoing,
                       slots, off, cur, ents);
    return hipGetLastError();
}
hipError_t launch_zero_u32(uint32_t* p, uint64_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_zero_u32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, p, n);
    return hipGetLastError();
}
hipError_t launch_scan(const uint32_t* cnt, uint32_t n, uint32_t* off, uint32_t* total, hipStream_t st) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, cnt, n, off, total);
    return hipGetLastError();
}
hipError_t launch_entry_prep(const ViewDev* views, uint32_t vi, const Entry* ents, const uint32_t* off, uint32_t n,
                             DEntry* dents, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_entry_prep, dim3((n + 127) / 128), dim3(128), 0, st, views, vi, ents, off, n, dents);
    return hipGetLastError();
}
hipError_t launch_score(const uint32_t* off, uint32_t M, DEntry* dents, Slot* slots, uint32_t* max_score_bits,
                        float two_sigA_sqr, float min_sim, hipStream_t st) {
    if (!M) return hipSuccess;
    SimConst sc{two_sigA_sqr, min_sim};
    hipLaunchKernelGGL(k_score, dim3((M + 3) / 4), dim3(256), 0, st, off, M, dents, slots, max_score_bits, sc);
    return hipGetLastError();
}
hipError_t launch_filter(const uint32_t* off, uint32_t M, DEntry* dents, const uint32_t* max_score_bits,
                         uint32_t* surv_cnt, uint32_t* has_best, uint32_t* best_pos, hipStream_t st) {
    if (!M) return hipSuccess;
    hipLaunchKernelGGL(k_filter, dim3((M + 127) / 128), dim3(128), 0, st, off, M, dents, max_score_bits, surv_cnt,
                       has_best, best_pos);
    return hipGetLastError();
}
hipError_t launch_filter_write(const ViewDev* views, uint32_t vi, const uint32_t* off, uint32_t M,
                               const DEntry* dents, const uint32_t* surv_off, const uint32_t* hyp_off,
                               const uint32_t* best_pos, Match* surv, uint32_t* surv_tv, int32_t* hyp_index,
                               uint32_t hyp_base, HypRec* hyps, float* depths, hipStream_t st) {
    if (!M) return hipSuccess;
    hipLaunchKernelGGL(k_filter_write, dim3((M + 127) / 128), dim3(128), 0, st, views, vi, off, M, dents, surv_off,
                       hyp_off, best_pos, surv, surv_tv, hyp_index, hyp_base, hyps, depths);
    return hipGetLastError();
}
hipError_t launch_median_depth(const float* depths, const uint32_t* n_hyp_ptr, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_median_depth, dim3(1), dim3(1024), 0, st, depths, n_hyp_ptr, out);
    return hipGetLastError();
}

}  // namespace l3d
