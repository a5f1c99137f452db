// k_affinity.hip -- Line3D::computingAffinityMatrix (line3D.cc:1852-1979, collinearity off) with
// Line3D::similarity (:1467-1553), unused() (:1982-2002) and getLocalID() (:2005-2023).
//
// The reference walks estimated_position3D_ x surviving matches sequentially (mutex-serialised
// under OpenMP) and its outputs depend on that order: the first candidate of an unordered segment
// pair whose similarity exceeds 0.5 emits the edge, and matrix row ids are handed out in
// first-touch order.  Here every candidate is evaluated in parallel and the sequential semantics
// (single-thread order: views ascending, segments ascending, list order) are reproduced by
// construction:
//   k_aff_sim    similarity of every (hypothesis, surviving match) candidate
//   k_aff_flag   a candidate emits unless the reverse candidate precedes it and also passes
//   k_scan       edge index = exclusive scan of the emit flags (k_views.hip)
//   k_aff_touch  first-touch position of every 2D segment = atomicMin over its edge endpoints
//   k_aff_mark / k_scan / k_aff_emit   row id = rank of the first touch; write CLEdge pairs
#include "l3d_dev.h"
#include "l3d_kernels.h"

namespace l3d {

namespace {

// Segment3D::distance_Point2Line, segment3D.h:69-73: P1 + (dir * v^T) * dir with Eigen's evaluation
// order (outer product first, then row sums left to right)
__device__ __forceinline__ float dist_p2l(const HypRec& s, const double* P) {
    const d3 p1{s.P1[0], s.P1[1], s.P1[2]}, dir{s.dir[0], s.dir[1], s.dir[2]}, p{P[0], P[1], P[2]};
    const d3 v = p - p1;
    const d3 h{p1.x + (((dir.x * v.x) * dir.x + (dir.x * v.y) * dir.y) + (dir.x * v.z) * dir.z),
               p1.y + (((dir.y * v.x) * dir.x + (dir.y * v.y) * dir.y) + (dir.y * v.z) * dir.z),
               p1.z + (((dir.z * v.x) * dir.x + (dir.z * v.y) * dir.y) + (dir.z * v.z) * dir.z)};
    return (float)norm(h - p);
}

__device__ __forceinline__ float expf_ref(float y) { return (float)exp((double)y); }

// Line3D::similarity(s1,m1,seg2,truncate=false), line3D.cc:1467-1553
__device__ __forceinline__ float sim_affinity(const HypRec& h1, const HypRec& h2, float k1, float md1, float k2,
                                              float md2, float med_scene_depth_lines, float two_sigA_sqr) {
    if (h1.length < kEps || h2.length < kEps) return 0.0f;
    const float dot_p = (float)dot(d3{h1.dir[0], h1.dir[1], h1.dir[2]}, d3{h2.dir[0], h2.dir[1], h2.dir[2]});
    float angle = (float)(acos((double)fmaxf(fminf(dot_p, 1.0f), -1.0f)) / M_PI * 180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    const float sim_a = expf_ref(-angle * angle / two_sigA_sqr);
    float cutoff1 = md1, cutoff2 = md2;
    if (med_scene_depth_lines > kEps) {
        cutoff1 = fminf(cutoff1, med_scene_depth_lines);
        cutoff2 = fminf(cutoff2, med_scene_depth_lines);
    }
    const float d11 = dist_p2l(h2, h1.P1), d12 = dist_p2l(h2, h1.P2);
    const float d21 = dist_p2l(h1, h2.P1), d22 = dist_p2l(h1, h2.P2);
    const float sig11 = (h1.m.dp1 > cutoff1) ? cutoff1 * k1 : h1.m.dp1 * k1;
    const float sig12 = (h1.m.dp2 > cutoff1) ? cutoff1 * k1 : h1.m.dp2 * k1;
    const float reg11 = 2.0f * sig11 * sig11, reg12 = 2.0f * sig12 * sig12;
    const float sig21 = (h2.m.dp1 > cutoff2) ? cutoff2 * k2 : h2.m.dp1 * k2;
    const float sig22 = (h2.m.dp2 > cutoff2) ? cutoff2 * k2 : h2.m.dp2 * k2;
    const float reg21 = 2.0f * sig21 * sig21, reg22 = 2.0f * sig22 * sig22;
    const float sim_p1 = fminf(expf_ref(-d11 * d11 / reg11), expf_ref(-d12 * d12 / reg12));
    const float sim_p2 = fminf(expf_ref(-d21 * d21 / reg21), expf_ref(-d22 * d22 / reg22));
    return fminf(sim_a, fminf(sim_p1, sim_p2));
}

}  // namespace

// candidate c = surviving match c of the global pool (views ascending, segments ascending, list order)
__global__ void k_aff_sim(uint32_t N, const uint32_t* __restrict__ surv_sg, const uint32_t* __restrict__ surv_tg,
                          const int32_t* __restrict__ hyp_of_seg, const HypRec* __restrict__ hyps,
                          const ViewAff* __restrict__ va, const float* __restrict__ medians,
                          const float* __restrict__ msdl_ptr, float two_sigA_sqr, float* __restrict__ simv,
                          int32_t* __restrict__ cand_a, int32_t* __restrict__ cand_b) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int32_t ha = hyp_of_seg[surv_sg[c]], hb = hyp_of_seg[surv_tg[c]];
    float sim = 0.0f;
    if (ha >= 0 && hb >= 0) {
        const uint32_t v1 = hyps[ha].view, v2 = hyps[hb].view;
        sim = sim_affinity(hyps[ha], hyps[hb], va[v1].k, medians[v1], va[v2].k, medians[v2], *msdl_ptr, two_sigA_sqr);
    }
    simv[c] = sim;
    cand_a[c] = ha;
    cand_b[c] = hb;
}

__global__ void k_aff_flag(uint32_t N, const uint32_t* __restrict__ surv_off, const uint32_t* __restrict__ surv_sg,
                           const uint32_t* __restrict__ surv_tg, const float* __restrict__ simv,
                           const int32_t* __restrict__ cand_a, const int32_t* __restrict__ cand_b,
                           uint32_t* __restrict__ flag) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    uint32_t f = 0;
    if (simv[c] > kMinAffinity) {
        f = 1;
        if (cand_b[c] < cand_a[c]) {
            // the target segment's own hypothesis was visited earlier: if its list holds the reverse
            // match and that one passed, the pair is already `used` (line3D.cc:1988)
            const uint32_t sg = surv_sg[c], tg = surv_tg[c];
            for (uint32_t i = surv_off[tg]; i < surv_off[tg + 1]; ++i) {
                if (surv_tg[i] == sg) {
                    if (simv[i] > kMinAffinity) f = 0;
                    break;
                }
            }
        }
    }
    flag[c] = f;
}

__global__ void k_fill_u32(uint32_t* p, uint32_t n, uint32_t val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = val;
}

__global__ void k_aff_touch(uint32_t N, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ epos,
                            const int32_t* __restrict__ cand_a, const int32_t* __restrict__ cand_b,
                            uint32_t* __restrict__ first_touch) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N || !flag[c]) return;
    const uint32_t k = epos[c];
    atomicMin(&first_touch[cand_a[c]], 2u * k);
    atomicMin(&first_touch[cand_b[c]], 2u * k + 1u);
}

__global__ void k_aff_mark(uint32_t H, const uint32_t* __restrict__ first_touch, uint32_t* __restrict__ touch_flag) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const uint32_t p = first_touch[h];
    if (p != kEmpty) touch_flag[p] = 1u;
}

struct CLEdgeDev { int32_t i, j; float w; };
struct Seg2D { uint32_t cam, seg; };

__global__ void k_aff_emit(uint32_t N, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ epos,
                           const int32_t* __restrict__ cand_a, const int32_t* __restrict__ cand_b,
                           const float* __restrict__ simv, const uint32_t* __restrict__ first_touch,
                           const uint32_t* __restrict__ touch_rank, const HypRec* __restrict__ hyps,
                           CLEdgeDev* __restrict__ edges, Seg2D* __restrict__ local2global) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N || !flag[c]) return;
    const uint32_t k = epos[c];
    const int32_t ha = cand_a[c], hb = cand_b[c];
    const uint32_t ta = first_touch[ha], tb = first_touch[hb];
    const int32_t id1 = (int32_t)touch_rank[ta], id2 = (int32_t)touch_rank[tb];
    const float w = simv[c];
    edges[2 * k] = CLEdgeDev{id1, id2, w};
    edges[2 * k + 1] = CLEdgeDev{id2, id1, w};
    if (ta == 2u * k) local2global[id1] = Seg2D{hyps[ha].m.src_cam, hyps[ha].m.src_seg};
    if (tb == 2u * k + 1u) local2global[id2] = Seg2D{hyps[hb].m.src_cam, hyps[hb].m.src_seg};
}

// ---- launchers ---------------------------------------------------------------------------------
static inline dim3 grid1(uint32_t n, uint32_t b = 256) { return dim3((n + b - 1) / b); }

hipError_t launch_aff_sim(uint32_t N, const uint32_t* surv_sg, const uint32_t* surv_tg, const int32_t* hyp_of_seg,
                          const HypRec* hyps, const ViewAff* va, const float* medians, const float* msdl,
                          float two_sigA_sqr, float* simv, int32_t* ca, int32_t* cb, hipStream_t st) {
    if (!N) return hipSuccess;
    hipLaunchKernelGGL(k_aff_sim, grid1(N, 128), dim3(128), 0, st, N, surv_sg, surv_tg, hyp_of_seg, hyps, va, medians,
                       msdl, two_sigA_sqr, simv, ca, cb);
    return hipGetLastError();
}
hipError_t launch_aff_flag(uint32_t N, const uint32_t* surv_off, const uint32_t* surv_sg, const uint32_t* surv_tg,
                           const float* simv, const int32_t* ca, const int32_t* cb, uint32_t* flag, hipStream_t st) {
    if (!N) return hipSuccess;
    hipLaunchKernelGGL(k_aff_flag, grid1(N), dim3(256), 0, st, N, surv_off, surv_sg, surv_tg, simv, ca, cb, flag);
    return hipGetLastError();
}
hipError_t launch_fill_u32(uint32_t* p, uint32_t n, uint32_t val, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_fill_u32, grid1(n), dim3(256), 0, st, p, n, val);
    return hipGetLastError();
}
hipError_t launch_aff_touch(uint32_t N, const uint32_t* flag, const uint32_t* epos, const int32_t* ca,
                            const int32_t* cb, uint32_t* first_touch, hipStream_t st) {
    if (!N) return hipSuccess;
    hipLaunchKernelGGL(k_aff_touch, grid1(N), dim3(256), 0, st, N, flag, epos, ca, cb, first_touch);
    return hipGetLastError();
}
hipError_t launch_aff_mark(uint32_t H, const uint32_t* first_touch, uint32_t* touch_flag, hipStream_t st) {
    if (!H) return hipSuccess;
    hipLaunchKernelGGL(k_aff_mark, grid1(H), dim3(256), 0, st, H, first_touch, touch_flag);
    return hipGetLastError();
}
hipError_t launch_aff_emit(uint32_t N, const uint32_t* flag, const uint32_t* epos, const int32_t* ca,
                           const int32_t* cb, const float* simv, const uint32_t* first_touch,
                           const uint32_t* touch_rank, const HypRec* hyps, void* edges, void* local2global,
                           hipStream_t st) {
    if (!N) return hipSuccess;
    hipLaunchKernelGGL(k_aff_emit, grid1(N), dim3(256), 0, st, N, flag, epos, ca, cb, simv, first_touch, touch_rank,
                       hyps, (CLEdgeDev*)edges, (Seg2D*)local2global);
    return hipGetLastError();
}

}  // namespace l3d
