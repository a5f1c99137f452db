// l3d_affinity_host.hip -- host side of the affinity part of Line3D::reconstruct3Dlines (line3D.cc:1702-1824):
// med_scene_depth_lines_, computingAffinityMatrix (:1852-1979) incl. the links to collinear segments, the matrix
// diffusion (performRDD, :2026-2076) and the hand-over to the clustering / reconstruction tail (l3d_recon.hip).
#include "l3d_ctx.h"

namespace l3d {

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
    L3D_HIP_CHECK(launch_scan(c->d_coll_cnt.p, G, c->d_coll_off.p, c->d_scan_ws.p, c->d_scal.p + 9, st));
    // (read-backs of this function: the stream is drained first and the copy is a blocking one, so that no early return
    // can leave a copy in flight towards a destination that has gone out of scope)
    uint32_t n_coll = 0;
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    L3D_HIP_CHECK(hipMemcpy(&n_coll, c->d_scal.p + 9, 4, hipMemcpyDeviceToHost));
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
        L3D_HIP_CHECK(c->d_scan_ws.reserve_zeroed(scan_ws_words(n, 4), st));
        L3D_HIP_CHECK(launch_aff_coll_count(mode, n, c->d_surv_tg.p, c->d_simv.p, c->d_hyps.p, c->d_seg_base.p,
                                            c->d_coll_off.p, c->d_item_cnt.p, st));
        L3D_HIP_CHECK(launch_scan(c->d_item_cnt.p, n, c->d_item_off.p, c->d_scan_ws.p, c->d_scal.p + 10, st));
        uint32_t total = 0;
        L3D_HIP_CHECK(hipStreamSynchronize(st));
        L3D_HIP_CHECK(hipMemcpy(&total, c->d_scal.p + 10, 4, hipMemcpyDeviceToHost));
        L3D_HIP_CHECK(c->d_item_seg.reserve(std::max<uint32_t>(total, 1)));
        L3D_HIP_CHECK(c->d_item_sim.reserve(std::max<uint32_t>(total, 1)));
        L3D_HIP_CHECK(launch_aff_coll_sim(mode, n, c->d_surv_sg.p, c->d_surv_tg.p, c->d_hyp_of_seg.p, c->d_hyps.p,
                                          c->d_views.p, c->d_seg_base.p, c->d_gseg_view.p, c->d_coll_off.p,
                                          c->d_coll_idx.p, c->d_item_off.p, c->d_vaff.p, (c->d_med + 8), (const float*)(c->d_vaff.p + V),
                                          c->two_sigA_sqr, c->d_item_seg.p, c->d_item_sim.p, st));
        off[mode].resize((size_t)n + 1); seg[mode].resize(total); sim[mode].resize(total);
        L3D_HIP_CHECK(hipStreamSynchronize(st));
        L3D_HIP_CHECK(hipMemcpy(off[mode].data(), c->d_item_off.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
        if (total) {
            L3D_HIP_CHECK(hipMemcpy(seg[mode].data(), c->d_item_seg.p, (size_t)total * 4, hipMemcpyDeviceToHost));
            L3D_HIP_CHECK(hipMemcpy(sim[mode].data(), c->d_item_sim.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        }
    }
    // ---- primary stream + hypothesis -> segment map ----
    std::vector<uint32_t> surv_off((size_t)G + 1), surv_tg(N);
    std::vector<float> simv(N);
    std::vector<int32_t> hyp_of_seg(G);
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    L3D_HIP_CHECK(hipMemcpy(surv_off.data(), c->d_surv_off.p, ((size_t)G + 1) * 4, hipMemcpyDeviceToHost));
    if (N) L3D_HIP_CHECK(hipMemcpy(surv_tg.data(), c->d_surv_tg.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (N) L3D_HIP_CHECK(hipMemcpy(simv.data(), c->d_simv.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (G) L3D_HIP_CHECK(hipMemcpy(hyp_of_seg.data(), c->d_hyp_of_seg.p, (size_t)G * 4, hipMemcpyDeviceToHost));
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

// med_scene_depth_lines_ + computingAffinityMatrix (line3D.cc:1759-1778) in the CURRENT (translated) frame, in three
// pieces so that the similarity pass -- the arithmetic of the step -- can be sharded by views over the ranks of a
// multi-GPU run (l3d_affinity_shard_begin / _finish below; SURVEY.md 8e "Affinity"):
//   affinity_prepare   med_scene_depth_lines_, the per-view table, the buffers
//   affinity_sim       similarity of the candidates [lo, hi) (all of them on one GPU)
//   affinity_rest      used_ / local ids / CLEdge pairs from the similarities of ALL candidates, the one read-back
static int affinity_prepare(l3d_ctx* c) {
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size();
    c->edges.clear(); c->l2g.clear();
    c->aff_n_edges = 0; c->aff_n_rows = 0; c->aff_host_valid = true;
    // med_scene_depth_lines_, line3D.cc:1759-1774
    std::vector<float> sd;
    for (auto* v : c->order) if (v->median_depth > kEps) sd.push_back(v->median_depth);
    if (!sd.empty()) { std::sort(sd.begin(), sd.end()); c->med_scene_depth_lines = sd[sd.size() / 2]; }
    else c->med_scene_depth_lines = 0.0f;
    // [ViewAff x V | med_scene_depth_lines]: one table, sent only when it differs from what the device holds
    std::vector<ViewAff> va(V + 1);
    for (uint32_t vi = 0; vi < V; ++vi) { va[vi].k = c->order[vi]->k; va[vi].pad = 0; }
    va[V].k = c->med_scene_depth_lines; va[V].pad = 0;
    const uint32_t N = c->n_surv, H = c->n_hyps;
    L3D_HIP_CHECK(c->d_scal.reserve(16));
    L3D_HIP_CHECK(c->d_scan_ws.reserve_zeroed(scan_ws_words((size_t)c->G + 2 * (size_t)N, 4), st));
    if (c->ev_on(6)) L3D_HIP_CHECK(hipEventRecord(c->ev[6], st));
    if (N > 0 && H > 0) {
        L3D_HIP_CHECK(c->d_vaff.reserve(V + 1));
        L3D_HIP_CHECK(c->d_simv.reserve(N)); L3D_HIP_CHECK(c->d_ca.reserve(N)); L3D_HIP_CHECK(c->d_cb.reserve(N));
        L3D_HIP_CHECK(c->d_flag.reserve(N + 1)); L3D_HIP_CHECK(c->d_epos.reserve(N + 1));
        L3D_HIP_CHECK(c->d_first_touch.reserve(H));
        L3D_HIP_CHECK(c->d_scan_ws.reserve_zeroed(scan_ws_words(2 * (size_t)N, 4), st));
        L3D_HIP_CHECK(upload_table(c->d_vaff, c->h_vaff, va.data(), ((size_t)V + 1) * sizeof(ViewAff), c->up_vaff, st));
    }
    return L3D_OK;
}

static int affinity_sim(l3d_ctx* c, uint32_t lo, uint32_t hi) {
    const uint32_t V = (uint32_t)c->order.size(), N = c->n_surv, H = c->n_hyps;
    if (!(N > 0 && H > 0)) return L3D_OK;
    L3D_HIP_CHECK(launch_aff_sim(N, lo, hi, c->d_surv_sg.p, c->d_surv_tg.p, c->d_hyp_of_seg.p, c->d_hyps.p, c->d_vaff.p,
                                 (c->d_med + 8), (const float*)(c->d_vaff.p + V), c->two_sigA_sqr, c->d_simv.p, c->d_ca.p, c->d_cb.p,
                                 c->stream));
    return L3D_OK;
}

static int affinity_rest(l3d_ctx* c) {
    hipStream_t st = c->stream;
    const uint32_t N = c->n_surv, H = c->n_hyps;
    bool counts_pending = false;
    if (N > 0 && H > 0) {
        if (c->collinearity_t > (float)kEps) {
            const int rc = affinity_collinear(c);
            if (rc) return rc;
            if (c->ev_on(7)) L3D_HIP_CHECK(hipEventRecord(c->ev[7], st));
            L3D_HIP_CHECK(hipStreamSynchronize(st));
            c->tm.affinity_ms = ev_ms(c, 6, 7);
            c->affinity_done = true;
            return L3D_OK;
        }
        L3D_HIP_CHECK(c->d_touch_flag.reserve(2 * (size_t)N + 1));
        L3D_HIP_CHECK(launch_aff_flag(N, c->d_surv_off.p, c->d_surv_sg.p, c->d_surv_tg.p, c->d_simv.p, c->d_ca.p,
                                      c->d_cb.p, c->d_flag.p, c->d_first_touch.p, H, c->d_touch_flag.p, st));
        L3D_HIP_CHECK(launch_scan(c->d_flag.p, N, c->d_epos.p, c->d_scan_ws.p, c->d_scal.p + 3, st));
        // no read-back of the edge count: everything downstream is sized by its upper bound N (flags beyond the
        // 2E touched positions stay zero), the two counts are read once at the end
        {
            L3D_HIP_CHECK(c->d_touch_flag.reserve(2 * (size_t)N + 1));
            L3D_HIP_CHECK(c->d_touch_rank.reserve(2 * (size_t)N + 1));
            L3D_HIP_CHECK(c->d_edges.reserve(2 * (size_t)N));
            L3D_HIP_CHECK(c->d_l2g.reserve(H));
            // (first_touch = none and touch_flag = 0 were set by k_aff_flag on its way)
            L3D_HIP_CHECK(launch_aff_touch(N, c->d_flag.p, c->d_epos.p, c->d_ca.p, c->d_cb.p, c->d_first_touch.p, st));
            L3D_HIP_CHECK(launch_aff_mark(H, c->d_first_touch.p, c->d_touch_flag.p, st));
            L3D_HIP_CHECK(launch_scan(c->d_touch_flag.p, 2 * N, c->d_touch_rank.p, c->d_scan_ws.p, c->d_scal.p + 4, st));
            L3D_HIP_CHECK(launch_aff_emit(N, c->d_flag.p, c->d_epos.p, c->d_ca.p, c->d_cb.p, c->d_simv.p,
                                          c->d_first_touch.p, c->d_touch_rank.p, c->d_hyps.p, c->d_edges.p,
                                          c->d_l2g.p, st));
            L3D_HIP_CHECK(c->h_cnt.reserve(16));
            L3D_HIP_CHECK(hipMemcpyAsync(c->h_cnt.p + 12, c->d_scal.p + 3, 8, hipMemcpyDeviceToHost, st));
            counts_pending = true;
        }
    }
    if (c->ev_on(7)) L3D_HIP_CHECK(hipEventRecord(c->ev[7], st));
    L3D_HIP_CHECK(hipStreamSynchronize(st));
    if (counts_pending) { c->aff_n_edges = 2 * c->h_cnt.p[12]; c->aff_n_rows = c->h_cnt.p[13]; c->aff_host_valid = false; }
    c->tm.affinity_ms = ev_ms(c, 6, 7);
    c->affinity_done = true;
    return L3D_OK;
}

int affinity_core(l3d_ctx* c) {
    int rc = affinity_prepare(c);
    if (rc == L3D_OK) rc = affinity_sim(c, 0u, c->n_surv);
    if (rc == L3D_OK) rc = affinity_rest(c);
    return rc;
}

// host copies of A_ / local2global_ (fetched on first use)
int ensure_affinity_host(l3d_ctx* c) {
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

}  // namespace l3d

using namespace l3d;

extern "C" {

int l3d_compute_affinity(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::MATCHED) return fail(L3D_ERR_STATE, "matchImages must precede the affinity step");
    if (c->aff_shard_open) return fail(L3D_ERR_STATE, "a sharded affinity fill is open: l3d_affinity_shard_finish (or _abort) first");
    // translate()/untranslate() (line3D.cc:1749,1820) only move camera centres, which the affinity terms never
    // read; they are applied to keep the host state identical to the reference's.
    translate(*c);
    const int rc = affinity_core(c);
    untranslate(*c);
    return rc;
}

// ---- the affinity fill sharded by views (N > 1 ranks; SURVEY.md 8e "Affinity": independent per hypothesis) ----------------
// After a call closed by l3d_tail_shard_commit every rank knows where every rank's views' surviving matches lie in the
// (replicated) arrays: tail_base_n.  The similarity of a candidate (line3D.cc:1467-1553: the fp64 acos / exp arithmetic
// of the step) is computed by the rank that owns the candidate's source view; the float values are exchanged in place
// (the caller: line3dpp_amd/dist.py compute_affinity_sharded), and the bookkeeping that follows (used_, local ids,
// CLEdge pairs: a few small passes over flags) runs on every rank, so every rank ends with the same A_.
int l3d_affinity_shard_begin(l3d_ctx* c, uint32_t rank, uint32_t world, void** simv, uint64_t* first, uint64_t* count) {
    if (!c || !simv || !first || !count) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::MATCHED) return fail(L3D_ERR_STATE, "matchImages must precede the affinity step");
    if (c->aff_shard_open) return fail(L3D_ERR_STATE, "l3d_affinity_shard_begin: a sharded affinity fill is already open (views translated)");
    if (world < 2 || rank >= world || c->aff_parts_world != world || c->tail_base_n.size() != (size_t)world + 1 ||
        c->tail_base_n[world] != c->n_surv)
        return fail(L3D_ERR_STATE, "l3d_affinity_shard_begin follows a call closed by l3d_tail_shard_commit with the same world size");
    if (c->collinearity_t > (float)kEps)
        return fail(L3D_ERR_LIMIT, "the links to collinear segments are sequential by definition (line3D.cc:1904-1974): use l3d_compute_affinity");
    translate(*c);
    int rc = affinity_prepare(c);
    if (rc == L3D_OK) rc = affinity_sim(c, c->tail_base_n[rank], c->tail_base_n[rank + 1]);
    // (the caller posts the exchange of the similarities right away, possibly on another stream or through a backend that
    // does not order itself after this stream: the part must be complete in device memory on return -- ADVICE round 5)
    if (rc == L3D_OK && !c->exch_ordered && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(L3D_ERR_HIP, "l3d_affinity_shard_begin: stream synchronisation failed");
    if (rc != L3D_OK) { untranslate(*c); return rc; }
    *simv = c->d_simv.p;
    for (uint32_t r = 0; r < world; ++r) { first[r] = c->tail_base_n[r]; count[r] = c->tail_base_n[r + 1] - c->tail_base_n[r]; }
    c->aff_shard_open = true;
    return L3D_OK;
}

int l3d_affinity_shard_finish(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::MATCHED || !c->aff_shard_open)
        return fail(L3D_ERR_STATE, "l3d_affinity_shard_finish follows l3d_affinity_shard_begin and the exchange of the similarities");
    c->aff_shard_open = false;
    const int rc = affinity_rest(c);
    untranslate(*c);
    return rc;
}

// closes an open sharded fill WITHOUT the bookkeeping pass (a rank whose peer failed: the other ranks' similarities never
// arrived, the caller goes on to l3d_compute_affinity): views untranslated, nothing else touched
int l3d_affinity_shard_abort(l3d_ctx* c) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!c->aff_shard_open) return L3D_OK;
    c->aff_shard_open = false;
    (void)hipStreamSynchronize(c->stream);
    untranslate(*c);
    return L3D_OK;
}

// Line3D::reconstruct3Dlines, line3D.cc:1702-1824
int l3d_reconstruct_3d_lines(l3d_ctx* c, uint32_t visibility_t, int perform_diffusion, float collinearity_t,
                             int use_CERES, uint32_t max_iter_CERES) {
    (void)max_iter_CERES;
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::MATCHED || c->n_hyps == 0)
        return fail(L3D_ERR_STATE, "no clusterable segments! forgot to match lines?");   // line3D.cc:1712-1718
    if (c->aff_shard_open) return fail(L3D_ERR_STATE, "a sharded affinity fill is open: l3d_affinity_shard_finish (or _abort) first");
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

}  // extern "C"
