// l3d_access.hip -- accessors of the C-ABI: host copies of matches_, estimated_position3D_, A_ / local2global_ and the
// SparseMatrix COO form in the reference's own layouts (commons.h:186-203, segment3D.h:99-115, clustering.h:47-51,
// sparsematrix.cc:8-60), per-view k / median depth, timings.
#include "l3d_ctx.h"

using namespace l3d;

extern "C" {

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
        if (c->ragged) {
            // ragged rows on the device (keep-all mode, k_keep_assemble): handed out in the padded Ms x K form of this
            // interface -- a row's matches in ascending target order, then empty slots up to the pair's longest row
            std::vector<uint32_t> rs((size_t)pd.Ms + 1);
            L3D_HIP_CHECK(hipMemcpy(rs.data(), c->d_row_start.p + pd.row_off, rs.size() * 4, hipMemcpyDeviceToHost));
            std::vector<l3d_slot> rows(rs[pd.Ms] - rs[0]);
            if (!rows.empty()) L3D_HIP_CHECK(hipMemcpy(rows.data(), c->d_slots.p + rs[0], rows.size() * sizeof(l3d_slot), hipMemcpyDeviceToHost));
            l3d_slot empty{};
            empty.tgt_seg = kEmpty;
            for (uint64_t i = 0; i < m; ++i) {
                const uint32_t r = (uint32_t)(i / pd.K), j = (uint32_t)(i % pd.K);
                out[i] = j < rs[r + 1] - rs[r] ? rows[rs[r] - rs[0] + j] : empty;
            }
            return L3D_OK;
        }
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

// test hook: process-wide counters that let a test see WHICH form of a kernel ran (the global-cursor form
// of k_pair_csr was selected by an environment variable that had been latched before the test set it)
unsigned long long l3d_debug_counter(const char* name) {
    if (name && std::string(name) == "csr_global_launches") return g_csr_global_launches.load(std::memory_order_relaxed);
    if (name && std::string(name) == "knn_replay_calls") return g_knn_replay_calls.load(std::memory_order_relaxed);
    if (name && std::string(name) == "keep_all_repeats") return g_keep_all_repeats.load(std::memory_order_relaxed);
    return ~0ull;
}

int l3d_set_timing_level(l3d_ctx* c, int level) {
    if (!c || level < 0 || level > 2) return fail(L3D_ERR_ARG, "l3d_set_timing_level: level 0, 1 or 2");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->timing_level = level;
    return L3D_OK;
}

int l3d_get_timings(l3d_ctx* c, l3d_timings* t) {
    if (!c || !t) return fail(L3D_ERR_ARG, "null argument");
    *t = c->tm;   // (nothing here waits for the GPU: tied_rows arrives with the read-back of l3d_match_finish)
    return L3D_OK;
}

}  // extern "C"
