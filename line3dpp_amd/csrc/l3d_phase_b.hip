// l3d_phase_b.hip -- the host side of phase B behind l3d_match_finish (include/l3dpp_hip.h): line3D.cc:745-773 for every
// view -- scoringCPU, storeInverseMatches, filterMatches -- in the sparse form of k_lists.hip, on one GPU and sharded over
// ranks (list pass by views: l3d_lists_shard*; tail by views: l3d_tail_shard_*), and the plan of a sharded call
// (l3d_plan_shards).  Split from l3d_api.hip (context, views, l3d_match_begin, phase A).
#include "l3d_ctx.h"

extern "C" {

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
        if (!c->exch_ordered) L3D_HIP_CHECK(hipStreamSynchronize(c->stream));   // (l3d_shard_options: the caller's exchange orders itself)
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
    lp.row_start = c->ragged ? c->d_row_start.p : nullptr; lp.slot_row = c->ragged ? c->d_slot_row.p : nullptr;
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
        c->caps_saved[c->caps_mode] = l3d_ctx::PoolCaps{c->lp_ecap, c->lp_hcap, c->lp_scap, c->lp_ccap, c->huge_cap, c->huge_skip, c->list4_skip};
        const l3d_ctx::PoolCaps& pc = c->caps_saved[caps_mode];
        c->lp_ecap = pc.e; c->lp_hcap = pc.h; c->lp_scap = pc.s; c->lp_ccap = pc.c; c->huge_cap = pc.huge; c->huge_skip = pc.huge_skip;
        c->list4_skip = pc.list4_skip;
        c->caps_mode = caps_mode;
    }
    // a sharded list pass always runs k_lists_huge: whether a rank's views hold a list for it is not known to the other
    // ranks before the pass, and a rank that had to repeat the pass alone would leave the collectives of the others
    if (caps_mode == 1) { c->huge_skip = false; c->list4_skip = false; }
    hipStream_t st = c->stream;
    const uint32_t V = (uint32_t)c->order.size(), P = (uint32_t)c->pairs.size();
    // global segment ids
    c->seg_base.assign(V + 1, 0);
    for (uint32_t vi = 0; vi < V; ++vi) c->seg_base[vi + 1] = c->seg_base[vi] + c->order[vi]->M;
    const uint32_t G = c->G = c->seg_base[V];
    uint64_t max_slots = 0;
    for (uint32_t p = 0; p < P; ++p) max_slots = std::max<uint64_t>(max_slots, c->pair_slots(p));
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
    L3D_HIP_CHECK(c->d_hyp_p.reserve(std::max<uint64_t>(c->n_slots, 1))); L3D_HIP_CHECK(c->d_hyp_q.reserve(std::max<uint64_t>(c->n_slots, 1)));
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
                op.slot_off = pd.slot_off; op.tgt = pd.tgt; op.pair = p; op.K = pd.K; op.row_off = pd.row_off;
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
                                          c->tgt16, c->d_hyp_p.p, c->d_hyp_q.p, c->orient_lo, c->orient_hi, st));
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
    for (uint32_t p = 0; p < P; ++p) max_slots = std::max<uint64_t>(max_slots, c->pair_slots(p));
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
                                          c->d_csr_dummy.p, v0, v0 + nv, max_slots, c->ragged ? c->d_row_start.p : nullptr, st));
        }
    }
    g_trace.mark("pair CSRs enqueued");
    // mean list length: every alive slot is a hypothesis of its source segment and, towards a later view, of its target
    // segment too (~0.8 of the slots are alive, ~half of the pairs hand inverse matches over); exact after the first call
    const uint32_t mean_list = c->n_ents ? (uint32_t)(c->n_ents / std::max<uint32_t>(G, 1))
                                         : (uint32_t)(1.5 * (double)c->n_slots / std::max<uint32_t>(G, 1));
    const HugeScratchArgs hsa{c->d_huge_f32.p, c->d_huge_u32.p, (uint64_t*)c->d_huge_u64.p, c->huge_cap, mean_list,
                              c->huge_skip ? 0u : 1u, c->list4_skip ? 0u : 1u};
    c->huge_ran = !c->huge_skip; c->list4_ran = !c->list4_skip;
    uint32_t max_M = 0;
    for (uint32_t vi = v0; vi < v0 + nv; ++vi) max_M = std::max(max_M, c->order[vi]->M);
    L3D_HIP_CHECK(launch_lists(v0, nv, max_M, c->d_views.p, c->d_pairs.p, lviews, opairs, ipairs, c->d_gseg_view.p,
                               c->d_poff.p, c->d_inv_refs.p, c->d_hyp_p.p, c->d_hyp_q.p, c->d_slots.p, c->kNN > 0 ? (uint32_t)c->kNN : 0u, simc, lp,
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
    // the chain: over the records this shard's views depend on -- all pools on one GPU and in a replicated tail; the pools of
    // the ranks [shard_dep_rank0, rank] when the tail is sharded (l3d_shard_options; the pools of later ranks hold nothing this
    // rank's scores read, those of ranks it does not depend on need not even have arrived: their counters are zero)
    const bool dep = ts.npools < kListPools && c->shard_world > 1 && c->shard_dep_rank0 <= c->shard_rank;
    const uint32_t cp0 = dep ? c->shard_dep_rank0 * c->shard_ppr : 0u, cp1 = dep ? (c->shard_rank + 1) * c->shard_ppr : kListPools;
    const ListPools lp = list_pools(c, cp0, cp1 - cp0), lps = list_pools(c, ts.pool0, ts.npools);
    uint32_t* changed = c->d_lzero.p + z.changed;
    uint32_t* max_score = c->d_lzero.p + z.max_score;
    uint32_t* kept = c->d_lzero.p + z.kept;
    unsigned long long* best = (unsigned long long*)(c->d_lzero.p + z.best);
    if (!fresh) L3D_HIP_CHECK(hipMemsetAsync(changed, 0, (z.words - z.changed) * 4, st));
    // as many launches as the last calls needed + 3 (a launch is a no-op once nothing changes; the last one enqueued must
    // report "no change", else the host keeps sweeping -- a second host round trip, 0.16 ms on C1).  How many sweeps a
    // scene needs varies by one or two from call to call: whether a bit set in a launch is seen by a later thread of the
    // SAME launch is a race the fixed point is indifferent to, its depth is not.  Round 5 enqueued the last call's count
    // + 1 and the SECOND call of most scenes paid the extra round (profiles/r06_first_call.txt).
    const uint32_t need = std::max(std::max(c->chain_hist[0], c->chain_hist[1]), std::max(c->chain_hist[2], c->chain_hist[3]));
    const uint32_t n_sweeps = std::min(kChainSweeps, std::max(4u, need + 3));
    c->chain_enqueued = n_sweeps;
    // (the lists of undecided headers, one per pool: in the array of the surviving matches' target segments, which the tail
    // writes after the chain -- one word per header at most, reserved by lists_reserve; their lengths: word 7 of every pool's
    // counters, zero after the list pass)
    if (!fresh) L3D_HIP_CHECK(hipMemset2DAsync(c->d_lzero.p + 7, 64, 0, 4, kListPools, st));
    for (uint32_t s2 = 0; s2 < n_sweeps; ++s2)
        L3D_HIP_CHECK(launch_chain_sweep(lp, positive_of(c), changed, s2, c->d_surv_tg.p, st));
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
    const bool huge_missed = (fl[5] && !c->huge_ran) || (fl[4] && !c->list4_ran);   // (the same for the four-wave tier: flags[4])
    c->huge_skip = fl[5] == 0 && c->shard_world <= 1;
    c->list4_skip = fl[4] == 0 && c->shard_world <= 1;
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
        uint64_t ents = 0, inv = 0, cands = 0, hdrs = 0;
        for (uint32_t q = 0; q < kListPools; ++q) { ents += h[q * 16 + 5]; inv += h[q * 16 + 6]; cands += h[q * 16 + 3]; hdrs += h[q * 16 + 1]; }
        c->n_ents = (uint32_t)std::min<uint64_t>(ents, 0xFFFFFFFFu);
        c->tm.list_entries = c->n_ents;
        c->tm.list_inverse = (uint32_t)std::min<uint64_t>(inv, 0xFFFFFFFFu);
        c->tm.list_candidates = (uint32_t)std::min<uint64_t>(cands, 0xFFFFFFFFu);
        c->tm.list_headers = (uint32_t)std::min<uint64_t>(hdrs, 0xFFFFFFFFu);
        c->tm.slots_lo = (uint32_t)c->n_slots; c->tm.slots_hi = (uint32_t)(c->n_slots >> 32);
    }
    for (uint32_t s2 = 0; s2 < c->chain_enqueued; ++s2) c->tm.chain_sweeps += changed[s2] ? 1u : 0u;   // of the last round
    c->chain_need = c->tm.chain_extra_rounds ? kChainSweeps : c->tm.chain_sweeps;
    c->chain_hist[c->chain_hist_at++ & 3u] = c->chain_need;
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

int l3d_shard_options(l3d_ctx* c, uint32_t first_needed_rank, int exchanges_stream_ordered) {
    if (!c) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->shard_dep_rank0 = first_needed_rank;
    c->exch_ordered = exchanges_stream_ordered != 0;
    return L3D_OK;
}

int l3d_tail_shard_count(l3d_ctx* c, uint32_t counts[2]) {
    if (!c || !counts) return fail(L3D_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->state != l3d_ctx::BEGUN || !c->lists_ready || c->shard_world < 2)
        return fail(L3D_ERR_STATE, "l3d_tail_shard_count follows l3d_lists_shard* and the exchange of its slabs (world > 1)");
    (void)hipSetDevice(c->device);
    const int rc = [&]() -> int {
        if (c->shard_dep_rank0 > c->shard_rank) return fail(L3D_ERR_ARG, "l3d_shard_options: first_needed_rank lies above this rank");
        // (the pools whose records are present: those of the ranks this rank depends on, and its own)
        const ListPools lp = list_pools(c, c->shard_dep_rank0 * c->shard_ppr, (c->shard_rank + 1 - c->shard_dep_rank0) * c->shard_ppr);
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
        // (the caller exchanges the parts right away, possibly on another stream or through a backend that does not order
        // itself after this stream: they must be complete in device memory on return, as the slabs of l3d_lists_shard* are --
        // ADVICE round 5; not when the caller's exchanges order themselves behind the stream: l3d_shard_options)
        if (!c->exch_ordered) L3D_HIP_CHECK(hipStreamSynchronize(c->stream));
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
        const int fr = finish_commit(c);
        if (fr == L3D_OK) c->aff_parts_world = world;   // tail_base_n: where every rank's surviving matches lie (l3d_affinity_shard_begin)
        return fr;
    }();
    return close_failed_call(c, rc);
}

}  // extern "C"
