// l3d_seam.hip -- seam layer of the C-ABI: one host-pointer entry per wrapper of the accelerator seam the reference
// already has (cudawrapper.h:54-80: match_lines_GPU, score_matches_GPU, find_collinear_segments_GPU,
// replicator_dynamics_diffusion_GPU), all with the semantics of the reference's CPU path.
#include "l3d_ctx.h"

using namespace l3d;

extern "C" {

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
    DevBuf<float4> seg4; DevBuf<ViewDev> dv; DevBuf<uint32_t> base, cnt, off, tot, lists; DevBuf<unsigned long long> tmp;
    auto cleanup = [&]() { seg4.release(); dv.release(); base.release(); cnt.release(); off.release(); tmp.release();
                           tot.release(); lists.release(); };
    const int rc = [&]() -> int {
        L3D_HIP_CHECK(seg4.reserve(M)); L3D_HIP_CHECK(dv.reserve(1)); L3D_HIP_CHECK(base.reserve(2));
        L3D_HIP_CHECK(cnt.reserve(M + 1)); L3D_HIP_CHECK(off.reserve(M + 1)); L3D_HIP_CHECK(tmp.reserve_zeroed(scan_ws_words(M, 4), 0));
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
    DevBuf<uint32_t> d_off, d_boff, d_len, d_scal, d_gv, d_long, d_max; DevBuf<unsigned long long> d_tmp; DevBuf<DEntry> dents; DevBuf<uint64_t> bits;
    DevBuf<float> d_scores;
    auto cleanup = [&]() { seg4.release(); m4.release(); rt.release(); segf.release(); segx.release(); dv.release();
                           d_off.release(); d_boff.release(); d_len.release(); d_tmp.release(); d_scal.release();
                           d_gv.release(); d_long.release(); d_max.release(); dents.release(); bits.release();
                           d_scores.release(); };
    const int rc = [&]() -> int {
        L3D_HIP_CHECK(seg4.reserve(M)); L3D_HIP_CHECK(segf.reserve(M)); L3D_HIP_CHECK(segx.reserve(M));
        L3D_HIP_CHECK(m4.reserve(n)); L3D_HIP_CHECK(rt.reserve(n)); L3D_HIP_CHECK(dv.reserve(1));
        L3D_HIP_CHECK(d_off.reserve(M + 1)); L3D_HIP_CHECK(d_boff.reserve(M + 1)); L3D_HIP_CHECK(d_len.reserve(M + 1));
        L3D_HIP_CHECK(d_tmp.reserve_zeroed(scan_ws_words(M, 4), 0)); L3D_HIP_CHECK(d_scal.reserve(4)); L3D_HIP_CHECK(d_gv.reserve(M + 1));
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
    DevBuf<SegD32> segd32;   // (float copy of rays + normals, same indexing)
    DevBuf<SegX> segx;   // one array for both views (source first), like the context's global array: the match kernel
                         // derives global segment ids from it for the phase-B counters it feeds
    DevBuf<uint32_t> inv_tgt, tie_count; DevBuf<uint2> tie_list; DevBuf<uint64_t> tie_heap;
    DevBuf<double> consts; DevBuf<ViewDev> dv; DevBuf<PairDesc> dp; DevBuf<WorkItem> dw; DevBuf<Slot> ds;
    DevBuf<PairCull> dc; DevBuf<uint64_t> ckeys; DevBuf<uint32_t> sperm, tperm; DevBuf<float2> sband, tband, cband; DevBuf<float4> tsf, ts4; DevBuf<SegD32> tsd;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) { seg4[i].release(); segf[i].release(); }
        segx.release(); segd32.release(); inv_tgt.release();
        tie_count.release(); tie_list.release(); tie_heap.release();
        consts.release(); dv.release(); dp.release(); dw.release(); ds.release();
        dc.release(); ckeys.release(); sperm.release(); tperm.release(); sband.release(); tband.release(); cband.release(); tsf.release(); ts4.release(); tsd.release();
    };
    int rc = [&]() -> int {
        ViewDev hv[2];
        double hc[24];
        L3D_HIP_CHECK(consts.reserve(24));
        L3D_HIP_CHECK(segx.reserve((size_t)Ms + Mt)); L3D_HIP_CHECK(segd32.reserve((size_t)Ms + Mt));
        for (int i = 0; i < 2; ++i) {
            L3D_HIP_CHECK(seg4[i].reserve(M[i])); L3D_HIP_CHECK(segf[i].reserve(M[i]));
            L3D_HIP_CHECK(hipMemcpy(seg4[i].p, lines[i], (size_t)M[i] * 16, hipMemcpyHostToDevice));
            std::memcpy(hc + 12 * i, A[i], 72); std::memcpy(hc + 12 * i + 9, Cc[i], 24);
            std::memcpy(hv[i].C, Cc[i], 24); std::memcpy(hv[i].RtKinv, A[i], 72);
            hv[i].seg4 = seg4[i].p; hv[i].segf = segf[i].p; hv[i].segx = segx.p + (i ? Ms : 0u); hv[i].segd32 = segd32.p + (i ? Ms : 0u);
            hv[i].M = M[i]; hv[i].cam = (uint32_t)i; hv[i].k = 0;
            hv[i].cx = 0.5f * (float)width; hv[i].cy = 0.5f * (float)height; hv[i].pad = 0;
        }
        PairDesc pd;
        std::memcpy(pd.F, F, 72);
        pd.src = 0; pd.tgt = 1; pd.Ms = Ms; pd.Mt = Mt; pd.K = (uint32_t)kNN; pd.row_off = 0; pd.slot_off = 0;
        pd.cc_dist = 0.0f; pd.flags = 0;   // (the seam entry keeps the compiler's own division / sqrt expansions)
        pair_baseline(d3{C_src[0], C_src[1], C_src[2]}, d3{C_tgt[0], C_tgt[1], C_tgt[2]}, pd);
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
        pc.k_off = ~0ull; pc.sorted_copy = Mt >= kSortedCopyMinSegs ? 1u : 0u;
        uint64_t n_keys = 0;
        if (pc.enabled && std::max(Ms, Mt) > kCullLdsSegs) {
            uint32_t a = 64, b = 64;
            while (a < Ms) a <<= 1;
            while (b < Mt) b <<= 1;
            pc.k_off = 0; n_keys = (uint64_t)a + b;
        }
        CullPools pools{};
        if (pc.enabled) {
            L3D_HIP_CHECK(dc.reserve(1)); L3D_HIP_CHECK(sperm.reserve(Ms)); L3D_HIP_CHECK(sband.reserve(Ms));
            L3D_HIP_CHECK(tperm.reserve(Mt)); L3D_HIP_CHECK(tsf.reserve(Mt)); L3D_HIP_CHECK(tband.reserve(Mt));
            L3D_HIP_CHECK(cband.reserve((Mt + 63) / 64));
            L3D_HIP_CHECK(hipMemcpy(dc.p, &pc, sizeof(pc), hipMemcpyHostToDevice));
            L3D_HIP_CHECK(ckeys.reserve(std::max<uint64_t>(n_keys, 1)));
            L3D_HIP_CHECK(ts4.reserve(Mt)); L3D_HIP_CHECK(tsd.reserve(Mt));
            pools = CullPools{dc.p, sperm.p, sband.p, tperm.p, tsf.p, tband.p, cband.p, ckeys.p};
            pools.tgt_s4 = ts4.p; pools.tgt_sd = tsd.p;
            L3D_HIP_CHECK(launch_cull_prepare(dv.p, dp.p, 0, 1, std::max(Ms, Mt), pools, 0u, 0));   // (the seam keeps the row form of the kernel)
        }
        // the kernel also applies the orientation filter (slot flags) and writes the inverse-target stream: scratch here
        L3D_HIP_CHECK(inv_tgt.reserve((size_t)Ms * pd.K));
        L3D_HIP_CHECK(tie_count.reserve(4)); L3D_HIP_CHECK(tie_list.reserve(Ms));
        L3D_HIP_CHECK(tie_heap.reserve(2 * (size_t)match_tied_grid(Mt) * std::max(Mt, 1u)));   // (l3d_kernels.h: one scratch region per workgroup)
        L3D_HIP_CHECK(hipMemset(tie_count.p, 0, 16));
        OrientFuse of{inv_tgt.p, 0u, OrientThr{-1.0, 1.0}, tie_count.p, tie_list.p, Ms, tie_count.p + 1, tie_count.p + 2, nullptr, nullptr};
        orientation_thresholds(of.thr.lo, of.thr.hi);
        L3D_HIP_CHECK(launch_match_pairs(0, false, dv.p, dp.p, dw.p, (uint32_t)work.size(), pd.K, ds.p, nullptr, thr,
                                         pools, of, Mt < 65536u && pd.K < 65536u, 0u, 0));
        // rows with equal overlaps: the reference's priority_queue order (line3D.cc:982-1007)
        L3D_HIP_CHECK(launch_match_tied_rows(dv.p, dp.p, ds.p, pd.K, thr, of, pools, tie_heap.p, std::max(Mt, 1u), 0));
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
