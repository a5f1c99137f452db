// l3d_ctx.h -- the context behind the C-ABI (include/l3dpp_hip.h): the slice of class L3DPP::Line3D that drives the
// hot path (line3D.h:61-424), shared by the translation units of the host side:
//   l3d_api.hip            context layer: addImage / matchImages (begin, pairs, exchange, finish)
//   l3d_affinity_host.hip  the affinity part of reconstruct3Dlines, collinear links, diffusion, reconstruction tail
//   l3d_access.hip         accessors (matches_, estimated_position3D_, A_, SparseMatrix, timings)
//   l3d_output.hip         get3Dlines and the result writers (TXT / OBJ / STL)
//   l3d_seam.hip           the accelerator-seam entries (cudawrapper.h:54-80) with CPU-path semantics
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <unordered_set>

#include "l3d_host.h"
#include "l3d_lists.h"
#include "l3d_recon.h"

struct l3d_ctx;

namespace l3d {

// ---- prototypes of the launchers in k_views.hip / k_scan.hip / k_affinity.hip -------------------------------
// k_scan.hip: single-launch exclusive scans; ws = scan_ws_words(n, bytes per element) zeroed 64-bit words
size_t scan_ws_words(size_t n, uint32_t bytes_per_element);
hipError_t launch_scan(const uint32_t* in, uint32_t n, uint32_t* out, unsigned long long* ws, uint32_t* total, hipStream_t);
hipError_t launch_collin(int pass, const ViewDev* views, uint32_t n_views, uint32_t max_M, const uint32_t* seg_base,
                         float collin_t, uint32_t* cnt, const uint32_t* coll_off, uint32_t* coll_idx, hipStream_t);
hipError_t launch_aff_coll_count(int mode, uint32_t n_items, const uint32_t* surv_tg, const float* simv, const HypRec*,
                                 const uint32_t* seg_base, const uint32_t* coll_off, uint32_t* cnt, hipStream_t);
hipError_t launch_aff_coll_sim(int mode, uint32_t n_items, const uint32_t* surv_sg, const uint32_t* surv_tg,
                               const int32_t* hyp_of_seg, const HypRec*, const ViewDev*, const uint32_t* seg_base,
                               const uint32_t* gseg_view, const uint32_t* coll_off, const uint32_t* coll_idx,
                               const uint32_t* item_off, const ViewAff*, const float* medians, const float* msdl,
                               float two_sigA_sqr, uint32_t* out_seg, float* out_sim, hipStream_t);
hipError_t launch_seam_entries(uint32_t n, const float4* m4, const float2* rt, const ViewDev*, float k, DEntry*, hipStream_t);
hipError_t launch_seam_all_present(uint32_t G, const uint32_t* off, const uint32_t* boff, uint64_t* bits, hipStream_t);
hipError_t launch_seam_scores_out(uint32_t n, const DEntry*, float* scores, hipStream_t);
hipError_t launch_fill_gseg_view(const uint32_t* seg_base, uint32_t V, uint32_t max_M, uint32_t* gseg_view, hipStream_t);
hipError_t launch_orient_pairs(const ViewDev*, const PairDesc*, uint32_t n_pairs, uint64_t max_slots, Slot* slots,
                               uint32_t* inv_tgt, uint32_t tgt16, float2* hyp_p, float2* hyp_q, double thr_lo, double thr_hi, hipStream_t);
hipError_t launch_bits_len(uint32_t G, const uint32_t* off, uint32_t* len, uint32_t* long_list, uint32_t* n_long,
                           hipStream_t);
hipError_t launch_support_long(uint32_t n_long, const uint32_t* long_list, const uint32_t* off, const uint32_t* boff,
                               const DEntry*, uint64_t* bits, const ViewDev*, const uint32_t* gseg_view, SimConst,
                               hipStream_t);
hipError_t launch_support_all(uint32_t g0, uint32_t G, const uint32_t* off, const uint32_t* boff, const DEntry*, uint64_t* bits,
                              const ViewDev*, const uint32_t* seg_base, const uint32_t* gseg_view, SimConst, hipStream_t);
hipError_t launch_score_all(uint32_t g0, uint32_t G, const uint32_t* off, const uint32_t* boff, const uint32_t* gseg_view, DEntry*,
                            const uint64_t* bits, Slot*, uint32_t* max_score_bits, const ViewDev*, const uint32_t* seg_base,
                            SimConst, hipStream_t);
hipError_t launch_median_all(uint32_t V, const float* depths, const uint32_t* hyp_off, const uint32_t* seg_base,
                             const uint32_t* tie_total, uint32_t* tie_out, float* out, const uint32_t* rb_src,
                             uint32_t* rb_host, uint32_t rb_words, uint32_t* rb_count, hipStream_t, uint32_t v0 = 0);
hipError_t launch_aff_sim(uint32_t N, uint32_t lo, uint32_t hi, const uint32_t* surv_sg, const uint32_t* surv_tg, const int32_t* hyp_of_seg,
                          const HypRec*, const ViewAff*, const float* medians, const float* msdl, float two_sigA_sqr,
                          float* simv, int32_t* ca, int32_t* cb, hipStream_t);
hipError_t launch_aff_flag(uint32_t N, const uint32_t* surv_off, const uint32_t* surv_sg, const uint32_t* surv_tg,
                           const float* simv, const int32_t* ca, const int32_t* cb, uint32_t* flag,
                           uint32_t* first_touch, uint32_t H, uint32_t* touch_flag, hipStream_t);
hipError_t launch_fill_u32(uint32_t*, uint32_t n, uint32_t val, hipStream_t);
hipError_t launch_aff_touch(uint32_t N, const uint32_t* flag, const uint32_t* epos, const int32_t* ca,
                            const int32_t* cb, uint32_t* first_touch, hipStream_t);
hipError_t launch_aff_mark(uint32_t H, const uint32_t* first_touch, uint32_t* touch_flag, hipStream_t);
hipError_t launch_aff_emit(uint32_t N, const uint32_t* flag, const uint32_t* epos, const int32_t* ca,
                           const int32_t* cb, const float* simv, const uint32_t* first_touch,
                           const uint32_t* touch_rank, const HypRec*, void* edges, void* local2global, hipStream_t);

// start-up launches, one per translation unit with kernels (see l3d_create)
hipError_t warm_affinity(hipStream_t); hipError_t warm_lists(hipStream_t); hipError_t warm_match(hipStream_t);
hipError_t warm_rdd(hipStream_t); hipError_t warm_scan(hipStream_t); hipError_t warm_views(hipStream_t);

// ---- host helpers shared by the translation units (defined in l3d_api.hip) ----
int fail(int code, const std::string& msg);
const char* last_error_cstr();
void init_view(HostView& v, const double K[9], const double R[9], const double t[3]);
void translate_view(HostView& v, const d3& d);
d3 scene_translation(const std::vector<HostView*>& order);
void neighbors_from_worldpoints(const std::map<uint32_t, HostView*>& views, uint32_t num_neighbors);   // l3d_neighbors.hip
void translate(::l3d_ctx& c);      // Line3D::translate, line3D.cc:500-536
void untranslate(::l3d_ctx& c);    // line3D.cc:539-545
void make_cull(const double F[9], double ws, double hs, double wt, double ht, PairCull& pc);
void pair_baseline(const d3& Cs, const d3& Ct, PairDesc& pd);
void orientation_thresholds(double& lo, double& hi);
SimConst sim_thresholds(float two_sigA_sqr);
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
extern HostTrace g_trace;   // l3d_api.hip

float ev_ms(hipEvent_t a, hipEvent_t b);
float ev_ms(const ::l3d_ctx* c, int a, int b);   // 0 when the timing level left one of the two events out
int affinity_core(::l3d_ctx* c);           // l3d_affinity_host.hip
int ensure_affinity_host(::l3d_ctx* c);    // l3d_affinity_host.hip
std::string output_filename(::l3d_ctx* c, int max_image_width);   // l3d_output.hip

}  // namespace l3d

// shared by l3d_api.hip and l3d_phase_b.hip (C linkage: defined inside their extern "C" blocks)
extern "C" void abort_match(::l3d_ctx* c);
extern "C" void collect_match_timing(::l3d_ctx* c);

using namespace l3d;   // host-side translation units of this library only (never included by users of the C-ABI)

// (global namespace: the C-ABI's opaque handle type; its members are l3d:: types)
struct l3d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::recursive_mutex mu;                                  // view_mutex_/view_reserve_mutex_ stand-in
    std::map<uint32_t, std::unique_ptr<HostView>> views;   // views_ (ascending camID)
    std::vector<HostView*> order;                   // view index -> view (ascending camID)
    std::vector<float> views_avg_depths;            // views_avg_depths_
    // params (matchImages)
    float sigma_p = 2.5f, sigma_a = 10.0f, two_sigA_sqr = 200.0f, epipolar_overlap = 0.25f;
    float const_regularization_depth = -1.0f, med_scene_depth = (float)kEps, med_scene_depth_lines = 0.0f;
    int kNN = 10, num_neighbors = 10;
    bool fixed3Dregularizer = false;
    bool brute = false;                             // test hook: disable the fp32 pre-filter
    bool knn_replay = false;                        // kNN beyond the LDS top-K tables of k_match_pairs: every row goes
                                                    // through k_match_tied_rows (the reference's heap, any kNN <= 4096)
    double orient_lo = -1.0, orient_hi = 1.0;       // dp window equivalent to acos(dp) in (PI/32, 31PI/32)
    d3 translation{0, 0, 0};
    // state
    enum { IDLE, BEGUN, MATCHED } state = IDLE;
    bool affinity_done = false;
    std::vector<PairDesc> pairs;
    std::vector<uint32_t> pair_src_cam, pair_tgt_cam;
    std::vector<char> pair_done;
    DevBuf<uint8_t> d_pair_present;                 // pair_done on the device for the tail of a sharded call (k_hyp_scores)
    uint64_t n_slots = 0, pair_tests = 0;
    uint32_t n_rows_total = 0;
    // device
    DevBuf<ViewDev> d_views;
    DevBuf<PairDesc> d_pairs;
    DevBuf<WorkItem> d_work;
    DevBuf<Slot> d_slots;
    DevBuf<uint32_t> d_slot_idx;   // compact exchange form of d_slots (N > 1 ranks): target index per slot
    // epipolar-band culling pools (l3d_kernels.h)
    std::vector<PairCull> cull;
    DevBuf<PairCull> d_cull;
    DevBuf<uint32_t> d_src_perm, d_tgt_perm;
    DevBuf<float2> d_src_band, d_chunk_band, d_tgt_band;
    DevBuf<float4> d_tgt_sf, d_tgt_s4;
    DevBuf<SegD32> d_tgt_sd;
    DevBuf<uint32_t> d_item_bucket, d_item_order, d_order_done;   // longest-first launch order (k_order_items)
    DevBuf<uint64_t> d_cull_keys;   // sort scratch of pairs whose views exceed the LDS sort capacity
    bool use_cull = true;
    unsigned visibility_t = 3;                      // visibility_t_ / perform_RDD_ of the last reconstruct3Dlines
    bool perform_rdd = false;
    // A_ / local2global_ stay on the device; the host copies (edges, l2g) are fetched on first use
    uint32_t aff_n_edges = 0, aff_n_rows = 0;
    bool aff_host_valid = true;
    PinnedBuf<uint32_t> h_cnt;
    std::vector<uint8_t> pair_counted;   // the pair's slots carry orientation flags and are in the phase-B counters
    DevBuf<SegX> d_gsegx;                           // SegX of every segment, global segment order
    DevBuf<SegD32> d_gsegd32;                       // float copy of rays + plane normal (stage 1 of the match kernel's pipeline)
    float collinearity_t = -1.0f;                   // collinearity_t_ (reconstruct3Dlines); > 0: collinear links
    DevBuf<uint32_t> d_coll_cnt, d_coll_off, d_coll_idx, d_item_cnt, d_item_off, d_item_seg;
    DevBuf<float> d_item_sim;
    // pinned staging of the small host->device tables (reused across calls; every public call ends synchronised)
    PinnedBuf<ViewDev> h_views;
    PinnedBuf<PairDesc> h_pairs;
    PinnedBuf<PairCull> h_cull;
    PinnedBuf<WorkItem> h_work;
    UploadTag up_views, up_pairs, up_cull, up_seg_base, up_ltab;   // what the device tables hold (upload_table)
    std::vector<unsigned char> begin_sig; bool begin_sig_valid = false;   // what the pair list was built from (l3d_match_begin)
    uint64_t cull_tot[4] = {0, 0, 0, 0};            // pool sizes of the culling set-up of that list
    uint32_t layout_rows = 0;                       // padded class layout of the source pools: rows per work item (0: Ms rows per pair, legacy)
    uint32_t tile_rows = 0;                         // rows per work item of the bounded-kNN launches of this call (0: the row form; k_match.hip)
    uint64_t pairs_version = 0;                     // bumped whenever the pair list on the device changes
    struct { uint64_t version = ~0ull; uint32_t first = 0, count = 0, rows = 0; size_t items = 0; const void* dev = nullptr; } work_key;   // d_work holds the items of these pairs
    PinnedBuf<uint32_t> h_segb;
    const void* gseg_view_for = nullptr;            // d_gseg_view was filled for the seg_base the device holds
    bool timing_pending = false;                    // phase-A events recorded but not read yet
    uint32_t pending_launches = 0;
    DevBuf<uint32_t> d_row_counts;
    // keep-all mode (kNN <= 0), single culled pass with RAGGED rows (round 6; k_match.hip: k_keep_assemble)
    bool ragged = false;                            // this call's slot buffer is ragged: row r of pair p = slots
                                                    // [d_row_start[row_off + r], d_row_start[row_off + r + 1])
    uint32_t keep_cap = 0;                          // records per row of the scratch (grows to the longest row seen)
    DevBuf<Slot> d_keep_rec;                        // [n_rows_total * keep_cap] accepted matches by (row, arrival index)
    DevBuf<uint32_t> d_row_pair, d_blk_row;         // [n_rows_total] pair of a row; [slot blocks] row that holds a block's first slot
    DevBuf<uint32_t> d_row_start, d_slot_row;       // [n_rows_total + 2], [n_slots]
    DevBuf<uint4> d_keep_info;                      // [pairs] {first slot, longest row, slots lo, slots hi}
    std::vector<uint64_t> pair_nslots;              // slots of every pair (ragged calls)
    uint64_t keep_last_total = 0;                   // slots of the previous keep-all call (sizes the outputs before the pass)
    PinnedBuf<uint4> h_keep_info;
    uint64_t pair_slots(uint32_t p) const { return ragged ? pair_nslots[p] : (uint64_t)pairs[p].Ms * pairs[p].K; }
    // rows with equal overlaps (k_match_tied_rows): counter, (pair, row) list, heap scratch of the replay kernel
    DevBuf<uint32_t> d_tie_count;
    uint32_t tie_seq = 0;                           // match launches so far: which of the two queue counters is current
    DevBuf<uint2> d_tie_list;
    DevBuf<uint64_t> d_tie_heap;
    // phase B (global over all views; G = sum of M)
    uint32_t G = 0, n_ents = 0, n_surv = 0, n_hyps = 0;
    std::vector<uint32_t> seg_base;                 // [V+1]
    DevBuf<uint32_t> d_seg_base, d_gseg_view, d_scal;
    DevBuf<uint32_t> d_surv_off, d_hyp_off, d_surv_tg, d_surv_sg;
    DevBuf<uint32_t> d_inv_tgt;                      // [n_slots] target segment of an inverse-alive slot, kEmpty otherwise
    // the compact hypothesis streams of phase B (round 6; l3d_kernels.h: OrientFuse): 8 bytes per slot each, written beside
    // the slots
    DevBuf<float2> d_hyp_p, d_hyp_q;
    uint32_t tgt16 = 0;                             // ... as 16-bit entries (every view below 65 535 segments; set by l3d_match_begin)
    DevBuf<uint32_t> d_poff, d_csr_dummy;           // per-pair CSR offsets over the target's segments (k_pair_csr)
    uint32_t poff_total = 0, n_in_pairs = 0;        // entries of d_poff / of the InPair table of the running call
    DevBuf<Match> d_surv;
    DevBuf<int32_t> d_hyp_of_seg;
    DevBuf<float> d_depths;
    float* d_med = nullptr;                         // 8 words of 64-bit totals, then [V] medians: inside the zero block (d_lzero),
                                                    // behind its head, so that ONE copy reads both back (l3d_api.hip: zero_layout)
    DevBuf<HypRec> d_hyps;
    // sparse phase B (k_lists.hip, l3d_lists.h)
    DevBuf<unsigned long long> d_cnt64, d_off64s, d_scan_ws, d_huge_u64;
    DevBuf<uint32_t> d_inv_refs;                    // inverse hypotheses of each pair sorted by target segment: slot indices
    DevBuf<uint32_t> d_lzero, d_list2, d_list4, d_listH, d_seg_of_g, d_huge_u32;
    DevBuf<float> d_huge_f32;
    DevBuf<EdgeRec> d_ledges;
    DevBuf<HypHdr> d_lhyps;
    DevBuf<SegHdr> d_lsegs;
    DevBuf<CandRec> d_lcands;
    DevBuf<CandHdr> d_lchdrs;
    uint32_t lp_ecap = 0, lp_hcap = 0, lp_scap = 0, lp_ccap = 0, huge_cap = 0;
    uint32_t lp_attempts = 0;
    bool huge_skip = false, huge_ran = true;        // k_lists_huge left out while the passes hand it no lists
    bool list4_skip = false, list4_ran = true;      // the same for the four-wave tier k_lists<4>
    // The pool capacities (and huge_skip) are kept across calls, separately for unsharded calls [0] and for calls whose
    // list pass is sharded over ranks [1]: the slabs the ranks all-gather must have one size on every rank, and the
    // sharded set only ever changes by decisions every rank takes alike (check_pass sees all ranks' counters), whatever
    // unsharded calls a rank's context has served in between.  caps_mode = the set the members above hold.
    struct PoolCaps { uint32_t e = 0, h = 0, s = 0, c = 0, huge = 0; bool huge_skip = false, list4_skip = false; } caps_saved[2];
    int caps_mode = 0;
    uint32_t chain_need = 8, chain_enqueued = 10;   // chain launches that changed something last time / enqueued this time
    uint32_t chain_hist[4] = {6, 6, 6, 6}, chain_hist_at = 0;   // ... in the last four calls (a fresh context: 6 + 2 launches)
    // list pass sharded over ranks (l3d_lists_shard): world size of the running call, slabs received
    uint32_t shard_world = 0;
    // this rank's share of a sharded list pass (l3d_lists_shard*): views [shard_v0, shard_v1), pools [shard_pool0, + shard_ppr)
    uint32_t shard_rank = 0, shard_v0 = 0, shard_v1 = 0, shard_pool0 = 0, shard_ppr = 0;
    // sharded tail (l3d_tail_shard_*): counts of all ranks -> where every rank's outputs start in the full arrays
    std::vector<uint32_t> tail_base_n, tail_base_h;
    bool tail_counted = false, tail_written = false;
    uint32_t shard_dep_rank0 = 0;                   // l3d_shard_options: lowest rank whose records this rank's chain needs
    bool exch_ordered = false;                      // ... the caller's exchanges order themselves behind the stream: no host waits
    uint32_t aff_parts_world = 0;                    // world size of the last call closed by l3d_tail_shard_commit (tail_base_n valid), else 0
    bool aff_shard_open = false;                    // between l3d_affinity_shard_begin and _finish (views translated)
    bool lists_ready = false, lists_prepared = false;   // per-pool capacities (grow on overflow, kept across calls)
    PinnedBuf<uint32_t> h_fin;
    PinnedBuf<char> h_ltab;
    DevBuf<char> d_ltab;   // [ListView x V | OutPair x P | InPair x (pairs that hand inverse matches over) | pair_poff x P]
    std::vector<uint32_t> h_surv_off, h_hyp_off;    // lazily fetched for the accessors
    bool host_offsets_valid = false;
    // affinity
    DevBuf<ViewAff> d_vaff;
    DevBuf<float> d_simv;
    PinnedBuf<ViewAff> h_vaff; UploadTag up_vaff;
    DevBuf<int32_t> d_ca, d_cb;
    DevBuf<uint32_t> d_flag, d_epos, d_first_touch, d_touch_flag, d_touch_rank;
    DevBuf<l3d_cledge> d_edges;
    DevBuf<l3d_segment2d> d_l2g;
    std::vector<l3d_cledge> edges;
    std::vector<l3d_segment2d> l2g;
    std::vector<ReconLine> lines3D;                 // lines3D_ (original frame)
    bool lines_done = false;
    // timings.  Every hipEventRecord between two kernels costs a ~6 us bubble on the stream (rocprofv3 kernel trace of C1:
    // gaps exactly where the ten events of a call sit, none between other back-to-back kernels): timing_level 2 records
    // all of them (profiling, opt-in), 1 -- THE DEFAULT since round 5: what a facade user of matchImages gets is the fast
    // setting -- only the pair around the match kernel (ev[4], ev[5]), 0 none (l3d_set_timing_level)
    hipEvent_t ev[10] = {};
    int timing_level = 1;
    bool ev_on(int k) const { return timing_level >= 2 || (timing_level == 1 && (k == 4 || k == 5)); }
    l3d_timings tm{};
};
