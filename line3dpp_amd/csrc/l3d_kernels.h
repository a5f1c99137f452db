// l3d_kernels.h -- launch prototypes shared by the .hip translation units of libl3dpp_hip.so
#pragma once
#include <atomic>

#include "l3d_dev.h"

struct l3d_cledge;

namespace l3d {

extern std::atomic<uint64_t> g_knn_replay_calls;      // l3d_api.hip
extern std::atomic<uint64_t> g_keep_all_repeats;      // l3d_api.hip
extern std::atomic<uint64_t> g_csr_global_launches;   // k_lists.hip, test hook read through l3d_debug_counter

constexpr int kMatchRows = 64;   // source rows per work item of k_match_pairs (one wave64): the ROW form (keep-all modes, brute-force
                                 // hook, the accelerator seam, L3D_MATCH_TILE=0)
// The TILE form of the bounded-kNN kernel (round 5): a work item is R = 16 or 32 source rows of ONE width class, a lane is
// a (row, target) pair -- 64 / R targets per step -- and the targets come out of an LDS FIFO of the records whose band
// meets the hull of the R rows (k_match.hip).  k_cull_prepare lays the rows of a pair out class by class, every class
// padded to a multiple of R (kEmpty rows), in a region of tile_src_cap(Ms, R) positions: the host's work list is that
// region cut into items of R positions, whatever the class sizes turn out to be on the device.
constexpr uint32_t kTileClasses = 5;   // unbounded band | width beyond 1/16, 1/32, 1/64 of the target image | narrow
// Width classes only pay where a class is cut into enough items: a view of 1000 segments (C3) has 16 items of 64 rows per
// pair, and splitting them into classes that each end in a partially filled item cost its match kernel 10 % (3.11 against
// 2.81 ms), while C1's 31 items gain.  Below 24 items per pair: the unbounded rows and ONE class of bounded rows.
constexpr uint32_t kTileMinItems = 24;
__host__ __device__ constexpr uint32_t tile_classes(uint32_t Ms, uint32_t R) { return Ms / R < kTileMinItems ? 2u : kTileClasses; }
__host__ __device__ constexpr uint32_t tile_src_cap(uint32_t Ms, uint32_t R) {
    return ((Ms + tile_classes(Ms, R) * (R - 1) + R - 1) / R) * R;
}
constexpr uint64_t kMatchTileMaxItems = 2048;     // launches of up to this many 64-row items take the tile form (C0: 1 430 items,
                                                  // kernel -7 %; C1's 10 560 and C3's 81 920 lose 20 % in it)
uint32_t match_tile_rows(int mode, bool brute, uint64_t est_row_items);   // 0: row form; 16: tile form (L3D_MATCH_TILE overrides)
uint32_t match_layout_rows(int mode, bool brute, uint32_t tile_rows); // rows per work item of the padded class layout: the tile form's R, 64 for the
                                                  // row form (L3D_MATCH_CLASSES=0: 0 = its legacy layout without padding)
struct WorkItem {
    uint32_t pair;  // index into the pair array
    uint32_t src0;  // first source row (position in the pair's row order) of the work item
};

// Epipolar-band culling (k_match.hip): per directed pair, the pencil of epipolar lines in the target image
// is parametrised by tau = (A.x)/(B.x), the coordinate at which the pencil line through x crosses a fixed
// transversal through the image centre.  Source rows and target segments are ordered by tau so that a wave
// can skip whole 64-segment chunks of targets whose tau band cannot meet the bands of its rows.
struct PairCull {
    double As[3], Bs[3];   // tau of the epipolar line F*p of a source point p: (As.p)/(Bs.p), Bs.centre == 1
    double At[3], Bt[3];   // tau of a target point / direction q: (At.q)/(Bt.q), Bt.centre == 1
    uint64_t s_off;        // first entry of this pair in the source pools
    uint64_t t_off;        // first entry in the target pools
    uint32_t c_off;        // first chunk band
    uint32_t enabled;      // 0: geometry not suited (epipole in/near the image, degenerate F) -> plain streaming
    uint64_t k_off;        // views beyond the LDS sort capacity: this pair's sort keys in CullPools::big_keys (source
                           // side first, target side behind it); ~0 when both sides sort in LDS
    uint32_t w_item0;      // first work item (64 source rows each) of the pair, counted over all pairs of the context
    uint32_t sorted_copy;  // 1: k_cull_prepare also keeps the target's seg4 / SegD records in walk order (CullPools::tgt_s4 /
                           // tgt_sd) and the exact tests read those.  For views of a few thousand segments the per-view arrays
                           // stay in the L2s across the pairs of a view and gathers by original index hit; per-pair copies
                           // would only add traffic (C1: +30 % bytes for nothing).  From 4096 segments on the gathers miss
                           // (C4 read 132 x its segment records) and the copies pay.
};
constexpr uint32_t kSortedCopyMinSegs = 4096;
struct CullPools {
    const PairCull* cull;      // [n_pairs] or nullptr
    uint32_t* src_perm;        // [sum Ms] source row visited at sorted position i (tile form: [sum tile_src_cap], kEmpty = padding)
    float2* src_band;          // [sum Ms] its tau band (lo, hi)
    uint32_t* tgt_perm;        // [sum Mt] target segment at sorted position i
    float4* tgt_sf;            // [sum Mt] SegF records in sorted order
    float2* tgt_band;          // [sum Mt] their tau bands
    float2* chunk_band;        // [sum ceil(Mt/64)] tau band of each 64-record chunk
    uint64_t* big_keys;        // global sort scratch of the pairs whose views exceed kCullLdsSegs
    // longest-first order of the work items of a launch (nullptr: launch order = list order), see k_order_items
    uint32_t* item_bucket;     // [items of the launch] length class of each item
    uint32_t* item_order;      // [items of the launch] work item started by workgroup b
    uint32_t* order_done;      // workgroups of k_order_items that have filed their items (zero between launches)
    uint32_t w_base;           // PairCull::w_item0 of the first pair of the launch
    uint32_t cost_max;         // largest Mt of the launch
    // what the exact tests read of a target, in sorted order as well (the candidates of a wave are neighbours in that
    // order: gathers by original index miss the L2s on large views -- C4 read 132 x its segment records)
    float4* tgt_s4;            // [sum Mt] raw segments (x1,y1,x2,y2)
    SegD32* tgt_sd;            // [sum Mt] rays and plane normal in float (the depth decision's share of SegX, l3d_dev.h SegD32)
    uint32_t row_cache;        // set by launch_match_pairs: the work items stage their source rows' records in LDS
    uint32_t padded_rows;      // 0: src_perm / src_band hold Ms rows per pair (row form, legacy classes); R: the padded class layout
                               // of tile_src_cap(Ms, R) positions per pair (tile form: R = 16 / 32; row form with classes: 64)
};
constexpr uint32_t kOrderBuckets = 1024;
constexpr uint32_t kCullLdsSegs = 16384;   // LDS sort capacity of k_cull_prepare (larger views sort in global memory)
constexpr uint32_t kCullBandCacheSegs = 4096;   // ... sides of up to this many elements keep their bands in LDS (8 bytes each)
constexpr uint32_t kCullMaxSegs = 1u << 20; // 20 index bits in the sort keys beside class and band

// the orientation filter of phase B fused into whatever produces a slot (match epilogue, exchange expansion)
struct OrientFuse {
    uint32_t* inv_tgt;              // [n_slots] target segment of a slot that hands an inverse match to its target view
                                    // (kSlotInvAlive), kEmpty otherwise: the stream k_pair_csr sorts by target
    uint32_t tgt16;                 // 1: the stream holds 16-bit entries (0xFFFF: none) -- every view of the scene has fewer
                                    // than 65 535 segments: half the bytes the epilogue writes and k_pair_csr reads twice
    OrientThr thr;
    // rows in which the match kernel saw equal overlaps (the reference's heap order decides there): (pair, source row),
    // replayed by k_match_tied_rows
    uint32_t* tie_count;            // rows queued by this match launch
    uint2* tie_list;
    uint32_t tie_cap;
    uint32_t* tie_next;             // the counter the next match launch will use (zeroed by k_match_tied_rows)
    uint32_t* tie_total;            // rows replayed so far (diagnostics)
    // The compact hypothesis streams of phase B (round 6), written beside the slots by whatever produces a slot (nullptr: not
    // kept -- the accelerator seam): the list pass of phase B reads 8 bytes per slot where it read the 32-byte record for 8.
    float2* hyp_p;                  // [n_slots] (depth_p1, depth_p2) of an ALIVE slot (kSlotAlive), NaN otherwise: the fresh
                                    // hypotheses of the source segment
    float2* hyp_q;                  // [n_slots] (depth_q1, depth_q2): the depths of a slot's INVERSE hypothesis (read where
                                    // inv_tgt names a target); k_pair_csr carries them into the sorted order of its records
    // keep-all mode (kNN <= 0), single culled pass (round 6): the count pass leaves every accepted match of row r at
    // keep_rec[(row_off + r) * keep_cap + arrival index] as a whole 32-byte slot (flags 0: k_keep_assemble sets them and puts
    // the rows into ascending target order)
    Slot* keep_rec;                 // nullptr: counts only (the legacy two-pass form)
    uint32_t keep_cap;              // records kept per row (a longer row: the pass is repeated with a larger scratch)
    uint32_t* slot_row;             // [n_slots] source segment of a slot (ragged rows), written by k_keep_assemble
};

// ---- k_match.hip ----
size_t match_lds_bytes(int mode, uint32_t K, bool ix16 = false, uint32_t waves = 1, bool brute = false, uint32_t tile_rows = 0,
                       bool row_cache = false);
bool match_row_cache(int mode, bool brute, uint32_t nwork, uint32_t tile_rows);   // the source rows' records staged in LDS (k_match.hip)
constexpr uint32_t kMatchOrderMaxItems = 16384;   // launches up to this many work items: two waves per item
constexpr uint32_t kMatchOrderMinItems = 3584;    // ... and, from this many on (more waves than wave slots), longest-first order
uint32_t match_waves_per_group(int mode, bool brute, uint32_t nwork);   // waves that share one 64-row work item of k_match_pairs
hipError_t launch_cull_prepare(const ViewDev* views, const PairDesc* pairs, uint32_t first, uint32_t count,
                               uint32_t max_M, CullPools pools, uint32_t tile_rows, hipStream_t stream);
hipError_t launch_order_items(const PairDesc* pairs, uint32_t first, uint32_t count, uint32_t max_Mt, CullPools pools,
                              uint32_t nwork, hipStream_t stream);
hipError_t launch_match_pairs(int mode, bool brute, const ViewDev* views, const PairDesc* pairs,
                              const WorkItem* work, uint32_t nwork, uint32_t maxK, Slot* slots,
                              uint32_t* row_counts, float thr, CullPools pools, OrientFuse of, bool ix16,
                              uint32_t tile_rows, hipStream_t stream);
// rows flagged by the match kernel (equal overlaps): the reference's priority_queue order, replayed; scratch = one
// region of 2 * scratch_stride (stride >= max Mt) packed entries per workgroup of match_tied_grid(stride); cp: the culling pools of the
// match launch (cp.cull == nullptr: every target is visited)
uint32_t match_tied_grid(uint32_t scratch_stride);
// kNN beyond the LDS tables of k_match_pairs: every row of the pairs [first, first + count) is queued for
// k_match_tied_rows instead (list_base = PairDesc::row_off of pair `first`)
hipError_t launch_queue_all_rows(const PairDesc* pairs, uint32_t first, uint32_t count, uint32_t max_Ms, uint32_t list_base,
                                 OrientFuse of, hipStream_t stream);
hipError_t launch_match_tied_rows(const ViewDev* views, const PairDesc* pairs, Slot* slots, uint32_t maxK, float thr,
                                  OrientFuse of, CullPools cp, uint64_t* scratch, uint32_t scratch_stride,
                                  hipStream_t stream);
// compact exchange of slots between ranks: target indices out, full records back (bit-identical re-derivation)
hipError_t launch_pack_slot_idx(const Slot* slots, uint32_t* idx, uint64_t lo, uint64_t hi, hipStream_t stream);
hipError_t launch_expand_slot_idx(const ViewDev* views, const PairDesc* pairs, uint32_t first, uint32_t count,
                                  uint32_t max_row_slots, const uint32_t* idx, Slot* slots, OrientFuse of,
                                  hipStream_t stream);
// keep-all mode (kNN <= 0), after the count pass that kept its records (OrientFuse::keep_rec) and the scan of the row counts
// info[p] = {first slot, longest row, slots lo, slots hi}; row_pair[row] = its pair; blk_row[b] = the row that holds slot b * kKeepBlock
constexpr uint32_t kKeepBlock = 1024;
hipError_t launch_keep_pair_info(const PairDesc* pairs, uint32_t n_pairs, const uint32_t* row_counts, const uint32_t* row_start,
                                 uint4* info, uint32_t* row_pair, uint32_t* blk_row, uint32_t n_blk, uint32_t* longest, hipStream_t stream);   // *longest: the longest row of all
// (slot_cap: capacity of the output arrays -- the number of slots is read on the device, a pass with more writes nothing)
hipError_t launch_keep_assemble(const ViewDev* views, const PairDesc* pairs, uint32_t n_rows, uint64_t slot_cap, const uint32_t* row_start,
                                const uint32_t* row_pair, const uint32_t* blk_row, const uint32_t* longest, Slot* slots, OrientFuse of,
                                hipStream_t stream);
// per-segment invariants of all views
hipError_t launch_prep_views(const ViewDev* views, uint32_t n_views, uint32_t max_M, hipStream_t stream);

// ---- k_lists.hip: the sparse phase B (l3d_lists.h) ----
struct InvRec; struct ListPools; struct PairCsr;
struct HugeScratchArgs { float* f32; uint32_t* u32; uint64_t* u64; uint32_t cap; uint32_t mean_list; uint32_t run_huge; uint32_t run_tier4; };   // [2 cap] [3 cap] [cap]; mean list length (estimate); which of the rarely needed tiers are launched
hipError_t launch_scan64(const unsigned long long* in, uint32_t n, unsigned long long* out, unsigned long long* tmp,
                         unsigned long long* total, hipStream_t st);
// the inverse hypotheses of every pair that hands matches to a later view, sorted by target segment (counting sort per
// pair, one workgroup each): poff = per-pair CSR offsets over the target's segments, refs = the slot indices in that order
hipError_t launch_pair_csr(const PairDesc* pairs, uint32_t n_pairs, uint32_t max_Mt, const PairCsr* pair_csr,
                           const uint32_t* inv_tgt, uint32_t tgt16, uint32_t* poff, uint32_t* refs,
                           uint32_t* dummy, uint32_t tgt_v0, uint32_t tgt_v1, uint64_t max_pair_slots,
                           const uint32_t* row_start, hipStream_t st);   // row_start: ragged rows (l3d_lists.h), else nullptr
hipError_t launch_seg_index(ListPools lp, uint32_t* seg_of_g, uint32_t G, uint32_t world, hipStream_t st);
struct ListView; struct OutPair; struct InPair;
hipError_t launch_lists(uint32_t v0, uint32_t nv, uint32_t max_M, const ViewDev* views, const PairDesc* pairs,
                        const ListView* lviews, const OutPair* opairs, const InPair* ipairs, const uint32_t* gseg_view,
                        const uint32_t* poff, const uint32_t* inv_refs, const float2* hyp_p, const float2* hyp_q,
                        const Slot* slots, uint32_t uniform_K,
                        SimConst sc, ListPools lp, uint32_t* seg_of_g, HugeScratchArgs hsa, hipStream_t st);
hipError_t launch_chain_sweep(ListPools lp, uint8_t* positive, uint32_t* changed, uint32_t sweep, uint32_t* undecided,
                              hipStream_t st);   // sweep 0: all headers of lp, files the undecided ones per pool; later sweeps: those lists
hipError_t launch_hyp_scores(ListPools lp, const uint8_t* positive, const uint32_t* gseg_view, Slot* slots,
                             const uint8_t* pair_present, uint32_t* max_score_bits, hipStream_t st);
hipError_t launch_hyp_filter(ListPools lp, uint32_t g0, uint32_t g1, const uint32_t* gseg_view, const uint32_t* max_score_bits,
                             uint32_t* kept_cnt, unsigned long long* best_pack, unsigned long long* cnt64,
                             hipStream_t st);
hipError_t launch_seg_write(uint32_t g0, uint32_t g1, unsigned long long base64, const ViewDev* views, const PairDesc* pairs, const uint32_t* seg_base,
                            const uint32_t* gseg_view, const unsigned long long* off64s,
                            const unsigned long long* best_pack, const uint32_t* seg_of_g, ListPools lp,
                            const Slot* slots, uint32_t* surv_off, uint32_t* hyp_off, Match* surv, uint32_t* surv_tg,
                            uint32_t* surv_sg, int32_t* hyp_of_seg, HypRec* hyps, float* depths, hipStream_t st);

// ---- k_rdd.hip ----
size_t rdd_workspace_bytes(uint32_t nnz, uint32_t n_rows);
hipError_t launch_rdd(const struct ::l3d_cledge* edges_in, uint32_t nnz, uint32_t n_rows, uint32_t iterations,
                      struct ::l3d_cledge* edges_out, void* workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace l3d
