// l3d_lists.h -- records of the sparse phase B (k_lists.hip).
//
// A 2D segment of view v owns a list of L hypotheses (fresh matches of v's outgoing pairs + inverse matches handed
// over by earlier views, line3D.cc:745-773).  scoringCPU (line3D.cc:1208-1294) compares every hypothesis with every
// other one of the list, but similarityForScoring exceeds L3D_DEF_MIN_SIMILARITY_3D only for hypotheses that agree in
// depth within a fraction of a percent: on the BASELINE scenes 10^7 hypotheses have 5*10^5 supporting pairs and 98 %
// of the hypotheses have none.  Phase B therefore keeps nothing per hypothesis: one dense pass over the lists
// (k_lists) finds the supporting pairs and emits them -- EDGES, with their similarity -- together with a HEADER per
// hypothesis that has at least one; the chain of the reference (which inverse hypotheses exist), the scores, the
// filter and the outputs are all functions of those few records.
#pragma once
#include "l3d_dev.h"

namespace l3d {

// A slot (pair, src_row, j) whose match survives both orientation tests is a potential INVERSE hypothesis of its target
// segment (storeInverseMatches, line3D.cc:1672-1699).  Round 4: the slot indices of one directed pair are SORTED BY
// TARGET SEGMENT (k_pair_csr: a counting sort per pair in LDS -- no device atomics, no position word per slot, no
// 16-byte records) into `inv_refs`, with a CSR per pair over the target view's segments, stored transposed per target
// view (`poff`, ListView::pbase).  The canonical order of a segment's inverse hypotheses -- (source view, source
// segment) ascending -- is the order of their slot indices (slots are laid out by pair = by source view, then source
// row; two slots of one row never share a target): the incoming pairs of a view in ascending pair index, and inside a
// pair's run ascending slot index.  k_pair_csr leaves a run in the order its LDS atomics were served; the list pass
// ranks an entry among the handful of entries of its run and reads the two depths it needs from the slot itself.

// Round 6: the depths a hypothesis needs no longer come out of the 32-byte slots but out of two 8-byte streams written beside
// them (l3d_kernels.h: OrientFuse): hyp_p[slot] = (dp1, dp2) of an alive slot (NaN otherwise) for the fresh hypotheses, read in
// slot order; hyp_q[slot] = (dq1, dq2) for the inverse ones, gathered by the slot index of the sorted entry.  Carrying the two
// depths through the sort (16-byte entries, one contiguous load per run in the list pass) was built and measured: the list pass
// read 0.32 x the bytes and gained 5 %, the sort lost more than that whichever way the depths reached it
// (profiles/r06_ab_pair_csr.txt).

// per view / per outgoing pair of a view: what the list pass needs of ViewDev / PairDesc, packed so that a wave gets
// it with one or two loads instead of a chain of dependent ones (view -> pair list -> pair -> slot)
struct ListView {
    uint32_t seg_base, M;     // first global segment id, segments
    uint32_t q0, nq;          // outgoing pairs [q0, q0 + nq) in the OutPair table, ascending target view
    float k;                  // View::k_
    uint32_t i0, ni;          // incoming pairs that hand inverse matches over: [i0, i0 + ni) in the InPair table,
                              // ascending pair index (= ascending source view)
    uint32_t pbase;           // the view's transposed CSR offsets: poff[pbase + t * ni + q], t = 0 .. M, q = 0 .. ni - 1
};
// an incoming pair of a view: where its records start
struct InPair {
    uint32_t rec_base;        // first record of the pair (= its first slot: a pair never has more records than slots)
    uint32_t src, pair;       // source view, pair index
    uint32_t pad;
};
static_assert(sizeof(InPair) == 16, "InPair is 16 bytes");
// per directed pair, for k_pair_csr: where its column of the target view's offset table is
struct PairCsr {
    uint32_t base;            // ListView::pbase of the target view (kEmpty: the pair hands nothing over)
    uint32_t ni, q;           // incoming pairs of the target view, this pair's column
    uint32_t pad;
};
static_assert(sizeof(PairCsr) == 16, "PairCsr is 16 bytes");
struct OutPair {
    uint64_t slot_off;        // first slot of the pair
    uint32_t tgt, pair, K;    // target view, pair index, slots per source segment
    uint32_t row_off;         // first row of the pair in ListPools::row_start (ragged rows: the keep-all mode)
    uint32_t pad[2];
};
static_assert(sizeof(ListView) == 32 && sizeof(OutPair) == 32, "table records are 32 bytes");

// hypothesis i of a list is supported by hypothesis j (other camera, similarityForScoring > 0.5): sim is the value
struct EdgeRec {
    uint32_t ref_j;      // slot of the supporter (its existence bit lives in positive[ref_j] when it is inverse)
    float sim;
    uint32_t j_cam;      // canonical index of j in the list (low 16 bits) | j inverse << 16 ; cam in hyp-local order
    uint32_t tv_j;       // camera (view index) of the supporter: the per-camera maximum of scoringCPU (:1255-1274)
};
static_assert(sizeof(EdgeRec) == 16, "EdgeRec is 16 bytes");
constexpr uint32_t kEdgeInv = 1u << 16;

// a hypothesis with at least one edge.  The headers of one segment are contiguous and in canonical order, and so
// are the edges of one header (ascending j).
struct HypHdr {
    uint32_t g;          // global segment id
    uint32_t ref;        // its slot
    uint32_t pair_flags; // pair | inverse << 31
    uint32_t canon;      // canonical index in the list
    float dp1, dp2;      // its depths in this view
    uint32_t edge_begin; // global edge index
    uint32_t edge_cnt;
    float score3D;       // k_hyp_scores
    uint32_t state;      // bit0 exists (fresh, or inverse with positive source), bit1 kept by filterMatches
    // what the surviving Match record needs of the hypothesis' slot (k_edges copies it here): the tail of phase B --
    // chain, scores, filterMatches, outputs -- then works on the records alone, so that a rank of a multi-GPU run needs
    // the slots of the pairs that touch ITS views only (the records of all ranks are all-gathered, the slots are not)
    uint32_t tgt_seg;    // segment of the other view (fresh: the slot's target; inverse: the slot's source row)
    float overlap;       // Match::overlap_score_
    float oq1, oq2;      // depths of the other view's end points (fresh: dq1, dq2; inverse: the slot's dp1, dp2)
    uint32_t pad[2];
};
static_assert(sizeof(HypHdr) == 64, "HypHdr is 64 bytes");
constexpr uint32_t kHypInv = 1u << 31;
constexpr uint32_t kHypExists = 1u, kHypKeep = 2u;

// a segment with at least one header
struct SegHdr {
    uint32_t g, hyp_begin, hyp_cnt, pad;
};

// candidate pair of the list pass: hypothesis i might be supported by hypothesis j (other camera, both depths inside
// i's conservative windows).  Carries what the exact test and the records above need, so that nothing is gathered
// twice.  The candidates of one segment are contiguous.
struct CandRec {
    uint32_t ij;         // canonical indices i << 16 | j
    uint32_t ref_i, ref_j;
    uint32_t pf_i;       // pair | inverse << 31 of i
    uint32_t tvj;        // camera (view index) of j | j inverse << 31
    float a1, a2, b1, b2;   // depths of i and of j
    float sim;           // from the list pass: the bits of the segment's global index g; after k_cand_exact: the
                         // similarity if the pair passed the exact test, else -1
};
static_assert(sizeof(CandRec) == 40, "CandRec is 40 bytes");
struct CandHdr {
    uint32_t g, begin, cnt, pad;
};

// The records are allocated from kListPools independent pools (a workgroup uses pool blockIdx % kListPools): one
// hot atomic counter would serialise 10^5 reservations per call.  cnt[pool * 16 + k] = records reserved in that pool,
// k = 0 edges, 1 headers, 2 segments, 3 candidates, 4 candidate headers (may exceed the capacity: then `overflow` is
// set, the call is repeated with larger pools, and consumers clamp).
constexpr uint32_t kListPools = 256;
struct ListPools {
    uint32_t* cnt;         // [kListPools * 16]
    EdgeRec* edges;        // [kListPools * ecap]
    HypHdr* hyps;          // [kListPools * hcap]
    SegHdr* segs;          // [kListPools * scap]
    CandRec* cands;        // [kListPools * ccap]
    CandHdr* chdrs;        // [kListPools * scap]
    uint32_t ecap, hcap, scap, ccap;
    uint32_t* flags;       // [0] pool overflow, [1] list longer than 65535 hypotheses, [2] scratch overflow,
                           // [3] counters and slots disagree (internal error), [4] lists handed to the 4-wave kernel,
                           // [5] lists handed to the global-memory kernel, [6] scratch cursor of the latter,
                           // [7] lists handed to the 2-wave kernel
    uint32_t* list2;       // [G] segments for the 2-wave kernel
    uint32_t* list4;       // [G] segments for the 4-wave kernel
    uint32_t* listH;       // [G] segments for the global-memory kernel
    uint32_t count_entries; // 1: the list pass adds every list's length to cnt[pool * 16 + 5] (statistics; L3D_NO_LIST_STAT=1: off)
    uint32_t pool0, npools; // the pools this pass allocates from (all of them on one GPU; a rank's share when the list
                            // pass is sharded: the filled pools of all ranks are all-gathered slab by slab)
    // RAGGED rows (kNN <= 0, round 6): a row holds exactly its accepted matches -- row r of a pair occupies the slots
    // [row_start[row_off + r], row_start[row_off + r + 1]) and slot_row[slot] is the source segment of a slot.  nullptr: every
    // row of a pair has PairDesc::K slots (bounded kNN; the legacy keep-all path).
    const uint32_t* row_start;
    const uint32_t* slot_row;
};
// the slots of source segment `seg` in an outgoing pair: first slot and count
__host__ __device__ inline void out_row_slots(const OutPair& op, uint32_t seg, const ListPools& lp, uint64_t& row0, uint32_t& n) {
    if (lp.row_start) { const uint32_t a = lp.row_start[op.row_off + seg]; row0 = a; n = lp.row_start[op.row_off + seg + 1] - a; }
    else { row0 = op.slot_off + (uint64_t)seg * op.K; n = op.K; }
}

}  // namespace l3d
