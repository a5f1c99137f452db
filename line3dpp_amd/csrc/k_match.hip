// k_match.hip -- phase A: all-pairs epipolar matching of 2D segments for directed view pairs.
//
// Replaces Line3D::matchingCPU (line3D.cc:900-1015) + mutualOverlap (:1086-1165) +
// triangulationDepths (:1168-1193) and the reference's K_match_lines/host-heap pipeline
// (cudawrapper.cu:186-253, 549-658) with a design that never materialises the Ms x Mt matrix:
//
//   workgroup  = (directed pair, 64 source segments in epipolar-band order), one wave64 -- or two that split the
//                target chunks and share the rows' top-K tables (template parameter WPG)
//   lane       = one source segment; its two epipolar lines live in VGPRs (fp32, unit normal,
//                image-centre origin)
//   k_cull_prepare orders rows and targets of the pair by epipolar band (tau); the wave visits only the
//                64-target chunks, and inside them only the targets, whose band meets its rows' bands
//                (exact culling), centre-out from where the targets' bands begin where its rows' do;
//                a target record reaches all lanes as scalar operands (s_load through the constant address space:
//                the record index is wave-uniform) -- no LDS tile, no barrier, no lane broadcast.
//                Round 5: the source rows are laid out by WIDTH CLASS of their bands, every class padded to a multiple
//                of the item size, so that a work item's hull never straddles two classes (C4 -8.8 %, C2 -3.9 %, C1
//                -2.6 %); with two waves per item the rows' own records are staged in LDS once per item (row cache).
//                The TILE form (template parameter TILE = 16: 16 rows x 4 targets per step, the records through an LDS
//                FIFO instead of the scalar cache) halves the pre-filter's lane-tests and takes the scalar-cache misses
//                from 27 % to 4 %; it is 20 % slower on C1 and C3 because every wave carries fixed work
//                (profiles/r05_ab_match_forms.txt) and 7 % FASTER where the row form's launch leaves the machine
//                half empty (C0, the reference's testdata: 1 430 items; profiles/r06_ab_small_views.txt): the form of
//                launches of up to kMatchTileMaxItems row-form items (round 6).  R = 32 lost everywhere and is gone.
//   k_order_items (launches with few items per wave slot) starts the longest work items first
//   fp32 pre-filter (conservative) -> __ballot -> popcount-prefix compaction of
//                the few survivors into a per-wave LDS ring
//   ring holds >= 64 candidates -> the wave drains 64 of them, one per lane, through the EXACT
//                double-precision pair test (l3d_dev.h exact_pair, reference CPU semantics)
//   accepted candidates are inserted into the per-source-segment top-K in LDS (kNN selection by
//                (overlap desc, tgt asc)); the K-th best overlap feeds back into the lane's
//                pre-filter threshold, so the candidate rate decays as the row fills
//   epilogue   = rank the <= K winners of each row and write the fixed-slot row (32 B slots), with the orientation
//                filter of phase B and its hypothesis counters fused in; rows with equal overlaps are replayed in the
//                reference's heap order by k_match_tied_rows
//
// Roofline: compulsory HBM traffic per directed pair is 16*(Ms+Mt) B read + 32*K*Ms B written;
// per pair test that is < 0.2 B against ~30 fp32 VALU issues, so the kernel is VALU-issue bound
// (DESIGN.md §roofline).  No MFMA: there is no contraction here.
#include <algorithm>
#include <cstdlib>

#include "l3d_dev.h"
#include "l3d_heap.h"
#include "l3d_kernels.h"

namespace l3d {

namespace {

constexpr int kBlock = kMatchRows;  // rows per work item = lanes of a wave64; a workgroup is WPG (1 or 2) such waves
constexpr int kRing = 256;          // candidate ring (entries); >= 63 + 2*64 (drained after every two pushes)
constexpr int kRing1 = 128;         // the same for the staged one-wave-per-item variant, which drains after EVERY push
                                    // (>= 63 + 64): its LDS is then 5.5 KiB per wave at K = 10 = 11 allocation granules
                                    // of 512 B, and 28 such waves fit a CU's 160 KiB (7 per SIMD; 6.3 KiB gave 6)
__host__ __device__ constexpr int ring_entries(bool staged, uint32_t waves) { return (staged && waves == 1) ? kRing1 : kRing; }
constexpr int kRing2 = 128;         // second ring (bounded kNN): candidates that passed the depth test; >= 63 + 64
constexpr int kFifo = 128;          // tile form: FIFO of in-band target records (>= 64 / R - 1 left over + 64 of a chunk)

// All LDS pointers carry the LDS address space in their TYPE: a generic pointer that travels through a struct
// or a lambda capture loses it and every access becomes a flat_load/flat_store (plus, for volatile, sc0 sc1
// and an immediate s_waitcnt) instead of ds_read/ds_write.
#define L3D_LDS __attribute__((address_space(3)))
// HIP's __ballot(p) is icmp(int(p) != 0): the flag is first turned into 0/1 in a VGPR and compared again (two VALU
// instructions per ballot); the builtin takes the comparison's lane mask as it is
#define L3D_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
// diagnostics builds only (-DL3D_STATS: candidate counters, slow; -DL3D_CYCLES: per-work-item timeline)
#if defined(L3D_STATS) || defined(L3D_CYCLES)
__device__ unsigned long long g_stats[16];   // 0 pre-filter tests, 1 candidates, 2 passed overlap, 3 accepted, 4 drains,
                                            // 5 (row, target) pairs whose OWN bands intersect (what a per-row walk would
                                            // test), 6 slots kept, 7 work items, 8 stage-1 drains (depth decision), 9 candidates
                                            // that passed it (1 = candidates into stage 1, 4 = stage-2 drains (exact overlap))
__device__ unsigned long long g_cycles[1 << 16][8];   // per work item (its first wave): start, main-loop duration, time in
                                                      // stage 1 / stage 2 of the candidate pipeline, epilogue, wave end - start, in
                                                      // shader cycles (s_memtime); [6] = XCC/CU id, [7] = SIMD/wave slot (HW_ID)
#endif
#ifdef L3D_STATS
#define L3D_STAT(i, n) atomicAdd(&g_stats[i], (unsigned long long)(n))
#else
#define L3D_STAT(i, n) ((void)0)
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// IX16: target indices, row counts and slot positions fit 16 bits (every Mt and K below 65536 -- always true when
// the pair is culled, kCullMaxSegs = 16384): 6.5 KiB of LDS per wave at K = 10 instead of 8 KiB, i.e. 24 resident
// waves per CU (6 per SIMD) instead of 20
template <bool IX16> struct IdxT { typedef uint32_t type; };
template <> struct IdxT<true> { typedef uint16_t type; };
template <bool IX16>
struct Lds {
    typedef typename IdxT<IX16>::type idx_t;
    L3D_LDS volatile uint32_t* ring;   // [waves][ring_entries]
    L3D_LDS volatile uint32_t* ring2;  // [waves][kRing2] (bounded kNN only)
    L3D_LDS volatile float* minov;     // [kBlock]
    L3D_LDS volatile uint32_t* claim;  // [kBlock]
    L3D_LDS volatile float* top_ov;    // [kBlock*K]
    L3D_LDS volatile idx_t* cnt;       // [kBlock]
    L3D_LDS volatile idx_t* minpos;    // [kBlock] slot of the worst entry of a full row; top bit (kTie): the row saw equal
                                       // overlaps where the reference's heap order decides (written under the row's lock)
    L3D_LDS volatile idx_t* top_ix;    // [kBlock*K]
    L3D_LDS v4f* fifo_rec;             // tile form: [kFifo] records of the targets whose band meets the wave's rows' hull
    L3D_LDS uint32_t* fifo_pos;        // tile form: [kFifo] their walk-order positions
    L3D_LDS v4f* row_sd;               // row cache: [rows][3] = SegD32 of the item's source rows, seg4.xyz in its padding
    L3D_LDS float* row_w;              // row cache: [rows] seg4.w
    static constexpr idx_t kTie = (idx_t)((idx_t)1 << (8 * sizeof(idx_t) - 1));
};

// rows: rows of a work item (kBlock in the row form, R in the tile form); fifo: the tile form's record FIFO (first: the
// base of the dynamic LDS is 16-byte aligned, and so are its 16-byte records)
template <bool IX16>
__device__ __forceinline__ Lds<IX16> carve(L3D_LDS char* base, uint32_t K, uint32_t waves, bool two_rings, uint32_t rows = kBlock,
                                           bool fifo = false, bool row_cache = false) {
    typedef typename IdxT<IX16>::type idx_t;
    Lds<IX16> l;
    l.fifo_rec = (L3D_LDS v4f*)base; if (fifo) base += kFifo * sizeof(v4f);
    l.row_sd = (L3D_LDS v4f*)base; if (row_cache) base += (size_t)rows * 3 * sizeof(v4f);
    l.row_w = (L3D_LDS float*)base; if (row_cache) base += (size_t)rows * sizeof(float);
    l.fifo_pos = (L3D_LDS uint32_t*)base; if (fifo) base += kFifo * sizeof(uint32_t);
    l.ring = (L3D_LDS volatile uint32_t*)base; base += waves * ring_entries(two_rings, waves) * sizeof(uint32_t);
    l.ring2 = (L3D_LDS volatile uint32_t*)base; if (two_rings) base += waves * kRing2 * sizeof(uint32_t);
    l.minov = (L3D_LDS volatile float*)base; base += rows * 4;
    l.claim = (L3D_LDS volatile uint32_t*)base; base += rows * 4;
    l.top_ov = (L3D_LDS volatile float*)base; base += (size_t)rows * K * 4;
    l.cnt = (L3D_LDS volatile idx_t*)base; base += rows * sizeof(idx_t);
    l.minpos = (L3D_LDS volatile idx_t*)base; base += rows * sizeof(idx_t);
    l.top_ix = (L3D_LDS volatile idx_t*)base; base += (size_t)rows * K * sizeof(idx_t);
    return l;
}

// conservative fp32 test "could overlap(src, tgt) exceed thr?": l3d_dev.h prefilter_products, the form without reciprocals
// (the round-2 form with two v_rcp_f32 and the batch levels / delivery variants of rounds 3-5 that lost their A/Bs are
// documented in profiles/r03_v3_ab_prefilter.txt, r03_v2_ab_target_delivery.txt, r05_ab_match_forms.txt -- and no longer built)
// scalar-unit bit operations the compiler does not select by itself (wave-uniform operands)
__device__ __forceinline__ uint32_t s_ff1(uint64_t m) { uint32_t r; asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m)); return r; }
__device__ __forceinline__ uint64_t s_bitset0(uint64_t m, uint32_t bit) { asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit)); return m; }

__device__ __forceinline__ bool prefilter(float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                                          const v4f q, float thr) {
    return prefilter_products(e1x, e1y, e1z, e2x, e2y, e2z, q.x, q.y, q.z, q.w, thr);
}

// checkMatchOrientation (line3D.cc:811-858) of one freshly computed slot, in its source frame and -- for a pair that
// hands inverse matches to its target (tgt processed later, :1680) -- in the target frame.  Returns the slot flags.
__device__ __forceinline__ uint32_t fuse_orientation(const OrientFuse& of, const double* Cs, const double* Ct,
                                                     const SegX& sx, const SegX& tx, const PairResult& res,
                                                     bool hands_inverse) {
    if (!orientation_ok_fast(Cs, sx, res.dp1, res.dp2, of.thr)) return 0u;
    uint32_t flags = kSlotAlive;
    if (hands_inverse && orientation_ok_fast(Ct, tx, res.dq1, res.dq2, of.thr)) flags |= kSlotInvAlive;
    return flags;
}
// the 4-byte stream k_pair_csr sorts: target segment of a slot that hands an inverse match over, kEmpty otherwise
__device__ __forceinline__ uint32_t inverse_target(const Slot& o) { return (o.flags & kSlotInvAlive) ? o.tgt_seg : kEmpty; }
__device__ __forceinline__ void store_inverse_target(const OrientFuse& of, uint64_t at, const Slot& o) {
    const uint32_t t = inverse_target(o);
    if (of.tgt16) ((uint16_t*)of.inv_tgt)[at] = (uint16_t)(t == kEmpty ? 0xFFFFu : t);   // (wave-uniform choice)
    else of.inv_tgt[at] = t;
    // the hypothesis streams of phase B (l3d_kernels.h: OrientFuse): what the list pass reads of a slot, 8 bytes each
    if (of.hyp_p) {
        const float nan = __builtin_nanf("");
        const bool alive = o.tgt_seg != kEmpty && (o.flags & kSlotAlive);
        of.hyp_p[at] = alive ? make_float2(o.dp1, o.dp2) : make_float2(nan, nan);
        of.hyp_q[at] = make_float2(o.dq1, o.dq2);
    }
}

}  // namespace

// MODE 0: bounded kNN (top-K rows)   MODE 1: count accepted matches per row (kNN <= 0, pass 1)
// MODE 2: write every accepted match in ascending target order (kNN <= 0, pass 2)
// BRUTE: skip the pre-filter (every pair goes through the exact test) -- on-GPU check that the
//        pre-filter never loses a match.
// WPG: waves per workgroup (1 or 2).  Both waves of a workgroup hold the SAME 64 source rows and share their top-K
//      tables in LDS; wave q visits the target chunks with index = q (mod 2).  A work item is then half as long, which
//      shortens the ramp-down tail of the launch (waves cannot migrate: at the end some SIMDs still hold a full set of
//      long items while others are empty).  The waves meet at two barriers only (tables initialised / all candidates
//      inserted); a row's table is guarded by a compare-and-swap lock in LDS.
// amdgpu_waves_per_eu: with two waves per work item the LDS tables leave room for 7 waves per SIMD; the kernel would
// otherwise take 75 VGPRs (6 waves) -- held to 72 it spills one register and runs 5 % faster on C1 (0.718 against
// 0.754 ms).  One wave per item (the large scenes) is limited to 6 per SIMD by its LDS (6.3 KiB per wave) whatever the
// registers: there the 84-register budget without the spill is the faster one (C2 13.88 against 14.26 ms, C4 25.6
// against 26.2; profiles/r03_v2_ab_target_delivery.txt, rows sl6 / sl7).
// TILE (round 5; 0 = the row form above, 16 = rows per work item): the tile form of the bounded-kNN kernel.
//   * a work item is R = TILE source rows of ONE width class in band order (k_cull_prepare pads every class to a multiple
//     of R), handled by one wave64; lane l = (row l % R, target slot l / R): a step of the walk tests 64 / R targets against
//     R rows.  The wave visits the targets whose band meets the hull of ITS R rows -- a quarter (an eighth) of the rows of
//     the row form, all of one width class: on C1 the pre-filter tests 1.5 x the (row, target) pairs whose own bands
//     intersect instead of 2.9 x (tools/hull_sim.py; C2 1.4 x instead of 2.1 x, C4 1.24 x instead of 1.7 x);
//   * the target records no longer come through the scalar cache one s_load_dwordx4 each (a 16-byte record stream misses
//     it on every fourth load by construction, and half of a wave's life was waiting): in a visited chunk lane l fetches
//     record l and band l (one coalesced 1 KiB + 512 B load per chunk), the in-band records are compacted into an LDS
//     FIFO (v_mbcnt), and a step reads its 64 / R records back as ds_read_b128 -- one address per R lanes, broadcast.
//     The FIFO carries what is left over (< 64 / R records) to the next chunk, so sparse chunks cost no partial steps;
//   * one ballot + one compaction per 64 lane-tests (two per 128 in the row form), no scalar bit scan, no record address
//     arithmetic on the scalar unit.
#ifndef L3D_MATCH_WAVES
#define L3D_MATCH_WAVES (TILE ? L3D_TILE_WAVES : MODE == 1 ? 5 : (WPG == 2 ? 7 : 6))   // (MODE 1: the keep-all pass holds both ray records)
#endif
#ifndef L3D_ROW_CACHE
#define L3D_ROW_CACHE 1   // 0: never stage the source rows' records in LDS (A/B)
#endif
#ifndef L3D_TILE_WAVES
#define L3D_TILE_WAVES 6
#endif
#ifndef L3D_ROW_CLASSES_DEFAULT
#define L3D_ROW_CLASSES_DEFAULT 1   // row form: padded class layout (1) or the legacy layout (0)
#endif
template <int MODE, bool BRUTE, bool IX16, int WPG, bool STAGED, int TILE = 0>
__global__ __launch_bounds__(kBlock * WPG) __attribute__((amdgpu_waves_per_eu(L3D_MATCH_WAVES))) void k_match_pairs(const ViewDev* __restrict__ views,
                                                           const PairDesc* __restrict__ pairs,
                                                           const WorkItem* __restrict__ work, uint32_t nwork,
                                                           Slot* __restrict__ slots,
                                                           uint32_t* __restrict__ row_counts, float thr,
                                                           const CullPools cp, const OrientFuse of) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware order: consecutive work items (same pair => same target view) go to one XCD
    // (block b runs on XCD b % 8, MI355X_MICROARCH.md), so the target view stays in that XCD's L2.
    // With a longest-first order (k_order_items) each XCD's part of the list is walked longest item first.
    const uint32_t per_xcd = (nwork + 7) / 8;
    uint32_t w = (blockIdx.x >> 3) < per_xcd ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : nwork;
    if (w >= nwork) return;
    if (cp.item_order) w = cp.item_order[w];
    const WorkItem wi = work[w];
#if defined(L3D_STATS) || defined(L3D_CYCLES)
    const unsigned long long t_start = clock64(), w_start = wall_clock64();
    unsigned long long t_s1 = 0, t_s2 = 0;
#define L3D_TIC const unsigned long long tic_ = clock64()
#define L3D_TOC(acc) acc += clock64() - tic_
#else
#define L3D_TIC ((void)0)
#define L3D_TOC(acc) ((void)0)
#endif
    const PairDesc& pd = pairs[wi.pair];
    const ViewDev& vs = views[pd.src];
    const ViewDev& vt = views[pd.tgt];
    const uint32_t K = pd.K, Ms = pd.Ms, Mt = pd.Mt;
    const bool fastm = (pd.flags & kPairFastMath) != 0;     // l3d_dev.h: IEEE division / sqrt without operand scaling
    typedef typename IdxT<IX16>::type idx_t;
    static_assert(!STAGED || (MODE == 0 && !BRUTE), "the two-stage candidate pipeline exists for the bounded-kNN variant");
    static_assert(TILE == 0 || (TILE == 16 && STAGED && WPG == 1), "the tile form: bounded kNN, staged, one wave per item");
    constexpr uint32_t ROWS = TILE ? (uint32_t)TILE : (uint32_t)kBlock;   // rows of a work item
    constexpr uint32_t TPS = 64u / ROWS;                                  // targets per step of the tile form's walk
    // row cache (two waves per item only: with one wave per item its 3.3 KiB would cost a wave of occupancy): what the two
    // candidate stages read of a SOURCE row -- the float depth record and the raw segment -- is staged in LDS once per
    // work item instead of being gathered from the view's arrays by every candidate
    // (C1: kernel 0.601 -> 0.585 ms; not for launches that do not fill the machine -- C0 0.179 -> 0.182: the launcher sets
    // CullPools::row_cache by the number of work items; profiles/r05_ab_match_forms.txt)
    constexpr bool ROWCACHE_BUILT = L3D_ROW_CACHE != 0 && STAGED && WPG == 2 && TILE == 0;
    const bool ROWCACHE = ROWCACHE_BUILT && cp.row_cache != 0;
    Lds<IX16> L = carve<IX16>((L3D_LDS char*)smem, MODE == 0 ? K : 0, WPG, STAGED, ROWS, TILE != 0, ROWCACHE);
    // wave of the workgroup -- through readfirstlane: the compiler must know it is wave-uniform, or the chunk loop
    // below (its mask depends on q) is compiled as a divergent loop with vector addresses
    const uint32_t q = WPG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0u;
    const uint32_t tid = threadIdx.x & 63u, lane = tid;          // row of the work item = lane (row form)
    const uint32_t rs = TILE ? (tid & (ROWS - 1u)) : tid;        // this lane's row of the work item
    const uint32_t grp = TILE ? tid / ROWS : 0u;                 // tile form: which of a step's TPS targets this lane tests
    constexpr uint32_t kRingN = (uint32_t)ring_entries(STAGED, WPG);
    constexpr idx_t kTie = Lds<IX16>::kTie;
    L3D_LDS volatile uint32_t* ring = L.ring + q * kRingN;
    L3D_LDS volatile uint32_t* ring2 = L.ring2 + q * kRing2;
    // epipolar-band culling.  Keep-all mode (round 4): the COUNT pass takes the culled walk as well (a count does not
    // depend on the order); the FILL pass streams unculled, because a row must come out in ascending target order
    // (line3D.cc:987-992) and that is the order in which an unculled row's matches arrive.  Measured on C1
    // (profiles/r04_keepall.txt): count 1388 -> 660 us; a culled fill pass (1737 -> 1329 us) followed by a sort of every
    // row (744 us as a kernel of its own, 3500 us inside the epilogue) was slower than streaming.
    const PairCull* pc = (MODE != 2 && !BRUTE && cp.cull && cp.cull[wi.pair].enabled) ? &cp.cull[wi.pair] : nullptr;
    const bool cull = pc != nullptr;
    bool active = wi.src0 + rs < Ms;
    uint32_t src = wi.src0 + rs;
    float blo = __builtin_inff(), bhi = -__builtin_inff();   // this lane's tau band (empty for a dead lane)
    // padded class layout of the source rows (the tile form always; the row form when k_cull_prepare laid the launch's
    // rows out that way, CullPools::padded_rows): the pair's region holds tile_src_cap(Ms, R) positions, every width class
    // padded to a multiple of R with kEmpty rows
    const bool pad = TILE != 0 || cp.padded_rows != 0;
    if (pad && cull) {
        src = cp.src_perm[pc->s_off + wi.src0 + rs];
        active = src != kEmpty;
        if (active) { const float2 b = cp.src_band[pc->s_off + wi.src0 + rs]; blo = b.x; bhi = b.y; }
    } else if (cull && active) {
        src = cp.src_perm[pc->s_off + wi.src0 + tid];
        const float2 b = cp.src_band[pc->s_off + wi.src0 + tid];
        blo = b.x; bhi = b.y;
    }
    if (pad && !active) src = kEmpty;                         // (the epilogue skips such rows)
    if (pad && L3D_BALLOT(active) == 0) return;               // an item of padding only (every wave of the workgroup holds
                                                              // the same rows: uniform, before any barrier / LDS use)

    double F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = pd.F[i];

    // ---- prologue: this lane's two epipolar lines, normalised, image-centre origin, fp32 ----
    float e1x = 0, e1y = 0, e1z = 0, e2x = 0, e2y = 0, e2z = 0;
    float thrL = thr;
    bool live = false;               // dead lane (row >= Ms or degenerate epipolar line): never a candidate
    if (ROWCACHE && q == 0) {        // lane = row: its float depth record and raw segment (zeros for a dead row: never read)
        SegD32 sd{}; float4 s4{0.0f, 0.0f, 0.0f, 0.0f};
        if (active) { sd = vs.segd32[src]; s4 = vs.seg4[src]; }
        L.row_sd[3 * tid + 0] = v4f{sd.r1[0], sd.r1[1], sd.r1[2], sd.r2[0]};
        L.row_sd[3 * tid + 1] = v4f{sd.r2[1], sd.r2[2], sd.n[0], sd.n[1]};
        L.row_sd[3 * tid + 2] = v4f{sd.n[2], s4.x, s4.y, s4.z};
        L.row_w[tid] = s4.w;
    }
    if (active) {
        const float4 s = vs.seg4[src];
        d3 e1 = mul33(F, d3{(double)s.x, (double)s.y, 1.0});
        d3 e2 = mul33(F, d3{(double)s.z, (double)s.w, 1.0});
        double n1 = sqrt(e1.x * e1.x + e1.y * e1.y), n2 = sqrt(e2.x * e2.x + e2.y * e2.y);
        // a vanishing (ex,ey) makes every intersection invalid in the reference (|z| <= 1e-12 up to
        // scale); such a row only ever produces NaNs here and is rejected by the exact test.
        if (n1 > 0.0 && n2 > 0.0) {
            double cx = (double)vt.cx, cy = (double)vt.cy;
            e1x = (float)(e1.x / n1); e1y = (float)(e1.y / n1);
            e1z = (float)((e1.z + (e1.x * cx + e1.y * cy)) / n1);
            e2x = (float)(e2.x / n2); e2y = (float)(e2.y / n2);
            e2z = (float)((e2.z + (e2.x * cx + e2.y * cy)) / n2);
            live = true;
        }
    }
    if (q == 0 && tid < ROWS) {
        L.cnt[tid] = 0;
        L.minov[tid] = thr;
        L.claim[tid] = kEmpty;
        L.minpos[tid] = 0;
    }
    if (WPG > 1) __syncthreads();
    uint32_t head = 0, tail = 0;   // wave-uniform ring cursors
    // tau band of the wave (hull of the live lanes), kept in SGPRs
    float wlo = live ? blo : __builtin_inff(), whi = live ? bhi : -__builtin_inff();
    if (cull) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            wlo = fminf(wlo, __shfl_xor(wlo, o));
            whi = fmaxf(whi, __shfl_xor(whi, o));
        }
    }
    wlo = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(wlo)));
    whi = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(whi)));
    const uint32_t* __restrict__ tperm = cull ? cp.tgt_perm + pc->t_off : nullptr;
    const float2* __restrict__ cband = cull ? cp.chunk_band + pc->c_off : nullptr;
    const float2* __restrict__ tband = cull ? cp.tgt_band + pc->t_off : nullptr;

    // worst entry of a full row: the smallest overlap (pipelined, non-volatile LDS reads).  Among EQUAL overlaps any entry
    // will do: whenever equal overlaps decide what a row keeps -- inside the final table, or between the K-th best and an
    // entry that lost to it or was evicted -- the row is flagged and replayed in the reference's heap order
    // (k_match_tied_rows), so this kernel's own choice among them never reaches the result.  That keeps the target index
    // out of the comparisons of the hot insertion path (half of its VALU instructions).
    auto rescan_worst = [&](uint32_t sl) {
        L3D_LDS const float* ov = (L3D_LDS const float*)L.top_ov + (size_t)sl * K;
        uint32_t wj = 0; float wo = ov[0];
        // (K is a run-time value, so the loop is not unrolled and every entry was its own LDS round trip -- ten dependent
        // ~100-cycle waits per insertion into a full row; four reads are requested together, compared in the same order)
        const idx_t mp = L.minpos[sl];
        uint32_t j = 1;
        for (; j + 4 <= K; j += 4) {
            const float o0 = ov[j], o1 = ov[j + 1], o2 = ov[j + 2], o3 = ov[j + 3];
            if (o0 < wo) { wo = o0; wj = j; }
            if (o1 < wo) { wo = o1; wj = j + 1; }
            if (o2 < wo) { wo = o2; wj = j + 2; }
            if (o3 < wo) { wo = o3; wj = j + 3; }
        }
        for (; j < K; ++j) {
            const float o = ov[j];
            if (o < wo) { wo = o; wj = j; }
        }
        L.minov[sl] = wo;
        L.minpos[sl] = (idx_t)(wj | (mp & kTie));
    };
    auto flag_tie = [&](uint32_t sl) { L.minpos[sl] = (idx_t)(L.minpos[sl] | kTie); };

    auto prefix = [&](uint64_t m) -> uint32_t {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    // ---- bounded kNN: the candidates of the pre-filter pass through TWO full-lane stages ----
    //   stage 1  the DECISION of the depth test (all four triangulated depths > L3D_EPS, line3D.cc:960-980) without its
    //            divisions (l3d_dev.h depths_positive: ~60 VALU instead of ~95): a third of the candidates of the
    //            BASELINE scenes fail it, and they used to go through the ~350-instruction overlap first
    //   stage 2  the exact epipolar overlap of the survivors (compacted into a second ring, so again one per lane) and
    //            the kNN insertion
    // The depth VALUES are only computed for the winners (epilogue), as before.  What the exact tests read of a target
    // comes from the copies k_cull_prepare keeps in walk order (tgt_s4 / tgt_sd): a wave's candidates are neighbours
    // there.  Table entries are positions in that order; the epilogue translates the winners (tgt_perm).
    uint32_t head2 = 0, tail2 = 0;
    // (walk-order copies only where k_cull_prepare keeps them: large target views; otherwise the view's own arrays, by
    // original index)
    const bool sorted = cull && pc->sorted_copy != 0;
    const float4* __restrict__ ts4 = sorted ? cp.tgt_s4 + pc->t_off : vt.seg4;
    const SegD32* __restrict__ tsd = sorted ? cp.tgt_sd + pc->t_off : vt.segd32;
    auto target_index = [&](uint32_t tp) -> uint32_t { return (cull && !sorted) ? tperm[tp] : tp; };
    // (round 4: stage 1 decides in FLOAT on 48-byte records -- depths_positive32, l3d_dev.h: the decision is a matter of
    // signs once the six dot products are clear of zero, which they are for all but a few candidates in a million;
    // those go through the double-precision decision of round 3 on the spot.  Halves the gathered bytes of the stage
    // and replaces ~60 fp64 instructions by ~35 fp32 ones at half the issue cost.)
    const float Bx = pd.B[0], By = pd.B[1], Bz = pd.B[2], tolB = pd.tolB;
    auto stage1 = [&]() {
        const uint32_t n = min(64u, tail - head);
        const bool has = lane < n;
        if (lane == 0) { L3D_STAT(1, n); L3D_STAT(8, 1); }
        const uint32_t ent = ring[(head + lane) & (kRingN - 1)];
        head += n;
        const uint32_t sl = ent >> 23, tp = ent & 0x7FFFFFu;
        const uint32_t sg = __shfl(src, sl);   // the ring of a wave only holds rows of that wave
        bool pass = false, fallback = false;
        if (has) {
            SegD32 sd;
            if (ROWCACHE) {
                const v4f a = L.row_sd[3 * sl + 0], b = L.row_sd[3 * sl + 1], c = L.row_sd[3 * sl + 2];
                sd.r1[0] = a.x; sd.r1[1] = a.y; sd.r1[2] = a.z; sd.r2[0] = a.w; sd.r2[1] = b.x; sd.r2[2] = b.y;
                sd.n[0] = b.z; sd.n[1] = b.w; sd.n[2] = c.x;
            } else sd = vs.segd32[sg];
            const SegD32 td = tsd[target_index(tp)];
            const float B[3] = {Bx, By, Bz};
            bool certain;
            pass = depths_positive32(sd, td, B, tolB, certain);
            fallback = !certain;
        }
        if (L3D_BALLOT(fallback)) {   // the sliver (wave-uniform test): the double-precision decision on the full records
            if (fallback) {
                const uint32_t to = cull ? tperm[tp] : tp;    // (by original target index)
                // (the camera centres are fetched HERE, through laundered pointers: hoisted out of the walk they occupied
                // 12 scalar registers of a loop that spills them -- this branch runs for a few candidates in a million)
                const ViewDev* pvs = &vs; const ViewDev* pvt = &vt;
                asm volatile("" : "+r"(pvs), "+r"(pvt));
                pass = depths_positive(*(const SegD*)&pvs->segx[sg], *(const SegD*)&pvt->segx[to], pvs->C, pvt->C);
                L3D_STAT(12, 1);                                // candidates decided in double (the float certificate failed)
            }
        }
        const uint64_t m = L3D_BALLOT(pass);
        if (pass) ring2[(tail2 + prefix(m)) & (kRing2 - 1)] = ent;
        tail2 += (uint32_t)__popcll(m);
        if (lane == 0) L3D_STAT(9, __popcll(m));
    };
    auto stage2 = [&]() {
        const uint32_t n = min(64u, tail2 - head2);
        const bool has = lane < n;
        if (lane == 0) L3D_STAT(4, 1);
        const uint32_t ent = ring2[(head2 + lane) & (kRing2 - 1)];
        head2 += n;
        const uint32_t sl = ent >> 23, tg = ent & 0x7FFFFFu;
        const uint32_t sg = __shfl(src, sl);
        bool pending = false;
        float ovv = 0.0f;
        // F is fetched per drain through a laundered pointer (wave-uniform control flow: scalar loads): hoisted out of the
        // walk its 18 scalar registers -- with everything else the stages need -- made the walk's loop reload spilled
        // scalars from vector lanes at every step (four v_readlane per two targets: the record base pointer and the lane
        // mask).  Round 3 measured this switch as neutral; since the depth decision of stage 1 runs on floats (no camera
        // centres in scalar registers either) it removes the reloads: C4 -8.7 %, C2 -3.5 %, C1 -1 % (profiles/r04_ab_match.txt).
        const double* Fp = pd.F;
        asm volatile("" : "+s"(Fp));
        if (has) {
            float4 s4;
            if (ROWCACHE) { const v4f c = L.row_sd[3 * sl + 2]; s4 = make_float4(c.y, c.z, c.w, L.row_w[sl]); }
            else s4 = vs.seg4[sg];
            const float4 t4 = ts4[target_index(tg)];
            const float ov = exact_overlap(Fp, s4.x, s4.y, s4.z, s4.w, t4.x, t4.y, t4.z, t4.w, fastm);
            // a full row only admits overlaps that reach its K-th best (minov == thr while the row is not full); an overlap
            // EQUAL to the K-th best goes on to the insertion: a tie at the kNN-th place flags the row for the exact replay
            const float need = L.minov[sl];
            if (ov > thr && ov >= need) { pending = true; ovv = ov; L3D_STAT(2, 1); L3D_STAT(3, 1); }
#ifdef L3D_STATS
            else if (!(ov > thr)) L3D_STAT(10, 1);          // not a match at all: what the pre-filter's slack lets through
            else L3D_STAT(11, 1);                           // a match that no longer reaches the row's K-th best
#endif
        }
        // several candidates of one drain may belong to the same row: one at a time (compare-and-swap lock with two waves
        // per row group; a wave's own contenders are serialised by the LDS atomic unit just the same)
        while (L3D_BALLOT(pending)) {
            bool win;
            if (WPG > 1) {
                win = false;
                if (pending) {
                    uint32_t expect = kEmpty;
                    win = __hip_atomic_compare_exchange_strong((L3D_LDS uint32_t*)&L.claim[sl], &expect, threadIdx.x,
                                                               __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                if (pending) __hip_atomic_fetch_min((L3D_LDS uint32_t*)&L.claim[sl], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                win = pending && (L.claim[sl] == lane);
            }
            if (win) {
                if (WPG == 1) L.claim[sl] = kEmpty;
                pending = false;
                const uint32_t c = L.cnt[sl];
                const idx_t mp_now = L.minpos[sl];      // (requested with the count: one wait instead of three)
                const float mo_now = L.minov[sl];
                L3D_LDS volatile float* ov = L.top_ov + (size_t)sl * K;
                L3D_LDS volatile idx_t* ix = L.top_ix + (size_t)sl * K;
                if (c < K) {
                    ov[c] = ovv; ix[c] = tg;
                    L.cnt[sl] = c + 1;
                    if (c + 1 == K) rescan_worst(sl);
                } else {
                    const uint32_t wj = mp_now & (idx_t)~kTie;
                    const float mo = mo_now;
                    if (ovv > mo) {
                        ov[wj] = ovv; ix[wj] = tg;
                        rescan_worst(sl);
                        if (L.minov[sl] == mo) flag_tie(sl);    // the evicted entry ties with the new K-th best
                    } else if (ovv == mo) {
                        flag_tie(sl);                           // a tie at the K-th place (whichever index would win)
                    }
                }
                if (WPG > 1)
                    __hip_atomic_store((L3D_LDS uint32_t*)&L.claim[sl], kEmpty, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // feed the K-th best overlap back into the owning lane's pre-filter threshold
        if (live) thrL = L.minov[rs];
    };

    // one exact test per lane on up to 64 queued candidates, then kNN insertion
    auto drain = [&]() {
        const uint32_t n = min(64u, tail - head);
        const bool has = lane < n;
        if (lane == 0) { L3D_STAT(1, n); L3D_STAT(4, 1); }
        const uint32_t ent = ring[(head + lane) & (kRingN - 1)];
        head += n;
        const uint32_t sl = ent >> 23;
        uint32_t tg = ent & 0x7FFFFFu;
        bool pending = false;
        PairResult res{};
        uint32_t keep_flags = 0;
        // the ring of a wave only holds rows of that wave: the source segment comes from the owning lane
        const uint32_t sg = __shfl(src, sl);
        if (has) {
            if (cull) tg = tperm[tg];
            const float4 s4 = vs.seg4[sg], t4 = vt.seg4[tg];
            const float ov = exact_overlap(F, s4.x, s4.y, s4.z, s4.w, t4.x, t4.y, t4.z, t4.w);
            // a full row only admits overlaps that reach its K-th best; minov == thr while the row is not full.  An
            // overlap EQUAL to the K-th best goes on to the insertion as well: whether it wins or loses there under
            // (overlap desc, tgt asc), it is a tie at the kNN-th place, which the reference resolves by the pop order
            // of its priority_queue -- the row is flagged and replayed exactly by k_match_tied_rows.
            const float need = (MODE == 0) ? L.minov[sl] : thr;
            if (ov > thr && ov >= need) {
                res.overlap = ov;
                L3D_STAT(2, 1);
                if (MODE == 1) {
                    // (the single pass of the keep-all mode keeps the whole slot: checkMatchOrientation here, on the ray records
                    // the depth test has loaded -- the target's is a gather of 104 bytes per candidate, paid once)
                    const SegX sxx = vs.segx[sg], txx = vt.segx[tg];
                    pending = exact_depths(sxx, txx, vs.C, vt.C, res);
                    if (pending && of.keep_rec) keep_flags = fuse_orientation(of, vs.C, vt.C, sxx, txx, res, pd.tgt > pd.src);
                } else
                pending = exact_depths(vs.segx[sg], vt.segx[tg], vs.C, vt.C, res);
                if (pending) L3D_STAT(3, 1);
            }
        }
        // several candidates of one drain may belong to the same row: one at a time, lowest lane first (without
        // culling a row therefore sees its candidates in ascending target order, which MODE 2 relies on)
        // With several waves per row group the row's table is a critical section: compare-and-swap lock (a wave's
        // own contenders are serialised by the LDS atomic unit just the same), released after the update.
        while (L3D_BALLOT(pending)) {
            bool win;
            if (WPG > 1) {
                win = false;
                if (pending) {
                    uint32_t expect = kEmpty;
                    win = __hip_atomic_compare_exchange_strong((L3D_LDS uint32_t*)&L.claim[sl], &expect, threadIdx.x,
                                                               __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                if (pending) __hip_atomic_fetch_min((L3D_LDS uint32_t*)&L.claim[sl], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                win = pending && (L.claim[sl] == lane);
            }
            if (win) {
                if (WPG == 1) L.claim[sl] = kEmpty;
                pending = false;
                const uint32_t c = L.cnt[sl];
                if (MODE == 1) {
                    // (the single culled pass of the keep-all mode: the accepted match is kept at its arrival index, in the
                    // row's scratch; k_keep_assemble puts the rows into ascending target order)
                    if (of.keep_rec && c < of.keep_cap) {
                        Slot o;
                        o.tgt_seg = tg; o.overlap = res.overlap;
                        o.dp1 = res.dp1; o.dp2 = res.dp2; o.dq1 = res.dq1; o.dq2 = res.dq2;
                        o.score3D = 0.0f; o.flags = keep_flags;
                        of.keep_rec[(size_t)(pd.row_off + sg) * of.keep_cap + c] = o;
                    }
                    L.cnt[sl] = c + 1;
                } else if (MODE == 2) {
                    Slot o;
                    o.tgt_seg = tg; o.overlap = res.overlap;
                    o.dp1 = res.dp1; o.dp2 = res.dp2; o.dq1 = res.dq1; o.dq2 = res.dq2;
                    o.score3D = 0.0f; o.flags = 0;
                    if (c < K) slots[pd.slot_off + (uint64_t)sg * K + c] = o;
                    L.cnt[sl] = c + 1;
                } else {
                    L3D_LDS volatile float* ov = L.top_ov + (size_t)sl * K;
                    L3D_LDS volatile idx_t* ix = L.top_ix + (size_t)sl * K;
                    if (c < K) {
                        ov[c] = res.overlap; ix[c] = tg;
                        L.cnt[sl] = c + 1;
                        if (c + 1 == K) rescan_worst(sl);
                    } else {
                        const uint32_t wj = L.minpos[sl] & (idx_t)~kTie;
                        const float mo = L.minov[sl];
                        if (res.overlap > mo) {
                            ov[wj] = res.overlap; ix[wj] = tg;
                            rescan_worst(sl);
                            if (L.minov[sl] == mo) flag_tie(sl);    // the evicted entry ties with the new K-th best
                        } else if (res.overlap == mo) {
                            flag_tie(sl);                           // a tie at the K-th place (whichever index would win)
                        }
                    }
                }
                if (WPG > 1)
                    __hip_atomic_store((L3D_LDS uint32_t*)&L.claim[sl], kEmpty, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // feed the K-th best overlap back into the owning lane's pre-filter threshold
        if (MODE == 0 && live) thrL = L.minov[rs];
    };

    // ---- main loop: stream the target view through LDS ----
    // Branch-free per test: dead lanes evaluate the pre-filter on zeros and are masked out of the ballot;
    // the compaction prefix is v_mbcnt (population count of the ballot below this lane).
    const v4f* __restrict__ tf = cull ? (const v4f*)(cp.tgt_sf + pc->t_off) : (const v4f*)vt.segf;
    const uint32_t ent_hi = rs << 23;
    const bool lane_on = BRUTE ? active : live;           // lanes that can produce candidates
    const uint64_t lanes_on = L3D_BALLOT(lane_on);
    // the candidate pipeline is run whenever the first ring holds a full drain (flush: until both rings are empty)
    auto pump = [&](bool flush) {
        if (STAGED) {
            while (tail - head >= (flush ? 1u : 64u)) {
                { L3D_TIC; stage1(); L3D_TOC(t_s1); }
                while (tail2 - head2 >= 64) { L3D_TIC; stage2(); L3D_TOC(t_s2); }
            }
            if (flush) while (tail2 != head2) { L3D_TIC; stage2(); L3D_TOC(t_s2); }
        } else {
            while (flush ? (tail != head) : (tail - head >= 64)) drain();
        }
    };
    // The target view is visited in chunks of 64 records.  With culling, 32 chunk bands are tested at once (one per
    // lane), then inside a visited chunk one target band per lane; only the targets whose band meets the wave's band
    // are walked (ascending order inside a chunk), their records fetched as scalar operands: the index is wave-uniform
    // and the sorted copy of the records was written by k_cull_prepare before this launch, so a load through the
    // constant address space is an s_load_dwordx4 -- no LDS tile, no barrier, no lane broadcast.
    typedef const __attribute__((address_space(4))) v4f* RecPtr;
    typedef const __attribute__((address_space(4))) char* RecBytes;
    RecBytes tfb = (RecBytes)(unsigned long)tf;
    const uint32_t nch = (Mt + 63) / 64;
    // CENTRE-OUT: rows and targets are both ordered by the lower end of their bands, and a row's best matches are the
    // targets whose band nearly coincides with its own.  Walking the chunks in ascending order fills every row's table
    // with poor matches from the far left first, which the good ones in the middle then evict one by one -- each through
    // the double-precision test.  So the walk starts at the chunk where the targets' bands begin where the wave's
    // median row's band does (cc), goes up to the end, then from cc - 1 down to the start: the tables fill with good
    // matches at once, the K-th best overlap fed back into the pre-filter is high early, and fewer candidates reach the
    // exact test.  The result does not depend on the order (total order of the insertion; ties are replayed).
    uint32_t cc = 0;
    if (cull) {
        // (tile form: the item's rows are lanes 0 .. R-1, the padding of its class -- if any -- at the end)
        const uint64_t row_lanes = ROWS == 64u ? ~0ull : ((1ull << (ROWS & 63u)) - 1ull);
        const uint32_t n_rows = pad ? (uint32_t)__popcll(L3D_BALLOT(active) & row_lanes) : min((uint32_t)kBlock, Ms - wi.src0);
        const float mid = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(blo), n_rows / 2));
        for (uint32_t c0 = 0; c0 < nch; c0 += 64) {
            const uint32_t c = c0 + lane;
            cc += (uint32_t)__popcll(L3D_BALLOT(c < nch && cband[c].x < mid));
        }
    }
    // ---- tile form: the FIFO of in-band target records and one step of the walk ----
    uint32_t fh = 0, ft = 0;             // wave-uniform FIFO cursors (records fh .. ft-1 are queued)
    auto tile_step = [&]() {
        // lane (row rs, slot grp) tests record fh + grp; a slot beyond the queue (the flush at the end of the walk) idles
        const uint32_t p = fh + grp;
        const bool valid = p < ft;
        const uint32_t ix = p & (uint32_t)(kFifo - 1);
        const v4f q = L.fifo_rec[ix];                 // ds_read_b128, one address per R lanes
        const uint32_t tp = L.fifo_pos[ix];
        fh = min(fh + TPS, ft);
        const bool cb = prefilter(e1x, e1y, e1z, e2x, e2y, e2z, q, thrL);
        // (lane masks from the comparisons themselves, combined by scalar ANDs)
        const uint64_t mm = L3D_BALLOT(cb) & L3D_BALLOT(valid) & lanes_on;
        if (lane == 0) L3D_STAT(0, 64);
#ifdef L3D_STATS
        {   // the row's own band against the target's: the pairs a walk without the hull would test
            uint32_t nb = (uint32_t)__popcll(L3D_BALLOT(valid) & lanes_on);
            if (cull) { const float2 b0 = tband[valid ? tp : 0u]; nb = (uint32_t)__popcll(L3D_BALLOT(valid && !(b0.y < blo || b0.x > bhi)) & lanes_on); }
            if (lane == 0) L3D_STAT(5, nb);
        }
#endif
        if (mm) {
            if ((mm >> lane) & 1ull) ring[(tail + prefix(mm)) & (kRingN - 1)] = ent_hi | tp;
            tail += (uint32_t)__popcll(mm);
            pump(false);
        }
    };
    const uint32_t ngrp = (nch + 31) / 32, gc = cc / 32;
    if constexpr (TILE != 0) {
        // ---- the tile form's walk: the same chunk order (centre-out), one chunk AHEAD ----
        // A work item of R rows meets a handful of targets per chunk (C1, R = 16: 7.5 steps per visited chunk, R = 64: 35), so
        // the latency of a chunk's two loads -- record l and band l per lane, L2 hits -- is no longer small against the work
        // between them: the loads of the NEXT visited chunk are issued before the current one is compacted and walked (six
        // registers across the candidate pipeline; loads return in order, so waiting for the current chunk leaves them in
        // flight).
        uint32_t it_pass = 0, it_gi = 0, it_wm = 0, it_g0 = 0, it_dir = 0;
        const uint32_t n_pass = cull ? 2u : 1u;
        auto next_chunk = [&]() -> uint32_t {          // first target of the next visited chunk, kEmpty at the end (wave-uniform)
            while (!it_wm) {
                if (it_pass >= n_pass) return kEmpty;
                const uint32_t n_g = it_pass == 0 ? ngrp - min(gc, ngrp) : gc + 1;
                if (it_gi >= n_g) { ++it_pass; it_gi = 0; continue; }
                const uint32_t g0 = (it_pass == 0 ? gc + it_gi : gc - it_gi) * 32;
                uint32_t wm;
                if (cull) {
                    bool vis = false;
                    const uint32_t c = g0 + lane;
                    if (lane < 32 && c < nch) {
                        const float2 cb = cband[c];
                        vis = !(cb.y < wlo || cb.x > whi);
                    }
                    wm = (uint32_t)L3D_BALLOT(vis);
                    if (it_gi == 0) { const uint32_t below = (1u << (cc & 31u)) - 1u; wm &= it_pass == 0 ? ~below : below; }
                } else {
                    wm = (nch - g0 >= 32) ? 0xFFFFFFFFu : ((1u << (nch - g0)) - 1u);
                }
                ++it_gi;
                it_wm = wm; it_g0 = g0; it_dir = it_pass;
            }
            const uint32_t bit = it_dir == 0 ? (uint32_t)__builtin_ctz(it_wm) : 31u - (uint32_t)__builtin_clz(it_wm);
            it_wm &= ~(1u << bit);
            return (it_g0 + bit) * 64;
        };
        // record and band of target tb + lane (an empty band for a lane beyond the view; without culling every target is in)
        auto fetch = [&](uint32_t tb, v4f& rec, float& lo, float& hi) {
            const uint32_t ti = tb + lane;
            rec = v4f{0.0f, 0.0f, 0.0f, 0.0f}; lo = __builtin_inff(); hi = -__builtin_inff();
            if (ti < Mt) {
                rec = tf[ti];
                if (cull) { const float2 b = tband[ti]; lo = b.x; hi = b.y; }
                else { lo = -__builtin_inff(); hi = __builtin_inff(); }
            }
        };
        uint32_t tb_c = next_chunk();
        v4f rec_c; float lo_c, hi_c;
        if (tb_c != kEmpty) fetch(tb_c, rec_c, lo_c, hi_c);
        while (tb_c != kEmpty) {
            const uint32_t tb_n = next_chunk();
            v4f rec_n = {0.0f, 0.0f, 0.0f, 0.0f}; float lo_n = __builtin_inff(), hi_n = -__builtin_inff();
            if (tb_n != kEmpty) fetch(tb_n, rec_n, lo_n, hi_n);
            // the records whose band meets the hull of the wave's rows are appended to the FIFO, and the walk advances while
            // a full step's worth is queued
            // (the range test is explicit: an item of unbounded rows -- hull (-inf, inf) -- would take the empty band of a lane
            // beyond the view for a hit; unculled: wlo = +inf, whi = -inf never reject the (-inf, inf) band of a real target)
            const bool in = tb_c + lane < Mt && !(hi_c < wlo || lo_c > whi);
            const uint64_t m = L3D_BALLOT(in);
            if (m) {
                if (in) {
                    const uint32_t ix = (ft + prefix(m)) & (uint32_t)(kFifo - 1);
                    L.fifo_rec[ix] = rec_c;
                    L.fifo_pos[ix] = tb_c + lane;
                }
                ft += (uint32_t)__popcll(m);
                // (LDS operations of a wave complete in order: the reads of tile_step see these writes; the compiler must
                // not move them across)
                asm volatile("" ::: "memory");
                while (ft - fh >= TPS) tile_step();
            }
            tb_c = tb_n; rec_c = rec_n; lo_c = lo_n; hi_c = hi_n;
        }
        while (ft != fh) tile_step();                         // what is left in the FIFO (fewer than TPS records)
    } else
    for (uint32_t pass = 0; pass < (cull ? 2u : 1u); ++pass) {
      const uint32_t n_g = pass == 0 ? ngrp - min(gc, ngrp) : gc + 1;
      for (uint32_t gi = 0; gi < n_g; ++gi) {
        const uint32_t g0 = (pass == 0 ? gc + gi : gc - gi) * 32;
        uint32_t wm;
        if (cull) {
            bool vis = false;
            const uint32_t c = g0 + lane;
            if (lane < 32 && c < nch) {
                const float2 cb = cband[c];
                vis = !(cb.y < wlo || cb.x > whi);
            }
            wm = (uint32_t)L3D_BALLOT(vis);
            if (gi == 0) { const uint32_t below = (1u << (cc & 31u)) - 1u; wm &= pass == 0 ? ~below : below; }
        } else {
            wm = (nch - g0 >= 32) ? 0xFFFFFFFFu : ((1u << (nch - g0)) - 1u);
        }
        if (WPG > 1) wm &= 0x55555555u << q;   // this wave's chunks: index = q (mod 2)
        while (wm) {
            const uint32_t bit = pass == 0 ? (uint32_t)__builtin_ctz(wm) : 31u - (uint32_t)__builtin_clz(wm);
            const uint32_t tb = (g0 + bit) * 64;
            wm &= ~(1u << bit);
            const uint32_t ti = tb + lane;
            bool in = ti < Mt;
            if (cull && in) {
                const float2 b = tband[ti];
                in = !(b.y < wlo || b.x > whi);
            }
            uint64_t m = L3D_BALLOT(in);
            while (m) {
                // two targets per step: after their pushes the ring holds at most 63 + 2*64 candidates (kRing = 256, which
                // is what lets a seventh wave per SIMD fit the LDS); the candidate flags go from the comparison straight
                // into the execution mask of their push
                // (scalar-unit economy: the walk costs ~30 scalar instructions per step at 4.3 issue cycles each, more than
                // its vector instructions -- s_bitset0 instead of the three-instruction m &= m - 1; an empty m gives bit
                // index -1, which clears bit 63 of zero)
                const uint32_t j0 = s_ff1(m); m = s_bitset0(m, j0);
                const bool v1 = m != 0; const uint32_t j1r = s_ff1(m); m = s_bitset0(m, j1r);
                const uint32_t j1 = v1 ? j1r : j0;
                // (32-bit byte offsets: s_load_dwordx4 base, offset -- no 64-bit address arithmetic on the scalar unit)
                v4f q0 = *(RecPtr)(tfb + ((tb + j0) << 4)), q1 = *(RecPtr)(tfb + ((tb + j1) << 4));
                asm volatile("" : "+s"(q0), "+s"(q1));   // both records requested before the first is used (C2 / C4 -1 %)
                // (the lane masks are built from the comparison itself and the uniform masks by scalar ANDs: a ballot of
                // `live & test` would first turn the flag into 0/1 in a VGPR)
                const bool c0b = BRUTE || prefilter(e1x, e1y, e1z, e2x, e2y, e2z, q0, thrL);
                const bool c1b = BRUTE || prefilter(e1x, e1y, e1z, e2x, e2y, e2z, q1, thrL);
                const uint64_t m0 = L3D_BALLOT(c0b) & lanes_on, m1 = L3D_BALLOT(c1b) & lanes_on & (v1 ? ~0ull : 0ull);
                if (lane == 0) L3D_STAT(0, 64 * (1 + v1));
#ifdef L3D_STATS
                {   // the row's own band against the target's: the pairs a walk without the 64-row hull would test
                    uint32_t nb = 0;
                    if (cull) {
                        const float2 b0 = tband[tb + j0], b1 = tband[tb + j1];
                        nb += (uint32_t)__popcll(L3D_BALLOT(!(b0.y < blo || b0.x > bhi)) & lanes_on);
                        if (v1) nb += (uint32_t)__popcll(L3D_BALLOT(!(b1.y < blo || b1.x > bhi)) & lanes_on);
                    } else nb = (uint32_t)__popcll(lanes_on) * (1 + v1);
                    if (lane == 0) L3D_STAT(5, nb);
                }
#endif
                if (m0 | m1) {
                    if (kRingN >= 63 + 2 * 64) {
                        if (m0) { if (c0b & lane_on) ring[(tail + prefix(m0)) & (kRingN - 1)] = ent_hi | (tb + j0); tail += __popcll(m0); }
                        if (m1) { if (c1b & lane_on) ring[(tail + prefix(m1)) & (kRingN - 1)] = ent_hi | (tb + j1); tail += __popcll(m1); }
                        pump(false);
                    } else {
                        // the short ring holds one push beyond a partial drain: push, pump, push, pump (a loop, so that the
                        // candidate pipeline is inlined once)
                        for (uint32_t h = 0; h < 2; ++h) {
                            const uint64_t mh = h ? m1 : m0;
                            if (!mh) continue;
                            const bool ch = h ? c1b : c0b;
                            const uint32_t jh = h ? j1 : j0;
                            if (ch & lane_on) ring[(tail + prefix(mh)) & (kRingN - 1)] = ent_hi | (tb + jh);
                            tail += __popcll(mh);
                            pump(false);
                        }
                    }
                }
            }
        }
      }
    }
    pump(true);
#if defined(L3D_STATS) || defined(L3D_CYCLES)
    const unsigned long long t_loop = clock64();
    if (threadIdx.x == 0) L3D_STAT(7, 1);
#endif

    // ---- epilogue ----
    if (MODE != 0) {
        const uint32_t c = active ? (uint32_t)L.cnt[tid] : 0u;
        if (MODE == 1) {
            if (active) row_counts[pd.row_off + src] = c;
            return;
        }
        if (active) {
            Slot* row = slots + pd.slot_off + (uint64_t)src * K;
            Slot empty;
            empty.tgt_seg = kEmpty; empty.overlap = 0; empty.dp1 = empty.dp2 = empty.dq1 = empty.dq2 = 0;
            empty.score3D = 0; empty.flags = 0;
            for (uint32_t j = min(c, K); j < K; ++j) row[j] = empty;
        }
        return;                // (streamed unculled: a row's matches arrived in ascending target order -- drain, lowest lane first)
    }
    // MODE 0.  The wave writes its rows TOGETHER: item = (row, entry), one item per lane and pass, so that the K
    // 32-byte slots of a row (K*32 B contiguous, 64-byte aligned for even K) leave in one store instruction as whole
    // cache lines -- a lane writing its own row slot by slot leaves every line partially written between two stores.
    // Per item: rank among the row's winners by (overlap desc, tgt asc) (the overlaps are in LDS), depths recomputed
    // with the arithmetic of the acceptance test (identical values), and the orientation filter of phase B
    // (checkMatchOrientation, line3D.cc:811-858), which is a function of the slot alone: its flags are written with
    // the slot and the hypothesis counters of phase B are fed from here instead of by a pass that re-reads every slot.
    if (WPG > 1) __syncthreads();                       // every wave's candidates are in the tables
    {   // equal overlaps INSIDE a row's table: their order is the reference's heap order as well (every wave of the
        // group checks all 64 rows itself: same flags, no further barrier)
        // (tile form: lanes R .. 63 repeat the rows of lanes 0 .. R-1 -- same table, same flag, same value written)
        const uint32_t c = min((uint32_t)L.cnt[rs], K);
        L3D_LDS const float* ov = (L3D_LDS const float*)L.top_ov + (size_t)rs * K;
        bool t = false;
        if (K <= 16) {
            // (the usual kNN: the row's entries are fetched four per LDS round trip and compared in registers; entries beyond
            // the row's count read as distinct NaN-free sentinels that equal nothing)
            float o[16];
#pragma unroll
            for (uint32_t b = 0; b < 16; b += 4) {
                if (b < c) {
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) o[b + i] = (b + i < c) ? ov[b + i] : -1.0f - (float)(b + i);
                } else {
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) o[b + i] = -1.0f - (float)(b + i);
                }
            }
#pragma unroll
            for (uint32_t i = 1; i < 16; ++i)
#pragma unroll
                for (uint32_t j = 0; j < i; ++j) t |= o[j] == o[i];
        } else
        for (uint32_t i = 1; i < c; ++i) {
            const float oi = ov[i];
            for (uint32_t j = 0; j < i; ++j) t |= ov[j] == oi;
        }
        if (t && tid < ROWS) flag_tie(tid);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // (tile form: all R rows of the item are walked; padding rows -- src == kEmpty -- are skipped item by item)
    const uint32_t n_rows = pad ? ROWS : min((uint32_t)kBlock, Ms - wi.src0);
    const bool hands_inverse = pd.tgt > pd.src;   // inverse copy only towards a view processed later (:1680)
    const uint32_t n_items = n_rows * K;
    // (the camera centres and the orientation thresholds are needed from here on only -- since round 4 the depth decision
    // of stage 1 runs on floats and the baseline --: fetched through laundered pointers, so that they do not occupy 16
    // scalar registers of the walk, whose loop spills scalars into vector lanes and reloads them every step)
    const ViewDev* evs = &vs; const ViewDev* evt = &vt;
    asm volatile("" : "+s"(evs), "+s"(evt));
    for (uint32_t base0 = 0; base0 < n_items; base0 += 64 * WPG) {
        const uint32_t base = base0 + q * 64;           // first item of this wave in this pass
        const uint32_t it = base + lane;
        const bool in_range = it < n_items;
        const uint32_t r = in_range ? it / K : 0u, j = it - r * K;
        const uint32_t rsrc = __shfl(src, r);           // every wave of the group holds the same 64 rows, row = lane
        const bool in = in_range && (!pad || rsrc != kEmpty);
        // a flagged row is left to k_match_tied_rows entirely (slots, orientation flags, counters)
        const bool tied = in && (L.minpos[r] & kTie) != 0;
        if (tied && j == 0) {
            const uint32_t pos = atomicAdd(of.tie_count, 1u);
            if (pos < of.tie_cap) of.tie_list[pos] = make_uint2(wi.pair, rsrc);
        }
        const uint32_t c = (in && !tied) ? (uint32_t)L.cnt[r] : 0u;
        Slot o;
        o.tgt_seg = kEmpty; o.overlap = 0; o.dp1 = o.dp2 = o.dq1 = o.dq2 = 0; o.score3D = 0.0f; o.flags = 0;
        uint32_t dst = j;
        if (j < c) {
            L3D_LDS const float* ov = (L3D_LDS const float*)L.top_ov + (size_t)r * K;
            L3D_LDS const idx_t* ix = (L3D_LDS const idx_t*)L.top_ix + (size_t)r * K;
            const float oj = ov[j];
            const uint32_t xj = (STAGED && cull) ? tperm[ix[j]] : (uint32_t)ix[j];   // staged, culled pair: the table holds walk-order positions
            // (a row that gets here has no two equal overlaps -- the pass above flagged those --, so the order
            // (overlap desc, target asc) is the order by overlap)
            uint32_t rank = 0;
            {
                uint32_t i = 0;
                for (; i + 4 <= c; i += 4) {
                    const float o0 = ov[i], o1 = ov[i + 1], o2 = ov[i + 2], o3 = ov[i + 3];
                    rank += (o0 > oj ? 1u : 0u) + (o1 > oj ? 1u : 0u) + (o2 > oj ? 1u : 0u) + (o3 > oj ? 1u : 0u);
                }
                for (; i < c; ++i) rank += ov[i] > oj ? 1u : 0u;
            }
            dst = rank;
            // Three short stages that each fetch only the invariants they use (depths: rays + plane; orientation:
            // rays + mid ray).  The pointers are laundered between the stages so that the compiler does not keep both
            // 104-byte records live across them: that would cost the kernel a wave of occupancy for its epilogue.
            const SegX* psx = evs->segx + rsrc;
            const SegX* ptx = evt->segx + xj;
            asm volatile("" : "+v"(psx), "+v"(ptx));
            PairResult res{};
            exact_depths(*psx, *ptx, evs->C, evt->C, res, fastm);
            o.tgt_seg = xj; o.overlap = oj;
            o.dp1 = res.dp1; o.dp2 = res.dp2; o.dq1 = res.dq1; o.dq2 = res.dq2;
            asm volatile("" : "+v"(psx));
            if (orientation_ok_fast(evs->C, *psx, res.dp1, res.dp2, of.thr)) {
                o.flags = kSlotAlive;
                asm volatile("" : "+v"(ptx));
                if (hands_inverse && orientation_ok_fast(evt->C, *ptx, res.dq1, res.dq2, of.thr)) o.flags |= kSlotInvAlive;
            }
        }
        // (round 4: nothing else leaves the epilogue -- no counters, no positions: the inverse hypotheses are sorted by
        // target segment per pair afterwards (k_pair_csr) and the list pass counts its own fresh hypotheses)
        if (in && !tied) {
            const uint64_t at = pd.slot_off + (uint64_t)rsrc * K + dst;
            slots[at] = o;
            store_inverse_target(of, at, o);
        }
#ifdef L3D_STATS
        { const uint32_t nk = (uint32_t)__popcll(L3D_BALLOT(j < c)); if (lane == 0 && nk) L3D_STAT(6, nk); }
#endif
    }
#if defined(L3D_STATS) || defined(L3D_CYCLES)
    if (threadIdx.x == 0 && w < (1u << 16)) {
        const unsigned long long t_end = clock64();
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // ([0], [6]: the 100 MHz wall counter -- s_memtime is not comparable between compute units)
        g_cycles[w][0] = w_start; g_cycles[w][1] = t_loop - t_start; g_cycles[w][2] = t_s1; g_cycles[w][3] = t_s2;
        g_cycles[w][4] = t_end - t_loop; g_cycles[w][5] = t_end - t_start; g_cycles[w][6] = wall_clock64(); g_cycles[w][7] = hw | ((unsigned long long)xcc << 32);
    }
#endif
}

// the two-stage candidate pipeline (default) or the single-stage one of round 2 (L3D_MATCH_STAGED=0: A/B switch; the
// brute-force test hook always takes the single-stage path, so the two check each other)
bool match_staged(int mode, bool brute) {
    static const bool off = [] { const char* e = std::getenv("L3D_MATCH_STAGED"); return e && std::atoi(e) == 0; }();
    return mode == 0 && !brute && !off;
}
bool match_row_cache(int mode, bool brute, uint32_t nwork, uint32_t tile_rows) {
    return L3D_ROW_CACHE != 0 && match_staged(mode, brute) && !tile_rows && match_waves_per_group(mode, brute, nwork) == 2 &&
           nwork > kMatchOrderMinItems;
}
size_t match_lds_bytes(int mode, uint32_t K, bool ix16, uint32_t waves, bool brute, uint32_t tile_rows, bool row_cache) {
    const size_t ib = ix16 ? 2 : 4;
    const bool staged = match_staged(mode, brute);
    if (tile_rows)   // (one wave per item: FIFO + the two rings + the tables of R rows)
        return (size_t)kFifo * (sizeof(v4f) + 4) + (size_t)(ring_entries(true, 1) + kRing2) * 4 + 2 * (size_t)tile_rows * 4 +
               2 * (size_t)tile_rows * ib + (size_t)tile_rows * K * (4 + ib);
    const size_t rc_bytes = (row_cache && staged && waves == 2) ? (size_t)kBlock * (3 * sizeof(v4f) + 4) : 0;
    return rc_bytes + (size_t)waves * (ring_entries(staged, waves) + (staged ? kRing2 : 0)) * 4 + 2 * kBlock * 4 + 2 * kBlock * ib +
           (mode == 0 ? (size_t)kBlock * K * (4 + ib) : 0);
}

// The tile form serves the bounded-kNN launches of the two-stage pipeline (everything else -- keep-all passes, the
// brute-force hook, the single-stage A/B switch -- keeps the row form) whose row form would leave the machine half empty:
// up to kMatchTileMaxItems items of 64 rows (est_row_items: what the caller expects the launch to hold).  L3D_MATCH_TILE = 0 | 16
// overrides.
uint32_t match_tile_rows(int mode, bool brute, uint64_t est_row_items) {
    if (!match_staged(mode, brute)) return 0;
    // (read per call -- once per l3d_match_begin --, not latched: a test switches forms inside one process)
    const char* e = std::getenv("L3D_MATCH_TILE");
    const int forced = e ? std::atoi(e) : -1;
    if (forced == 0 || forced == 16) return (uint32_t)forced;
    return est_row_items && est_row_items <= kMatchTileMaxItems ? 16u : 0u;
}
uint32_t match_layout_rows(int mode, bool brute, uint32_t tile_rows) {
    if (mode != 0 || brute) return 0;
    const uint32_t t = tile_rows;
    if (t) return t;
    const char* e = std::getenv("L3D_MATCH_CLASSES");
    const int classes = e ? std::atoi(e) : L3D_ROW_CLASSES_DEFAULT;
    return classes ? (uint32_t)kMatchRows : 0u;
}

// Two waves per work item pay off while the launch has few items for the machine (C0: kernel 0.34 -> 0.24 ms, C1 with
// 1.4 items per wave slot: -2 %); with many rounds of items the duplicated per-wave work costs more than the shorter
// tail gains (C4: +4 %; four waves per item were slower everywhere: each wave drains its own ring, so the K-th-best
// feedback arrives later and more candidates reach the exact test).
uint32_t match_waves_per_group(int mode, bool brute, uint32_t nwork) {
    if (mode != 0 || brute) return 1;   // keep-all rows need ascending target order; the brute path is a test hook
    static const int forced = [] { const char* e = std::getenv("L3D_MATCH_WPG"); return e ? std::atoi(e) : 0; }();
    if (forced == 1 || forced == 2) return (uint32_t)forced;
    return nwork <= kMatchOrderMaxItems ? 2u : 1u;
}

hipError_t launch_match_pairs(int mode, bool brute, const ViewDev* views, const PairDesc* pairs,
                              const WorkItem* work, uint32_t nwork, uint32_t maxK, Slot* slots,
                              uint32_t* row_counts, float thr, CullPools pools, OrientFuse of, bool ix16,
                              uint32_t tile_rows, hipStream_t stream) {
    if (nwork == 0) return hipSuccess;
    if (mode == 0 && (!of.inv_tgt || !of.tie_count || !of.tie_list)) return hipErrorInvalidValue;   // MODE 0 always fuses
    const uint32_t grid = ((nwork + 7) / 8) * 8;
    if (!(mode == 0 && !brute)) ix16 = false;       // the compact layout is only instantiated for the hot variant
    if (tile_rows && tile_rows != 16u) return hipErrorInvalidValue;
    if (tile_rows && !match_staged(mode, brute)) return hipErrorInvalidValue;   // (the tile form exists for the staged bounded-kNN kernel)
    const uint32_t wpg = tile_rows ? 1u : match_waves_per_group(mode, brute, nwork);
    pools.row_cache = match_row_cache(mode, brute, nwork, tile_rows) ? 1u : 0u;
    const size_t lds = match_lds_bytes(mode, maxK, ix16, wpg, brute, tile_rows, pools.row_cache != 0);
#define L3D_LAUNCH(M, B, X, W, S, T)                                                                          \
    do {                                                                                                      \
        hipError_t e = hipFuncSetAttribute((const void*)k_match_pairs<M, B, X, W, S, T>,                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        if (e != hipSuccess) return e;                                                                        \
        hipLaunchKernelGGL((k_match_pairs<M, B, X, W, S, T>), dim3(grid), dim3(kBlock * W), lds, stream, views, \
                           pairs, work, nwork, slots, row_counts, thr, pools, of);                            \
    } while (0)
#define L3D_LAUNCH_HOT(X, W) do { if (match_staged(mode, brute)) L3D_LAUNCH(0, false, X, W, true, 0); else L3D_LAUNCH(0, false, X, W, false, 0); } while (0)
    if (mode == 0 && tile_rows) {
        if (ix16) L3D_LAUNCH(0, false, true, 1, true, 16); else L3D_LAUNCH(0, false, false, 1, true, 16);
    } else if (mode == 0) {
        if (brute) L3D_LAUNCH(0, true, false, 1, false, 0);
        else if (ix16) { if (wpg == 2) L3D_LAUNCH_HOT(true, 2); else L3D_LAUNCH_HOT(true, 1); }
        else { if (wpg == 2) L3D_LAUNCH_HOT(false, 2); else L3D_LAUNCH_HOT(false, 1); }
    }
    else if (mode == 1) { if (brute) L3D_LAUNCH(1, true, false, 1, false, 0); else L3D_LAUNCH(1, false, false, 1, false, 0); }
    else { if (brute) L3D_LAUNCH(2, true, false, 1, false, 0); else L3D_LAUNCH(2, false, false, 1, false, 0); }
#undef L3D_LAUNCH_HOT
#undef L3D_LAUNCH
    return hipGetLastError();
}

// ---- compact slot exchange (N > 1 ranks) ---------------------------------------------------------------------
// Between ranks only the target index of every slot travels (4 B instead of the 32-byte record): overlap and
// depths are functions of (pair, source row, target index) alone, so the receiving rank re-derives them with the
// device functions the match kernel itself uses (exact_overlap / exact_depths, l3d_dev.h).
__global__ __launch_bounds__(256) void k_pack_slot_idx(const Slot* __restrict__ slots, uint32_t* __restrict__ idx,
                                                       uint64_t lo, uint64_t hi) {
    const uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hi) idx[i] = slots[i].tgt_seg;
}

// grid (ceil(max Ms*K / 256), pairs): one thread per slot of pair first + blockIdx.y
__global__ __launch_bounds__(256) void k_expand_slot_idx(const ViewDev* __restrict__ views,
                                                         const PairDesc* __restrict__ pairs, uint32_t first,
                                                         const uint32_t* __restrict__ idx, Slot* __restrict__ slots,
                                                         const OrientFuse of) {
    const PairDesc& pd = pairs[first + blockIdx.y];
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = pd.Ms * pd.K;
    if (blockIdx.x * blockDim.x >= n) return;   // whole block past the end
    const uint32_t K = pd.K, row = s / K;
    const uint64_t at = pd.slot_off + s;
    if (s >= n) return;
    const uint32_t tg = idx[at];
    Slot o;
    o.tgt_seg = tg; o.overlap = 0; o.dp1 = o.dp2 = o.dq1 = o.dq2 = 0; o.score3D = 0.0f; o.flags = 0;
    if (tg != kEmpty && tg < pd.Mt) {
        const ViewDev& vs = views[pd.src];
        const ViewDev& vt = views[pd.tgt];
        double F[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = pd.F[i];
        const float4 s4 = vs.seg4[row], t4 = vt.seg4[tg];
        o.overlap = exact_overlap(F, s4.x, s4.y, s4.z, s4.w, t4.x, t4.y, t4.z, t4.w);
        PairResult res{};
        const SegX sx = vs.segx[row], tx = vt.segx[tg];
        exact_depths(sx, tx, vs.C, vt.C, res);
        o.dp1 = res.dp1; o.dp2 = res.dp2; o.dq1 = res.dq1; o.dq2 = res.dq2;
        o.flags = fuse_orientation(of, vs.C, vt.C, sx, tx, res, pd.tgt > pd.src);
    } else {
        o.tgt_seg = kEmpty;
    }
    if (s < n) { slots[at] = o; store_inverse_target(of, at, o); }
}

hipError_t launch_pack_slot_idx(const Slot* slots, uint32_t* idx, uint64_t lo, uint64_t hi, hipStream_t stream) {
    if (hi <= lo) return hipSuccess;
    const uint64_t blocks = (hi - lo + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pack_slot_idx, dim3((uint32_t)blocks), dim3(256), 0, stream, slots, idx, lo, hi);
    return hipGetLastError();
}

hipError_t launch_expand_slot_idx(const ViewDev* views, const PairDesc* pairs, uint32_t first, uint32_t count,
                                  uint32_t max_row_slots, const uint32_t* idx, Slot* slots, OrientFuse of,
                                  hipStream_t stream) {
    if (!max_row_slots) return hipSuccess;
    if (!of.inv_tgt) return hipErrorInvalidValue;
    for (uint32_t p0 = 0; p0 < count; p0 += 65535u) {
        const uint32_t n = count - p0 < 65535u ? count - p0 : 65535u;
        hipLaunchKernelGGL(k_expand_slot_idx, dim3((max_row_slots + 255) / 256, n), dim3(256), 0, stream, views, pairs,
                           first + p0, idx, slots, of);
    }
    return hipGetLastError();
}

// ---- epipolar-band culling: order the source rows and the target segments of each pair by tau ------------
namespace {

constexpr int kCullBlock = 512;
constexpr double kCullMinDen = 0.05;   // B.x of a usable point (B.centre == 1): the pencil line is not near-parallel
                                       // to the transversal

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

struct Band { float lo, hi; uint32_t cls; };   // cls 0: unbounded, 1: widened by the direction term, 2: plain

__device__ __forceinline__ Band make_band(double lo, double hi, uint32_t cls) {
    Band b;
    if (!(lo <= hi) || !(fabs(lo) < 1e30) || !(fabs(hi) < 1e30)) {   // NaN / overflow: never culled
        b.lo = -__builtin_inff(); b.hi = __builtin_inff(); b.cls = 0;
        return b;
    }
    // pad well beyond the fp64 rounding of tau and the fp32 rounding of the stored band
    b.lo = (float)(lo - (0.01 + 1e-6 * fabs(lo)));
    b.hi = (float)(hi + (0.01 + 1e-6 * fabs(hi)));
    b.cls = cls;
    return b;
}

// wedge of epipolar lines of a source segment: tau of F*p1 and F*p2
__device__ __forceinline__ Band src_band(const PairCull& pc, const float4 s) {
    const double d1 = pc.Bs[0] * s.x + pc.Bs[1] * s.y + pc.Bs[2];
    const double d2 = pc.Bs[0] * s.z + pc.Bs[1] * s.w + pc.Bs[2];
    if (!(d1 > kCullMinDen) || !(d2 > kCullMinDen)) return make_band(1.0, 0.0, 0);
    const double t1 = (pc.As[0] * s.x + pc.As[1] * s.y + pc.As[2]) / d1;
    const double t2 = (pc.As[0] * s.z + pc.As[1] * s.w + pc.As[2]) / d2;
    return make_band(fmin(t1, t2), fmax(t1, t2), 2);
}

// pencil lines met by a target segment; widened to the pencil line parallel to it (tau of its direction) when
// that line lies inside the span [slo, shi] of all source wedges: only then can a wedge contain the direction,
// the one case in which the two intersection points enclose the segment from outside the tau interval
__device__ __forceinline__ Band tgt_band(const PairCull& pc, const float4 s, float slo, float shi) {
    const double d1 = pc.Bt[0] * s.x + pc.Bt[1] * s.y + pc.Bt[2];
    const double d2 = pc.Bt[0] * s.z + pc.Bt[1] * s.w + pc.Bt[2];
    if (!(d1 > kCullMinDen) || !(d2 > kCullMinDen)) return make_band(1.0, 0.0, 0);
    const double t1 = (pc.At[0] * s.x + pc.At[1] * s.y + pc.At[2]) / d1;
    const double t2 = (pc.At[0] * s.z + pc.At[1] * s.w + pc.At[2]) / d2;
    double lo = fmin(t1, t2), hi = fmax(t1, t2);
    const double dx = (double)s.z - (double)s.x, dy = (double)s.w - (double)s.y;
    const double dd = pc.Bt[0] * dx + pc.Bt[1] * dy;
    const double td = (pc.At[0] * dx + pc.At[1] * dy) / dd;   // +-inf: direction parallel to the transversal
    uint32_t cls = 2;
    if (!(td == td)) return make_band(1.0, 0.0, 0);            // zero-length segment
    if (td >= (double)slo - 1.0 && td <= (double)shi + 1.0) { lo = fmin(lo, td); hi = fmax(hi, td); cls = 1; }
    return make_band(lo, hi, cls);
}

// bitonic sort of n2 (power of two) 64-bit keys in place, one barrier per stage: the views beyond the LDS capacity, whose keys
// (class | exact band start | element) live in global scratch
__device__ void lds_sort(uint64_t* keys, uint32_t n2) {
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
        }
    __syncthreads();
}

// ---- round 6: 32-BIT KEYS for the views that sort in LDS ---------------------------------------------------------------
// key = class (3 bits) | band start, quantised linearly over the span of the source wedges (29 - B bits) | element (B = log2 n2
// bits).  The order of the rows / targets only decides how tight the hulls of 64 consecutive elements are -- every band that is
// stored, compared or reduced afterwards is the element's own exact band --, so a quantised start is as good as the exact one
// as long as its resolution is far below a chunk's extent (16 384 targets: 15 bits over the span, ~0.1 px, against ~15 px per
// chunk; equal starts fall back to the element index).  Half the LDS (two workgroups per CU at 16 384 segments where one
// fitted), half the lane exchanges of the sort's in-wave stages, and a compare-exchange is v_min_u32 / v_max_u32 + one select.
// The bitonic network for n2 = KPT * kCullBlock keys in LDS with the keys held in REGISTERS, thread t owning the KPT consecutive
// elements t*KPT ..: a stage whose partner distance j is below KPT is a register exchange (v_min_u32 / v_max_u32), up to 32
// threads away a lane exchange inside the wave, and only the last three distances (64, 128, 256 threads) go through LDS with
// barriers -- for 2048 keys 6 stages with barriers instead of 66.  The keys are distinct (the element index is part of them).
// (A first 32-bit version selected by lane masks built on the scalar unit, as the list pass of phase B does with its two to
// four keys per thread: with 32 keys per thread the scalar unit became the bound and the kernel was 25 % SLOWER than the
// 64-bit one, profiles/r06_ab_cull_prepare.txt.)
template <int KPT>
__device__ void reg_sort32(uint32_t* keys) {
    constexpr uint32_t n2 = KPT * kCullBlock;
    const uint32_t t = threadIdx.x;
    uint32_t v[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) v[r] = keys[t * KPT + r];
    for (uint32_t k = 2; k <= n2; k <<= 1) {
        // ---- partner in another wave ----
        for (uint32_t j = k >> 1; j >= 64u * KPT; j >>= 1) {
            __syncthreads();                                   // the reads of the previous exchange are done
#pragma unroll
            for (int r = 0; r < KPT; ++r) keys[t * KPT + r] = v[r];
            __syncthreads();
            const uint32_t pt = t ^ (j / KPT);
            const bool lower = (t & (j / KPT)) == 0;
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                const uint32_t o = keys[pt * KPT + r];
                const bool up = ((t * KPT + r) & k) == 0;
                v[r] = (lower == up) ? min(v[r], o) : max(v[r], o);
            }
        }
        // ---- partner in this wave ----
        for (uint32_t j = min(k >> 1, 32u * KPT); j >= (uint32_t)KPT; j >>= 1) {
            const uint32_t d = j / KPT;
            const bool lower = (t & d) == 0;
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                const uint32_t o = __shfl_xor(v[r], (int)d);
                const bool up = ((t * KPT + r) & k) == 0;
                v[r] = (lower == up) ? min(v[r], o) : max(v[r], o);
            }
        }
        // ---- partner in this thread ----
#pragma unroll
        for (int jj = KPT / 2; jj > 0; jj >>= 1) {
            if ((uint32_t)jj < k) {
#pragma unroll
                for (int r = 0; r < KPT; ++r) {
                    if ((r & jj) == 0) {
                        const uint32_t a = v[r], b = v[r | jj];
                        const bool up = ((t * KPT + r) & k) == 0;
                        v[r] = up ? min(a, b) : max(a, b); v[r | jj] = up ? max(a, b) : min(a, b);
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KPT; ++r) keys[t * KPT + r] = v[r];
    __syncthreads();
}
template <int KMAX>
__device__ void cull_sort32(uint32_t* keys, uint32_t n2) {
    const uint32_t kpt = n2 / kCullBlock;
    if (kpt == 1) { reg_sort32<1>(keys); return; }
    if constexpr (KMAX >= 2) if (kpt == 2) { reg_sort32<2>(keys); return; }
    if constexpr (KMAX >= 4) if (kpt == 4) { reg_sort32<4>(keys); return; }
    if constexpr (KMAX >= 8) if (kpt == 8) { reg_sort32<8>(keys); return; }
    if constexpr (KMAX >= 16) if (kpt == 16) { reg_sort32<16>(keys); return; }
    if constexpr (KMAX >= 32) if (kpt == 32) { reg_sort32<32>(keys); return; }
}

}  // namespace

// ---- longest-first launch order -----------------------------------------------------------------------------------
// A work item's length is set by the targets its wave walks: those whose band meets the hull of its 64 rows' bands.
// The spread is large (a group of wide-band rows walks the whole view, a group of narrow ones a twentieth of it), a
// workgroup cannot migrate, and with one to three items per wave slot the launch ends when the longest late starter
// does.  k_order_items (after k_cull_prepare; grid = pairs of the launch) counts, for every item of its pair, the
// targets (beyond 4096 per view: the 64-target chunks) the item will walk and files it in one of kOrderBuckets
// classes; the last workgroup to finish lays the items out by descending class (histogram, scan and cursors in its
// LDS) and the match kernel's workgroups follow that order.  Launches with many rounds of items keep the list
// order: their tail is short against the whole, and the list order keeps the waves of a pair on one XCD.
__global__ __launch_bounds__(kCullBlock) void k_order_items(const PairDesc* __restrict__ pairs, uint32_t first,
                                                            const CullPools cp, uint32_t nwork) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_last;
    const uint32_t p = first + blockIdx.x;
    const PairCull& pc = cp.cull[p];
    const uint32_t Ms = pairs[p].Ms, Mt = pairs[p].Mt;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, n_waves = kCullBlock / 64;
    {   // ---- this pair's items: one wave per item, the pair's target (or chunk) bands staged once in LDS ----
        float2* bands = (float2*)smem;
        const bool by_target = Mt <= 4096;
        const uint32_t n_bands = !pc.enabled ? 0u : by_target ? Mt : (Mt + 63) / 64;
        const float2* src = by_target ? cp.tgt_band + pc.t_off : cp.chunk_band + pc.c_off;
        for (uint32_t i = tid; i < n_bands; i += kCullBlock) bands[i] = src[i];
        __syncthreads();
        const uint32_t n_rows_tot = cp.padded_rows ? tile_src_cap(Ms, 64u) : Ms;   // (padded class layout: kEmpty rows carry an empty band)
        const uint32_t n_items = (n_rows_tot + 63) / 64;
        uint32_t* bucket = cp.item_bucket + (pc.w_item0 - cp.w_base);
        const uint64_t cmax = cp.cost_max ? cp.cost_max : 1u;
        for (uint32_t it = wave; it < n_items; it += n_waves) {
            uint32_t cost = Mt;                  // streamed unculled: every item walks all Mt targets
            if (pc.enabled) {
                const uint32_t r = it * 64 + lane;
                float lo = __builtin_inff(), hi = -__builtin_inff();
                if (r < n_rows_tot) { const float2 b = cp.src_band[pc.s_off + r]; lo = b.x; hi = b.y; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
                uint32_t n = 0;
                for (uint32_t t = lane; t < n_bands; t += 64) {
                    const float2 tb = bands[t];
                    n += !(tb.y < lo || tb.x > hi) ? 1u : 0u;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
                cost = by_target ? n : n * 64;
            }
            if (lane == 0) {   // (agent-scope store: read by the ordering workgroup on another XCD, no L2 write-back needed)
                const uint64_t q = (uint64_t)cost * kOrderBuckets / cmax;
                __hip_atomic_store(&bucket[it], q < kOrderBuckets ? (uint32_t)q : kOrderBuckets - 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- the last workgroup orders all items of the launch ----
    __syncthreads();                             // this workgroup's bucket stores are complete
    if (tid == 0) s_last = atomicAdd(cp.order_done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    // The list is cut into 8 contiguous parts, one per XCD (workgroup b runs on XCD b % 8): the items of a pair stay on
    // one XCD, whose L2 then holds the pair's two views; each part is ordered by itself.
    uint32_t* hist = (uint32_t*)smem;            // [8][kOrderBuckets] counts, then first position of each class
    const uint32_t per_xcd = (nwork + 7) / 8;
    for (uint32_t i = tid; i < 8 * kOrderBuckets; i += kCullBlock) hist[i] = 0;
    // (the buckets are read twice rather than kept in registers: the registers would be every workgroup's, and this
    // kernel is all launch latency -- agent-scope loads: written by other CUs)
    auto key_of = [&](uint32_t w) {
        return (w / per_xcd) * kOrderBuckets + __hip_atomic_load(&cp.item_bucket[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    __syncthreads();
#pragma unroll 4
    for (uint32_t w = tid; w < nwork; w += kCullBlock) atomicAdd(&hist[key_of(w)], 1u);
    __syncthreads();
    {                                            // wave x: descending exclusive scan of the counters of part x
        static_assert(kCullBlock / 64 == 8, "one wave per XCD");
        uint32_t* hx = hist + wave * kOrderBuckets;
        constexpr uint32_t per = kOrderBuckets / 64;
        uint32_t h[per], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) { h[k] = hx[kOrderBuckets - 1 - (lane * per + k)]; sum += h[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
        uint32_t run = wave * per_xcd + incl - sum;
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) { hx[kOrderBuckets - 1 - (lane * per + k)] = run; run += h[k]; }
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t w = tid; w < nwork; w += kCullBlock) {
        const uint32_t pos = atomicAdd(&hist[key_of(w)], 1u);
        if (pos < nwork) cp.item_order[pos] = w;
    }
    if (tid == 0) *cp.order_done = 0;            // re-armed for the next launch
}

hipError_t launch_order_items(const PairDesc* pairs, uint32_t first, uint32_t count, uint32_t max_Mt, CullPools pools,
                              uint32_t nwork, hipStream_t stream) {
    if (!pools.item_order || !nwork || !count) return hipSuccess;
    const size_t lds = std::max<size_t>((size_t)(max_Mt <= 4096 ? max_Mt : (max_Mt + 63) / 64) * 8, 8 * kOrderBuckets * 4);
    hipError_t e = hipFuncSetAttribute((const void*)k_order_items, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_order_items, dim3(count), dim3(kCullBlock), lds, stream, pairs, first, pools, nwork);
    return hipGetLastError();
}

// tile_rows: 0 = the row form's layout (Ms positions, at most two width classes from 4096 segments on); R = 16 / 32 = the
// tile form's: kTileClasses width classes at every view size, each padded with kEmpty rows to a multiple of R, in a region
// of tile_src_cap(Ms, R) positions
template <int KMAX>
__global__ __launch_bounds__(kCullBlock) void k_cull_prepare(const ViewDev* __restrict__ views,
                                                             const PairDesc* __restrict__ pairs, uint32_t first,
                                                             const CullPools cp, const uint32_t tile_rows,
                                                             const uint32_t band_cache) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t p = first + blockIdx.x;
    const PairCull& pc = cp.cull[p];
    const PairDesc& pd = pairs[p];
    const uint32_t Ms = pd.Ms, Mt = pd.Mt;
    const bool tgt_side = blockIdx.y != 0;       // grid.y = 2: the two sorts of a pair run in different workgroups
    if (!pc.enabled) return;
    const ViewDev& vs = views[pd.src];
    const ViewDev& vt = views[pd.tgt];
    uint32_t n2 = kCullBlock;                    // (at least one key per thread: reg_sort)
    while (n2 < (tgt_side ? Mt : Ms)) n2 <<= 1;
    // keys in LDS up to kCullLdsSegs per side; beyond that in this pair's global scratch (same code, L2 instead of LDS:
    // slower, but a view of 16 385 segments no longer falls back to unculled matching)
    uint32_t n2s = 64;
    while (n2s < Ms) n2s <<= 1;
    const bool big = pc.k_off != ~0ull;
    uint64_t* keys = big ? cp.big_keys + pc.k_off + (tgt_side ? n2s : 0u) : (uint64_t*)smem;
    // (round 6) views that sort in LDS: 32-bit keys -- class | quantised band start | element
    uint32_t* keys32 = (uint32_t*)smem;
    const bool k32 = !big;
    uint32_t* cb = big ? (uint32_t*)smem : (keys32 + n2);                // [2 * n2/64] chunk bands, orderable floats
    // (round 6) BAND CACHE, sides of up to kCullBandCacheSegs elements: a band costs two or three double-precision divisions
    // and was evaluated three times per element (span / key, key, final write) -- half of the kernel's time; the side's own
    // bands are kept in LDS after their first evaluation (lo, hi: 8 bytes per element)
    const bool cache = k32 && band_cache != 0;   // (the launch's decision: its LDS is sized by its largest view)
    float2* bandc = (float2*)(cb + 2 * (n2 / 64));
    __shared__ uint32_t span[2];
    const uint32_t tid = threadIdx.x;
    uint32_t kbits = 0;
    while ((1u << kbits) < n2) ++kbits;                                  // element bits of a 32-bit key
    const uint32_t idx_mask = k32 ? (1u << kbits) - 1u : 0xFFFFFFu, qmax = (1u << (29u - kbits)) - 1u;
    auto key_elem = [&](uint32_t i) -> uint32_t { return k32 ? (keys32[i] & idx_mask) : (uint32_t)(keys[i] & 0xFFFFFFu); };

    // The range the 32-bit keys quantise a band start over, span[0 .. 1] as orderable floats: the target side takes the span of
    // all bounded source wedges (it needs that span anyway -- empty: no bounded row, nothing can be widened, nothing is culled
    // either --; a target outside it meets no wedge, where it sorts is of no consequence), the source side the range of its own
    // bounded starts, reduced while it builds its keys
    if (tid == 0) { span[0] = 0xFFFFFFFFu; span[1] = 0u; }
    __syncthreads();
    auto reduce_span = [&](uint32_t lo_o, uint32_t hi_o) {   // (one LDS atomic per wave: two per segment on the same two
                                                             // addresses serialised the whole pass)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo_o = min(lo_o, (uint32_t)__shfl_xor(lo_o, o)); hi_o = max(hi_o, (uint32_t)__shfl_xor(hi_o, o)); }
        if ((tid & 63u) == 0) { atomicMin(&span[0], lo_o); atomicMax(&span[1], hi_o); }
    };
    if (tgt_side) {
        uint32_t lo_o = 0xFFFFFFFFu, hi_o = 0u;
        for (uint32_t i = tid; i < Ms; i += kCullBlock) {
            const Band b = src_band(pc, vs.seg4[i]);
            if (b.cls) { lo_o = min(lo_o, f2ord(b.lo)); hi_o = max(hi_o, f2ord(b.hi)); }
        }
        reduce_span(lo_o, hi_o);
        __syncthreads();
    }
    // linear over [qlo, qhi], clamped; an empty or degenerate range quantises everything to 0 (element order)
    auto key32_of = [&](uint32_t cls, float lo, uint32_t i, float qlo, float qscale) -> uint32_t {
        float q = (lo - qlo) * qscale;
        q = q > 0.0f ? q : 0.0f;                                        // (NaN / -inf -> 0)
        const uint32_t qi = q < (float)qmax ? (uint32_t)q : qmax;
        return (cls << 29) | (qi << kbits) | i;
    };
    auto qscale_of = [&](uint32_t o_lo, uint32_t o_hi) -> float {
        const float a = ord2f(o_lo), b = ord2f(o_hi);
        return (o_lo <= o_hi && b > a && (b - a) < 1e30f) ? (float)qmax / (b - a) : 0.0f;
    };

    if (!tgt_side) {
        // ---- source rows: key = (class, lo, row) ----
        // A wave tests every target that meets the HULL of its 64 rows' bands, so one wide band makes 63 narrow rows
        // test targets that cannot concern them.  In large views (>= 4096 segments: the spread of 64 neighbouring lo
        // values is small there and the widest row decides the hull) rows with wide bands are therefore ordered -- and
        // thus grouped -- apart.  tau runs along a transversal through the target image, so widths are judged against
        // its size: one class beyond 1/16 of it, from 8192 segments on classes beyond 1/32 and 1/8 (C4: match kernel
        // 49.3 -> 40.6 ms, C2: 19.7 -> 19.0 ms).  Small views are left alone: a row group whose hull spans the whole
        // view is a work item four times the average, and with only one or two items per wave slot (C1) the launch
        // then waits for it (measured 1.06 -> 1.21 ms).  The row order does not affect the result.
        const float ref = vt.cx + vt.cy;   // (width + height) / 2
        const bool two = Ms >= 8192;
        const float w1 = Ms >= 4096 ? ref * (two ? 1.0f / 32.0f : 1.0f / 16.0f) : __builtin_inff(), w2 = ref * (1.0f / 8.0f);
        uint32_t own_lo = 0xFFFFFFFFu, own_hi = 0u;
        for (uint32_t i = tid; i < n2; i += kCullBlock) {
            uint64_t key = ~0ull;
            if (i < Ms) {
                const Band b = src_band(pc, vs.seg4[i]);
                if (cache) bandc[i] = make_float2(b.lo, b.hi);
                if (b.cls) { own_lo = min(own_lo, f2ord(b.lo)); own_hi = max(own_hi, f2ord(b.lo)); }
                // order: unbounded, widest, wide, plain -- the row groups with the largest hulls are the longest work
                // items and must start first (ordered last they lengthen the ramp-down tail of the launch)
                uint32_t cls = b.cls;   // 0: unbounded, 2: bounded
                if (cls == 2) {
                    const float w = b.hi - b.lo;
                    // (padded layout: how many width classes pay depends on how many items a class is cut into -- with few
                    // items per class 64 consecutive rows of a class spread further in tau than the class saves in width;
                    // tools/hull_sim.py on C1 / C2 / C4 for R = 16 / 32 / 64: no cut below 24 items per pair (tile_classes), one at 1/16 of
                    // the image below 48, 1/64 and 1/16 below 192, 1/64, 1/32 and 1/16 from there on)
                    if (tile_rows) {
                        const uint32_t items = Ms / tile_rows;
                        const float c16 = items >= kTileMinItems ? ref * (1.0f / 16.0f) : __builtin_inff(),   // (few items: one class)
                                    c32 = items >= 192u ? ref * (1.0f / 32.0f) : c16, c64 = items >= 48u ? ref * (1.0f / 64.0f) : c16;
                        cls = w > c16 ? 1u : w > c32 ? 2u : w > c64 ? 3u : 4u;
                    }
                    else cls = w > w1 ? ((two && w > w2) ? 1u : 2u) : 3u;
                }
                key = ((uint64_t)cls << 56) | ((uint64_t)f2ord(b.lo) << 24) | i;
                // (provisional: the start with its low three bits given to the class, until the range is known)
                if (k32) keys32[i] = (f2ord(b.lo) & ~7u) | cls;
            } else if (k32) keys32[i] = ~0u;
            if (!k32) keys[i] = key;
        }
        if (k32) {
            reduce_span(own_lo, own_hi);
            __syncthreads();
            const float qlo = ord2f(span[0]), qs = qscale_of(span[0], span[1]);
            for (uint32_t i = tid; i < Ms; i += kCullBlock) {
                const uint32_t tmp = keys32[i];
                keys32[i] = key32_of(tmp & 7u, ord2f(tmp & ~7u), i, qlo, qs);
            }
        }
        __syncthreads();
        if (k32) cull_sort32<KMAX>(keys32, n2); else lds_sort(keys, n2);   // (beyond the LDS capacity: 64-bit keys in global scratch)
        if (tile_rows) {
            // Tile form (k_match_pairs<..., TILE>): a work item is R consecutive positions, and its hull must not straddle
            // two classes (the last rows of one class and the first of the next are far apart in tau: such an item would
            // walk the whole view) -- so every class starts at a multiple of R; the gap is padded with kEmpty rows, which
            // the match kernel skips.  The class sizes are only known here, so the host cuts tile_src_cap(Ms, R) positions
            // into items and the positions beyond the last class stay kEmpty as well.
            __shared__ uint32_t cstart[kTileClasses + 1], cbase[kTileClasses + 1];
            const uint32_t R = tile_rows, cap = tile_src_cap(Ms, R);
            for (uint32_t i = tid; i < cap; i += kCullBlock) {
                cp.src_perm[pc.s_off + i] = kEmpty;
                cp.src_band[pc.s_off + i] = make_float2(__builtin_inff(), -__builtin_inff());
            }
            if (tid <= kTileClasses) {          // first sorted position of class tid (binary search; padding keys sort last)
                const uint64_t want = (uint64_t)tid << 56;
                uint32_t lo = 0, hi = Ms;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (k32 ? (keys32[mid] >> 29) < tid : keys[mid] < want) lo = mid + 1; else hi = mid;
                }
                cstart[tid] = lo;
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t base = 0;
                for (uint32_t k = 0; k < kTileClasses; ++k) { cbase[k] = base; base += ((cstart[k + 1] - cstart[k] + R - 1) / R) * R; }
                cbase[kTileClasses] = base;
            }
            __syncthreads();                    // (also: the kEmpty fill above is complete before the rows are placed)
            for (uint32_t i = tid; i < Ms; i += kCullBlock) {
                const uint32_t row = key_elem(i), k = k32 ? keys32[i] >> 29 : (uint32_t)(keys[i] >> 56);
                const Band b = cache ? Band{bandc[row].x, bandc[row].y, 0u} : src_band(pc, vs.seg4[row]);
                const uint32_t at = cbase[k] + (i - cstart[k]);
                cp.src_perm[pc.s_off + at] = row;
                cp.src_band[pc.s_off + at] = make_float2(b.lo, b.hi);
            }
            return;
        }
        for (uint32_t i = tid; i < Ms; i += kCullBlock) {
            const uint32_t row = key_elem(i);
            const Band b = cache ? Band{bandc[row].x, bandc[row].y, 0u} : src_band(pc, vs.seg4[row]);
            cp.src_perm[pc.s_off + i] = row;
            cp.src_band[pc.s_off + i] = make_float2(b.lo, b.hi);
        }
        return;
    }
    // ---- target segments ----
    const float slo = ord2f(span[0]), shi = ord2f(span[1]);
    const float tq = qscale_of(span[0], span[1]);
    const uint32_t nchunk = (Mt + 63) / 64;
    for (uint32_t i = tid; i < nchunk; i += kCullBlock) { cb[2 * i] = 0xFFFFFFFFu; cb[2 * i + 1] = 0u; }
    for (uint32_t i = tid; i < n2; i += kCullBlock) {
        uint64_t key = ~0ull;
        if (i < Mt) {
            const Band b = tgt_band(pc, vt.seg4[i], slo, shi);
            if (cache) bandc[i] = make_float2(b.lo, b.hi);
            key = ((uint64_t)b.cls << 56) | ((uint64_t)f2ord(b.lo) << 24) | i;
            if (k32) keys32[i] = key32_of(b.cls, b.lo, i, slo, tq);
        } else if (k32) keys32[i] = ~0u;
        if (!k32) keys[i] = key;
    }
    __syncthreads();
    if (k32) cull_sort32<KMAX>(keys32, n2); else lds_sort(keys, n2);   // (beyond the LDS capacity: 64-bit keys in global scratch)
    // (i = tid + k * 512: the 64 lanes of a wave hold exactly one 64-target chunk, so its band is a wave reduction and
    // one plain store -- 128 LDS atomics per chunk on two addresses before)
    for (uint32_t i0 = 0; i0 < Mt; i0 += kCullBlock) {
        const uint32_t i = i0 + tid;
        uint32_t lo_o = 0xFFFFFFFFu, hi_o = 0u;
        if (i < Mt) {
            const uint32_t seg = key_elem(i);
            const Band b = cache ? Band{bandc[seg].x, bandc[seg].y, 0u} : tgt_band(pc, vt.seg4[seg], slo, shi);
            cp.tgt_perm[pc.t_off + i] = seg;
            cp.tgt_sf[pc.t_off + i] = *(const float4*)&vt.segf[seg];
            if (pc.sorted_copy) {
                cp.tgt_s4[pc.t_off + i] = vt.seg4[seg];
                cp.tgt_sd[pc.t_off + i] = vt.segd32[seg];
            }
            cp.tgt_band[pc.t_off + i] = make_float2(b.lo, b.hi);
            lo_o = f2ord(b.lo); hi_o = f2ord(b.hi);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo_o = min(lo_o, (uint32_t)__shfl_xor(lo_o, o)); hi_o = max(hi_o, (uint32_t)__shfl_xor(hi_o, o)); }
        if ((tid & 63u) == 0 && i < Mt) { cb[2 * (i >> 6)] = lo_o; cb[2 * (i >> 6) + 1] = hi_o; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nchunk; i += kCullBlock)
        cp.chunk_band[pc.c_off + i] = make_float2(ord2f(cb[2 * i]), ord2f(cb[2 * i + 1]));
}

hipError_t launch_cull_prepare(const ViewDev* views, const PairDesc* pairs, uint32_t first, uint32_t count,
                               uint32_t max_M, CullPools pools, uint32_t tile_rows, hipStream_t stream) {
    if (!count || !pools.cull) return hipSuccess;
    uint32_t n2 = kCullBlock;
    while (n2 < max_M) n2 <<= 1;
    // (views beyond the LDS capacity keep their keys in global scratch: only the chunk bands stay in LDS)
    const size_t lds = n2 <= kCullLdsSegs ? (size_t)n2 * 4 + (size_t)(n2 / 64) * 8 + (n2 <= kCullBandCacheSegs ? (size_t)n2 * 8 : 0)   // (32-bit keys; band cache)
                                           : std::max<size_t>((size_t)kCullLdsSegs * 8 + (kCullLdsSegs / 64) * 8, (size_t)(n2 / 64) * 8);
#define L3D_CULL(K)                                                                                                        \
    do {                                                                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)k_cull_prepare<K>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                           (int)lds);                                                                      \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL(k_cull_prepare<K>, dim3(count, 2), dim3(kCullBlock), lds, stream, views, pairs, first, pools,   \
                           tile_rows, n2 <= kCullBandCacheSegs ? 1u : 0u);                                                 \
    } while (0)
    const uint32_t kmax = std::min(n2, kCullLdsSegs) / kCullBlock;
    if (kmax <= 1) L3D_CULL(1); else if (kmax == 2) L3D_CULL(2); else if (kmax == 4) L3D_CULL(4);
    else if (kmax == 8) L3D_CULL(8); else if (kmax == 16) L3D_CULL(16); else L3D_CULL(32);
#undef L3D_CULL
    return hipGetLastError();
}

// ---- rows with equal overlaps: the reference's kNN order, replayed ------------------------------------------------
// Line3D::matchingCPU keeps a row's accepted matches in a std::priority_queue keyed by the overlap alone and pops
// kNN of them (line3D.cc:982-1007): with equal overlaps, which entries come out and in which order is the pop order
// of libstdc++'s binary heap.  k_match_pairs selects by (overlap desc, target asc), which is the same thing for
// distinct overlaps, and hands every row in which it saw equal overlaps -- inside the table or at the kNN-th place --
// to this kernel (one wave per row): all Mt targets through the exact test in ascending target order, accepted ones
// pushed into the very heap the reference builds (l3d_heap.h), kNN pops, then the same slot / orientation / counter
// work as the match epilogue.  Such rows are rare (C1: none, C4: a few per 10^5 rows), duplicated segments make many.
constexpr uint32_t kTieBlock = 512;       // threads per tied row (all Mt exact tests in Mt/512 steps)
constexpr uint32_t kTieLdsHeap = 1024;    // accepted (overlap, target) entries kept in LDS (8 KiB + 8 KiB sort buffer: four
                                          // workgroups per CU); more: the workgroup's global scratch

// arrival order -> ascending target order (rank by counting: targets are distinct), then the reference's heap: push
// all, pop K (l3d_heap.h).  One thread walks the heap -- every access is a dependent round trip, which is why the
// entries are packed and why the LDS case is instantiated with LDS pointers rather than generic ones.
template <class Ptr>
__device__ __forceinline__ uint32_t tie_select(Ptr list, Ptr sorted, uint32_t n, uint32_t K, float* win_ov, uint32_t* win_ix,
                                               uint32_t tid) {
    for (uint32_t i = tid; i < n; i += kTieBlock) {
        const uint64_t me = list[i];
        uint32_t rank = 0;
        for (uint32_t y = 0; y < n; ++y) rank += (uint32_t)list[y] < (uint32_t)me ? 1u : 0u;
        sorted[rank] = me;
    }
    __threadfence_block();
    __syncthreads();
    uint32_t w = 0;
    if (tid == 0) {
        for (uint32_t i = 1; i < n; ++i) heap_push_packed(sorted, i, sorted[i]);   // in place: the heap is the prefix
        uint32_t left = n;
        while (w < K && left > 0) {
            const uint64_t e = heap_pop_packed(sorted, left);
            --left;
            win_ov[w] = heap_overlap(e); win_ix[w] = (uint32_t)e; ++w;
        }
    }
    return w;
}

// of.tie_count = rows queued by the match kernel that has just run
__global__ __launch_bounds__(kTieBlock) void k_match_tied_rows(const ViewDev* __restrict__ views,
                                                               const PairDesc* __restrict__ pairs,
                                                               Slot* __restrict__ slots, float thr, const OrientFuse of,
                                                               const CullPools cp, uint64_t* __restrict__ scratch,
                                                               uint32_t scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint64_t s_list[kTieLdsHeap], s_sorted[kTieLdsHeap];
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_nwin, s_nvis;
    float* win_ov = (float*)smem;                 // [K] dynamic
    const uint32_t tid = threadIdx.x;
    const uint32_t n_tied = min(*of.tie_count, of.tie_cap);
    uint64_t* glist = scratch + (size_t)blockIdx.x * 2 * scratch_stride;      // [2 stride]: list + sort buffer
    for (uint32_t t = blockIdx.x; t < n_tied; t += gridDim.x) {
        const uint2 item = of.tie_list[t];
        const PairDesc& pd = pairs[item.x];
        const uint32_t src = item.y, K = pd.K, Mt = pd.Mt;
        uint32_t* win_ix = (uint32_t*)(win_ov + K);
        const ViewDev& vs = views[pd.src];
        const ViewDev& vt = views[pd.tgt];
        double F[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = pd.F[i];
        const float4 s4 = vs.seg4[src];
        const SegX sx = vs.segx[src];
        // the row's two epipolar lines for the conservative fp32 pre-filter (as the prologue of k_match_pairs): only
        // its survivors go through the double-precision test
        float e1x = 0, e1y = 0, e1z = 0, e2x = 0, e2y = 0, e2z = 0;
        bool live = false;
        {
            const d3 e1 = mul33(F, d3{(double)s4.x, (double)s4.y, 1.0});
            const d3 e2 = mul33(F, d3{(double)s4.z, (double)s4.w, 1.0});
            const double n1 = sqrt(e1.x * e1.x + e1.y * e1.y), n2 = sqrt(e2.x * e2.x + e2.y * e2.y);
            if (n1 > 0.0 && n2 > 0.0) {
                const double cx = (double)vt.cx, cy = (double)vt.cy;
                e1x = (float)(e1.x / n1); e1y = (float)(e1.y / n1); e1z = (float)((e1.z + (e1.x * cx + e1.y * cy)) / n1);
                e2x = (float)(e2.x / n2); e2y = (float)(e2.y / n2); e2z = (float)((e2.z + (e2.x * cx + e2.y * cy)) / n2);
                live = true;
            }
        }
        // ---- every target that can match through the acceptance test.  A culled pair (k_cull_prepare ran for it):
        // only the chunks / targets whose epipolar band meets the row's, in band order -- the accepted ones are put
        // back into ascending target order afterwards, the order in which the reference pushes them ----
        const PairCull* pc = (cp.cull && cp.cull[item.x].enabled) ? &cp.cull[item.x] : nullptr;
        float blo = -__builtin_inff(), bhi = __builtin_inff();
        if (pc) { const Band bb = src_band(*pc, s4); blo = bb.lo; bhi = bb.hi; }
        if (tid == 0) s_n = 0;
        __syncthreads();
        // p: sorted position (culled pair) or target index
        auto visit = [&](uint32_t p) {
            bool acc = false;
            uint32_t cc = p;
            PairResult res{};
            if (p < Mt && live) {
                bool cand;
                if (pc) {
                    const float2 tb = cp.tgt_band[pc->t_off + p];
                    cand = !(tb.y < blo || tb.x > bhi);
                    if (cand) {
                        const float4 f = cp.tgt_sf[pc->t_off + p];
                        const v4f q = {f.x, f.y, f.z, f.w};
                        cand = prefilter(e1x, e1y, e1z, e2x, e2y, e2z, q, thr);
                        if (cand) cc = cp.tgt_perm[pc->t_off + p];
                    }
                } else {
                    const SegF f = vt.segf[p];
                    const v4f q = {f.qx, f.qy, f.dx, f.dy};
                    cand = prefilter(e1x, e1y, e1z, e2x, e2y, e2z, q, thr);
                }
                if (cand) acc = exact_pair(F, s4, vt.seg4[cc], sx, vt.segx[cc], vs.C, vt.C, thr, res);
            }
            // accepted ones are appended in any order (one LDS atomic each, no barrier in the loops): they are sorted
            // by target index below anyway
            if (acc) {
                const uint32_t k = atomicAdd(&s_n, 1u);
                const uint64_t e = heap_pack(res.overlap, cc);
                if (k < kTieLdsHeap) s_list[k] = e;
                else glist[k] = e;
            }
        };
        if (pc) {
            // the chunks whose band meets the row's are listed first (one pass over the chunk bands), then their targets
            // are visited 512 at a time: the row's neighbourhood is a fifth of the view, i.e. one or two rounds of
            // dependent loads instead of Mt / 512 (the list lives in the sort buffer, not needed yet)
            uint32_t* s_vis = (uint32_t*)s_sorted;
            constexpr uint32_t kVisCap = 2 * kTieLdsHeap;
            const uint32_t nch = (Mt + 63) / 64;
            for (uint32_t cb0 = 0; cb0 < nch; cb0 += kVisCap) {
                if (tid == 0) s_nvis = 0;
                __syncthreads();
                for (uint32_t c = cb0 + tid; c < min(nch, cb0 + kVisCap); c += kTieBlock) {
                    const float2 cb = cp.chunk_band[pc->c_off + c];
                    if (!(cb.y < blo || cb.x > bhi)) s_vis[atomicAdd(&s_nvis, 1u)] = c;
                }
                __syncthreads();
                const uint32_t n_units = s_nvis * 64;
                for (uint32_t u0 = 0; u0 < n_units; u0 += kTieBlock) {
                    const uint32_t u = u0 + tid;
                    if (u < n_units) visit(s_vis[u >> 6] * 64 + (u & 63u));
                }
                __syncthreads();               // before the list is rebuilt
            }
        } else {
            for (uint32_t c0 = 0; c0 < Mt; c0 += kTieBlock) visit(c0 + tid);
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t n = s_n;
        // ---- the reference's heap: push all in ascending target order, pop K ----
        uint32_t w;
        if (n <= kTieLdsHeap) {
            w = tie_select(s_list, s_sorted, n, K, win_ov, win_ix, tid);
        } else {                 // outgrown the LDS list: continue in this block's global scratch
            for (uint32_t i = tid; i < kTieLdsHeap; i += kTieBlock) glist[i] = s_list[i];
            __threadfence_block();
            __syncthreads();
            w = tie_select(glist, glist + scratch_stride, n, K, win_ov, win_ix, tid);
        }
        if (tid == 0) s_nwin = w;
        __syncthreads();
        const uint32_t n_win = s_nwin;
        const bool hands_inverse = pd.tgt > pd.src;
        // ---- the row's K slots: depths, orientation filter (as the match epilogue) ----
        for (uint32_t j0 = 0; j0 < K; j0 += kTieBlock) {
            const uint32_t j = j0 + tid;
            Slot o;
            o.tgt_seg = kEmpty; o.overlap = 0; o.dp1 = o.dp2 = o.dq1 = o.dq2 = 0; o.score3D = 0.0f; o.flags = 0;
            if (j < n_win) {
                const uint32_t xj = win_ix[j];
                const SegX tx = vt.segx[xj];
                PairResult res{};
                exact_depths(sx, tx, vs.C, vt.C, res);
                res.overlap = win_ov[j];
                o.tgt_seg = xj; o.overlap = res.overlap;
                o.dp1 = res.dp1; o.dp2 = res.dp2; o.dq1 = res.dq1; o.dq2 = res.dq2;
                o.flags = fuse_orientation(of, vs.C, vt.C, sx, tx, res, hands_inverse);
            }
            if (j < K) {
                const uint64_t at = pd.slot_off + (uint64_t)src * K + j;
                slots[at] = o;
                store_inverse_target(of, at, o);
            }
        }
        __syncthreads();
    }
    // The queue of the NEXT match launch is the other one of two alternating counters (tie_next): it is idle while this
    // kernel runs, so one thread zeroes it and no workgroup has to wait for, or count, the others (512 same-address
    // device atomics were a third of this kernel's time).
    if (blockIdx.x == 0 && tid == 0) {
        *of.tie_next = 0;
        of.tie_total[0] += n_tied;              // rows replayed since the context was created (diagnostics)
    }
}

// kNN beyond what the per-row top-K tables of k_match_pairs hold in LDS: EVERY row is replayed by k_match_tied_rows (the
// reference's own selection: all accepted matches pushed into its heap, kNN pops, line3D.cc:982-1007), whose winners
// live in dynamic LDS sized by kNN and whose heap spills to global scratch.  This kernel queues the rows of the pairs
// [first, first + count): grid (row blocks, pairs).
__global__ __launch_bounds__(256) void k_queue_all_rows(const PairDesc* __restrict__ pairs, uint32_t first,
                                                        const OrientFuse of, uint32_t list_base) {
    const PairDesc& pd = pairs[first + blockIdx.y];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= pd.Ms) return;
    // rows of a pair are contiguous in the list: position = rows of the earlier pairs of the launch + r
    const uint32_t pos = (pd.row_off - list_base) + r;
    if (pos < of.tie_cap) of.tie_list[pos] = make_uint2(first + blockIdx.y, r);
    atomicAdd(of.tie_count, 1u);
}
hipError_t launch_queue_all_rows(const PairDesc* pairs, uint32_t first, uint32_t count, uint32_t max_Ms, uint32_t list_base,
                                 OrientFuse of, hipStream_t stream) {
    if (!count || !max_Ms) return hipSuccess;
    if (!of.tie_count || !of.tie_list) return hipErrorInvalidValue;
    for (uint32_t p0 = 0; p0 < count; p0 += 65535u) {
        const uint32_t n = count - p0 < 65535u ? count - p0 : 65535u;
        hipLaunchKernelGGL(k_queue_all_rows, dim3((max_Ms + 255) / 256, n), dim3(256), 0, stream, pairs, first + p0, of, list_base);
    }
    return hipGetLastError();
}

// two workgroups per CU -- what its registers let be resident at once: more would only queue up behind them, and with
// the usual handful of rows the launch is all latency; fewer when the per-workgroup scratch (16 bytes per target)
// would pass 128 MiB in total
uint32_t match_tied_grid(uint32_t scratch_stride) {
    const uint64_t per_wg = 2ull * std::max(scratch_stride, 1u) * 8;
    return (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(64, (128ull << 20) / per_wg));
}

hipError_t launch_match_tied_rows(const ViewDev* views, const PairDesc* pairs, Slot* slots, uint32_t maxK, float thr,
                                  OrientFuse of, CullPools cp, uint64_t* scratch, uint32_t scratch_stride,
                                  hipStream_t stream) {
    if (!of.tie_count || !of.tie_next || !of.tie_total || !of.tie_list || !scratch) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_match_tied_rows, dim3(match_tied_grid(scratch_stride)), dim3(kTieBlock), (size_t)maxK * 8, stream,
                       views, pairs, slots, thr, of, cp, scratch, scratch_stride);
    return hipGetLastError();
}

// ---- keep-all mode (kNN <= 0, line3D.cc:982-992): the rows of the single culled pass ----------------------------------
// Round 6.  Until then the mode walked every pair twice -- a count pass through the culled walk, then (the rows sized by the
// host from the counts, K of a pair = its LONGEST row) a fill pass streamed unculled because the reference keeps a row's matches
// in ascending target order, writing every slot of the padded rows: C1 65.5 M slots of 32 bytes for 19.0 M matches.  Now the
// count pass keeps what it accepts -- the slot, as a 32-byte cell -- at (row, arrival index) in a per-row scratch, a scan of the
// counts gives every row its place in a RAGGED slot buffer (row r of a pair = slots [row_start[row_off + r], row_start[row_off
// + r + 1])), and k_keep_assemble ranks each record among its row's by target index, applies checkMatchOrientation and writes
// the slot, the streams of phase B and the slot's source row.
// k_keep_pair_info (grid = pairs): per pair {first slot, longest row, number of slots (64 bits)} for the host, which sizes the
// slot buffer; per row its pair; per block of kKeepBlock slots the row that holds the block's first slot.
__global__ __launch_bounds__(256) void k_keep_pair_info(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ row_counts,
                                                        const uint32_t* __restrict__ row_start, uint4* __restrict__ info,
                                                        uint32_t* __restrict__ row_pair, uint32_t* __restrict__ blk_row,
                                                        uint32_t n_blk, uint32_t* __restrict__ longest) {
    __shared__ uint32_t s_max[4];
    __shared__ unsigned long long s_sum[4];
    const PairDesc& pd = pairs[blockIdx.x];
    uint32_t mx = 0;
    unsigned long long sum = 0;
    for (uint32_t r = threadIdx.x; r < pd.Ms; r += 256) {
        const uint32_t gr = pd.row_off + r, c = row_counts[gr], s = row_start[gr];
        mx = max(mx, c); sum += c;
        row_pair[gr] = blockIdx.x;
        // (32-bit slot indices: a scene beyond 2^32 matches is refused by the host, which sees the 64-bit sums; its marks are unused)
        // (n_blk marks: rows x scratch cells -- only a pass with rows beyond the scratch, which is repeated, can want more)
        for (uint32_t b = (s + kKeepBlock - 1) / kKeepBlock; c && b < n_blk && (uint64_t)b * kKeepBlock < (uint64_t)s + c; ++b) blk_row[b] = gr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); sum += (unsigned long long)__shfl_xor((long long)sum, o); }
    if ((threadIdx.x & 63u) == 0) { s_max[threadIdx.x >> 6] = mx; s_sum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mx = max(mx, s_max[w]); sum += s_sum[w]; }
        info[blockIdx.x] = make_uint4(row_start[pd.row_off], mx, (uint32_t)sum, (uint32_t)(sum >> 32));
        atomicMax(longest, mx);                                   // (the assembly behind this kernel checks it against the scratch)
    }
}

// grid = blocks of kKeepBlock slots of the whole (ragged) slot buffer: one thread per accepted match = per slot.  A thread finds
// its row among the row starts its workgroup staged in LDS (from the block's first row, k_keep_pair_info) and its place in the
// row by counting the row's smaller targets, which its neighbours hold: through LDS.  (First form: one grid per pair, the row by
// a binary search in memory and the count through vector loads -- ~50 loads per wave of the counting loop, every one a
// quarter-rate address pass of 64 lanes whatever they coalesce to, behind 11 dependent loads of the search: 0.70 ms for C1's
// 19 M records.)
constexpr uint32_t kAsmRows = 2048;
__global__ __launch_bounds__(kKeepBlock) void k_keep_assemble(const ViewDev* __restrict__ views, const PairDesc* __restrict__ pairs,
                                                              uint32_t n_rows, uint32_t slot_cap,
                                                              const uint32_t* __restrict__ rs, const uint32_t* __restrict__ row_pair,
                                                              const uint32_t* __restrict__ blk_row,
                                                              const uint32_t* __restrict__ longest,
                                                              Slot* __restrict__ slots, const OrientFuse of) {
    __shared__ uint32_t s_tg[kKeepBlock];
    __shared__ uint32_t s_rs[kAsmRows];
    // The launch is enqueued right behind the scan, before the host knows the number of slots: the grid covers the CAPACITY of
    // the output arrays (sized from the previous call), the number of slots is read here; more slots than capacity: nothing is
    // written, the host sees the total with the pair table, enlarges the arrays and launches again.
    const uint32_t n_slots = rs[n_rows];
    const uint32_t at0 = blockIdx.x * kKeepBlock, tid = threadIdx.x;
    if (n_slots > slot_cap || at0 >= n_slots) return;             // (uniform)
    if (*longest > of.keep_cap) return;                           // a row outgrew the scratch: the pass is repeated, its records are incomplete
    const uint32_t rf = blk_row[blockIdx.x];                      // the row that holds slot at0 (rs[rf] <= at0 < rs[rf + 1])
    for (uint32_t i = tid; i < kAsmRows; i += kKeepBlock) s_rs[i] = rs[min(rf + i, n_rows)];   // (rs[n_rows] = n_slots)
    __syncthreads();
    const uint32_t at = at0 + tid;
    const bool on = at < n_slots;
    uint32_t gr = 0, r0 = 0, n = 0, tg = ~0u;
    Slot me{};
    const Slot* __restrict__ rec = of.keep_rec;
    if (on) {
        // the last row that starts at or before `at` (empty rows share their successor's start)
        uint32_t l = 0, h = kAsmRows;                             // s_rs[l] <= at; s_rs[h] > at or h = kAsmRows
        while (h - l > 1) { const uint32_t m = (l + h) >> 1; if (s_rs[m] <= at) l = m; else h = m; }
        gr = rf + l;
        if (l == kAsmRows - 1) {                                  // (more rows than are staged -- mostly empty ones: go on in memory)
            uint32_t lo = gr, hi = n_rows;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rs[mid] <= at) lo = mid; else hi = mid; }
            gr = lo;
            r0 = rs[gr]; n = rs[gr + 1] - r0;
        } else { r0 = s_rs[l]; n = s_rs[l + 1] - r0; }
        rec += (size_t)gr * of.keep_cap;
        me = rec[at - r0];
        tg = me.tgt_seg;
    }
    s_tg[tid] = tg;
    __syncthreads();
    if (!on) return;
    // place in the row = targets of the row below mine (the targets of a row are distinct).  The row's records inside the
    // block come from LDS, those of a row the block's ends cut from the scratch.
    const uint32_t j_lo = r0 < at0 ? at0 - r0 : 0u, j_hi = min(n, at0 + kKeepBlock - r0);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < j_lo; ++j) rank += rec[j].tgt_seg < tg ? 1u : 0u;
    {
        const uint32_t* __restrict__ q = s_tg + (r0 + j_lo - at0);   // the row's records j_lo .. j_hi - 1 (r0 + j_lo >= at0)
        const uint32_t m = j_hi > j_lo ? j_hi - j_lo : 0u;
        uint32_t k = 0;
        for (; k + 4 <= m; k += 4)
            rank += (q[k] < tg ? 1u : 0u) + (q[k + 1] < tg ? 1u : 0u) + (q[k + 2] < tg ? 1u : 0u) + (q[k + 3] < tg ? 1u : 0u);
        for (; k < m; ++k) rank += q[k] < tg ? 1u : 0u;
    }
    for (uint32_t j = j_hi; j < n; ++j) rank += rec[j].tgt_seg < tg ? 1u : 0u;
    const uint32_t row = gr - pairs[row_pair[gr]].row_off;
    const uint64_t to = (uint64_t)r0 + rank;
    slots[to] = me;
    const Slot& o = me;
    store_inverse_target(of, to, o);
    of.slot_row[to] = row;
}

hipError_t launch_keep_pair_info(const PairDesc* pairs, uint32_t n_pairs, const uint32_t* row_counts, const uint32_t* row_start,
                                 uint4* info, uint32_t* row_pair, uint32_t* blk_row, uint32_t n_blk, uint32_t* longest, hipStream_t stream) {
    if (!n_pairs) return hipSuccess;
    hipError_t e = hipMemsetAsync(longest, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_keep_pair_info, dim3(n_pairs), dim3(256), 0, stream, pairs, row_counts, row_start, info, row_pair, blk_row, n_blk, longest);
    return hipGetLastError();
}
hipError_t launch_keep_assemble(const ViewDev* views, const PairDesc* pairs, uint32_t n_rows, uint64_t slot_cap, const uint32_t* row_start,
                                const uint32_t* row_pair, const uint32_t* blk_row, const uint32_t* longest, Slot* slots, OrientFuse of,
                                hipStream_t stream) {
    if (!slot_cap) return hipSuccess;
    if (!of.keep_rec || !of.slot_row || !of.inv_tgt || !of.hyp_p || !of.hyp_q) return hipErrorInvalidValue;
    slot_cap = std::min<uint64_t>(slot_cap, 0xFFFFFFFFull);
    const uint32_t nb = (uint32_t)((slot_cap + kKeepBlock - 1) / kKeepBlock);
    hipLaunchKernelGGL(k_keep_assemble, dim3(nb), dim3(kKeepBlock), 0, stream, views, pairs, n_rows, (uint32_t)slot_cap, row_start, row_pair,
                       blk_row, longest, slots, of);
    return hipGetLastError();
}

// ---- per-view precompute (after translate), all views in one launch: grid = (segment blocks, views) -----
__global__ void k_prep_views(const ViewDev* __restrict__ views) {
    const ViewDev& v = views[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.M) return;
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = v.RtKinv[k];
    const d3 C{v.C[0], v.C[1], v.C[2]};
    const float4 s = v.seg4[i];
    const double ax = s.x, ay = s.y, bx = s.z, by = s.w;
    const d3 r1 = normalized(mul33(A, d3{ax, ay, 1.0}));
    const d3 r2 = normalized(mul33(A, d3{bx, by, 1.0}));
    const d3 n = normalized(cross(r1, r2));
    const d3 rm = normalized(mul33(A, d3{0.5 * (ax + bx), 0.5 * (ay + by), 1.0}));
    SegX o;
    o.r1[0] = r1.x; o.r1[1] = r1.y; o.r1[2] = r1.z;
    o.r2[0] = r2.x; o.r2[1] = r2.y; o.r2[2] = r2.z;
    o.n[0] = n.x; o.n[1] = n.y; o.n[2] = n.z;
    o.cn = dot(C, n);
    o.rm[0] = rm.x; o.rm[1] = rm.y; o.rm[2] = rm.z;
    const_cast<SegX*>(v.segx)[i] = o;
    if (v.segd32) {   // float copy of rays + plane normal: the depth DECISION of the match kernel (l3d_dev.h depths_positive32)
        SegD32 q;
        q.r1[0] = (float)r1.x; q.r1[1] = (float)r1.y; q.r1[2] = (float)r1.z;
        q.r2[0] = (float)r2.x; q.r2[1] = (float)r2.y; q.r2[2] = (float)r2.z;
        q.n[0] = (float)n.x; q.n[1] = (float)n.y; q.n[2] = (float)n.z;
        q.pad[0] = q.pad[1] = q.pad[2] = 0.0f;
        const_cast<SegD32*>(v.segd32)[i] = q;
    }
    SegF f;
    f.qx = (float)(ax - (double)v.cx); f.qy = (float)(ay - (double)v.cy);
    f.dx = (float)(ax - bx); f.dy = (float)(ay - by);
    const_cast<SegF*>(v.segf)[i] = f;
}

hipError_t launch_prep_views(const ViewDev* views, uint32_t n_views, uint32_t max_M, hipStream_t stream) {
    if (!n_views || !max_M) return hipSuccess;
    hipLaunchKernelGGL(k_prep_views, dim3((max_M + 255) / 256, n_views), dim3(256), 0, stream, views);
    return hipGetLastError();
}

}  // namespace l3d

#if defined(L3D_STATS) || defined(L3D_CYCLES)
extern "C" void l3d_debug_stats(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(l3d::g_stats), sizeof(l3d::g_stats));
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(l3d::g_stats), z, sizeof(z)); }
}
extern "C" void l3d_debug_cycles(unsigned long long* out, int n) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(l3d::g_cycles), (size_t)n * 64);
}
#endif


// ---- start-up (l3d_create): the runtime loads a translation unit's code object at the first launch of one of its
// kernels (~0.6 ms each, measured on the first matchImages of a process); an empty launch pays that at context creation
namespace l3d {
namespace { __global__ void k_warm_match() {} }
hipError_t warm_match(hipStream_t st) {
    hipLaunchKernelGGL(k_warm_match, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
}  // namespace l3d
