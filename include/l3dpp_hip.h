/* =============================================================================
 * l3dpp_hip.h -- C-ABI of the MI355X-native Line3D++ matching/scoring hot path.
 *
 * Shared library: line3dpp_amd/csrc/libl3dpp_hip.so (hand-written HIP, gfx950).
 * Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository manhofer/Line3Dpp).  Two layers are exported:
 *
 *   (1) context layer  -- replaces the slice of class L3DPP::Line3D that drives the hot
 *       path: addImage (explicit segments) / matchImages / the affinity part of
 *       reconstruct3Dlines, with accessors that hand back matches_, estimated_position3D_
 *       and A_ in the reference's own POD layouts (Match, Segment3D fields, CLEdge,
 *       SparseMatrix COO) so clustering / optimisation can run unchanged.
 *   (2) seam layer     -- replaces the accelerator seam the reference already has in
 *       cudawrapper.h:54-80 (match_lines_GPU / score_matches_GPU), for a maintainer who
 *       wants to keep Line3D's own driver loop and only swap the kernels.
 *
 * All functions return 0 on success or a negative l3d_status; l3d_last_error() gives text.
 * Semantics follow the reference CPU path (line3D.cc), not its CUDA path.
 * ===========================================================================*/
#ifndef L3DPP_HIP_H_
#define L3DPP_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- PODs with the reference's exact layout -------------------------------------- */

/* L3DPP::Match, commons.h:186-203 (40 bytes) */
typedef struct l3d_match {
    uint32_t src_camID_, src_segID_, tgt_camID_, tgt_segID_;
    float overlap_score_, score3D_;
    float depth_p1_, depth_p2_, depth_q1_, depth_q2_;
} l3d_match;

/* L3DPP::Segment2D, commons.h:104-131 */
typedef struct l3d_segment2d { uint32_t camID_, segID_; } l3d_segment2d;

/* the data members of L3DPP::Segment3D, segment3D.h:99-115 (P1,P2,dir double; length float; valid) */
typedef struct l3d_segment3d {
    double P1[3], P2[3], dir[3];
    float length_;
    uint32_t valid_;
} l3d_segment3d;

/* L3DPP::CLEdge, clustering.h:47-51 */
typedef struct l3d_cledge { int32_t i_, j_; float w_; } l3d_cledge;

/* float4 entry of L3DPP::SparseMatrix (sparsematrix.cc:40: (i, j, w, 0)) */
typedef struct l3d_float4 { float x, y, z, w; } l3d_float4;

/* one phase-A result slot: what a (src segment, rank) of a directed view pair holds.
 * 32 bytes; this is also the record the multi-GPU all-gather exchanges. */
typedef struct l3d_slot {
    uint32_t tgt_seg;   /* 0xFFFFFFFF = empty */
    float overlap;
    float depth_p1, depth_p2, depth_q1, depth_q2;
    float score3D;      /* written when the src view is processed (scoringCPU) */
    uint32_t flags;     /* bit0: survived the src view's orientation filter (checkMatchOrientation); bit1: so did
                         * its inverse copy in the tgt view.  Set by phase A itself when kNN > 0. */
} l3d_slot;

typedef enum l3d_status {
    L3D_OK = 0,
    L3D_ERR_ARG = -1,          /* bad argument */
    L3D_ERR_IMAGE_SMALL = -2,  /* line3D.cc:119 */
    L3D_ERR_ID_IN_USE = -3,    /* line3D.cc:130 */
    L3D_ERR_NO_NEIGHBORS = -4, /* line3D.cc:154 */
    L3D_ERR_NO_SEGMENTS = -5,  /* line3D.cc:188 */
    L3D_ERR_NO_VIEWS = -6,     /* line3D.cc:385 */
    L3D_ERR_STATE = -7,        /* call order */
    L3D_ERR_HIP = -8,          /* HIP runtime error */
    L3D_ERR_LIMIT = -9,        /* size limit of this build */
    L3D_ERR_RETRY = -10        /* l3d_match_finish after l3d_lists_shard: the record pools were enlarged, repeat
                                * l3d_lists_shard + the exchange of its slabs + l3d_match_finish (every rank gets it) */
} l3d_status;

typedef struct l3d_ctx l3d_ctx;

/* Line3D::matchImages arguments, line3D.h:143-148 (defaults commons.h:51-55) */
typedef struct l3d_match_params {
    float sigma_position;             /* 2.5  */
    float sigma_angle;                /* 10.0 */
    uint32_t num_neighbors;           /* 10   */
    float epipolar_overlap;           /* 0.25 */
    int32_t kNN;                      /* 10   */
    float const_regularization_depth; /* -1   */
} l3d_match_params;

const char* l3d_last_error(void);
/* build info: "gfx950;..." */
const char* l3d_build_info(void);
/* Device and pinned blocks that contexts release are kept in a process-wide cache for the next context (hipFree /
 * hipHostMalloc cost milliseconds per scene otherwise): at most a quarter of the device's memory (L3D_CACHE_MAX_MB) and
 * 1 GiB pinned; emptied automatically when an allocation fails.  l3d_trim_cache() hands everything cached back to the
 * runtime now (returns the bytes freed) -- for a process that shares the device with other allocators. */
uint64_t l3d_trim_cache(void);

/* ---- (1) context layer ------------------------------------------------------------- */

/* Line3D::Line3D (line3D.cc:6-69) with use_GPU=true (neighbours: per view, l3d_add_view / l3d_add_view_worldpoints).
 * `device` = HIP device ordinal.  `stream` = hipStream_t to launch on (0 = default stream). */
l3d_ctx* l3d_create(int device, void* stream);
void l3d_destroy(l3d_ctx*);

/* Line3D::addImage (line3D.cc:112-227) with explicit `line_segments` and explicit neighbour
 * list; the image contributes only width/height.  segs4 = M x (x1,y1,x2,y2) pixels. */
int l3d_add_view(l3d_ctx*, uint32_t camID, const float* segs4, uint32_t M, const double K[9],
                 const double R[9], const double t[3], uint32_t width, uint32_t height,
                 float median_depth, const uint32_t* neighbors, uint32_t n_neighbors);

/* The same with a WORLDPOINT list instead of the neighbour list: Line3D::addImage on an instance constructed with
 * neighbors_by_worldpoints=true (line3D.cc:216-219, processWPlist :230-240).  The view's visual neighbours are then
 * found from the worldpoint overlap inside every matchImages / l3d_match_begin: Line3D::findVisualNeighborsFromWPs
 * (line3D.cc:578-699; host code, line3dpp_amd/csrc/l3d_neighbors.hip). */
int l3d_add_view_worldpoints(l3d_ctx*, uint32_t camID, const float* segs4, uint32_t M, const double K[9],
                             const double R[9], const double t[3], uint32_t width, uint32_t height,
                             float median_depth, const uint32_t* worldpoints, uint32_t n_worldpoints);
/* visual_neighbors_[camID] (line3D.h:352) as the last matchImages / l3d_match_begin left it, ascending; *n = its size
 * (out may be NULL or shorter: the first `cap` are written) */
int l3d_get_visual_neighbors(l3d_ctx*, uint32_t camID, uint32_t* out, uint32_t cap, uint32_t* n);
/* Line3D::findVisualNeighborsFromWPs for a set of cameras without a context (and without a GPU): cameras as handed to
 * addImage (row-major K, R, t per view), worldpoint lists in CSR form (wp_offsets[n_views + 1]); the neighbour sets come
 * back in CSR form (nb_offsets[n_views + 1]; neighbors may be NULL to ask for the sizes).  The cameras are moved by
 * the median of their centres for the computation, as matchImages does (line3D.cc:436, 500-536). */
int l3d_neighbors_from_worldpoints(uint32_t n_views, const uint32_t* cam_ids, const double* K9, const double* R9,
                                   const double* t3, const uint64_t* wp_offsets, const uint32_t* worldpoints,
                                   uint32_t num_neighbors, uint64_t* nb_offsets, uint32_t* neighbors, uint64_t cap);

/* ---- input side (host code, no GPU) ---------------------------------------------------------------------------------
 * VisualSfM .nvm exactly as main_vsfm.cpp:144-250 reads it, with what it derives per camera (:188-206, :300-303).  The
 * camera index is the camID main_vsfm.cpp passes to addImage; a camera that sees no point has n_worldpoints == 0 and
 * is skipped there (:258). */
typedef struct l3d_nvm l3d_nvm;
typedef struct l3d_nvm_camera {
    const char* filename;             /* valid while the handle lives */
    float focal, distortion;          /* (float vectors in main_vsfm.cpp:164-169) */
    float median_depth;               /* sorted point distances [n/2], the value handed to addImage */
    uint32_t n_worldpoints;
    double R[9], t[3], C[3];          /* row-major R from the quaternion, t = -R C */
} l3d_nvm_camera;
int l3d_nvm_open(const char* path, l3d_nvm** out);    /* L3D_ERR_NO_VIEWS: "No aligned cameras in NVM file!" (:157-161) */
uint32_t l3d_nvm_num_cameras(const l3d_nvm*);
int l3d_nvm_get_camera(const l3d_nvm*, uint32_t index, l3d_nvm_camera* out);
int l3d_nvm_get_worldpoints(const l3d_nvm*, uint32_t index, uint32_t* out, uint32_t cap);   /* = addImage's wps list */
void l3d_nvm_close(l3d_nvm*);
void l3d_nvm_intrinsics(float focal, uint32_t width, uint32_t height, double K[9]);          /* main_vsfm.cpp:272-282 */
/* COLMAP text results (cameras.txt / images.txt / points3D.txt of a folder, main_colmap.cpp:136-348) and bundler files
 * (bundle.rd.out, main_bundler.cpp:147-252) with what those front ends derive per image: K (COLMAP; bundler builds it
 * from the image size: l3d_nvm_intrinsics), R, t, C = R^T (-t), the distortion coefficients handed to undistortImage,
 * the worldpoint list and the median worldpoint distance handed to addImage.  Images in file order; `id` is the camID
 * the front end passes to addImage (COLMAP: IMAGE_ID, bundler: the camera index); n_worldpoints == 0: skipped there. */
typedef struct l3d_sfm l3d_sfm;
typedef struct l3d_sfm_image {
    uint32_t id, camera;              /* camID for addImage; COLMAP CAMERA_ID (bundler: = id) */
    uint32_t width, height;           /* COLMAP: from cameras.txt (the reference takes the size from the image file); bundler: 0 */
    const char* name;                 /* COLMAP image name (valid while the handle lives); bundler: "" */
    float focal;                      /* bundler focal length (float there); COLMAP: (float)fx */
    float median_depth;
    uint32_t n_worldpoints;
    double K[9];                      /* COLMAP; bundler: zeros */
    double R[9], t[3], C[3];
    double radial[3], tangential[2];  /* k1 k2 k3, p1 p2 */
} l3d_sfm_image;
int l3d_sfm_open_colmap(const char* folder, l3d_sfm** out);
int l3d_sfm_open_bundler(const char* bundle_file, l3d_sfm** out);   /* L3D_ERR_NO_VIEWS: "No cameras and/or points in bundle file!" */
uint32_t l3d_sfm_num_images(const l3d_sfm*);
int l3d_sfm_get_image(const l3d_sfm*, uint32_t index, l3d_sfm_image* out);
int l3d_sfm_get_worldpoints(const l3d_sfm*, uint32_t index, uint32_t* out, uint32_t cap);
void l3d_sfm_close(l3d_sfm*);
/* The segment cache Line3D::detectLineSegments loads / stores per image when load_segments is set (line3D.cc:295-309,
 * 362-366): "<data folder>/segments_L3D++_<camID>_<width>x<height>_<max segments>.bin", the boost binary archive of a
 * one-row L3DPP::DataArray<float4> (dataArray.h:352-374).  segs4 = n x (x1, y1, x2, y2). */
int l3d_segment_cache_name(uint32_t camID, uint32_t width, uint32_t height, uint32_t max_segments, char* out, uint32_t cap);
int l3d_read_segment_cache(const char* path, float* segs4 /* may be NULL */, uint32_t cap, uint32_t* n);
int l3d_write_segment_cache(const char* path, const float* segs4, uint32_t n);

/* Line3D::matchImages (line3D.cc:375-497): the whole call on this context's GPU. */
int l3d_match_images(l3d_ctx*, const l3d_match_params*);

/* matchImages split in three for pair-sharded multi-GPU runs (one process per GPU):
 *   l3d_match_begin   param clamps, translate(), per-view k, neighbour sets, directed pair
 *                     list (line3D.cc:394-485, 704-741), upload + per-view precompute
 *   l3d_match_pairs   matchingCPU (line3D.cc:900-1015) for the pairs [first, first+count) of
 *                     the pair list -> their slots in the slot buffer
 *   (exchange: all-gather of the slot buffer, done by the caller, e.g. RCCL through
 *    torch.distributed on the device pointer l3d_slot_buffer() returns)
 *   l3d_match_finish  per view, ascending camID: checkMatchOrientation, scoringCPU,
 *                     storeInverseMatches, filterMatches (line3D.cc:745-773); untranslate() */
int l3d_match_begin(l3d_ctx*, const l3d_match_params*);
int l3d_num_pairs(l3d_ctx*, uint32_t* n);
int l3d_get_pairs(l3d_ctx*, uint32_t* src_cam, uint32_t* tgt_cam, uint64_t* slot_offset /* in slots */);
int l3d_match_pairs(l3d_ctx*, uint32_t first, uint32_t count);
int l3d_slot_buffer(l3d_ctx*, void** dev_ptr, uint64_t* n_slots);   /* (halo form of a multi-GPU call: only the regions of
                                                                       * the pairs this rank matched or received hold slots) */
/* (kNN <= 0, keep every match -- line3D.cc:982-992: the device buffer is RAGGED, a row holds exactly its matches in ascending
 *  target order and n_slots is their number; l3d_get_pairs' slot_offset is the pair's first slot; l3d_get_pair_slots below
 *  hands a pair out in the padded Ms x K form, K = its longest row.  The mode runs unsharded.) */
/* tell the context that the caller's exchange has filled in the slots of all other pairs */
int l3d_slots_exchanged(l3d_ctx*);
/* Compact form of the same exchange (kNN > 0): only the target index of every slot travels (4 B instead of 32);
 * overlap and depths are functions of (pair, source row, target index) and are re-derived on the receiving rank by
 * the match kernel's own device functions, bit-identically.
 *   l3d_pack_slot_indices    slots of the rank's own pairs [first, first+count) -> index buffer (returns when done)
 *   (all-gather of the u32 index buffer l3d_slot_index_buffer() returns, by the caller)
 *   l3d_expand_slot_indices  index buffer -> slots of the foreign pairs [first, first+count); marks them matched
 * Replaces the host-side gather of per-view match lists a multi-GPU run of the reference would need between
 * matchingCPU and the per-view chain (line3D.cc:728-773). */
int l3d_slot_index_buffer(l3d_ctx*, void** dev_ptr, uint64_t* n_slots);
int l3d_pack_slot_indices(l3d_ctx*, uint32_t first, uint32_t count);
int l3d_expand_slot_indices(l3d_ctx*, uint32_t first, uint32_t count);
int l3d_match_finish(l3d_ctx*);
/* Phase B sharded over ranks as well (optional step between the slot exchange and l3d_match_finish; every pair must be
 * present): the dense part of phase B -- one pass over every 2D segment's hypothesis list that finds the supporting
 * pairs of similarityForScoring (line3D.cc:1208-1294, 1417-1446) -- is run for the views of rank `rank` of `world`
 * only (contiguous view ranges of equal segment count).  Its output records live in pools; rank r fills the pools
 * [r * 256/world, (r+1) * 256/world) of four arrays (edges, headers, segment headers, pool counters).  On return
 * slab_ptr[k] / slab_bytes[k] describe THIS rank's slab of array k and full_ptr[k] the array itself (device
 * pointers; rank r's slab starts at full_ptr[k] + r * slab_bytes[k]): the caller all-gathers the four arrays slab by
 * slab (e.g. ncclAllGather) and then calls l3d_match_finish, which runs the cheap sparse remainder of phase B (the
 * chain of inverse matches, scores, filterMatches, outputs) on the complete records on every rank.  Returns when the
 * slabs are complete in device memory. */
int l3d_lists_shard(l3d_ctx*, uint32_t rank, uint32_t world, void* slab_ptr[4], uint64_t slab_bytes[4], void* full_ptr[4]);
/* The same for an explicit range of views [view0, view1) (indices in ascending camID order) -- the halo form of a
 * multi-GPU run (line3dpp_amd/dist.py, tests/cpp/rccl_driver.cpp): rank r owns the views l3d_plan_shards gives it,
 * matches the pairs whose SOURCE view it owns, receives the pairs whose TARGET view it owns from their owners (compact
 * index form, l3d_pack_slot_indices / l3d_expand_slot_indices per run of consecutive pairs) and runs the list pass of its
 * views: only the pairs that touch [view0, view1) have to be present.  The records are all-gathered as above; the
 * replicated remainder of phase B works on the records alone. */
int l3d_lists_shard_views(l3d_ctx*, uint32_t rank, uint32_t world, uint32_t view0, uint32_t view1, void* slab_ptr[4],
                          uint64_t slab_bytes[4], void* full_ptr[4]);
/* The tail of phase B sharded by views as well (optional, instead of l3d_match_finish after l3d_lists_shard_views and the
 * exchange of its slabs).  The chain of storeInverseMatches (line3D.cc:1672-1699) is a global fixed point over the
 * records of all ranks and every rank runs it; scoringCPU's scores, filterMatches (:1586-1669), the surviving matches_
 * lists, estimated_position3D_ and the view medians are per view and are computed by the rank that owns the view:
 *   l3d_tail_shard_count   chain + scores + filterMatches + counts of this rank's views: counts[0] = its surviving
 *                          matches, counts[1] = its best hypotheses (L3D_ERR_RETRY as l3d_match_finish gives it)
 *   -- the caller all-gathers the two counts of every rank --
 *   l3d_tail_shard_layout  this rank's outputs, written at their places in the full arrays, and where every rank's parts
 *                          are: nine arrays (Match, target / source segment of a match, HypRec, depth pairs, the three
 *                          per-segment arrays, the view medians): base_ptr[k] device pointer, elt_bytes[k] element size,
 *                          first[9 r + k] / count[9 r + k] the elements of rank r.  view_bounds[world + 1]: the view
 *                          ranges of the ranks (those of the list pass)
 *   -- the caller exchanges the parts: rank r's part of array k to every other rank, in place --
 *   l3d_tail_shard_commit  closes the call as l3d_match_finish does (medians to the host, totals, views untranslated) */
int l3d_tail_shard_count(l3d_ctx*, uint32_t counts[2]);
/* Options of the sharded entries of this context (round 6; they hold until changed):
 *   first_needed_rank   the chain of a rank only has to cover the records its views DEPEND on: view v depends on view u < v when a
 *                       pair (u -> v) hands inverse matches over (line3D.cc:1680), transitively.  With contiguous view ranges the
 *                       ranks a rank depends on lie below it; the caller (which knows the pair list: l3d_plan_shards) passes the
 *                       lowest one, and may then leave the record slabs of ranks outside [first_needed_rank, rank] unexchanged --
 *                       the COUNTER slab (array 3 of l3d_lists_shard*) must still reach every rank, it carries the pool
 *                       overflow flags all ranks decide on alike.  0 (default): all ranks below (and the records of all ranks
 *                       must be present, as before).
 *   exchanges_stream_ordered   nonzero: the caller's exchanges are ordered behind the context's stream by themselves (RCCL
 *                       on the same stream): l3d_lists_shard*, l3d_tail_shard_layout and l3d_affinity_shard_begin then return
 *                       WITHOUT waiting for the device (three host synchronisations less per call).  0 (default): they
 *                       return when their slabs / parts are complete in device memory. */
int l3d_shard_options(l3d_ctx*, uint32_t first_needed_rank, int exchanges_stream_ordered);
int l3d_tail_shard_layout(l3d_ctx*, uint32_t world, const uint32_t* counts_all, const uint32_t* view_bounds, void* base_ptr[9],
                          uint64_t elt_bytes[9], uint64_t* first, uint64_t* count);
int l3d_tail_shard_commit(l3d_ctx*);
/* The affinity fill of Line3D::reconstruct3Dlines (computingAffinityMatrix, line3D.cc:1852-1979) sharded by the same
 * views, for a call that was closed by l3d_tail_shard_commit (the similarity of a candidate depends on its
 * two hypotheses alone).  Instead of l3d_compute_affinity:
 *   l3d_affinity_shard_begin   similarity (line3D.cc:1467-1553) of the surviving matches of this rank's views, written at
 *                              their places in the full float array *simv; first[r] / count[r]: the elements of rank r
 *   -- the caller exchanges the parts: rank r's part to every other rank, in place --
 *   l3d_affinity_shard_finish  used_ / local ids / CLEdge pairs from all similarities (every rank: same A_ everywhere)
 * Not with collinearity_t > 0 (line3D.cc:1904-1974 is sequential by definition): L3D_ERR_LIMIT, use l3d_compute_affinity. */
int l3d_affinity_shard_begin(l3d_ctx*, uint32_t rank, uint32_t world, void** simv, uint64_t* first, uint64_t* count);
int l3d_affinity_shard_finish(l3d_ctx*);
/* closes an open sharded fill without the bookkeeping pass (a peer failed, its similarities never arrived: the caller goes
 * on to l3d_compute_affinity on every rank): views untranslated, nothing else touched.  L3D_OK also when none is open. */
int l3d_affinity_shard_abort(l3d_ctx*);
/* Partition of a call over `world` ranks (host only; a function of the pair list of l3d_get_pairs): contiguous view
 * ranges whose outgoing pairs carry equal shares of the cost (pair_cost[p], e.g. Ms * Mt); pair_src_view[p] = index of
 * the pair's source view (the list is ordered by it).  view_bounds / pair_bounds receive world + 1 entries: rank r owns
 * the views [view_bounds[r], view_bounds[r+1]) and the pairs [pair_bounds[r], pair_bounds[r+1]). */
int l3d_plan_shards(uint32_t n_views, uint32_t n_pairs, const uint32_t* pair_src_view, const uint64_t* pair_cost,
                    uint32_t world, uint32_t* view_bounds, uint32_t* pair_bounds);
/* Leaves an open l3d_match_begin without results: everything queued is drained, the views are moved back
 * (matchImages translates them for its duration, line3D.cc:436/493), the context is idle again.  A no-op when no
 * begin is open.  Every failing l3d_match_begin / l3d_match_images / l3d_match_finish does this itself; callers that
 * give up between the split calls (e.g. a failed exchange in a multi-GPU run) call it explicitly. */
int l3d_match_abort(l3d_ctx*);

/* The affinity part of Line3D::reconstruct3Dlines: translate(), med_scene_depth_lines_,
 * computingAffinityMatrix(), untranslate() (line3D.cc:1749-1778, 1852-2023; collinearity off). */
int l3d_compute_affinity(l3d_ctx*);

/* Line3D::reconstruct3Dlines (line3D.cc:1702-1824) up to the final 3D segments: translate(), affinity matrix
 * (as l3d_compute_affinity), graph clustering (clustering.cc), 3D line per cluster, collinear 3D segments,
 * filterTinySegments, untranslate().  perform_diffusion != 0 runs the replicator-dynamics diffusion of A_
 * (performRDD, line3D.cc:2026-2076) on the device-resident matrix; collinearity_t > 0 adds the per-image collinearity
 * tests (View::findCollinearSegments, view.cc:150-258) and the collinear affinity links (line3D.cc:1904-1974).
 * use_CERES is reported and ignored like in a reference build without Ceres (line3D.cc:1741-1743).
 * The clustering / reconstruction tail is small sequential host work in the reference and runs on the host
 * here as well. */
int l3d_reconstruct_3d_lines(l3d_ctx*, uint32_t visibility_t, int perform_diffusion, float collinearity_t,
                             int use_CERES, uint32_t max_iter_CERES);
/* Line3D::get3Dlines (line3D.cc:2455-2463): lines3D_ flattened.  Line i owns the 3D segments
 * [seg_offsets[i], seg_offsets[i+1]) (collinear3Dsegments_) and the residual 2D segments
 * [res_offsets[i], res_offsets[i+1]) (underlyingCluster_.residuals_); cluster_lines[i] is
 * underlyingCluster_.seg3D_, reference_views[i] its reference_view_.  Original (untranslated) frame. */
int l3d_num_3d_lines(l3d_ctx*, uint32_t* n_lines, uint32_t* n_segments, uint32_t* n_residuals);
int l3d_get_3d_lines(l3d_ctx*, uint32_t* seg_offsets, l3d_segment3d* segments, uint32_t* res_offsets,
                     l3d_segment2d* residuals, l3d_segment3d* cluster_lines, uint32_t* reference_views);

/* block the calling thread until everything queued on the context's stream has finished */
int l3d_synchronize(l3d_ctx*);

/* ---- accessors (host copies, reference layouts) ------------------------------------ */

/* number of directed pair tests matchImages performed: sum Ms*Mt */
int l3d_pair_tests(l3d_ctx*, uint64_t* n);
/* matches_[camID] after matchImages: CSR over the view's segments.
 * seg_offsets has M+1 entries.  Call with out=NULL to get the count. */
int l3d_get_matches(l3d_ctx*, uint32_t camID, l3d_match* out, uint64_t cap, uint32_t* seg_offsets,
                    uint64_t* n);
/* fresh phase-A slots of one directed pair: Ms x K slots (K = kNN; kNN <= 0: K = the pair's longest row, shorter rows
 * padded with empty slots) */
int l3d_get_pair_slots(l3d_ctx*, uint32_t pair_index, l3d_slot* out, uint64_t cap, uint32_t* Ms,
                       uint32_t* K);
/* estimated_position3D_ (+ entry_map_ keys), ordered by (camID, segID). Coordinates are in the
 * translated frame matchImages works in (the reference never untranslates them). */
int l3d_num_best(l3d_ctx*, uint32_t* n);
int l3d_get_best(l3d_ctx*, l3d_segment2d* seg2d, l3d_segment3d* seg3d, l3d_match* best);
/* per view: View::k(), View::median_depth() after matchImages */
int l3d_view_info(l3d_ctx*, uint32_t camID, float* k, float* median_depth);
int l3d_translation(l3d_ctx*, double t[3]);
/* A_ (CLEdge list, both (i,j) and (j,i)), local2global_ and med_scene_depth_lines_ */
int l3d_num_affinity(l3d_ctx*, uint32_t* n_edges, uint32_t* n_rows);
int l3d_get_affinity(l3d_ctx*, l3d_cledge* edges, l3d_segment2d* local2global, float* med_scene_depth_lines);
/* L3DPP::SparseMatrix(A_, n_rows, 1.0, sort_by_row) (sparsematrix.cc:8-60): entries float4(i,j,w,0)
 * sorted by row or column, start_indices[n_rows] with -1 for empty rows/columns */
int l3d_get_sparse_matrix(l3d_ctx*, int sort_by_row, l3d_float4* entries, int32_t* start_indices);

/* test hook (host only): principal direction of a row-major symmetric 3x3 scatter matrix as Line3D::get3DlineFromCluster
 * (line3D.cc:2196-2211) takes it from JacobiSVD -- this library's closed-form solver, checked against LAPACK on the CPU */
int l3d_principal_direction(const double S9[9], double dir3[3]);

/* test hook (device): the unscaled IEEE division / square root of the exact tests (l3d_dev.h: rcp_refined, div_by,
 * sqrt_unscaled) against the compiler's own expansions on n random operand sets; counts[3] = results whose bits differ
 * (single divisions, paired divisions, square roots) -- all zero on a correct build */
int l3d_selftest_arith(int device, uint64_t n, uint64_t seed, uint64_t counts[3]);

/* test hook: route every segment pair through the exact double-precision test (no fp32 pre-filter);
 * used by the tests to prove that the pre-filter never loses a match */
int l3d_set_brute_force(l3d_ctx*, int on);

/* timing of the last calls, milliseconds of GPU time from HIP events on the context's stream */
typedef struct l3d_timings {
    float begin_ms;        /* upload + per-view precompute kernels */
    float match_pairs_ms;  /* phase A: pair-matching kernel(s) */
    float finish_ms;       /* phase B: per-view chain */
    float affinity_ms;
    uint32_t match_kernel_launches;
    float match_kernel_ms; /* the pair-matching kernel alone */
    float cull_prepare_ms; /* ordering of rows/targets by epipolar band (part of match_pairs_ms) */
    uint32_t culled_pairs; /* directed pairs matched with epipolar-band culling in the last matchImages */
    uint32_t list_entries; /* phase B: total length of the per-segment hypothesis lists (fresh + inverse) */
    uint32_t support_words;/* phase B: supporting (hypothesis, supporter) pairs = edges of the sparse form */
    uint32_t tied_rows;    /* phase A: source rows with equal overlaps, replayed in the reference's priority_queue order
                            * (cumulative since l3d_create) */
    uint32_t chain_sweeps; /* phase B: sweeps of the chain fixed point that still changed something (last round) */
    uint32_t chain_extra_rounds; /* phase B: extra rounds of chain sweeps beyond the ones enqueued blindly (0 normally) */
    uint32_t pool_retries; /* phase B: list passes of the last matchImages that outgrew their record pools and were repeated
                            * with larger ones (0 once the pools have the scene's size; a first call may need one) */
    float lists_ms;        /* phase B: the list pass (inverse records, candidate pairs, edges) -- the part of finish_ms that a
                            * multi-GPU run shards by views; the rest of finish_ms is the tail every rank runs */
    uint32_t record_kbytes;/* phase B: size of the four record arrays at their current pool strides, KiB: what the ranks of a
                            * multi-GPU run all-gather */
    /* round 6: what the roofline of phase B is computed from (bench.py: roofline_phase_b) */
    uint32_t list_inverse; /* phase B: of list_entries, the inverse hypotheses (storeInverseMatches, line3D.cc:1672-1699) */
    uint32_t list_candidates; /* phase B: candidate pairs the list pass handed to the exact test (similarityForScoring) */
    uint32_t list_headers; /* phase B: hypotheses with at least one supporter (headers of the sparse form) */
    uint32_t slots_lo, slots_hi; /* phase A: slots of the call = sum over directed pairs of Ms x kNN (64 bits) */
} l3d_timings;
int l3d_get_timings(l3d_ctx*, l3d_timings*);
/* How many of the context's HIP events a call records (they feed l3d_timings; the reference has no such thing: its
 * timing is the wall clock of main_*.cpp).  1 (DEFAULT since round 5: the fast setting is what a drop-in user gets): only
 * the pair around the pair-matching kernel (match_kernel_ms; the other times read 0); 2 (profiling, opt-in): all ten --
 * every field above is filled; 0: none.  An event between two kernels costs a ~6 us bubble on the stream, which a
 * 1.4 ms call notices: bench.py times the library as shipped (level 1) and takes the phase breakdown from separate
 * untimed steps at level 2. */
int l3d_set_timing_level(l3d_ctx*, int level);
/* Test hook (no reference counterpart): process-wide counters that tell a test which form of a kernel ran.
 * "csr_global_launches": launches of the global-cursor form of k_pair_csr (views beyond 32 768 segments, or
 * L3D_CSR_GLOBAL=1); "knn_replay_calls": l3d_match_begin calls whose kNN exceeded the LDS tables of the match kernel
 * (every row then takes the exact replay path).  Unknown name: ~0. */
unsigned long long l3d_debug_counter(const char* name);

/* ---- (2) seam layer ------------------------------------------------------------------ */

/* Replaces match_lines_GPU (cudawrapper.h:54-63; caller Line3D::matchingGPU line3D.cc:1040-1074)
 * with CPU-path semantics (Line3D::matchingCPU line3D.cc:900-1015).  Host pointers in and out.
 * F, RtKinv_* are row-major 3x3 doubles, C_* camera centres (already translated, line3D.cc:436).
 * kNN must be > 0 here.  out_slots: Ms x kNN slots, every row in the order Line3D::matchingCPU appends it to
 * matches_[src][r] (descending overlap; equal overlaps in the pop order of its std::priority_queue).
 * Returns the number of matches in *num_matches. */
int l3d_match_lines(int device, const float* lines_src4, uint32_t Ms, const float* lines_tgt4, uint32_t Mt,
                    const double F[9], const double RtKinv_src[9], const double RtKinv_tgt[9],
                    const double C_src[3], const double C_tgt[3], uint32_t width, uint32_t height,
                    float epi_overlap, int32_t kNN, l3d_slot* out_slots, uint64_t* num_matches);

/* Line3D::createOutputFilename (line3D.cc:2853-2893): "Line3D++__W_FULL__N_10__sigmaP_2.5__..._vis_3" from the
 * parameters of the last matchImages / reconstruct3Dlines.  max_image_width <= 0 -> "W_FULL". */
int l3d_output_filename(l3d_ctx*, int max_image_width, char* buf, uint32_t cap);
/* Line3D::save3DLinesAsTXT (line3D.cc:2631-2688): <output_folder>/<output filename>.txt, one line per 3D line:
 * n_segments (P1 P2)* n_residuals (camID segID x1 y1 x2 y2)*, the format of the .txt files under testdata/Line3D++_ref. */
int l3d_save_3d_lines_txt(l3d_ctx*, const char* output_folder, int max_image_width);
/* Line3D::save3DLinesAsBIN (line3D.cc:2690-2711): <output filename>.bin = boost::archive::binary_oarchive of
 * std::vector<FinalLine3D> (serialization.h:38-45, segment3D.h:99-178), written without Boost in the byte layout of
 * the reference's own fixtures testdata/Line3D++_ref/\*vis_3.bin (doubles: the only result file that keeps full precision). */
int l3d_save_3d_lines_bin(l3d_ctx*, const char* output_folder, int max_image_width);
/* Line3D::getSegmentCoords2D (line3D.cc:2757-2772): (x1,y1,x2,y2) of a 2D segment, zeros if unknown */
int l3d_get_segment_coords2d(l3d_ctx*, uint32_t camID, uint32_t segID, float coords[4]);
/* Line3D::saveResultAsSTL (line3D.cc:2465-2531) and saveResultAsOBJ (:2579-2628): <output filename>.stl / .obj */
int l3d_save_result_stl(l3d_ctx*, const char* output_folder, int max_image_width);
int l3d_save_result_obj(l3d_ctx*, const char* output_folder, int max_image_width);

/* Replaces score_matches_GPU (cudawrapper.h:70-73; caller Line3D::scoringGPU, line3D.cc:1297-1414) with the
 * semantics of Line3D::scoringCPU (line3D.cc:1208-1294): score3D of every match of one view.  Host pointers,
 * arrays as scoringGPU marshals them: matches4[n] = (src segment, target camera, depth_p1, depth_p2), grouped per
 * segment and by target camera inside a segment (sortMatches); ranges2[M] = (first, last) inclusive, (-1,-1) if
 * none; reg_tgt2[n] = View::regularizerFrom3Dpoint of the two 3D end points in the target view; k = View::k(). */
int l3d_score_matches(int device, const float* lines4, uint32_t M, const float* matches4, const int32_t* ranges2,
                      const float* reg_tgt2, uint32_t n, const double RtKinv[9], const double C[3], float two_sigA_sqr,
                      float k, float* scores);

/* Replaces the body of View::findCollinGPU (view.cc:173-209; find_collinear_segments_GPU, cudawrapper.h:66-68) with
 * the semantics of View::findCollinCPU (view.cc:213-258).  Host pointers.  CSR output: offsets[M+1] and, if
 * cap >= *n, idx[*n] (ascending lists; call with idx = NULL first to learn *n). */
int l3d_find_collinear_segments(int device, const float* lines4, uint32_t M, float dist_t, uint32_t* offsets,
                                uint32_t* idx, uint64_t cap, uint64_t* n);

/* Replaces the body of Line3D::performRDD (line3D.cc:2026-2076): SparseMatrix(A_, n) +
 * replicator_dynamics_diffusion_GPU (cudawrapper.h:74-75, cudawrapper.cu:708-766: row normalisation + 10
 * diffusion steps P' = P o (P W)^T with the reference's lockstep row/column walk) + the min(w12, w21)
 * symmetrisation.  Host pointers in and out; `edges` in any order with ids < n_rows; `out` receives n_edges
 * CLEdges in (i, j) ascending order (the order performRDD rebuilds A_ in).  iterations = L3D_DEF_RDD_MAX_ITER (10)
 * in the reference.  l3d_reconstruct_3d_lines(perform_diffusion != 0) runs the same kernels on the context's
 * device-resident affinity matrix. */
int l3d_diffuse_affinity(int device, const l3d_cledge* edges, uint32_t n_edges, uint32_t n_rows, uint32_t iterations,
                         l3d_cledge* out);

#ifdef __cplusplus
}
#endif
#endif /* L3DPP_HIP_H_ */
