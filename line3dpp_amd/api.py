"""Host-side mirror of the reference's public interface for the hot path.

`Line3D` follows class L3DPP::Line3D (line3D.h:61-424): `addImage`, `matchImages`,
and the affinity part of `reconstruct3Dlines`, with the reference's argument names, defaults
(commons.h:40-70) and error behaviour: errors are printed with the `[L3D++] ERROR:` prefix and the
call returns without raising (line3D.cc:119-126, 385-391); `last_status` holds the l3d_status
code.  All compute runs in libl3dpp_hip.so (HIP, gfx950) through the C-ABI in
include/l3dpp_hip.h.  The reference is C++; this Python front-end exists for the tests, the
bench and torch.distributed plumbing -- the C++ facade over the same C-ABI is
include/line3dpp/line3D.h.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (CLEDGE_DTYPE, EMPTY, FLOAT4_DTYPE, MATCH_DTYPE, SEGMENT2D_DTYPE, SEGMENT3D_DTYPE, SLOT_DTYPE,
                   MatchParams, Timings, ptr)

# commons.h:40-70
L3D_DEF_MATCHING_NEIGHBORS = 10
L3D_DEF_EPIPOLAR_OVERLAP = 0.25
L3D_DEF_KNN = 10
L3D_DEF_SCORING_POS_REGULARIZER = 2.5
L3D_DEF_SCORING_ANG_REGULARIZER = 10.0
L3D_DEF_MIN_VISIBILITY_T = 3


class Line3D:
    PREFIX = "[L3D++] "

    def __init__(self, output_folder="", load_segments=False, max_img_width=-1, max_line_segments=3000,
                 neighbors_by_worldpoints=False, use_GPU=True, device=0, stream=0, verbose=False):
        # neighbors_by_worldpoints (line3D.cc:6-69): addImage's list is a WORLDPOINT list and the visual neighbours are found
        # from the worldpoint overlap at every matchImages (Line3D::findVisualNeighborsFromWPs, line3D.cc:578-699)
        self.neighbors_by_worldpoints = bool(neighbors_by_worldpoints)
        self.L = _lib.load()
        self.verbose = verbose
        self.last_status = 0
        self._M = {}
        self.stream = int(stream)          # the hipStream_t every launch of this context goes to (dist.py: ordering of collectives)
        self.h = self.L.l3d_create(int(device), C.c_void_p(stream))
        if not self.h:
            raise RuntimeError("l3d_create failed: " + _lib.last_error())
        self.h = C.c_void_p(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.l3d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- error behaviour of the reference: print and return ---------------------------------
    def _check(self, rc, what):
        self.last_status = rc
        if rc != 0:
            print(f"{self.PREFIX}ERROR: {what}: {_lib.last_error()} [{rc}]")
        return rc == 0

    # Line3D::addImage(camID, image, K, R, t, median_depth, wps_or_neighbors, line_segments), line3D.h:104-108
    def addImage(self, camID, image_size, K, R, t, median_depth, wps_or_neighbors, line_segments):
        """image_size = (cols, rows) stands in for the cv::Mat, which only contributes its size once
        `line_segments` are given (line3D.cc:119,198)."""
        segs = np.ascontiguousarray(line_segments, np.float32).reshape(-1, 4)
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        nb = np.ascontiguousarray(list(wps_or_neighbors), np.uint32)
        add = self.L.l3d_add_view_worldpoints if self.neighbors_by_worldpoints else self.L.l3d_add_view
        rc = add(self.h, int(camID), ptr(segs), len(segs), ptr(K), ptr(R), ptr(t),
                 int(image_size[0]), int(image_size[1]), float(median_depth), ptr(nb), len(nb))
        if self._check(rc, f"addImage [{camID}]"):
            self._M[int(camID)] = len(segs)

    def add_scene(self, scene):
        for v in scene.views:
            lst = v.worldpoints if self.neighbors_by_worldpoints else v.neighbors
            self.addImage(v.cam, (v.width, v.height), v.K, v.R, v.t, v.median_depth, lst, v.segs)

    def visualNeighbors(self, camID):
        """visual_neighbors_[camID] (line3D.h:352) as the last matchImages left it, ascending"""
        n = C.c_uint32(0)
        if not self._check(self.L.l3d_get_visual_neighbors(self.h, int(camID), None, 0, C.byref(n)), "visualNeighbors"):
            return None
        out = np.zeros(max(n.value, 1), np.uint32)
        self._check(self.L.l3d_get_visual_neighbors(self.h, int(camID), ptr(out), n.value, C.byref(n)), "visualNeighbors")
        return out[:n.value].copy()

    def _params(self, sigma_position, sigma_angle, num_neighbors, epipolar_overlap, kNN, const_regularization_depth):
        return MatchParams(float(sigma_position), float(sigma_angle), int(num_neighbors), float(epipolar_overlap),
                           int(kNN), float(const_regularization_depth))

    # Line3D::matchImages, line3D.h:143-148
    def matchImages(self, sigma_position=L3D_DEF_SCORING_POS_REGULARIZER, sigma_angle=L3D_DEF_SCORING_ANG_REGULARIZER,
                    num_neighbors=L3D_DEF_MATCHING_NEIGHBORS, epipolar_overlap=L3D_DEF_EPIPOLAR_OVERLAP,
                    kNN=L3D_DEF_KNN, const_regularization_depth=-1.0):
        p = self._params(sigma_position, sigma_angle, num_neighbors, epipolar_overlap, kNN, const_regularization_depth)
        return self._check(self.L.l3d_match_images(self.h, C.byref(p)), "matchImages")

    # split form for pair-sharded runs (see line3dpp_amd/dist.py)
    def matchBegin(self, sigma_position=L3D_DEF_SCORING_POS_REGULARIZER, sigma_angle=L3D_DEF_SCORING_ANG_REGULARIZER,
                   num_neighbors=L3D_DEF_MATCHING_NEIGHBORS, epipolar_overlap=L3D_DEF_EPIPOLAR_OVERLAP,
                   kNN=L3D_DEF_KNN, const_regularization_depth=-1.0):
        p = self._params(sigma_position, sigma_angle, num_neighbors, epipolar_overlap, kNN, const_regularization_depth)
        return self._check(self.L.l3d_match_begin(self.h, C.byref(p)), "matchBegin")

    def matchPairs(self, first, count):
        return self._check(self.L.l3d_match_pairs(self.h, int(first), int(count)), "matchPairs")

    def matchFinish(self):
        return self._check(self.L.l3d_match_finish(self.h), "matchFinish")

    def listsShard(self, rank, world):
        """phase B's list pass for this rank's share of the views (l3d_lists_shard): returns [(slab pointer, slab
        bytes, full-array pointer)] x 4 (edges, headers, segment headers, pool counters) or None"""
        sp = (C.c_void_p * 4)(); sb = (C.c_uint64 * 4)(); fp = (C.c_void_p * 4)()
        if not self._check(self.L.l3d_lists_shard(self.h, int(rank), int(world), sp, sb, fp), "listsShard"):
            return None
        return [(sp[k], int(sb[k]), fp[k]) for k in range(4)]

    def listsShardViews(self, rank, world, view0, view1):
        """the same for an explicit view range (l3d_lists_shard_views: the halo form, only the pairs that touch the
        range have to be present)"""
        sp = (C.c_void_p * 4)(); sb = (C.c_uint64 * 4)(); fp = (C.c_void_p * 4)()
        if not self._check(self.L.l3d_lists_shard_views(self.h, int(rank), int(world), int(view0), int(view1), sp, sb, fp),
                           "listsShardViews"):
            return None
        return [(sp[k], int(sb[k]), fp[k]) for k in range(4)]

    # the tail of phase B sharded by views (see line3dpp_amd/dist.py: match_images_halo)
    def tailShardCount(self):
        """l3d_tail_shard_count -> (status, surviving matches, best hypotheses of this rank's views)"""
        c = (C.c_uint32 * 2)()
        rc = self.L.l3d_tail_shard_count(self.h, c)
        self.last_status = rc
        return rc, int(c[0]), int(c[1])

    def shardOptions(self, first_needed_rank=0, exchanges_stream_ordered=False):
        """l3d_shard_options: the lowest rank whose records this rank's chain depends on; whether the caller's exchanges order
        themselves behind the context's stream (then the sharded entries return without waiting for the device)"""
        return self._check(self.L.l3d_shard_options(self.h, int(first_needed_rank), 1 if exchanges_stream_ordered else 0), "shardOptions")

    def tailShardLayout(self, world, counts_all, view_bounds):
        """l3d_tail_shard_layout -> [(device pointer of the full array, element bytes, [(first, count) per rank])] x 9, or None"""
        ca = (C.c_uint32 * (2 * world))(*[int(x) for pair in counts_all for x in pair])
        vb = (C.c_uint32 * (world + 1))(*[int(x) for x in view_bounds])
        bp = (C.c_void_p * 9)(); eb = (C.c_uint64 * 9)()
        first = (C.c_uint64 * (9 * world))(); count = (C.c_uint64 * (9 * world))()
        if not self._check(self.L.l3d_tail_shard_layout(self.h, int(world), ca, vb, bp, eb, first, count), "tailShardLayout"):
            return None
        return [(bp[k], int(eb[k]), [(int(first[9 * r + k]), int(count[9 * r + k])) for r in range(world)]) for k in range(9)]

    def tailShardCommit(self):
        rc = self.L.l3d_tail_shard_commit(self.h)
        self.last_status = rc
        return rc

    def matchAbort(self):
        """closes an open matchBegin without results (views untranslated, context idle); no-op otherwise"""
        return self.L.l3d_match_abort(self.h) == 0

    # the affinity part of Line3D::reconstruct3Dlines, line3D.h:162-166 / line3D.cc:1749-1778
    def computeAffinity(self):
        return self._check(self.L.l3d_compute_affinity(self.h), "computeAffinity")

    def affinityShardBegin(self, rank, world):
        """l3d_affinity_shard_begin -> (device pointer of the similarity array, 4, [(first, count) per rank]) in the form
        dist.exchange_parts takes, or None (not after a sharded tail / collinearity links asked for / an error)"""
        p = C.c_void_p(); first = (C.c_uint64 * world)(); count = (C.c_uint64 * world)()
        rc = self.L.l3d_affinity_shard_begin(self.h, int(rank), int(world), C.byref(p), first, count)
        self.last_status = rc
        if rc != 0:
            return None
        return (p.value, 4, [(int(first[r]), int(count[r])) for r in range(world)])

    def affinityShardFinish(self):
        return self._check(self.L.l3d_affinity_shard_finish(self.h), "affinityShardFinish")

    def affinityShardAbort(self):
        """l3d_affinity_shard_abort: close an open sharded fill without its bookkeeping pass (a peer could not shard)"""
        return self._check(self.L.l3d_affinity_shard_abort(self.h), "affinityShardAbort")

    # Line3D::reconstruct3Dlines, line3D.h:162-166 (defaults commons.h:63-70)
    def reconstruct3Dlines(self, visibility_t=L3D_DEF_MIN_VISIBILITY_T, perform_diffusion=False, collinearity_t=-1.0,
                           use_CERES=False, max_iter_CERES=250):
        if use_CERES:           # line3D.cc:1741-1743
            print(f"{self.PREFIX}ERROR: CERES was not found! no optimization will be performed...")
        return self._check(self.L.l3d_reconstruct_3d_lines(self.h, int(visibility_t), int(perform_diffusion),
                                                           float(collinearity_t), int(use_CERES), int(max_iter_CERES)),
                           "reconstruct3Dlines")

    # Line3D::createOutputFilename (private in the reference; line3D.cc:2853-2893)
    def outputFilename(self, max_image_width=-1):
        buf = C.create_string_buffer(512)
        if not self._check(self.L.l3d_output_filename(self.h, int(max_image_width), buf, 512), "outputFilename"):
            return None
        return buf.value.decode()

    # Line3D::save3DLinesAsTXT, line3D.h:176 -- <output_folder>/<outputFilename()>.txt
    def save3DLinesAsTXT(self, output_folder, max_image_width=-1):
        return self._check(self.L.l3d_save_3d_lines_txt(self.h, str(output_folder).encode(), int(max_image_width)),
                           "save3DLinesAsTXT")

    # Line3D::save3DLinesAsBIN, line3D.h:177 -- <output_folder>/<outputFilename()>.bin (boost binary archive layout)
    def save3DLinesAsBIN(self, output_folder, max_image_width=-1):
        return self._check(self.L.l3d_save_3d_lines_bin(self.h, str(output_folder).encode(), int(max_image_width)),
                           "save3DLinesAsBIN")

    # Line3D::getSegmentCoords2D, line3D.h:195-197
    def getSegmentCoords2D(self, camID, segID):
        out = np.zeros(4, np.float32)
        self._check(self.L.l3d_get_segment_coords2d(self.h, int(camID), int(segID), ptr(out)), "getSegmentCoords2D")
        return out

    # Line3D::saveResultAsSTL / saveResultAsOBJ, line3D.h:174-175
    def saveResultAsSTL(self, output_folder, max_image_width=-1):
        return self._check(self.L.l3d_save_result_stl(self.h, str(output_folder).encode(), int(max_image_width)),
                           "saveResultAsSTL")

    def saveResultAsOBJ(self, output_folder, max_image_width=-1):
        return self._check(self.L.l3d_save_result_obj(self.h, str(output_folder).encode(), int(max_image_width)),
                           "saveResultAsOBJ")

    # Line3D::get3Dlines, line3D.h:173: list of FinalLine3D as dicts
    def get3Dlines(self):
        nl = C.c_uint32(); ns = C.c_uint32(); nr = C.c_uint32()
        if not self._check(self.L.l3d_num_3d_lines(self.h, C.byref(nl), C.byref(ns), C.byref(nr)), "get3Dlines"):
            return []
        so = np.zeros(nl.value + 1, np.uint32); ro = np.zeros(nl.value + 1, np.uint32)
        segs = np.zeros(max(ns.value, 1), SEGMENT3D_DTYPE); res = np.zeros(max(nr.value, 1), SEGMENT2D_DTYPE)
        cl = np.zeros(max(nl.value, 1), SEGMENT3D_DTYPE); rv = np.zeros(max(nl.value, 1), np.uint32)
        self.L.l3d_get_3d_lines(self.h, ptr(so), ptr(segs), ptr(ro), ptr(res), ptr(cl), ptr(rv))
        return [dict(collinear3Dsegments=segs[so[i]:so[i + 1]], residuals=res[ro[i]:ro[i + 1]],
                     cluster_line=cl[i], reference_view=int(rv[i])) for i in range(nl.value)]

    def set_brute_force(self, on):
        self.L.l3d_set_brute_force(self.h, int(on))

    # ---- accessors ----------------------------------------------------------------------------
    def numImages(self):
        return len(self._M)

    def pairs(self):
        n = C.c_uint32()
        self.L.l3d_num_pairs(self.h, C.byref(n))
        s = np.zeros(n.value, np.uint32); t = np.zeros(n.value, np.uint32); off = np.zeros(n.value, np.uint64)
        self.L.l3d_get_pairs(self.h, ptr(s), ptr(t), ptr(off))
        return np.stack([s, t], 1), off

    def slot_buffer(self):
        p = C.c_void_p(); n = C.c_uint64()
        self.L.l3d_slot_buffer(self.h, C.byref(p), C.byref(n))
        return p.value, n.value

    def slot_index_buffer(self):
        """device pointer of the compact exchange buffer (uint32 target index per slot) and its length"""
        p = C.c_void_p(); n = C.c_uint64()
        if not self._check(self.L.l3d_slot_index_buffer(self.h, C.byref(p), C.byref(n)), "slot_index_buffer"):
            return None, 0
        return p.value, n.value

    def packSlotIndices(self, first, count):
        return self._check(self.L.l3d_pack_slot_indices(self.h, int(first), int(count)), "packSlotIndices")

    def expandSlotIndices(self, first, count):
        return self._check(self.L.l3d_expand_slot_indices(self.h, int(first), int(count)), "expandSlotIndices")

    def pair_tests(self):
        n = C.c_uint64()
        self.L.l3d_pair_tests(self.h, C.byref(n))
        return n.value

    def pair_slots(self, pair_index):
        Ms = C.c_uint32(); K = C.c_uint32()
        if not self._check(self.L.l3d_get_pair_slots(self.h, pair_index, None, 0, C.byref(Ms), C.byref(K)), "pair_slots"):
            return None
        out = np.zeros((Ms.value, K.value), SLOT_DTYPE)
        self._check(self.L.l3d_get_pair_slots(self.h, pair_index, ptr(out), out.size, C.byref(Ms), C.byref(K)),
                    "pair_slots")
        return out

    def matches(self, camID):
        """matches_[camID] as (Match records, CSR offsets over the view's segments)"""
        M = self._M[camID]
        n = C.c_uint64()
        off = np.zeros(M + 1, np.uint32)
        if not self._check(self.L.l3d_get_matches(self.h, camID, None, 0, ptr(off), C.byref(n)), "matches"):
            return None, None
        out = np.zeros(max(n.value, 1), MATCH_DTYPE)
        self._check(self.L.l3d_get_matches(self.h, camID, ptr(out), n.value, ptr(off), C.byref(n)), "matches")
        return out[:n.value], off

    def best(self):
        """estimated_position3D_: (Segment2D keys, Segment3D, best Match), ordered by (camID, segID)"""
        n = C.c_uint32()
        self.L.l3d_num_best(self.h, C.byref(n))
        s2 = np.zeros(n.value, SEGMENT2D_DTYPE); s3 = np.zeros(n.value, SEGMENT3D_DTYPE); m = np.zeros(n.value, MATCH_DTYPE)
        if n.value:
            self._check(self.L.l3d_get_best(self.h, ptr(s2), ptr(s3), ptr(m)), "best")
        return s2, s3, m

    def view_info(self, camID):
        k = C.c_float(); md = C.c_float()
        self.L.l3d_view_info(self.h, camID, C.byref(k), C.byref(md))
        return dict(k=np.float32(k.value), median_depth=np.float32(md.value))

    def translation(self):
        t = np.zeros(3)
        self.L.l3d_translation(self.h, ptr(t))
        return t

    def affinity(self):
        """A_ (CLEdge array), local2global_ (Segment2D per matrix row), med_scene_depth_lines_"""
        ne = C.c_uint32(); nr = C.c_uint32()
        if not self._check(self.L.l3d_num_affinity(self.h, C.byref(ne), C.byref(nr)), "affinity"):
            return None, None, None
        e = np.zeros(max(ne.value, 1), CLEDGE_DTYPE); l2g = np.zeros(max(nr.value, 1), SEGMENT2D_DTYPE)
        msdl = C.c_float()
        self.L.l3d_get_affinity(self.h, ptr(e), ptr(l2g), C.byref(msdl))
        return e[:ne.value], l2g[:nr.value], np.float32(msdl.value)

    def sparse_matrix(self, sort_by_row=False):
        """L3DPP::SparseMatrix(A_, n_rows, 1.0f, sort_by_row): (float4 entries, int start_indices)"""
        ne = C.c_uint32(); nr = C.c_uint32()
        self.L.l3d_num_affinity(self.h, C.byref(ne), C.byref(nr))
        ent = np.zeros(max(ne.value, 1), FLOAT4_DTYPE); start = np.zeros(max(nr.value, 1), np.int32)
        self._check(self.L.l3d_get_sparse_matrix(self.h, int(sort_by_row), ptr(ent), ptr(start)), "sparse_matrix")
        return ent[:ne.value], start[:nr.value]

    def setTimingLevel(self, level):
        """l3d_set_timing_level: 1 = the match kernel only (default), 2 = every phase timed with HIP events (profiling), 0 = none"""
        return self._check(self.L.l3d_set_timing_level(self.h, int(level)), "setTimingLevel")

    def timings(self):
        t = Timings()
        self.L.l3d_get_timings(self.h, C.byref(t))
        return {f: getattr(t, f) for f, _ in Timings._fields_}


def match_lines(lines_src, lines_tgt, F, RtKinv_src, RtKinv_tgt, C_src, C_tgt, width, height, epi_overlap=0.25,
                kNN=10, device=0):
    """Seam-level call replacing match_lines_GPU (cudawrapper.h:54-63) with CPU-path semantics."""
    L = _lib.load()
    a = np.ascontiguousarray(lines_src, np.float32).reshape(-1, 4)
    b = np.ascontiguousarray(lines_tgt, np.float32).reshape(-1, 4)
    arrs = [np.ascontiguousarray(x, np.float64) for x in (F, RtKinv_src, RtKinv_tgt, C_src, C_tgt)]
    out = np.zeros((len(a), kNN), SLOT_DTYPE)
    n = C.c_uint64()
    rc = L.l3d_match_lines(device, ptr(a), len(a), ptr(b), len(b), *[ptr(x) for x in arrs], width, height,
                           float(epi_overlap), int(kNN), ptr(out), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"l3d_match_lines failed [{rc}]: {_lib.last_error()}")
    return out, n.value


def diffuse_affinity(edges, n_rows, iterations=10, device=0):
    """Seam-level call replacing the body of Line3D::performRDD (line3D.cc:2026-2076): replicator-dynamics
    diffusion of the affinity matrix + min-symmetrisation; returns the CLEdges in (i, j) order."""
    from ._lib import CLEDGE_DTYPE
    L = _lib.load()
    e = np.ascontiguousarray(edges, CLEDGE_DTYPE)
    out = np.zeros(len(e), CLEDGE_DTYPE)
    rc = L.l3d_diffuse_affinity(device, ptr(e), len(e), int(n_rows), int(iterations), ptr(out))
    if rc != 0:
        raise RuntimeError(f"l3d_diffuse_affinity failed [{rc}]: {_lib.last_error()}")
    return out


def find_collinear_segments(lines, dist_t, device=0):
    """Seam-level call replacing View::findCollinGPU / find_collinear_segments_GPU (view.cc:173-209) with the
    semantics of View::findCollinCPU: CSR (offsets[M+1], idx) of the collinear segments of every segment."""
    L = _lib.load()
    a = np.ascontiguousarray(lines, np.float32).reshape(-1, 4)
    off = np.zeros(len(a) + 1, np.uint32)
    n = C.c_uint64()
    rc = L.l3d_find_collinear_segments(device, ptr(a), len(a), float(dist_t), ptr(off), None, 0, C.byref(n))
    idx = np.zeros(max(n.value, 1), np.uint32)
    if rc == 0 and n.value:
        rc = L.l3d_find_collinear_segments(device, ptr(a), len(a), float(dist_t), ptr(off), ptr(idx), n.value, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"l3d_find_collinear_segments failed [{rc}]: {_lib.last_error()}")
    return off, idx[:n.value]


def score_matches(lines, matches4, ranges2, reg_tgt2, RtKinv, C_, two_sigA_sqr, k, device=0):
    """Seam-level call replacing score_matches_GPU (cudawrapper.h:70-73) with the semantics of Line3D::scoringCPU;
    arrays as Line3D::scoringGPU marshals them (see include/l3dpp_hip.h)."""
    L = _lib.load()
    a = np.ascontiguousarray(lines, np.float32).reshape(-1, 4)
    m = np.ascontiguousarray(matches4, np.float32).reshape(-1, 4)
    r = np.ascontiguousarray(ranges2, np.int32).reshape(-1, 2)
    g = np.ascontiguousarray(reg_tgt2, np.float32).reshape(-1, 2)
    A = np.ascontiguousarray(RtKinv, np.float64); Cc = np.ascontiguousarray(C_, np.float64)
    out = np.zeros(len(m), np.float32)
    rc = L.l3d_score_matches(device, ptr(a), len(a), ptr(m), ptr(r), ptr(g), len(m), ptr(A), ptr(Cc),
                             float(two_sigA_sqr), float(k), ptr(out))
    if rc != 0:
        raise RuntimeError(f"l3d_score_matches failed [{rc}]: {_lib.last_error()}")
    return out


def neighbors_from_worldpoints(cams, K, R, t, worldpoints, num_neighbors=10):
    """Line3D::findVisualNeighborsFromWPs (line3D.cc:578-699) without a context or a GPU: cams = camera ids, K / R / t per
    camera as handed to addImage, worldpoints = one list of worldpoint ids per camera -> {cam: ascending neighbour ids}"""
    L = _lib.load()
    n = len(cams)
    ids = np.ascontiguousarray(cams, np.uint32)
    Ka = np.ascontiguousarray(K, np.float64).reshape(n, 9); Ra = np.ascontiguousarray(R, np.float64).reshape(n, 9)
    ta = np.ascontiguousarray(t, np.float64).reshape(n, 3)
    off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum([len(w) for w in worldpoints])
    wps = np.ascontiguousarray(np.concatenate([np.asarray(w, np.uint32) for w in worldpoints]) if off[-1] else np.zeros(1, np.uint32), np.uint32)
    nb_off = np.zeros(n + 1, np.uint64)
    rc = L.l3d_neighbors_from_worldpoints(n, ptr(ids), ptr(Ka), ptr(Ra), ptr(ta), ptr(off), ptr(wps), int(num_neighbors), ptr(nb_off), None, 0)
    if rc != 0:
        raise RuntimeError("l3d_neighbors_from_worldpoints: " + _lib.last_error())
    nb = np.zeros(max(int(nb_off[-1]), 1), np.uint32)
    rc = L.l3d_neighbors_from_worldpoints(n, ptr(ids), ptr(Ka), ptr(Ra), ptr(ta), ptr(off), ptr(wps), int(num_neighbors), ptr(nb_off), ptr(nb), len(nb))
    if rc != 0:
        raise RuntimeError("l3d_neighbors_from_worldpoints: " + _lib.last_error())
    return {int(c): nb[int(nb_off[i]):int(nb_off[i + 1])].copy() for i, c in enumerate(cams)}
