"""ctypes front-end of the CPU oracle (oracle/l3d_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Nothing under line3dpp_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libl3d_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libl3d_ref.so")   # the reference's own sources, compiled in place (-O2, asserts on)
_REL_SO = os.path.join(_HERE, "_ref", "libl3d_ref_release.so")   # the same at -O3 -DNDEBUG (the reference's Release build)
# the reference WITH its CUDA path, every CUDA construct executed on the host (oracle/ref_shim_cuda): pins performRDD
_CUDA_SO = os.path.join(_HERE, "_ref", "libl3d_ref_cuda.so")

# commons.h:186-203
MATCH_DTYPE = np.dtype([
    ("src_cam", "<u4"), ("src_seg", "<u4"), ("tgt_cam", "<u4"), ("tgt_seg", "<u4"),
    ("overlap", "<f4"), ("score3D", "<f4"),
    ("d_p1", "<f4"), ("d_p2", "<f4"), ("d_q1", "<f4"), ("d_q2", "<f4")])
assert MATCH_DTYPE.itemsize == 40
# clustering.h:47-51
CLEDGE_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("w", "<f4")])


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile). Building the checker is not using it."""
    src = os.path.join(_HERE, "l3d_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_libs = {}


def have_reference():
    """True if oracle/_ref/libl3d_ref.so (the reference's own line3D.cc/view.cc, built by oracle/Makefile
    where /root/reference exists) is available."""
    return os.path.exists(_REF_SO)


def have_release():
    return os.path.exists(_REL_SO)


def have_cuda_path():
    return os.path.exists(_CUDA_SO)


_cuda_lib = None


def rdd_reference(edges, n_rows):
    """Replicator-dynamics diffusion + symmetrisation of an affinity edge list by the reference's OWN code:
    Line3D::performRDD (line3D.cc:2026-2076) -> SparseMatrix (sparsematrix.cc) -> replicator_dynamics_diffusion_GPU and
    the K_sparseMat_* kernels (cudawrapper.cu:432-544, 708-766), compiled in place and run as host code
    (oracle/_ref/libl3d_ref_cuda.so).  10 iterations (L3D_DEF_RDD_MAX_ITER)."""
    global _cuda_lib
    if _cuda_lib is None:
        if not os.path.exists(_CUDA_SO):
            raise RuntimeError(_CUDA_SO + " is missing (needs /root/reference to build)")
        _cuda_lib = C.CDLL(_CUDA_SO)
        _cuda_lib.lo_ref_rdd.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        _cuda_lib.lo_ref_rdd.restype = C.c_uint32
    e = np.ascontiguousarray(edges, CLEDGE_DTYPE)
    out = np.zeros(2 * max(len(e), 1), CLEDGE_DTYPE)
    n = _cuda_lib.lo_ref_rdd(_p(e), len(e), int(n_rows), _p(out))
    return out[:n]


def lib(reference=False):
    """reference: False = the restatement, True = the reference's own code (-O2 build), "release" = its -O3 -DNDEBUG
    build (CMAKE_BUILD_TYPE Release of the reference's CMakeLists.txt:3)"""
    key = "rel" if reference == "release" else ("ref" if reference else "port")
    if key not in _libs:
        if reference:
            path = _REL_SO if key == "rel" else _REF_SO
            if not os.path.exists(path):
                raise RuntimeError(path + " is missing (needs /root/reference to build)")
            L = C.CDLL(path)
        else:
            build()
            L = C.CDLL(_SO)
        vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
        L.lo_create.restype = vp
        L.lo_destroy.argtypes = [vp]
        L.lo_set_record_scored.argtypes = [vp, i32]
        L.lo_set_threads.argtypes = [i32]
        L.lo_max_threads.restype = i32
        L.lo_add_view.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, u32, f32, vp, u32]
        L.lo_add_view.restype = i32
        for fn in (L.lo_match_images, L.lo_begin_match):
            fn.argtypes = [vp, f32, f32, u32, f32, i32, f32]
            fn.restype = None
        L.lo_end_match.argtypes = [vp]
        L.lo_match_pair.argtypes = [vp, u32, u32, vp, u64, vp]
        L.lo_match_pair.restype = u64
        L.lo_fundamental.argtypes = [vp, u32, u32, vp]
        L.lo_compute_affinity.argtypes = [vp]
        L.lo_num_pairs.argtypes = [vp]; L.lo_num_pairs.restype = u32
        L.lo_get_pairs.argtypes = [vp, vp, vp]
        L.lo_pair_tests.argtypes = [vp]; L.lo_pair_tests.restype = u64
        for fn in (L.lo_get_matches, L.lo_get_scored):
            fn.argtypes = [vp, u32, vp, u64, vp]
            fn.restype = u64
        L.lo_num_best.argtypes = [vp]; L.lo_num_best.restype = u32
        L.lo_get_best.argtypes = [vp, vp, vp, vp, vp]
        L.lo_view_info.argtypes = [vp, u32, vp, vp, vp, vp]
        L.lo_translation.argtypes = [vp, vp]
        L.lo_med_scene_depth_lines.argtypes = [vp]; L.lo_med_scene_depth_lines.restype = f32
        L.lo_num_edges.argtypes = [vp]; L.lo_num_edges.restype = u32
        L.lo_num_rows.argtypes = [vp]; L.lo_num_rows.restype = u32
        L.lo_get_affinity.argtypes = [vp, vp, vp]
        L.lo_set_collinearity.argtypes = [vp, f32]
        L.lo_get_collinear.argtypes = [vp, u32, vp, vp, u32]; L.lo_get_collinear.restype = u32
        if not reference:
            L.lo_rdd.argtypes = [vp, u32, u32, u32, vp]; L.lo_rdd.restype = u32
        if reference:
            L.lo_ref_view_matrices.argtypes = [vp, u32, vp, vp]
            L.lo_ref_fundamental.argtypes = [vp, u32, u32, vp]
            L.lo_ref_principal_direction.argtypes = [vp, vp]
            L.lo_reconstruct.argtypes = [vp, u32]
            L.lo_reconstruct_collin.argtypes = [vp, u32, f32]
            L.lo_save_txt.argtypes = [vp, C.c_char_p]
            L.lo_num_lines.argtypes = [vp, vp, vp, vp]
            L.lo_get_lines.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Mirror of the slice of L3DPP::Line3D the hot path needs (explicit segments + neighbours)."""

    def __init__(self, record_scored=False, threads=1, reference=False, by_worldpoints=False):
        """reference=True drives the reference's own translation units (oracle/_ref) instead of the
        restatement; stage-level calls (begin_match/match_pair/scored) exist only in the restatement.
        by_worldpoints=True (reference only): the instance is constructed with neighbors_by_worldpoints=true -- add_view's
        list is a worldpoint list and matchImages finds the visual neighbours itself (line3D.cc:578-699)."""
        self.reference = reference
        self.by_worldpoints = by_worldpoints
        self.L = lib(reference)
        if by_worldpoints:
            assert reference, "worldpoint-derived neighbours are pinned on the reference's own code only"
            self.L.lo_create_worldpoints.restype = C.c_void_p
            self.L.lo_get_visual_neighbors.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
            self.L.lo_get_visual_neighbors.restype = C.c_uint32
            self.h = C.c_void_p(self.L.lo_create_worldpoints())
        else:
            self.h = C.c_void_p(self.L.lo_create())
        self.L.lo_set_record_scored(self.h, int(record_scored))
        self.L.lo_set_threads(int(threads))
        self.M = {}

    def __del__(self):
        try:
            self.L.lo_destroy(self.h)
        except Exception:
            pass

    def add_view(self, cam, segs, K, R, t, width, height, median_depth, neighbors):
        segs = np.ascontiguousarray(segs, np.float32).reshape(-1, 4)
        K = np.ascontiguousarray(K, np.float64); R = np.ascontiguousarray(R, np.float64)
        t = np.ascontiguousarray(t, np.float64)
        nb = np.ascontiguousarray(list(neighbors), np.uint32)
        rc = self.L.lo_add_view(self.h, cam, _p(segs), len(segs), _p(K), _p(R), _p(t), width, height,
                                float(median_depth), _p(nb), len(nb))
        if rc == 0:
            self.M[cam] = len(segs)
        return rc

    def add_scene(self, scene):
        for v in scene.views:
            rc = self.add_view(v.cam, v.segs, v.K, v.R, v.t, v.width, v.height, v.median_depth,
                               v.worldpoints if self.by_worldpoints else v.neighbors)
            assert rc == 0, rc

    def visual_neighbors(self, cam):
        """visual_neighbors_[cam] after match_images (reference, by_worldpoints)"""
        n = self.L.lo_get_visual_neighbors(self.h, cam, None, 0)
        out = np.zeros(max(n, 1), np.uint32)
        self.L.lo_get_visual_neighbors(self.h, cam, _p(out), n)
        return out[:n].copy()

    def match_images(self, sigma_p=2.5, sigma_a=10.0, num_neighbors=10, epi_overlap=0.25, kNN=10,
                     const_reg_depth=-1.0):
        self.L.lo_match_images(self.h, sigma_p, sigma_a, num_neighbors, epi_overlap, kNN, const_reg_depth)

    def begin_match(self, sigma_p=2.5, sigma_a=10.0, num_neighbors=10, epi_overlap=0.25, kNN=10,
                    const_reg_depth=-1.0):
        self.L.lo_begin_match(self.h, sigma_p, sigma_a, num_neighbors, epi_overlap, kNN, const_reg_depth)

    def end_match(self):
        self.L.lo_end_match(self.h)

    def match_pair(self, src, tgt, cap=None):
        Ms = self.M[src]
        cap = cap or Ms * 64
        while True:
            out = np.zeros(cap, MATCH_DTYPE)
            off = np.zeros(Ms + 1, np.uint32)
            n = self.L.lo_match_pair(self.h, src, tgt, _p(out), cap, _p(off))
            if n <= cap:
                return out[:n], off
            cap = int(n)

    def fundamental(self, src, tgt):
        F = np.zeros(9, np.float64)
        self.L.lo_fundamental(self.h, src, tgt, _p(F))
        return F.reshape(3, 3)

    def set_collinearity(self, t):
        """collinearity_t_ of reconstruct3Dlines (line3D.cc:1725); > 0 adds the collinear-segment links"""
        self.L.lo_set_collinearity(self.h, float(t))

    def collinear(self, cam, M):
        """View::collinearSegments for every segment of view `cam` as CSR (offsets[M+1], idx)"""
        off = np.zeros(M + 1, np.uint32)
        n = self.L.lo_get_collinear(self.h, int(cam), _p(off), None, 0)
        idx = np.zeros(max(n, 1), np.uint32)
        self.L.lo_get_collinear(self.h, int(cam), _p(off), _p(idx), n)
        return off, idx[:n]

    def compute_affinity(self):
        self.L.lo_compute_affinity(self.h)

    def pairs(self):
        n = self.L.lo_num_pairs(self.h)
        s = np.zeros(n, np.uint32); t = np.zeros(n, np.uint32)
        self.L.lo_get_pairs(self.h, _p(s), _p(t))
        return np.stack([s, t], 1)

    def pair_tests(self):
        return int(self.L.lo_pair_tests(self.h))

    def _rows(self, fn, cam):
        Ms = self.M[cam]
        off = np.zeros(Ms + 1, np.uint32)
        n = fn(self.h, cam, None, 0, _p(off))
        out = np.zeros(max(int(n), 1), MATCH_DTYPE)
        n = fn(self.h, cam, _p(out), len(out), _p(off))
        return out[:n], off

    def matches(self, cam):
        """surviving matches_[cam] as (records, CSR offsets over segments)"""
        return self._rows(self.L.lo_get_matches, cam)

    def scored(self, cam):
        return self._rows(self.L.lo_get_scored, cam)

    def best(self):
        n = self.L.lo_num_best(self.h)
        camseg = np.zeros((n, 2), np.uint32); geo = np.zeros((n, 9), np.float64)
        length = np.zeros(n, np.float32); m = np.zeros(n, MATCH_DTYPE)
        if n:
            self.L.lo_get_best(self.h, _p(camseg), _p(geo), _p(length), _p(m))
        return camseg, geo, length, m

    def view_info(self, cam):
        k = C.c_float(); md = C.c_float()
        Cc = np.zeros(3); t = np.zeros(3)
        self.L.lo_view_info(self.h, cam, C.byref(k), C.byref(md), _p(Cc), _p(t))
        return dict(k=np.float32(k.value), median_depth=np.float32(md.value), C=Cc, t=t)

    def translation(self):
        t = np.zeros(3)
        self.L.lo_translation(self.h, _p(t))
        return t

    def med_scene_depth_lines(self):
        return np.float32(self.L.lo_med_scene_depth_lines(self.h))

    def affinity(self):
        ne = self.L.lo_num_edges(self.h); nr = self.L.lo_num_rows(self.h)
        e = np.zeros(max(ne, 1), CLEDGE_DTYPE); l2g = np.zeros((max(nr, 1), 2), np.uint32)
        if ne:
            self.L.lo_get_affinity(self.h, _p(e), _p(l2g))
        return e[:ne], l2g[:nr]

    @staticmethod
    def rdd(edges, n_rows, iterations=10):
        """Replicator-dynamics diffusion + symmetrisation of an affinity edge list (restatement of the reference's
        CUDA-only performRDD; see l3d_oracle.cpp lo_rdd)."""
        L = lib(False)
        e = np.ascontiguousarray(edges, CLEDGE_DTYPE)
        out = np.zeros(2 * max(len(e), 1), CLEDGE_DTYPE)
        n = L.lo_rdd(_p(e), len(e), int(n_rows), int(iterations), _p(out))
        return out[:n]

    # ---- the reference's linear algebra as compiled here (oracle/ref_shim Eigen subset), for tests/test_shim_vs_lapack.py
    def ref_view_matrices(self, cam):
        assert self.reference
        a = np.zeros((3, 3)); b = np.zeros((3, 3))
        self.L.lo_ref_view_matrices(self.h, int(cam), _p(a), _p(b))
        return a, b

    def ref_fundamental(self, src, tgt):
        assert self.reference
        F = np.zeros((3, 3))
        self.L.lo_ref_fundamental(self.h, int(src), int(tgt), _p(F))
        return F

    @staticmethod
    def ref_principal_direction(S):
        L = lib(True)
        S = np.ascontiguousarray(S, np.float64).reshape(3, 3); d = np.zeros(3)
        L.lo_ref_principal_direction(_p(S), _p(d))
        return d

    # ---- reconstruction tail: only through the reference's own code (oracle/_ref) -------------------------
    def reconstruct(self, visibility_t=3, collinearity_t=-1.0):
        assert self.reference, "reconstruct3Dlines is run by the reference's own code only (oracle/_ref)"
        if collinearity_t > 0:
            self.L.lo_reconstruct_collin(self.h, int(visibility_t), float(collinearity_t))
        else:
            self.L.lo_reconstruct(self.h, int(visibility_t))

    def save_txt(self, folder):
        """the reference's own Line3D::save3DLinesAsTXT into `folder`"""
        assert self.reference
        self.L.lo_save_txt(self.h, str(folder).encode())

    def lines(self):
        """lines3D_ as a list of dicts: collinear3Dsegments [n,9] (P1,P2,dir), residuals [m,2], cluster_line [9],
        reference_view"""
        assert self.reference
        nl = C.c_uint32(); ns = C.c_uint32(); nr = C.c_uint32()
        self.L.lo_num_lines(self.h, C.byref(nl), C.byref(ns), C.byref(nr))
        so = np.zeros(nl.value + 1, np.uint32); ro = np.zeros(nl.value + 1, np.uint32)
        segs = np.zeros((max(ns.value, 1), 9)); res = np.zeros((max(nr.value, 1), 2), np.uint32)
        cl = np.zeros((max(nl.value, 1), 9)); rv = np.zeros(max(nl.value, 1), np.uint32)
        self.L.lo_get_lines(self.h, _p(so), _p(segs), _p(ro), _p(res), _p(cl), _p(rv))
        return [dict(collinear3Dsegments=segs[so[i]:so[i + 1]], residuals=res[ro[i]:ro[i + 1]], cluster_line=cl[i],
                     reference_view=int(rv[i])) for i in range(nl.value)]
