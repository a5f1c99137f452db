"""ctypes binding of libl3dpp_hip.so (C-ABI: include/l3dpp_hip.h).

There is no CPU fallback: if the HIP library is missing this module raises, and every compute
entry point fails with L3D_ERR_HIP when no MI355X/HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libl3dpp_hip.so")
# L3D_LIB=<path>: load another build of the same library (diagnostic builds with -DL3D_STATS / -DL3D_CYCLES)
SO_PATH = os.environ.get("L3D_LIB", SO_PATH)

# reference PODs (include/l3dpp_hip.h)
MATCH_DTYPE = np.dtype([
    ("src_cam", "<u4"), ("src_seg", "<u4"), ("tgt_cam", "<u4"), ("tgt_seg", "<u4"),
    ("overlap", "<f4"), ("score3D", "<f4"),
    ("d_p1", "<f4"), ("d_p2", "<f4"), ("d_q1", "<f4"), ("d_q2", "<f4")])
SEGMENT2D_DTYPE = np.dtype([("cam", "<u4"), ("seg", "<u4")])
SEGMENT3D_DTYPE = np.dtype([("P1", "<f8", 3), ("P2", "<f8", 3), ("dir", "<f8", 3), ("length", "<f4"), ("valid", "<u4")])
CLEDGE_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("w", "<f4")])
SLOT_DTYPE = np.dtype([
    ("tgt_seg", "<u4"), ("overlap", "<f4"), ("d_p1", "<f4"), ("d_p2", "<f4"), ("d_q1", "<f4"), ("d_q2", "<f4"),
    ("score3D", "<f4"), ("flags", "<u4")])
FLOAT4_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
assert MATCH_DTYPE.itemsize == 40 and SLOT_DTYPE.itemsize == 32 and SEGMENT3D_DTYPE.itemsize == 80
EMPTY = 0xFFFFFFFF


class MatchParams(C.Structure):
    _fields_ = [("sigma_position", C.c_float), ("sigma_angle", C.c_float), ("num_neighbors", C.c_uint32),
                ("epipolar_overlap", C.c_float), ("kNN", C.c_int32), ("const_regularization_depth", C.c_float)]


class SfmImage(C.Structure):
    """l3d_sfm_image (include/l3dpp_hip.h)"""
    _fields_ = [("id", C.c_uint32), ("camera", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("name", C.c_char_p), ("focal", C.c_float), ("median_depth", C.c_float), ("n_worldpoints", C.c_uint32),
                ("K", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3), ("C", C.c_double * 3),
                ("radial", C.c_double * 3), ("tangential", C.c_double * 2)]


class Timings(C.Structure):
    _fields_ = [("begin_ms", C.c_float), ("match_pairs_ms", C.c_float), ("finish_ms", C.c_float),
                ("affinity_ms", C.c_float), ("match_kernel_launches", C.c_uint32), ("match_kernel_ms", C.c_float),
                ("cull_prepare_ms", C.c_float), ("culled_pairs", C.c_uint32),
                ("list_entries", C.c_uint32), ("support_words", C.c_uint32), ("tied_rows", C.c_uint32), ("chain_sweeps", C.c_uint32),
                ("chain_extra_rounds", C.c_uint32), ("pool_retries", C.c_uint32),
                ("lists_ms", C.c_float), ("record_kbytes", C.c_uint32),
                ("list_inverse", C.c_uint32), ("list_candidates", C.c_uint32), ("list_headers", C.c_uint32),
                ("slots_lo", C.c_uint32), ("slots_hi", C.c_uint32)]


EXPORTS = [
    "l3d_last_error", "l3d_build_info", "l3d_create", "l3d_destroy", "l3d_add_view", "l3d_match_images",
    "l3d_match_begin", "l3d_num_pairs", "l3d_get_pairs", "l3d_match_pairs", "l3d_slot_buffer", "l3d_match_finish",
    "l3d_compute_affinity", "l3d_synchronize", "l3d_pair_tests", "l3d_get_matches", "l3d_get_pair_slots",
    "l3d_num_best", "l3d_get_best", "l3d_view_info", "l3d_translation", "l3d_num_affinity", "l3d_get_affinity",
    "l3d_get_sparse_matrix", "l3d_get_timings", "l3d_match_lines", "l3d_set_brute_force", "l3d_slots_exchanged",
    "l3d_reconstruct_3d_lines", "l3d_num_3d_lines", "l3d_get_3d_lines", "l3d_diffuse_affinity",
    "l3d_output_filename", "l3d_save_3d_lines_txt", "l3d_save_result_stl", "l3d_save_result_obj",
    "l3d_get_segment_coords2d", "l3d_find_collinear_segments", "l3d_score_matches",
    "l3d_slot_index_buffer", "l3d_pack_slot_indices", "l3d_expand_slot_indices", "l3d_match_abort", "l3d_save_3d_lines_bin", "l3d_lists_shard",
    "l3d_principal_direction", "l3d_selftest_arith", "l3d_lists_shard_views", "l3d_plan_shards",
    "l3d_add_view_worldpoints", "l3d_get_visual_neighbors", "l3d_neighbors_from_worldpoints",
    "l3d_nvm_open", "l3d_nvm_num_cameras", "l3d_nvm_get_camera", "l3d_nvm_get_worldpoints", "l3d_nvm_close",
    "l3d_nvm_intrinsics", "l3d_segment_cache_name", "l3d_read_segment_cache", "l3d_write_segment_cache",
    "l3d_trim_cache", "l3d_set_timing_level", "l3d_tail_shard_count", "l3d_tail_shard_layout", "l3d_tail_shard_commit",
    "l3d_sfm_open_colmap", "l3d_sfm_open_bundler", "l3d_sfm_num_images", "l3d_sfm_get_image", "l3d_sfm_get_worldpoints",
    "l3d_sfm_close", "l3d_debug_counter", "l3d_affinity_shard_begin", "l3d_affinity_shard_finish", "l3d_affinity_shard_abort", "l3d_shard_options",
]

_lib = None


def load():
    """Load the HIP library; fail loudly if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: the HIP extension has not been built (run `make -C line3dpp_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    L.l3d_last_error.restype = C.c_char_p
    L.l3d_build_info.restype = C.c_char_p
    L.l3d_trim_cache.argtypes = []; L.l3d_trim_cache.restype = u64
    L.l3d_create.argtypes = [i32, vp]; L.l3d_create.restype = vp
    L.l3d_destroy.argtypes = [vp]; L.l3d_destroy.restype = None
    L.l3d_add_view.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, u32, f32, vp, u32]
    L.l3d_add_view_worldpoints.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, u32, f32, vp, u32]
    L.l3d_get_visual_neighbors.argtypes = [vp, u32, vp, u32, vp]
    L.l3d_neighbors_from_worldpoints.argtypes = [u32, vp, vp, vp, vp, vp, vp, u32, vp, vp, u64]
    L.l3d_sfm_open_colmap.argtypes = [C.c_char_p, vp]; L.l3d_sfm_open_bundler.argtypes = [C.c_char_p, vp]
    L.l3d_sfm_num_images.argtypes = [vp]; L.l3d_sfm_num_images.restype = u32
    L.l3d_sfm_get_image.argtypes = [vp, u32, vp]; L.l3d_sfm_get_worldpoints.argtypes = [vp, u32, vp, u32]
    L.l3d_sfm_close.argtypes = [vp]; L.l3d_sfm_close.restype = None
    L.l3d_nvm_open.argtypes = [C.c_char_p, vp]
    L.l3d_nvm_num_cameras.argtypes = [vp]; L.l3d_nvm_num_cameras.restype = u32
    L.l3d_nvm_get_camera.argtypes = [vp, u32, vp]
    L.l3d_nvm_get_worldpoints.argtypes = [vp, u32, vp, u32]
    L.l3d_nvm_close.argtypes = [vp]; L.l3d_nvm_close.restype = None
    L.l3d_nvm_intrinsics.argtypes = [f32, u32, u32, vp]; L.l3d_nvm_intrinsics.restype = None
    L.l3d_segment_cache_name.argtypes = [u32, u32, u32, u32, C.c_char_p, u32]
    L.l3d_read_segment_cache.argtypes = [C.c_char_p, vp, u32, vp]
    L.l3d_write_segment_cache.argtypes = [C.c_char_p, vp, u32]
    L.l3d_match_images.argtypes = [vp, C.POINTER(MatchParams)]
    L.l3d_match_begin.argtypes = [vp, C.POINTER(MatchParams)]
    L.l3d_num_pairs.argtypes = [vp, C.POINTER(u32)]
    L.l3d_get_pairs.argtypes = [vp, vp, vp, vp]
    L.l3d_match_pairs.argtypes = [vp, u32, u32]
    L.l3d_slot_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.l3d_match_finish.argtypes = [vp]
    L.l3d_match_abort.argtypes = [vp]
    L.l3d_lists_shard.argtypes = [vp, u32, u32, vp, vp, vp]
    L.l3d_lists_shard_views.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp]
    L.l3d_plan_shards.argtypes = [u32, u32, vp, vp, u32, vp, vp]
    L.l3d_compute_affinity.argtypes = [vp]
    L.l3d_synchronize.argtypes = [vp]
    L.l3d_pair_tests.argtypes = [vp, C.POINTER(u64)]
    L.l3d_get_matches.argtypes = [vp, u32, vp, u64, vp, C.POINTER(u64)]
    L.l3d_get_pair_slots.argtypes = [vp, u32, vp, u64, C.POINTER(u32), C.POINTER(u32)]
    L.l3d_num_best.argtypes = [vp, C.POINTER(u32)]
    L.l3d_get_best.argtypes = [vp, vp, vp, vp]
    L.l3d_view_info.argtypes = [vp, u32, C.POINTER(f32), C.POINTER(f32)]
    L.l3d_translation.argtypes = [vp, vp]
    L.l3d_num_affinity.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.l3d_get_affinity.argtypes = [vp, vp, vp, C.POINTER(f32)]
    L.l3d_get_sparse_matrix.argtypes = [vp, i32, vp, vp]
    L.l3d_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.l3d_set_timing_level.argtypes = [vp, C.c_int]
    L.l3d_debug_counter.argtypes = [C.c_char_p]
    L.l3d_debug_counter.restype = C.c_ulonglong
    L.l3d_tail_shard_count.argtypes = [vp, vp]
    L.l3d_tail_shard_layout.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.l3d_tail_shard_commit.argtypes = [vp]
    L.l3d_affinity_shard_begin.argtypes = [vp, u32, u32, vp, vp, vp]
    L.l3d_affinity_shard_finish.argtypes = [vp]
    L.l3d_affinity_shard_abort.argtypes = [vp]
    L.l3d_shard_options.argtypes = [vp, C.c_uint32, C.c_int]
    L.l3d_match_lines.argtypes = [i32, vp, u32, vp, u32, vp, vp, vp, vp, vp, u32, u32, f32, C.c_int32, vp,
                                  C.POINTER(u64)]
    L.l3d_set_brute_force.argtypes = [vp, i32]
    L.l3d_slots_exchanged.argtypes = [vp]
    L.l3d_slot_index_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.l3d_pack_slot_indices.argtypes = [vp, u32, u32]
    L.l3d_expand_slot_indices.argtypes = [vp, u32, u32]
    L.l3d_reconstruct_3d_lines.argtypes = [vp, u32, i32, f32, i32, u32]
    L.l3d_num_3d_lines.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.l3d_get_3d_lines.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.l3d_diffuse_affinity.argtypes = [i32, vp, u32, u32, u32, vp]
    L.l3d_output_filename.argtypes = [vp, i32, C.c_char_p, u32]
    L.l3d_save_3d_lines_txt.argtypes = [vp, C.c_char_p, i32]
    L.l3d_save_result_stl.argtypes = [vp, C.c_char_p, i32]
    L.l3d_save_3d_lines_bin.argtypes = [vp, C.c_char_p, i32]
    L.l3d_save_result_obj.argtypes = [vp, C.c_char_p, i32]
    L.l3d_get_segment_coords2d.argtypes = [vp, u32, u32, vp]
    L.l3d_find_collinear_segments.argtypes = [i32, vp, u32, f32, vp, vp, u64, vp]
    L.l3d_principal_direction.argtypes = [vp, vp]
    L.l3d_selftest_arith.argtypes = [i32, u64, u64, vp]
    L.l3d_score_matches.argtypes = [i32, vp, u32, vp, vp, vp, u32, vp, vp, f32, f32, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("l3d_last_error", "l3d_build_info", "l3d_create", "l3d_destroy"):
            fn.restype = C.c_int
    _lib = L
    return L


def last_error():
    return load().l3d_last_error().decode()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
