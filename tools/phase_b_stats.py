"""Phase-B list statistics of a BASELINE config from the -DL3D_STATS build (tools/build_stats_lib.sh):
   L3D_LIB=gpurun_scratch/libl3dpp_hip_stats.so python tools/phase_b_stats.py C1"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd import _lib
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
g = Line3D(); g.add_scene(make_config(cfg))
L = _lib.load()
out = (C.c_ulonglong * 8)()
L.l3d_debug_bstats(out, 1)
assert g.matchImages()
L.l3d_debug_bstats(out, 0)
names = ["lists", "hypotheses", "present", "present_pairs", "hyps_with_supporters", "longest_pair_sequence", "support_bits_all"]
print(cfg, {n: int(v) for n, v in zip(names, out)}, g.timings())
