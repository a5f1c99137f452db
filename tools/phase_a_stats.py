"""Candidate counters of phase A from the -DL3D_STATS build, as JSON on stdout:
   L3D_LIB=gpurun_scratch/libl3dpp_hip_stats.so python tools/phase_a_stats.py C1"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd import _lib
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
sc = make_config(cfg)
g = Line3D(); g.add_scene(sc)
L = _lib.load()
out = (C.c_ulonglong * 16)()
L.l3d_debug_stats(out, 1)
assert g.matchBegin() and g.matchPairs(0, len(g.pairs()[0]))
L.l3d_debug_stats(out, 0)
nominal = sc.pair_tests()[0]
print(json.dumps({"config": cfg, "build_info": L.l3d_build_info().decode(), "nominal_pair_tests": nominal,
                  "prefilter_tests": int(out[0]), "exact_tests": int(out[1]), "passed_overlap": int(out[2]),
                  "accepted": int(out[3]), "drains": int(out[4]), "band_pairs": int(out[5]), "kept_slots": int(out[6]),
                  "work_items": int(out[7]), "stage1_drains": int(out[8]), "depth_passed": int(out[9]),
                  "stage2_not_a_match": int(out[10]), "stage2_below_kth_best": int(out[11]), "stage1_decided_in_double": int(out[12]), "prefilter_fraction_of_nominal": out[0] / nominal,
                  "band_fraction_of_nominal": out[5] / nominal}))
