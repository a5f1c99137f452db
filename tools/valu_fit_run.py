"""Phase A on one scene under varied parameters (epipolar-overlap threshold x kNN): the same kernel variant with very
different shares of pre-filter walk, exact tests and epilogue work.  Run twice by tools/valu_fit.sh:
  * product library under `rocprofv3 --pmc SQ_INSTS_VALU` -> VALU wave-instructions of every k_match_pairs dispatch
  * -DL3D_STATS library                                   -> the unit counts of every run (JSON on stdout)
tools/valu_fit.py fits  VALU = a*steps + b*drains + e*epilogue_passes + w*waves  over the runs: the per-unit instruction
costs behind bench.py's `roofline.useful_frac`."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd import _lib  # noqa: E402
from line3dpp_amd.api import Line3D  # noqa: E402
from line3dpp_amd.scene import make_scene  # noqa: E402

RUNS = [(thr, k) for thr in (0.05, 0.25, 0.6, 0.9) for k in (2, 10, 30)]
sc = make_scene(16, 2000, n_neighbors=10, seed=7)
g = Line3D(); g.add_scene(sc)
L = _lib.load()
stats = hasattr(L, "l3d_debug_stats")
out = (C.c_ulonglong * 16)()
rows = []
for thr, k in RUNS:
    if stats:
        L.l3d_debug_stats(out, 1)
    assert g.matchBegin(epipolar_overlap=thr, kNN=k) and g.matchPairs(0, len(g.pairs()[0]))
    tm = g.timings()
    g.matchAbort()
    row = {"epipolar_overlap": thr, "kNN": k, "match_kernel_ms": tm["match_kernel_ms"]}
    if stats:
        L.l3d_debug_stats(out, 0)
        row.update(prefilter_tests=int(out[0]), exact_tests=int(out[1]), passed_overlap=int(out[2]), accepted=int(out[3]),
                   drains=int(out[4]), band_pairs=int(out[5]), kept_slots=int(out[6]), work_items=int(out[7]),
                   stage1_drains=int(out[8]), depth_passed=int(out[9]))
    rows.append(row)
print(json.dumps({"build_info": L.l3d_build_info().decode(), "stats_build": bool(stats), "runs": rows}))
