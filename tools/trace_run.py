"""Host-side timeline of matchImages / computeAffinity on a BASELINE config (L3D_TRACE=1 prints the checkpoints of
l3d_match_images on stderr):  L3D_TRACE=1 python tools/trace_run.py [C1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
g = Line3D(); g.add_scene(make_config(sys.argv[1] if len(sys.argv) > 1 else "C1"))
for i in range(4):
    t = time.perf_counter(); assert g.matchImages(); t1 = time.perf_counter(); assert g.computeAffinity(); t2 = time.perf_counter()
    print("step", i, "matchImages %.3f ms affinity %.3f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3), g.timings(), file=sys.stderr)
