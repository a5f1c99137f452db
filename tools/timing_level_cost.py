"""What the HIP events of a call cost: wall clock per matchImages + affinity step at the three timing levels
(l3d_set_timing_level), alternating:  python tools/timing_level_cost.py [C1] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = Line3D(); g.add_scene(make_config(cfg))
for _ in range(5):
    assert g.matchImages() and g.computeAffinity()
for level in (2, 1, 0, 2, 1, 0):
    g.setTimingLevel(level)
    assert g.matchImages() and g.computeAffinity()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.matchImages(); g.computeAffinity()
    print("%s level %d: %.4f ms per step" % (cfg, level, 1e3 * (time.perf_counter() - t0) / steps), flush=True)
