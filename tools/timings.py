"""l3d_timings of one matchImages + affinity on a BASELINE config (after a warm-up call): python tools/timings.py C1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
for cfg in sys.argv[1:] or ["C1"]:
    g = Line3D(); g.add_scene(make_config(cfg))
    for _ in range(3):
        assert g.matchImages() and g.computeAffinity()
    print(cfg, g.timings())
