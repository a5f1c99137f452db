"""Runs only matchBegin + matchPairs (phase A) on a BASELINE config, for focused profiling:
   python tools/run_phase_a.py [C1] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = make_config(cfg)
g = Line3D(); g.add_scene(sc)
for r in range(reps):
    assert g.matchBegin() and g.matchPairs(0, len(g.pairs()[0]))
    print(cfg, "rep", r, g.timings())
