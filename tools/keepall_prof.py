"""keep-all mode (kNN <= 0), C1 phase A, culled and unculled: run under rocprofv3 --kernel-trace --stats for the two
passes' kernel times (count = k_match_pairs<1,...>, fill = <2,...>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
sc = make_config(sys.argv[1] if len(sys.argv) > 1 else "C1")
g = Line3D(); g.add_scene(sc)
for rep in range(3):
    assert g.matchBegin(kNN=0) and g.matchPairs(0, len(g.pairs()[0]))
    print(g.timings()["match_kernel_ms"], g.timings()["cull_prepare_ms"], g.timings()["match_pairs_ms"])
    g.matchAbort()
