"""keep-all mode (kNN <= 0), C1 phase A: run under rocprofv3 --kernel-trace --stats for the kernel times of the mode
(tools/gpu_ab_kernels.sh style summary: `python tools/kernel_stats.py <dir>`); prints the wall time of matchPairs per call,
the slots of the padded layout and the accepted matches they hold"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
sc = make_config(sys.argv[1] if len(sys.argv) > 1 else "C1")
g = Line3D(); g.add_scene(sc)
for rep in range(3):
    assert g.matchBegin(kNN=0)
    t0 = time.perf_counter()
    assert g.matchPairs(0, len(g.pairs()[0]))
    wall = time.perf_counter() - t0
    tm = g.timings()
    print("call", rep, "match_pairs wall ms", round(wall * 1e3, 3), "kernels ms", round(tm["match_kernel_ms"], 3), "cull set-up ms",
          round(tm["cull_prepare_ms"], 3), "match_pairs_ms", round(tm["match_pairs_ms"], 3))
    if rep == 2:
        pairs, offs = g.pairs()
        n_slots = int(offs[-1]) if len(offs) > len(pairs) else None
        acc = 0
        for pi in range(0, len(pairs), max(1, len(pairs) // 16)):
            s = g.pair_slots(pi)
            acc += int((s["tgt_seg"] != 0xFFFFFFFF).sum()); 
            print("pair", pi, "slots", s.shape, "accepted", int((s["tgt_seg"] != 0xFFFFFFFF).sum()))
        print("n_slots", g.slot_buffer()[1])
    if rep == 2 and len(sys.argv) > 2 and sys.argv[2] == "finish":
        t0 = time.perf_counter()
        ok = g.matchFinish()
        print("matchFinish", ok, "wall ms", round((time.perf_counter() - t0) * 1e3, 3), {k: round(v, 3) for k, v in g.timings().items() if k.endswith("_ms")})
    else:
        g.matchAbort()
