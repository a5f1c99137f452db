"""Randomised stress (run on the GPU box: python tests/stress/stress_culling.py): phase A with epipolar-band culling + fp32
pre-filter against the brute-force path (every pair through the exact test) on random ring geometries, image
scalings, kNN and overlap thresholds (args: n_scenes seed).  Round 1: 180 scenes, 2615 directed pairs, 17.5 M matches,
0 differences; round 2: 300 scenes on the late builds and 300 on the final one (4574 directed pairs, 29.6 M matches), 0 differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_scene
from line3dpp_amd._lib import EMPTY
from tests.stress.cases import culling_case
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 123)
bad = 0; total = 0; culled = 0; npairs = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    sc, knn, epi = culling_case(rng)
    nv, ns, nn, radius = len(sc.views), len(sc.views[0].segs), len(sc.views[0].neighbors), 0.0
    out = []
    for brute in (0, 1):
        g = Line3D(); g.add_scene(sc); g.set_brute_force(brute)
        assert g.matchBegin(kNN=knn, epipolar_overlap=epi) and g.matchPairs(0, len(g.pairs()[0]))
        if not brute: culled += g.timings()["culled_pairs"]; npairs += len(g.pairs()[0])
        out.append([g.pair_slots(pi) for pi in range(len(g.pairs()[0]))])
    for a, b in zip(*out):
        total += int((b["tgt_seg"] != EMPTY).sum())
        if not np.array_equal(a, b): bad += 1
    print(it, nv, ns, nn, round(radius, 1), knn, epi, "pairs", len(out[0]), "bad so far", bad, flush=True)
print("RESULT bad pairs:", bad, "matches compared:", total, "culled pairs", culled, "of", npairs)
