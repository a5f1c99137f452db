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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 123)
bad = 0; total = 0; culled = 0; npairs = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    nv = int(rng.integers(3, 14)); ns = int(rng.integers(50, 2500)); nn = int(rng.integers(2, min(nv, 8)))
    radius = float(rng.uniform(8, 60)); knn = int(rng.choice([1, 3, 10, 20])); epi = float(rng.choice([0.1, 0.25, 0.5, 0.8]))
    sc = make_scene(nv, ns, n_neighbors=nn, seed=int(rng.integers(1, 1 << 30)), radius=radius, noise_px=float(rng.uniform(0, 2)))
    if rng.random() < 0.3:   # anisotropic rescale of the image
        sx, sy = rng.uniform(0.5, 2.0, 2)
        for v in sc.views:
            v.segs = (v.segs * np.array([sx, sy, sx, sy])).astype(np.float32); v.K = v.K.copy(); v.K[0] *= sx; v.K[1] *= sy
            v.width = int(v.width * sx); v.height = int(v.height * sy)
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
