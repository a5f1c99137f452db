"""Randomised full-pipeline stress (run on the GPU box: python tests/stress/stress_pipeline.py [n]): matchImages + affinity
of the HIP path against the CPU oracle on random ring geometries / parameters: surviving-match sets, best
hypotheses and affinity edges must be identical, float values within 1e-4 (incl. metric regulariser, keep-all kNN,
ragged views, asymmetric neighbour lists; args: n_scenes seed).  Round 1: 114 scenes (v7 build) + 45 scenes (v8: orientation fused into the match epilogue), 0 mismatches; round 2: 120 scenes mid-round, 200 on the late
builds and 300 on the final one (sparse phase B, centre-out walk, longest-first order, overlap-only insertion), 0 mismatches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_scene
from oracle.oracle import Oracle
from tests import helpers as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 321)
bad = 0
for it in range(n):
    nv = int(rng.integers(3, 16)); ns = int(rng.integers(40, 700)); nn = int(rng.integers(2, min(nv, 12)))
    knn = int(rng.choice([1, 5, 10, 25])); epi = float(rng.choice([0.1, 0.25, 0.5])); sa = float(rng.choice([5.0, 10.0, 20.0]))
    sp = float(rng.choice([1.0, 2.5, 5.0, -0.05, -0.2])); radius = float(rng.uniform(10, 50))   # < 0: metric regulariser
    if rng.random() < 0.15: knn = 0                                                         # keep-all mode
    sc = make_scene(nv, ns, n_neighbors=nn, seed=int(rng.integers(1, 1 << 30)), radius=radius, noise_px=float(rng.uniform(0, 1.5)),
                    real_fraction=float(rng.uniform(0.3, 0.9)))
    if rng.random() < 0.5:    # ragged views and asymmetric neighbour lists
        for v in sc.views:
            v.segs = v.segs[:max(1, int(len(v.segs) * rng.uniform(0.3, 1.0)))].copy()
            if len(v.neighbors) > 1 and rng.random() < 0.5: v.neighbors = v.neighbors[:-1]
    g = Line3D(); g.add_scene(sc)
    assert g.matchImages(sigma_position=sp, sigma_angle=sa, kNN=knn, epipolar_overlap=epi) and g.computeAffinity()
    o = Oracle(threads=16); o.add_scene(sc)
    o.match_images(sigma_p=sp, sigma_a=sa, kNN=knn, epi_overlap=epi); o.compute_affinity()
    ok = True; worst = 0.0
    for v in sc.views:
        r = H.compare_matches(g.matches(v.cam)[0], o.matches(v.cam)[0])
        ok &= not r["missing"] and not r["extra"]; worst = max(worst, r["max_rel"])
    ge, gl, _ = g.affinity(); oe, ol = o.affinity()
    ok &= len(ge) == len(oe) and np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])
    if len(ge) == len(oe) and len(ge): worst = max(worst, float(np.max(np.abs(ge["w"] - oe["w"]) / oe["w"])))
    ok &= worst < 1e-4
    bad += not ok
    print(it, nv, ns, nn, knn, epi, sa, sp, "entries", g.timings()["list_entries"], "edges", len(ge), "max_rel %.2e" % worst, "OK" if ok else "MISMATCH", flush=True)
print("RESULT mismatching scenes:", bad, "of", n)
