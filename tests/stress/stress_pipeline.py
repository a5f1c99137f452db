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
from tests.stress.cases import pipeline_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 321)
bad = 0
for it in range(n):
    sc, kw = pipeline_case(rng)
    nv, ns, nn = len(sc.views), max(len(v.segs) for v in sc.views), max(len(v.neighbors) for v in sc.views)
    knn, epi, sa, sp = kw['kNN'], kw['epi_overlap'], kw['sigma_a'], kw['sigma_p']
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
