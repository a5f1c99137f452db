"""Verbose GPU-vs-oracle comparison on one synthetic scene (run on the GPU box):
   python tests/stress/gpu_check.py [n_views n_segs n_neighbors kNN seed]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from line3dpp_amd.scene import make_scene
from line3dpp_amd.api import Line3D
from oracle.oracle import Oracle
from tests import helpers as H

a = [int(x) for x in sys.argv[1:]]
nv, ns, nn, knn, seed = (a + [12, 500, 4, 10, 1][len(a):])[:5]
sc = make_scene(nv, ns, n_neighbors=nn, seed=seed)
g = Line3D()
g.add_scene(sc)
t = time.time(); ok = g.matchImages(kNN=knn); print("gpu matchImages", ok, time.time() - t, g.timings())
t = time.time(); ok = g.computeAffinity(); print("gpu affinity", ok, time.time() - t)
o = Oracle(threads=8); o.add_scene(sc)
t = time.time(); o.match_images(kNN=knn); print("cpu match_images", time.time() - t)
o.compute_affinity()
gp, _ = g.pairs(); op = o.pairs()
print("pairs equal", np.array_equal(gp, op), len(gp))
# phase A per pair
o2 = Oracle(threads=8); o2.add_scene(sc); o2.begin_match(kNN=knn)
tot = dict(n_gpu=0, n_cpu=0, missing=0, extra=0, order=0); worst = 0; exact = True
for pi, (s, t_) in enumerate(gp):
    om, _ = o2.match_pair(int(s), int(t_))
    r = H.compare_pair(g.pair_slots(pi), om)
    tot["n_gpu"] += r["n_gpu"]; tot["n_cpu"] += r["n_cpu"]; tot["missing"] += len(r["missing"]); tot["extra"] += len(r["extra"])
    tot["order"] += r["order_mismatch"]; worst = max(worst, r["max_rel"]); exact &= r["bit_exact"]
print("phase A", tot, "max_rel", worst, "bit_exact", exact)
# final
tm = dict(n_gpu=0, n_cpu=0, missing=0, extra=0); worst = 0
for v in sc.views:
    gm, goff = g.matches(v.cam); om, ooff = o.matches(v.cam)
    r = H.compare_matches(gm, om)
    tm["n_gpu"] += r["n_gpu"]; tm["n_cpu"] += r["n_cpu"]; tm["missing"] += len(r["missing"]); tm["extra"] += len(r["extra"])
    worst = max(worst, r["max_rel"])
    gi, oi = g.view_info(v.cam), o.view_info(v.cam)
    if gi["k"] != oi["k"] or gi["median_depth"] != oi["median_depth"]:
        print("view", v.cam, "k/median differ", gi, oi["k"], oi["median_depth"])
print("surviving matches", tm, "max_rel", worst)
s2, s3, bm = g.best(); cs, geo, ln, obm = o.best()
print("best", len(s2), len(cs), "keys equal", np.array_equal(np.stack([s2["cam"], s2["seg"]], 1), cs) if len(s2) == len(cs) else False)
if len(s2) == len(cs):
    gg = np.concatenate([s3["P1"], s3["P2"], s3["dir"]], 1)
    print("best geo max abs diff", np.abs(gg - geo).max(), "score max rel", np.max(np.abs(bm["score3D"] - obm["score3D"]) / obm["score3D"]))
ge, gl, gms = g.affinity(); oe, ol = o.affinity()
gmap = H.affinity_map(ge, np.stack([gl["cam"], gl["seg"]], 1)); omap = H.affinity_map(oe, ol)
print("affinity edges", len(ge), len(oe), "rows", len(gl), len(ol), "msdl", gms, o.med_scene_depth_lines())
print("aff keys equal", set(gmap) == set(omap), "max w diff", max([abs(gmap[k] - omap[k]) for k in gmap if k in omap] + [0]))
print("A_ identical (order+ids)", len(ge) == len(oe) and np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"]))
