"""Generates tests/golden/small_scene.npz: golden input/output vectors of the hot path on a small
deterministic scene, produced by THE REFERENCE'S OWN CODE (oracle/_ref = line3D.cc / view.cc compiled in
place from /root/reference against oracle/ref_shim; see oracle/Makefile).  /root/reference does not exist on
the GPU box, so the vectors are committed; the oracle restatement and the HIP path are both tested against
this file.  Regenerate (in the container that has /root/reference):
    python tests/golden/make_golden.py
`python tests/golden/make_golden.py --txt-excerpt` re-creates tests/golden/ref_lines3d_excerpt.txt: the first 40 records of
the reference's own result fixture testdata/Line3D++_ref/...kNN_10__vis_3.txt (Line3D::save3DLinesAsTXT format).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from line3dpp_amd.scene import make_scene  # noqa: E402

PARAMS = dict(sigma_p=2.5, sigma_a=10.0, num_neighbors=10, epi_overlap=0.25, kNN=10, const_reg_depth=-1.0)


def golden_scene():
    return make_scene(10, 300, n_neighbors=4, seed=20260925)


def flat_matches(get, cams):
    rows = []
    for c in cams:
        m, off = get(c)
        for r in m:
            rows.append((r["src_cam"], r["src_seg"], r["tgt_cam"], r["tgt_seg"], r["overlap"], r["score3D"],
                         r["d_p1"], r["d_p2"], r["d_q1"], r["d_q2"]))
    return np.array(rows, np.float64).reshape(-1, 10)


def run_oracle(scene, reference=False):
    from oracle.oracle import Oracle
    o = Oracle(threads=1, reference=reference)
    o.add_scene(scene)
    o.match_images(**PARAMS)
    o.compute_affinity()
    cams = [v.cam for v in scene.views]
    cs, geo, ln, bm = o.best()
    e, l2g = o.affinity()
    return dict(matches=flat_matches(o.matches, cams), best_keys=cs.astype(np.int64), best_geo=geo,
                edges=np.stack([e["i"].astype(np.float64), e["j"].astype(np.float64), e["w"].astype(np.float64)], 1),
                l2g=l2g.astype(np.int64),
                medians=np.array([o.view_info(c)["median_depth"] for c in cams], np.float64),
                ks=np.array([o.view_info(c)["k"] for c in cams], np.float64))


REF_TXT = ("/root/reference/testdata/Line3D++_ref/"
           "Line3D++__W_FULL__N_10__sigmaP_2.5__sigmaA_10__epiOverlap_0.25__kNN_10__vis_3.txt")


if __name__ == "__main__":
    if "--txt-excerpt" in sys.argv:
        with open(REF_TXT) as f:
            head = [next(f) for _ in range(40)]
        path = os.path.join(ROOT, "tests", "golden", "ref_lines3d_excerpt.txt")
        open(path, "w").write("".join(head))
        print("wrote", path)
        # the `v` records of the same 3D segments in the reference's .obj result (two per segment, same order)
        n_seg = sum(int(l.split()[0]) for l in head)
        with open(REF_TXT[:-4] + ".obj") as f:
            v = [next(f) for _ in range(2 * n_seg)]
        path = os.path.join(ROOT, "tests", "golden", "ref_lines3d_excerpt.obj")
        open(path, "w").write("".join(v))
        print("wrote", path)
        sys.exit(0)
    from oracle.oracle import have_reference
    assert have_reference(), "oracle/_ref is not built: run `make -C oracle` where /root/reference exists"
    out = run_oracle(golden_scene(), reference=True)
    out["generator"] = np.array("reference line3D.cc/view.cc via oracle/_ref (shim headers)")
    path = os.path.join(ROOT, "tests", "golden", "small_scene.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
