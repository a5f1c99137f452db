"""GPU parity tests (pytest -m gpu, real MI355X): the HIP path behind the C-ABI against the CPU oracle.

Bar (north_star): affinity entries and 3D end points within 1e-4 relative.  What is actually asserted
is stronger wherever the arithmetic allows it: phase-A matches (sets, overlap, four depths) are
bit-exact, sets of surviving matches / best hypotheses / affinity edges are identical, and float
values that pass through expf/acos agree to REL_TOL.
"""
import os

import numpy as np
import pytest

from line3dpp_amd._lib import EMPTY
from line3dpp_amd.scene import ViewData, Scene, make_scene
from tests import helpers as H

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "small_scene.npz")


def _gpu(scene, **kw):
    from line3dpp_amd.api import Line3D
    g = Line3D()
    g.add_scene(scene)
    assert g.numImages() == scene.n_views
    return g


def _oracle(scene, **kw):
    from oracle.oracle import Oracle
    o = Oracle(threads=kw.pop("threads", 8), **kw)
    o.add_scene(scene)
    return o


def _check_phase_a(g, scene, kNN, epi=0.25, allow_tie_order=True):
    o = _oracle(scene)
    o.begin_match(kNN=kNN, epi_overlap=epi)
    pairs, _ = g.pairs()
    assert np.array_equal(pairs, np.array(scene.pair_tests()[1], np.uint32).reshape(-1, 2))
    total = 0
    for pi, (s, t) in enumerate(pairs):
        om, _ = o.match_pair(int(s), int(t))
        r = H.compare_pair(g.pair_slots(pi), om)
        assert not r["missing"] and not r["extra"], (pi, r["missing"][:3], r["extra"][:3])
        assert r["bit_exact"], f"pair {pi}: overlap/depths differ from the oracle (max rel {r['max_rel']})"
        assert r["order_mismatch"] == 0, f"pair {pi}: rows ordered differently from the reference's priority_queue"
        total += r["n_cpu"]
    o.end_match()
    return total


@pytest.mark.parametrize("n_views,n_segs,nn,kNN,seed", [
    (6, 300, 4, 10, 1),
    (5, 257, 2, 1, 2),      # ragged: one lane past a full 256 block
    (4, 63, 2, 3, 3),       # less than one wave
    (4, 1000, 2, 25, 4),    # K larger than the default
    (3, 1, 2, 10, 5),       # single segment per view
])
def test_phase_a_bit_exact(n_views, n_segs, nn, kNN, seed):
    sc = make_scene(n_views, n_segs, n_neighbors=nn, seed=seed)
    g = _gpu(sc)
    assert g.matchBegin(kNN=kNN) and g.matchPairs(0, len(g.pairs()[0]))
    total = _check_phase_a(g, sc, kNN)
    assert total > 0 or n_segs == 1
    assert g.matchFinish()


@pytest.mark.parametrize("kNN", [10, 3, 1])
def test_equal_overlaps_follow_the_reference_heap_order(kNN):
    """Every segment occurs twice in its view, so nearly every source row sees pairs of EQUAL overlaps -- inside its
    top-kNN and at the kNN-th place.  The reference resolves them by the pop order of its std::priority_queue
    (line3D.cc:982-1007, commons.h:217-231); the HIP path replays exactly that (k_match_tied_rows, l3d_heap.h): same
    slots in the same order, and the whole pipeline stays identical."""
    sc = make_scene(5, 300, n_neighbors=4, seed=41)
    for v in sc.views:
        v.segs[1::2] = v.segs[0::2]
    g = _gpu(sc)
    assert g.matchBegin(kNN=kNN) and g.matchPairs(0, len(g.pairs()[0]))
    o = _oracle(sc, threads=2)
    o.begin_match(kNN=kNN)
    pairs, _ = g.pairs()
    total = 0
    for pi, (s, t) in enumerate(pairs):
        om, _ = o.match_pair(int(s), int(t))
        r = H.compare_pair_fast(g.pair_slots(pi), om)
        assert r["set_diff"] == 0 and r["order_rows"] == 0 and r["inexact_fields"] == 0, (pi, r)
        total += r["n_cpu"]
    o.end_match()
    assert total > 1000
    assert g.matchFinish() and g.computeAffinity()
    assert g.timings()["tied_rows"] > 200
    o1 = _oracle(sc, threads=1); o1.match_images(kNN=kNN); o1.compute_affinity()
    _compare_final(g, o1, sc)


def test_phase_a_views_of_different_size_and_overlap_threshold():
    sc = make_scene(5, 400, n_neighbors=4, seed=7)
    for i, v in enumerate(sc.views):      # ragged views: 400, 333, 266, ...
        v.segs = v.segs[:400 - 67 * i].copy()
    g = _gpu(sc)
    assert g.matchBegin(kNN=7, epipolar_overlap=0.6) and g.matchPairs(0, len(g.pairs()[0]))
    _check_phase_a(g, sc, 7, epi=0.6)


def test_prefilter_never_loses_a_match():
    """fp32 pre-filter + ring compaction vs the brute-force path (every pair through the exact test),
    at a size the CPU oracle would need minutes for; also at 4x image scale (larger coordinates)."""
    for scale, seed in ((1.0, 21), (4.0, 22)):
        sc = make_scene(4, 3000, n_neighbors=2, seed=seed)
        if scale != 1.0:
            for v in sc.views:
                v.segs = (v.segs * scale).astype(np.float32); v.K = v.K.copy(); v.K[:2] *= scale
                v.width = int(v.width * scale); v.height = int(v.height * scale)
        out = []
        for brute in (0, 1):
            g = _gpu(sc)
            g.set_brute_force(brute)
            assert g.matchBegin(kNN=10) and g.matchPairs(0, len(g.pairs()[0]))
            out.append([g.pair_slots(pi) for pi in range(len(g.pairs()[0]))])
        for a, b in zip(*out):
            assert np.array_equal(a, b)
        assert sum(int((a["tgt_seg"] != EMPTY).sum()) for a in out[0]) > 1000


def _forward_motion_scene(n_segs=600, seed=31):
    """Two cameras on (almost) the same optical axis: the epipoles lie inside the images, the geometry the
    epipolar-band culling must refuse (make_cull, l3d_api.hip) -- plus a third, sideways camera."""
    rng = np.random.default_rng(seed)
    w, h, f = 1600, 1200, 1400.0
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1.0]])
    P = rng.uniform([-6, -4, 14], [6, 4, 22], (4 * n_segs, 3))
    Q = P + rng.normal(0, 0.8, P.shape)
    views = []
    for cam, C in enumerate([np.array([0.0, 0.0, 0.0]), np.array([0.15, -0.1, 2.0]), np.array([2.5, 0.2, 0.3])]):
        R = np.eye(3); t = -R @ C
        a = (K @ (P - C).T).T; b = (K @ (Q - C).T).T
        a = a[:, :2] / a[:, 2:]; b = b[:, :2] / b[:, 2:]
        ok = ((a >= 1).all(1) & (b >= 1).all(1) & (a[:, 0] < w - 1) & (b[:, 0] < w - 1) & (a[:, 1] < h - 1) &
              (b[:, 1] < h - 1) & (np.hypot(*(a - b).T) > 8))
        segs = np.concatenate([a, b], 1)[ok][:n_segs] + rng.normal(0, 0.3, (min(n_segs, int(ok.sum())), 4))
        views.append(ViewData(cam, segs.astype(np.float32), K.copy(), R, t, w, h, 18.0, [c for c in range(3) if c != cam]))
    return Scene(views, "forward-motion")


def test_epipolar_culling_active_and_refused_where_unsound():
    """Ring scenes match with epipolar-band culling on every pair; a forward-motion pair (epipole inside the
    image) must fall back to plain streaming.  Both are bit-exact against the oracle."""
    sc = make_scene(16, 400, n_neighbors=4, seed=41)
    g = _gpu(sc)
    assert g.matchBegin(kNN=10) and g.matchPairs(0, len(g.pairs()[0]))
    assert g.timings()["culled_pairs"] > len(g.pairs()[0]) // 2
    assert _check_phase_a(g, sc, 10) > 1000
    fm = _forward_motion_scene()
    g = _gpu(fm)
    assert g.matchBegin(kNN=10) and g.matchPairs(0, len(g.pairs()[0]))
    n_culled = g.timings()["culled_pairs"]
    assert n_culled < len(g.pairs()[0])        # pair (0,1): epipole in the image -> refused
    assert _check_phase_a(g, fm, 10) > 500
    assert g.matchFinish()


def test_keep_all_mode_knn_zero():
    sc = make_scene(4, 260, n_neighbors=2, seed=8)
    g = _gpu(sc)
    assert g.matchBegin(kNN=0) and g.matchPairs(0, len(g.pairs()[0]))
    o = _oracle(sc); o.begin_match(kNN=0)
    for pi, (s, t) in enumerate(g.pairs()[0]):
        om, off = o.match_pair(int(s), int(t))
        sl = g.pair_slots(pi)
        r = H.compare_pair(sl, om)
        assert not r["missing"] and not r["extra"] and r["bit_exact"] and r["order_mismatch"] == 0
    o.end_match()
    assert g.matchFinish() and g.computeAffinity()
    o2 = _oracle(sc); o2.match_images(kNN=0); o2.compute_affinity()
    _compare_final(g, o2, sc)


def test_keep_all_single_pass_against_the_two_pass_form(monkeypatch):
    """kNN <= 0 since round 6: ONE culled pass that keeps what it accepts (row scratch) + k_keep_assemble into RAGGED rows
    (k_match.hip).  The form of rounds 3-5 (count pass, rows sized by the host, streamed fill pass: L3D_KEEPALL_TWO_PASS=1) must
    give the same slots -- the accessor hands both out in the padded Ms x K form -- and the same final result; a row scratch that
    is too small (L3D_KEEPALL_CAP=2: first size of a context) repeats the pass with a larger one."""
    from line3dpp_amd import _lib
    sc = make_scene(5, 300, n_neighbors=3, seed=41)
    L = _lib.load()
    runs = {}
    for name, env in (("two_pass", {"L3D_KEEPALL_TWO_PASS": "1"}), ("single", {}), ("single_small_scratch", {"L3D_KEEPALL_CAP": "2"}),
                      ("single_brute_force", {})):
        for k in ("L3D_KEEPALL_TWO_PASS", "L3D_KEEPALL_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        before = L.l3d_debug_counter(b"keep_all_repeats")
        g = _gpu(sc)
        if name == "single_brute_force":           # (every pair through the exact test, no culling, no pre-filter: same rows)
            g.set_brute_force(1)
        assert g.matchBegin(kNN=0) and g.matchPairs(0, len(g.pairs()[0]))
        slots = [g.pair_slots(pi) for pi in range(len(g.pairs()[0]))]
        repeats = L.l3d_debug_counter(b"keep_all_repeats") - before
        assert g.matchFinish() and g.computeAffinity()
        runs[name] = (g, slots, repeats)
    assert runs["two_pass"][2] == 0 and (runs["single_small_scratch"][2] & 0xFFFF) >= 1
    ref = runs["two_pass"][1]
    n = 0
    for name in ("single", "single_small_scratch", "single_brute_force"):
        for a, b in zip(ref, runs[name][1]):
            assert a.shape == b.shape
            for f in ("tgt_seg", "overlap", "d_p1", "d_p2", "d_q1", "d_q2"):
                assert np.array_equal(a[f], b[f]), (name, f)
            n += int((a["tgt_seg"] != 0xFFFFFFFF).sum())
    assert n > 10_000
    g0 = runs["two_pass"][0]
    for name in ("single", "single_small_scratch", "single_brute_force"):
        g1 = runs[name][0]
        for v in sc.views:
            m0, o0 = g0.matches(v.cam); m1, o1 = g1.matches(v.cam)
            assert np.array_equal(o0, o1) and m0.tobytes() == m1.tobytes(), (name, v.cam)
        e0, l0, w0 = g0.affinity(); e1, l1, w1 = g1.affinity()
        assert np.array_equal(e0, e1) and np.array_equal(l0, l1) and np.array_equal(w0, w1)
    o = _oracle(sc); o.match_images(kNN=0); o.compute_affinity()
    _compare_final(runs["single"][0], o, sc)
    # tiny views at a strict overlap threshold: (almost) no accepted match -- an empty or near-empty ragged buffer, rows without
    # slots, pairs without slots; the call still succeeds and agrees with the oracle
    st = make_scene(4, 3, n_neighbors=3, seed=5)
    g = _gpu(st)
    assert g.matchBegin(kNN=0, epipolar_overlap=0.99) and g.matchPairs(0, len(g.pairs()[0]))
    o = _oracle(st); o.begin_match(kNN=0, epi_overlap=0.99)
    n_ref = 0
    for pi, (s_, t_) in enumerate(g.pairs()[0]):
        om, off = o.match_pair(int(s_), int(t_))
        sl = g.pair_slots(pi)
        r = H.compare_pair(sl, om)
        assert not r["missing"] and not r["extra"] and r["order_mismatch"] == 0
        n_ref += len(om)
    o.end_match()
    assert g.slot_buffer()[1] == n_ref
    assert g.matchFinish() and g.computeAffinity()


def _compare_final(g, o, sc, exact_sets=True):
    n_surv = 0
    for v in sc.views:
        gm, goff = g.matches(v.cam); om, ooff = o.matches(v.cam)
        r = H.compare_matches(gm, om)
        assert not r["missing"] and not r["extra"], (v.cam, r["missing"][:3], r["extra"][:3])
        assert r["max_rel"] <= H.REL_TOL
        assert np.array_equal(goff, ooff)
        # same order inside every list (canonical order == reference single-thread order)
        assert np.array_equal(gm["tgt_cam"], om["tgt_cam"]) and np.array_equal(gm["tgt_seg"], om["tgt_seg"])
        gi, oi = g.view_info(v.cam), o.view_info(v.cam)
        assert gi["k"] == oi["k"]
        assert H.rel_close(gi["median_depth"], oi["median_depth"])
        n_surv += len(om)
    s2, s3, bm = g.best(); cs, geo, ln, obm = o.best()
    assert np.array_equal(np.stack([s2["cam"], s2["seg"]], 1), cs)
    gg = np.concatenate([s3["P1"], s3["P2"], s3["dir"]], 1)
    assert np.all(H.rel_close(gg, geo)), "3D end points of the best hypotheses"
    assert np.all(H.rel_close(bm["score3D"], obm["score3D"]))
    assert np.array_equal(bm["tgt_cam"], obm["tgt_cam"]) and np.array_equal(bm["tgt_seg"], obm["tgt_seg"])
    ge, gl, gms = g.affinity(); oe, ol = o.affinity()
    assert H.rel_close(gms, o.med_scene_depth_lines())
    gmap = H.affinity_map(ge, np.stack([gl["cam"], gl["seg"]], 1)); omap = H.affinity_map(oe, ol)
    assert set(gmap) == set(omap)
    assert all(H.rel_close(gmap[k], omap[k]) for k in omap)
    # row ids and edge order are the reference's single-thread first-touch order
    assert np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])
    assert np.array_equal(np.stack([gl["cam"], gl["seg"]], 1), ol)
    return n_surv, len(cs), len(oe)


@pytest.mark.parametrize("n_views,n_segs,nn,seed", [(8, 300, 4, 1), (12, 500, 6, 2), (7, 129, 2, 3)])
def test_full_pipeline_parity(n_views, n_segs, nn, seed):
    sc = make_scene(n_views, n_segs, n_neighbors=nn, seed=seed)
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    o = _oracle(sc); o.match_images(); o.compute_affinity()
    n_surv, n_best, n_edges = _compare_final(g, o, sc)
    assert n_surv > 5 and n_best > 2 and (n_edges > 2 or n_segs < 200)
    assert g.pair_tests() == o.pair_tests() == sc.pair_tests()[0]


def test_scored_lists_before_filter_match_oracle():
    """score3D of every hypothesis (fresh + inverse) right after scoring, via the slot score write-back
    of fresh matches: compares with the oracle's pre-filter snapshot."""
    sc = make_scene(6, 250, n_neighbors=4, seed=13)
    g = _gpu(sc)
    assert g.matchImages()
    o = _oracle(sc, record_scored=True); o.match_images()
    pairs, _ = g.pairs()
    checked = 0
    for pi, (s, t) in enumerate(pairs):
        sl = g.pair_slots(pi)
        sm, soff = o.scored(int(s))
        od = {(int(x["src_seg"]), int(x["tgt_seg"])): float(x["score3D"]) for x in sm if x["tgt_cam"] == t}
        for r in range(sl.shape[0]):
            for x in sl[r]:
                if x["tgt_seg"] == EMPTY:
                    continue
                key = (r, int(x["tgt_seg"]))
                alive = bool(x["flags"] & 1)
                assert alive == (key in od), "orientation filter decision"
                if alive:
                    assert H.rel_close(x["score3D"], od[key]) or abs(x["score3D"] - od[key]) < 1e-6
                    checked += 1
    assert checked > 1000


def test_fixed_regularizer_mode_sigma_in_metres():
    sc = make_scene(6, 200, n_neighbors=4, seed=17)
    g = _gpu(sc)
    assert g.matchImages(sigma_position=-0.05) and g.computeAffinity()
    o = _oracle(sc); o.match_images(sigma_p=-0.05); o.compute_affinity()
    _compare_final(g, o, sc)


def test_asymmetric_neighbours_and_sparse_cam_ids():
    """neighbour lists need not be symmetric and camIDs need not be dense (line3D.cc:704-741 direction rule,
    inverse matches only towards unprocessed views)."""
    sc = make_scene(7, 220, n_neighbors=4, seed=19)
    remap = {i: 10 + 7 * i for i in range(7)}
    for v in sc.views:
        v.neighbors = [remap[n] for n in v.neighbors if (v.cam + n) % 3 != 0] or [remap[(v.cam + 1) % 7]]
        v.cam = remap[v.cam]
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    o = _oracle(sc); o.match_images(); o.compute_affinity()
    assert np.array_equal(g.pairs()[0], o.pairs())
    _compare_final(g, o, sc)


def test_golden_fixture_gpu():
    from tests.golden.make_golden import PARAMS, flat_matches, golden_scene
    gold = np.load(GOLDEN)
    sc = golden_scene()
    g = _gpu(sc)
    assert g.matchImages(sigma_position=PARAMS["sigma_p"], sigma_angle=PARAMS["sigma_a"], kNN=PARAMS["kNN"],
                         epipolar_overlap=PARAMS["epi_overlap"]) and g.computeAffinity()
    m = flat_matches(g.matches, [v.cam for v in sc.views])
    assert m.shape == gold["matches"].shape
    assert np.array_equal(m[:, :4], gold["matches"][:, :4])
    assert np.all(H.rel_close(m[:, 4:], gold["matches"][:, 4:]))
    s2, s3, bm = g.best()
    assert np.array_equal(np.stack([s2["cam"], s2["seg"]], 1), gold["best_keys"])
    assert np.all(H.rel_close(np.concatenate([s3["P1"], s3["P2"], s3["dir"]], 1), gold["best_geo"]))
    ge, gl, _ = g.affinity()
    assert np.array_equal(np.stack([ge["i"], ge["j"]], 1), gold["edges"][:, :2].astype(np.int64))
    assert np.all(H.rel_close(ge["w"], gold["edges"][:, 2]))
    assert np.array_equal(np.stack([gl["cam"], gl["seg"]], 1), gold["l2g"])


def test_seam_level_match_lines():
    """l3d_match_lines (replaces match_lines_GPU, cudawrapper.h:54-63) against matchingCPU of the oracle."""
    from line3dpp_amd.api import match_lines
    sc = make_scene(3, 350, n_neighbors=2, seed=23)
    o = _oracle(sc); o.begin_match(kNN=6)
    F = o.fundamental(0, 1)
    vs, vt = sc.views[0], sc.views[1]
    info0, info1 = o.view_info(0), o.view_info(1)
    A0 = vs.R.T @ np.linalg.inv(vs.K); A1 = vt.R.T @ np.linalg.inv(vt.K)
    slots, n = match_lines(vs.segs, vt.segs, F, A0, A1, info0["C"], info1["C"], vs.width, vs.height, 0.25, 6)
    om, _ = o.match_pair(0, 1)
    o.end_match()
    r = H.compare_pair(slots, om)
    assert n == len(om) and not r["missing"] and not r["extra"]
    assert r["max_rel"] < 1e-5      # RtKinv passed in comes from numpy's inverse, not the cofactor formula


def test_error_behaviour_mirrors_reference():
    from line3dpp_amd.api import Line3D
    sc = make_scene(3, 40, n_neighbors=2, seed=2)
    v = sc.views[0]
    g = Line3D()
    assert not g.matchImages() and g.last_status == -6                 # no images (line3D.cc:385)
    g.addImage(0, (700, 500), v.K, v.R, v.t, 1.0, [1], v.segs); assert g.last_status == -2   # too small (:119)
    g.addImage(0, (v.width, v.height), v.K, v.R, v.t, 1.0, [], v.segs); assert g.last_status == -4  # (:154)
    g.addImage(0, (v.width, v.height), v.K, v.R, v.t, 1.0, [1], v.segs); assert g.last_status == 0
    g.addImage(0, (v.width, v.height), v.K, v.R, v.t, 1.0, [1], v.segs); assert g.last_status == -3  # ID in use (:130)
    assert g.numImages() == 1
    assert not g.computeAffinity() and g.last_status == -7             # reconstruct before match (:1712)
    # a view whose neighbours do not exist is matched with nobody and yields nothing
    assert g.matchImages()
    m, off = g.matches(0)
    assert len(m) == 0 and g.best()[0].size == 0


def test_determinism_and_idempotence():
    sc = make_scene(8, 400, n_neighbors=4, seed=29)
    g = _gpu(sc)
    runs = []
    for _ in range(2):
        assert g.matchImages() and g.computeAffinity()
        runs.append(([g.matches(v.cam)[0].tobytes() for v in sc.views], g.best()[1].tobytes(), g.affinity()[0].tobytes()))
    assert runs[0] == runs[1], "a second matchImages on the same context reproduces the first bit for bit"
    g2 = _gpu(sc)
    assert g2.matchImages() and g2.computeAffinity()
    assert runs[0][2] == g2.affinity()[0].tobytes()


def test_sparse_matrix_export():
    sc = make_scene(8, 300, n_neighbors=4, seed=31)
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    e, l2g, _ = g.affinity()
    for by_row in (False, True):
        ent, start = g.sparse_matrix(sort_by_row=by_row)
        assert len(ent) == len(e) and len(start) == len(l2g) and np.all(ent["w"] == 0)
        key = ent["x"] if by_row else ent["y"]
        assert np.all(np.diff(key) >= 0)
        for rc in range(len(l2g)):
            pos = np.nonzero(key == rc)[0]
            assert start[rc] == (pos[0] if len(pos) else -1)
        got = sorted(zip(ent["x"].astype(int), ent["y"].astype(int), ent["z"]))
        assert got == sorted(zip(e["i"], e["j"], e["w"]))


def test_properties_at_full_c1_size():
    """BASELINE config C1 (64 views x 2000 segments, 10 neighbours) -- size-independent properties."""
    from line3dpp_amd.scene import make_config
    sc = make_config("C1")
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    assert g.pair_tests() == 1_280_000_000
    pairs, _ = g.pairs()
    assert len(pairs) == 320
    rng = np.random.default_rng(0)
    for pi in rng.choice(len(pairs), 12, replace=False):
        sl = g.pair_slots(int(pi))
        valid = sl["tgt_seg"] != EMPTY
        assert np.all(valid[:, :-1] >= valid[:, 1:]), "valid slots form a prefix of every row"
        ov = np.where(valid, sl["overlap"], -1.0)
        assert np.all(ov[:, :-1] >= ov[:, 1:]), "rows sorted by overlap, best first"
        assert np.all(sl["overlap"][valid] > 0.25) and np.all(sl["overlap"][valid] <= 1.0 + 1e-6)
        for f in ("d_p1", "d_p2", "d_q1", "d_q2"):
            assert np.all(sl[f][valid] > 1e-12)
        for r in range(0, sl.shape[0], 97):
            t = sl["tgt_seg"][r][valid[r]]
            assert len(set(t.tolist())) == len(t)
    e, l2g, msdl = g.affinity()
    amap = H.affinity_map(e, np.stack([l2g["cam"], l2g["seg"]], 1))
    assert len(amap) * 2 == len(e) and min(amap.values()) > 0.5 and max(amap.values()) <= 1.0
    s2, s3, bm = g.best()
    assert np.all(bm["score3D"] > 0.75)
    keys = s2["cam"].astype(np.int64) * (1 << 32) + s2["seg"]
    assert np.all(np.diff(keys) > 0), "best hypotheses ordered by (camID, segID), at most one per segment"
    hyp = set(map(tuple, np.stack([s2["cam"], s2["seg"]], 1).tolist()))
    assert set(map(tuple, np.stack([l2g["cam"], l2g["seg"]], 1).tolist())) <= hyp
    lim_ok = 0
    for v in sc.views[::8]:
        m, off = g.matches(v.cam)
        if len(m):
            assert np.all(m["score3D"] > 0)
            lim_ok += 1
    assert lim_ok > 0


@pytest.mark.parametrize("mode", ["neighbors", "worldpoints"])
def test_cpp_facade_matches_python_front_end(tmp_path, mode):
    """include/line3dpp/line3D.h (C++ mirror of L3DPP::Line3D) driven like a reference main_*.cpp -- with explicit
    neighbour lists (main_mavmap.cpp) and with worldpoint lists on an instance constructed with
    neighbors_by_worldpoints=true (main_vsfm.cpp)."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "facade_smoke")
    lib_dir = os.path.join(root, "line3dpp_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "facade_smoke.cpp"), "-o", exe, "-L" + lib_dir,
                           "-ll3dpp_hip", "-Wl,-rpath," + lib_dir])
    sc = make_scene(8, 300, n_neighbors=4, seed=1)
    wps = mode == "worldpoints"
    if wps:
        from line3dpp_amd.scene import add_worldpoints
        add_worldpoints(sc, n_points=1500, seed=4, keep=0.5)
    path = str(tmp_path / "scene.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", sc.n_views))
        for v in sc.views:
            lst = v.worldpoints if wps else v.neighbors
            f.write(struct.pack("<5I", v.cam, len(v.segs), v.width, v.height, len(lst)))
            f.write(np.ascontiguousarray(v.K, np.float64).tobytes()); f.write(np.ascontiguousarray(v.R, np.float64).tobytes())
            f.write(np.ascontiguousarray(v.t, np.float64).tobytes()); f.write(struct.pack("<f", v.median_depth))
            f.write(np.asarray(lst, np.uint32).tobytes()); f.write(np.ascontiguousarray(v.segs, np.float32).tobytes())
    out = subprocess.check_output([exe, path, mode]).decode()
    line = [l for l in out.splitlines() if l.startswith("RESULT")][0]
    kv = dict(x.split("=") for x in line.split()[1:])
    from line3dpp_amd.api import Line3D
    g = Line3D(neighbors_by_worldpoints=wps)
    g.add_scene(sc)
    assert g.matchImages() and g.computeAffinity()
    assert sum(len(m) for m in [g.matches(v.cam)[0] for v in sc.views]) > 0
    ms = [g.matches(v.cam)[0] for v in sc.views]
    assert int(kv["images"]) == 8 and int(kv["matches"]) == sum(len(m) for m in ms)
    assert abs(float(kv["score_sum"]) - sum(float(m["score3D"].astype(np.float64).sum()) for m in ms)) < 1e-3
    e, l2g, _ = g.affinity()
    assert int(kv["hypotheses"]) == len(g.best()[0]) and int(kv["edges"]) == len(e) and int(kv["rows"]) == len(l2g)
    assert abs(float(kv["wsum"]) - float(e["w"].astype(np.float64).sum())) < 1e-3
    # FinalLine3D through the reference's accessors (underlyingCluster_.seg3D() / .residuals() / .reference_view(),
    # segment3D.h:120-178) against the Python front end's lines
    ll = dict(x.split("=") for x in [l for l in out.splitlines() if l.startswith("LINES")][0].split()[1:])
    assert g.reconstruct3Dlines(3)
    lines = g.get3Dlines()
    assert int(ll["lines"]) == len(lines) and int(ll["segments"]) == sum(len(L["collinear3Dsegments"]) for L in lines)
    assert int(ll["residuals"]) == sum(len(L["residuals"]) for L in lines)
    assert int(ll["refsum"]) == sum(int(L["reference_view"]) for L in lines)
    len2 = sum(float(((np.asarray(L["cluster_line"]["P1"], np.float64) - np.asarray(L["cluster_line"]["P2"], np.float64)) ** 2).sum()) for L in lines)
    assert abs(float(ll["len2"]) - len2) < 1e-3 * max(len2, 1.0)


def test_full_pipeline_vs_reference_own_code():
    """HIP path against oracle/_ref = the reference's own line3D.cc/view.cc compiled in place (prebuilt library
    shipped to the GPU box; skipped if it is absent)."""
    from oracle import oracle as O
    assert O.have_reference(), "oracle/_ref is missing: no downgrade of the checker on the GPU box"
    for (nv, ns, nn, seed, params) in ((10, 400, 6, 41, {}), (7, 260, 4, 42, dict(kNN=0)),
                                       (8, 300, 4, 43, dict(sigma_p=-0.05))):
        sc = make_scene(nv, ns, n_neighbors=nn, seed=seed)
        g = _gpu(sc)
        kw = dict(params)
        if "sigma_p" in kw:
            kw["sigma_position"] = kw.pop("sigma_p")
        assert g.matchImages(**kw) and g.computeAffinity()
        r = O.Oracle(threads=1, reference=True)
        r.add_scene(sc); r.match_images(**params); r.compute_affinity()
        _compare_final(g, r, sc)


def test_large_views_tile_loop_parity():
    """One pair of 16k-segment views (BASELINE C4 size): many LDS tiles per pass, top-K threshold feedback at
    high candidate counts; bit-exact against matchingCPU of the oracle (8 OpenMP threads, ~2.7e8 pair tests)."""
    sc = make_scene(2, 16384, n_neighbors=2, seed=51)
    for v in sc.views:
        v.neighbors = [1 - v.cam]
    g = _gpu(sc)
    assert g.matchBegin() and g.matchPairs(0, 1)
    o = _oracle(sc); o.begin_match()
    om, _ = o.match_pair(0, 1)
    o.end_match()
    r = H.compare_pair(g.pair_slots(0), om)
    assert not r["missing"] and not r["extra"] and r["bit_exact"] and r["n_cpu"] > 100000


@pytest.mark.parametrize("n_segs", [5000, 9000, 16385, 20000])
def test_width_classes_of_large_views_vs_brute_force(n_segs):
    """From 4096 segments per view on, k_cull_prepare groups source rows with wide epipolar bands apart (one class,
    two from 8192 on).  The row order must not change the result: culled + pre-filtered path against the brute-force
    path (every pair through the exact test) of the same library, slot for slot.  16 385 and 20 000 segments: one more
    than the LDS sort capacity of k_cull_prepare and well beyond it -- the keys then sort in global memory and the pair
    is still culled (round 1 matched such views unculled, three times slower)."""
    sc = make_scene(16, n_segs, n_neighbors=2, seed=57, max_views=3)     # neighbouring views of a 16-view ring: epipoles far outside
    sc.views = sc.views[:3]
    for v in sc.views:
        v.neighbors = [c for c in (0, 1, 2) if c != v.cam]
    out = []
    for brute in (0, 1):
        g = _gpu(sc); g.set_brute_force(brute)
        assert g.matchBegin(kNN=10) and g.matchPairs(0, len(g.pairs()[0]))
        if not brute:
            assert g.timings()["culled_pairs"] == len(g.pairs()[0])
        out.append([g.pair_slots(pi) for pi in range(len(g.pairs()[0]))])
    n = 0
    for a, b in zip(*out):
        assert np.array_equal(a, b)
        n += int((b["tgt_seg"] != EMPTY).sum())
    assert n > 10 * n_segs


def test_many_views_chain_properties():
    """BASELINE C3-like: a long chain (256 views x 250 segments, two rings): full parity with the oracle."""
    sc = make_scene(256, 250, n_neighbors=6, seed=53, rings=2, radius=22.0)
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    o = _oracle(sc); o.match_images(); o.compute_affinity()
    n_surv, n_best, n_edges = _compare_final(g, o, sc)
    assert n_best > 1000


def test_device_tensor_wraps_slot_buffer():
    """line3dpp_amd.dist.device_tensor: the slot buffer as a torch tensor without a copy (what the RCCL
    all-gather of the N>1 path operates on)."""
    import torch
    from line3dpp_amd import dist
    from line3dpp_amd._lib import SLOT_DTYPE
    sc = make_scene(4, 200, n_neighbors=2, seed=57)
    g = _gpu(sc)
    assert g.matchBegin() and g.matchPairs(0, len(g.pairs()[0]))
    ptr, n_slots = g.slot_buffer()
    t = dist.device_tensor(ptr, n_slots * 32, torch.device("cuda", 0))
    assert t.dtype == torch.uint8 and t.numel() == n_slots * 32 and t.data_ptr() == ptr
    host = np.frombuffer(t.cpu().numpy().tobytes(), SLOT_DTYPE)
    pairs, off = g.pairs()
    for pi in range(len(pairs)):
        sl = g.pair_slots(pi)
        assert np.array_equal(host[int(off[pi]):int(off[pi]) + sl.size], sl.reshape(-1))
    # in-place write through the tensor is seen by the library (what a broadcast into the buffer does)
    t[:32] = 0xFF
    assert g.pair_slots(0).reshape(-1)[0]["tgt_seg"] == EMPTY


def test_final_3d_lines_vs_reference_own_code():
    """matchImages -> reconstruct3Dlines -> get3Dlines against the reference's own
    clustering.cc / line3D.cc tail (oracle/_ref).  Same clusters (residual 2D segments), same reference view,
    3D end points within 1e-4 relative (the 3x3 principal direction comes from two different Jacobi codes,
    and its sign is free: end points are compared as unordered pairs)."""
    from oracle import oracle as O
    assert O.have_reference(), "oracle/_ref is missing: no downgrade of the checker on the GPU box"
    for (nv, ns, nn, seed, vis) in ((12, 500, 6, 61, 3), (16, 400, 8, 62, 4)):
        sc = make_scene(nv, ns, n_neighbors=nn, seed=seed)
        g = _gpu(sc)
        assert g.matchImages() and g.reconstruct3Dlines(visibility_t=vis)
        gl = g.get3Dlines()
        r = O.Oracle(threads=1, reference=True)
        r.add_scene(sc); r.match_images(); r.reconstruct(vis)
        rl = r.lines()
        assert len(gl) == len(rl) and len(gl) > 10
        key = lambda res: tuple(sorted(map(tuple, np.asarray(res).reshape(-1, 2).tolist())))
        rmap = {key(L["residuals"]): L for L in rl}
        scale = 30.0   # scene extent: tolerance is relative to coordinates of this magnitude
        for L in gl:
            k = key(np.stack([L["residuals"]["cam"], L["residuals"]["seg"]], 1))
            assert k in rmap, "cluster membership differs"
            R = rmap[k]
            assert L["reference_view"] == R["reference_view"]
            assert len(L["collinear3Dsegments"]) == len(R["collinear3Dsegments"])
            for a, b in zip(L["collinear3Dsegments"], R["collinear3Dsegments"]):
                p = np.concatenate([a["P1"], a["P2"]]); q = b[:6]; qs = np.concatenate([b[3:6], b[0:3]])
                err = min(np.abs(p - q).max(), np.abs(p - qs).max())
                assert err <= H.REL_TOL * scale, err
        # the affinity accessors still work after reconstruct3Dlines, and every residual has a hypothesis
        hyp = set(map(tuple, np.stack([g.best()[0]["cam"], g.best()[0]["seg"]], 1).tolist()))
        assert all(set(map(tuple, np.stack([L["residuals"]["cam"], L["residuals"]["seg"]], 1).tolist())) <= hyp for L in gl)


def test_rdd_diffusion_matches_the_reference_code():
    """Replicator-dynamics diffusion: l3d_diffuse_affinity (seam) and
    reconstruct3Dlines(perform_diffusion=True) against THE REFERENCE'S OWN performRDD: line3D.cc:2026-2076,
    sparsematrix.cc and the K_sparseMat_* kernels of cudawrapper.cu:432-544 / 708-766 compiled in place and executed as
    host code (oracle/_ref/libl3d_ref_cuda.so through oracle/ref_shim_cuda; its absence fails the test).  The
    restatement lo_rdd is pinned byte for byte against the same code in tests/test_reference_pin.py."""
    from line3dpp_amd.api import diffuse_affinity
    from oracle import oracle as O

    class Oracle:   # the checker of this test: the reference's own code
        @staticmethod
        def rdd(edges, n_rows):
            assert O.have_cuda_path(), "oracle/_ref/libl3d_ref_cuda.so is missing (oracle/Makefile, needs /root/reference)"
            return O.rdd_reference(edges, n_rows)
    sc = make_scene(10, 400, n_neighbors=4, seed=51)
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    e, l2g, _ = g.affinity()
    assert len(e) > 200
    ref = Oracle.rdd(e, len(l2g))
    out = diffuse_affinity(e, len(l2g))
    assert len(out) == len(ref) == len(e)
    assert np.array_equal(out["i"], ref["i"]) and np.array_equal(out["j"], ref["j"])
    assert np.max(np.abs(out["w"] - ref["w"]) / ref["w"]) < 1e-4
    assert np.array_equal(out["w"], ref["w"])          # same float operations in the same order: bit-identical in practice
    # (i, j) ascending, symmetric values
    key = out["i"].astype(np.int64) * (1 << 32) + out["j"]
    assert np.all(np.diff(key) > 0)
    w = {(int(a), int(b)): float(c) for a, b, c in zip(out["i"], out["j"], out["w"])}
    assert all(w[(j, i)] == v for (i, j), v in w.items())
    # through the pipeline: A_ after reconstruct3Dlines(diffusion) is the diffused matrix, lines come out
    assert g.reconstruct3Dlines(3, True)
    e2, _, _ = g.affinity()
    assert np.array_equal(e2["i"], ref["i"]) and np.array_equal(e2["w"], ref["w"])
    lines_rdd = g.get3Dlines()
    assert g.reconstruct3Dlines(3, False)
    lines_plain = g.get3Dlines()
    assert len(lines_rdd) > 0 and len(lines_plain) > 0
    # random sparse symmetric-pattern matrix with uneven degrees (exercises the lockstep walk and empty rows)
    rng = np.random.default_rng(5)
    n = 300
    pairs = {(int(a), int(b)) for a, b in rng.integers(0, n, (1500, 2)) if a != b}
    pairs |= {(b, a) for a, b in pairs}
    ee = np.array([(a, b, rng.uniform(0.05, 1.0)) for a, b in sorted(pairs)], dtype=e.dtype)
    ee = ee[rng.permutation(len(ee))]
    ref = Oracle.rdd(ee, n); out = diffuse_affinity(ee, n)
    assert np.array_equal(out["i"], ref["i"]) and np.array_equal(out["j"], ref["j"])
    assert np.max(np.abs(out["w"] - ref["w"]) / ref["w"]) < 1e-4


def test_collinearity_links_and_lines():
    """collinearity_t > 0: per-view collinear lists + the extra affinity links, against the
    restatement (A_ ids/order identical, weights within tolerance) and, for the final 3D lines, against the
    reference's own code (oracle/_ref)."""
    from oracle import oracle as O
    sc = H.split_scene(make_scene(8, 400, n_neighbors=4, seed=9))
    collin_t = 2.0
    g = _gpu(sc)
    assert g.matchImages() and g.reconstruct3Dlines(3, False, collin_t)
    ge, gl, _ = g.affinity()
    o = _oracle(sc, threads=1)
    o.match_images(); o.set_collinearity(collin_t); o.compute_affinity()
    oe, ol = o.affinity()
    o2 = _oracle(sc, threads=1); o2.match_images(); o2.compute_affinity()
    assert len(oe) > len(o2.affinity()[0]) + 200          # the links add edges
    assert len(ge) == len(oe) and len(gl) == len(ol)
    assert np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])
    assert np.array_equal(np.stack([gl["cam"], gl["seg"]], 1), ol)
    assert np.max(np.abs(ge["w"] - oe["w"]) / oe["w"]) < H.REL_TOL
    lines = g.get3Dlines()
    assert len(lines) > 10
    assert O.have_reference(), "oracle/_ref is missing: no downgrade of the checker on the GPU box"
    if True:
        r = O.Oracle(threads=1, reference=True)
        r.add_scene(sc); r.match_images(); r.reconstruct(3, collin_t)
        rl = r.lines()
        key = lambda res: tuple(sorted(map(tuple, np.asarray(res).reshape(-1, 2).tolist())))
        assert len(lines) == len(rl)
        assert {key(np.stack([L["residuals"]["cam"], L["residuals"]["seg"]], 1)) for L in lines} == {key(L["residuals"]) for L in rl}
    # switching it off again restores the plain matrix
    assert g.reconstruct3Dlines(3, False, -1.0)
    assert len(g.affinity()[0]) == len(o2.affinity()[0])


@pytest.mark.parametrize("n_views,n_segs,nn,kNN,epi,seed,rf,min_avg", [
    (18, 300, 16, 30, 0.1, 71, 0.5, 50),       # ~200 lists beyond one wave's staging (192), longest ~360
    (12, 1100, 11, 80, 0.05, 73, 0.9, 200),    # thousands of long lists, some beyond 768 (sort-only path)
])
def test_long_hypothesis_lists_parity(n_views, n_segs, nn, kNN, epi, seed, rf, min_avg):
    """Many neighbours / large kNN make per-segment hypothesis lists longer than one wave's LDS staging (192):
    the workgroup-per-list support kernel (staged up to 768, sort-only beyond) against the oracle, full pipeline."""
    sc = make_scene(n_views, n_segs, n_neighbors=nn, seed=seed, real_fraction=rf)
    g = _gpu(sc)
    assert g.matchImages(kNN=kNN, epipolar_overlap=epi) and g.computeAffinity()
    tm = g.timings()
    assert tm["list_entries"] / (n_views * n_segs) > min_avg
    o = _oracle(sc, threads=16)
    o.match_images(kNN=kNN, epi_overlap=epi); o.compute_affinity()
    worst = 0.0
    for v in sc.views:
        r = H.compare_matches(g.matches(v.cam)[0], o.matches(v.cam)[0])
        assert not r["missing"] and not r["extra"], (v.cam, r["missing"][:3], r["extra"][:3])
        worst = max(worst, r["max_rel"])
    assert worst < H.REL_TOL
    ge, gl, _ = g.affinity(); oe, ol = o.affinity()
    assert len(ge) == len(oe) and np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_phase_a_emulated_on_one_gpu(world):
    """The N>1 path without RCCL: `world` contexts on one GPU each match their range of the pair list
    (l3d_match_pairs(first, count)), the slot slices are copied device-to-device exactly where the all-gather would
    put them, and every context finishes phase B on its own -- all must equal the single-context result."""
    import torch
    from line3dpp_amd import dist
    sc = make_scene(9, 350, n_neighbors=4, seed=81)
    ref = _gpu(sc)
    assert ref.matchImages() and ref.computeAffinity()
    ctxs = [_gpu(sc) for _ in range(world)]
    for g in ctxs:
        assert g.matchBegin()
    pairs, slot_off = ctxs[0].pairs()
    M = {v.cam: len(v.segs) for v in sc.views}
    ranges = dist.pair_ranges([M[int(s)] * M[int(t)] for s, t in pairs], world)
    for r, g in enumerate(ctxs):
        first, count = ranges[r]
        if count:
            assert g.matchPairs(first, count)
    ptr0, n_slots = ctxs[0].slot_buffer()
    bufs = [dist.device_tensor(g.slot_buffer()[0], n_slots * 32, torch.device("cuda", 0)) for g in ctxs]
    byte_ranges = dist.slot_byte_ranges(ranges, slot_off, n_slots)
    for r, (lo, hi) in enumerate(byte_ranges):       # "all-gather": rank r's slice to everybody else
        for q in range(world):
            if q != r and hi > lo:
                bufs[q][lo:hi].copy_(bufs[r][lo:hi])
    torch.cuda.synchronize()
    for g in ctxs:
        g.L.l3d_slots_exchanged(g.h)
        assert g.matchFinish() and g.computeAffinity()
        for v in sc.views:
            a, ao = g.matches(v.cam); b, bo = ref.matches(v.cam)
            assert np.array_equal(ao, bo) and a.tobytes() == b.tobytes()
        ge, gl, _ = g.affinity(); re_, rl, _ = ref.affinity()
        assert ge.tobytes() == re_.tobytes() and gl.tobytes() == rl.tobytes()


@pytest.mark.parametrize("world", [2, 3])
def test_compact_exchange_emulated_on_one_gpu(world):
    """The default N>1 exchange (line3dpp_amd/dist.py): only the uint32 target index of every slot travels and the
    receiving context re-derives the 32-byte records (l3d_pack_slot_indices / l3d_expand_slot_indices).  `world`
    contexts on one GPU; the index slices are copied where the all-gather would put them.  The expanded slot buffer
    must equal, byte for byte, the one a single context produces, and so must the final matches and affinities."""
    import torch
    from line3dpp_amd import dist
    sc = H.split_scene(make_scene(9, 350, n_neighbors=4, seed=83))
    for i, v in enumerate(sc.views):                 # ragged views: uneven slices
        v.segs = v.segs[:len(v.segs) - 11 * i].copy()
    ref = _gpu(sc)
    assert ref.matchBegin()
    pairs, slot_off = ref.pairs()
    assert ref.matchPairs(0, len(pairs))
    ptr, n_slots = ref.slot_buffer()
    dev = torch.device("cuda", 0)
    want = dist.device_tensor(ptr, n_slots * 32, dev).clone()
    assert ref.matchFinish() and ref.computeAffinity()
    ctxs = [_gpu(sc) for _ in range(world)]
    for g in ctxs:
        assert g.matchBegin()
    M = {v.cam: len(v.segs) for v in sc.views}
    ranges = dist.pair_ranges([M[int(s)] * M[int(t)] for s, t in pairs], world)
    for r, g in enumerate(ctxs):
        first, count = ranges[r]
        if count:
            assert g.matchPairs(first, count)
        assert g.packSlotIndices(first, count)
    bufs = [dist.device_tensor(g.slot_index_buffer()[0], n_slots * 4, dev) for g in ctxs]
    byte_ranges = dist.slot_byte_ranges(ranges, slot_off, n_slots, slot_bytes=4)
    assert sum(hi - lo for lo, hi in byte_ranges) == n_slots * 4
    for r, (lo, hi) in enumerate(byte_ranges):
        for q in range(world):
            if q != r and hi > lo:
                bufs[q][lo:hi].copy_(bufs[r][lo:hi])
    torch.cuda.synchronize()
    for r, g in enumerate(ctxs):
        first, count = ranges[r]
        assert g.expandSlotIndices(0, first) and g.expandSlotIndices(first + count, len(pairs) - first - count)
        g.L.l3d_synchronize(g.h)
        got = dist.device_tensor(g.slot_buffer()[0], n_slots * 32, dev)
        assert torch.equal(got, want), "expanded slots differ from the match kernel's"
        assert g.matchFinish() and g.computeAffinity()
        for v in sc.views:
            a, ao = g.matches(v.cam); b, bo = ref.matches(v.cam)
            assert np.array_equal(ao, bo) and a.tobytes() == b.tobytes()
        ge, gl, _ = g.affinity(); re_, rl, _ = ref.affinity()
        assert ge.tobytes() == re_.tobytes() and gl.tobytes() == rl.tobytes()
    # misuse: packing a pair this context has not matched; matching or expanding a pair twice (its slots already
    # feed the phase-B counters) -- refused, and the call sequence still completes
    g = _gpu(sc)
    assert g.matchBegin() and not g.packSlotIndices(0, 1)
    assert g.matchPairs(0, len(pairs))
    assert not g.matchPairs(0, 1) and not g.expandSlotIndices(0, 1)
    assert g.matchFinish() and g.computeAffinity()
    ge, gl, _ = g.affinity(); re_, rl, _ = ref.affinity()
    assert ge.tobytes() == re_.tobytes()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_list_pass_emulated_on_one_gpu(world):
    """Phase B's list pass sharded by views (l3d_lists_shard, line3dpp_amd/dist.py): `world` contexts on one GPU; every
    context holds all slots (as after the slot exchange), runs the list pass for ITS views into ITS pools, the pool
    slabs are copied where the all-gather would put them, and every context finishes on the complete records.
    Surviving matches, best hypotheses, view medians and A_ must equal a single context's, byte for byte."""
    import torch
    from line3dpp_amd import dist
    sc = H.split_scene(make_scene(11, 420, n_neighbors=6, seed=29))
    for i, v in enumerate(sc.views):                 # ragged views: view ranges of unequal length
        v.segs = v.segs[:len(v.segs) - 9 * i].copy()
    ref = _gpu(sc)
    assert ref.matchImages() and ref.computeAffinity()
    dev = torch.device("cuda", 0)
    ctxs = [_gpu(sc) for _ in range(world)]
    for g in ctxs:
        assert g.matchBegin() and g.matchPairs(0, len(g.pairs()[0]))
    for attempt in range(8):
        slabs = []
        for r, g in enumerate(ctxs):
            sl = g.listsShard(r, world)
            assert sl is not None and len(sl) == 4
            slabs.append(sl)
        for k in range(4):
            sizes = {sl[k][1] for sl in slabs}
            assert len(sizes) == 1, "equal slab sizes on every rank"
            sb = sizes.pop()
            fulls = [dist.device_tensor(sl[k][2], sb * world, dev) for sl in slabs]
            for r in range(world):
                assert slabs[r][k][0] == slabs[r][k][2] + r * sb
                for q in range(world):
                    if q != r:
                        fulls[q][r * sb:(r + 1) * sb].copy_(fulls[r][r * sb:(r + 1) * sb])
        torch.cuda.synchronize()
        rcs = [g.L.l3d_match_finish(g.h) for g in ctxs]
        assert len(set(rcs)) == 1, "every rank takes the same decision (all of them see all pool counters)"
        if rcs[0] == 0:
            break
        assert rcs[0] == -10, rcs               # L3D_ERR_RETRY: pools enlarged on every rank alike, repeat the step
    else:
        raise AssertionError("the pools never became large enough")
    for g in ctxs:
        assert g.computeAffinity()
        for v in sc.views:
            a, ao = g.matches(v.cam); b, bo = ref.matches(v.cam)
            assert np.array_equal(ao, bo) and a.tobytes() == b.tobytes()
            assert g.view_info(v.cam) == ref.view_info(v.cam)
        for x, y in zip(g.best(), ref.best()):
            assert x.tobytes() == y.tobytes()
        ge, gl, _ = g.affinity(); re_, rl, _ = ref.affinity()
        assert ge.tobytes() == re_.tobytes() and gl.tobytes() == rl.tobytes()
    # order: the list pass needs every pair's slots
    g = _gpu(sc)
    assert g.matchBegin() and g.matchPairs(0, 1) and g.listsShard(0, 2) is None and g.last_status == -7
    assert g.matchImages()                           # the failed call left a clean context


@pytest.mark.parametrize("world", [2, 3, 8])
def test_halo_form_emulated_on_one_gpu_at_c2_slice_size(world):
    """The halo form of the multi-GPU call (line3dpp_amd/dist.py match_images_halo) with `world` contexts on one GPU, on
    a 24-view slice of BASELINE C2 at the configured size (4096 segments per view, 20 neighbours): every context
    matches the pairs whose source view it owns (l3d_plan_shards), the compact indices of the pairs whose target view
    another context owns are copied where the point-to-point exchange would put them and expanded there, every context
    runs the list pass of ITS views with only the pairs that touch them present, the record slabs are copied where the
    all-gather would put them, and every context finishes on the records alone.  Everything a user can read back must
    equal a single context's result byte for byte."""
    import torch
    from line3dpp_amd import dist
    from line3dpp_amd.scene import make_config
    sc = H.ring_slice(make_config("C2", max_views=24), 0, 24)
    ref = _gpu(sc)
    assert ref.matchImages() and ref.computeAffinity()
    dev = torch.device("cuda", 0)
    ctxs = [_gpu(sc) for _ in range(world)]
    for g in ctxs:
        assert g.matchBegin()
    pairs, slot_off = ctxs[0].pairs()
    plan = dist.plan_halo(pairs, ctxs[0]._M, world)
    vb, pb, runs = plan["view_bounds"], plan["pair_bounds"], plan["runs"]
    assert sum(len(r) for r in runs) > 0, "a ring slice cut into ranges has pairs across the cuts"
    bufs = []
    for r, g in enumerate(ctxs):
        first, count = int(pb[r]), int(pb[r + 1] - pb[r])
        halo = [p for (_, f, n) in runs[r] for p in range(f, f + n)]
        early, late = dist.early_ranges(first, count, halo)
        assert sum(n for _, n in early) + late[1] == count
        for f, n in early + [late]:
            assert n == 0 or g.matchPairs(f, n)
        for _, f, n in runs[r]:
            assert g.packSlotIndices(f, n)
        ptr, n_slots = g.slot_index_buffer()
        bufs.append(dist.device_tensor(ptr, n_slots * 4, dev))
    off = [int(o) for o in slot_off] + [int(n_slots)]
    moved = 0
    for r in range(world):
        for q, f, n in runs[r]:                      # what isend / irecv would move: rank r -> rank q
            bufs[q][4 * off[f]:4 * off[f + n]].copy_(bufs[r][4 * off[f]:4 * off[f + n]])
            moved += 4 * (off[f + n] - off[f])
    torch.cuda.synchronize()
    assert moved < 4 * n_slots * (world - 1) / 2, "the halo is a fraction of what an all-gather moves"
    for q, g in enumerate(ctxs):
        for r in range(world):
            for qq, f, n in runs[r]:
                if qq == q:
                    assert g.expandSlotIndices(f, n)
    for attempt in range(8):
        slabs = []
        for r, g in enumerate(ctxs):
            sl = g.listsShardViews(r, world, int(vb[r]), int(vb[r + 1]))
            assert sl is not None and len(sl) == 4
            slabs.append(sl)
        for k in range(4):
            sizes = {sl[k][1] for sl in slabs}
            assert len(sizes) == 1, "equal slab sizes on every rank"
            sb = sizes.pop()
            fulls = [dist.device_tensor(sl[k][2], sb * world, dev) for sl in slabs]
            for r in range(world):
                for q in range(world):
                    if q != r:
                        fulls[q][r * sb:(r + 1) * sb].copy_(fulls[r][r * sb:(r + 1) * sb])
        torch.cuda.synchronize()
        rcs = [g.L.l3d_match_finish(g.h) for g in ctxs]
        assert len(set(rcs)) == 1, "every rank takes the same decision (all of them see all pool counters)"
        if rcs[0] == 0:
            break
        assert rcs[0] == -10, rcs
    else:
        raise AssertionError("the pools never became large enough")
    for g in ctxs:
        assert g.computeAffinity()
        for v in sc.views:
            a, ao = g.matches(v.cam); b, bo = ref.matches(v.cam)
            assert np.array_equal(ao, bo) and a.tobytes() == b.tobytes()
            assert g.view_info(v.cam) == ref.view_info(v.cam)
        for x, y in zip(g.best(), ref.best()):
            assert x.tobytes() == y.tobytes()
        ge, gl, _ = g.affinity(); re_, rl, _ = ref.affinity()
        assert ge.tobytes() == re_.tobytes() and gl.tobytes() == rl.tobytes()
    # a pair that touches the rank's views but never arrived is noticed (rank 1 receives the pairs across the first cut)
    assert any(q == 1 for (q, _, _) in runs[0])
    g = _gpu(sc)
    assert g.matchBegin() and g.matchPairs(int(pb[1]), int(pb[2] - pb[1]))
    assert g.listsShardViews(1, world, int(vb[1]), int(vb[2])) is None and g.last_status == -7
    assert g.matchImages()                           # the failed call left a clean context


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_tail_emulated_on_one_gpu_at_c2_slice_size(world):
    """The same halo form with the TAIL of phase B sharded by views as well (l3d_tail_shard_count / _layout / _commit):
    after the record slabs have been copied where the exchange would put them, every context runs the chain on the
    records its views depend on, computes scores, filterMatches, outputs and medians of ITS views only and writes them at
    their places in the full arrays; the parts are copied where the exchange would put them; the commit closes the call.
    Everything a user can read back must equal a single context's result byte for byte.  Round 6: a context only receives
    the record slabs of the ranks its views DEPEND on (dist.shard_needs; the counter slab of every rank) and its chain only
    covers those (l3d_shard_options)."""
    from line3dpp_amd.scene import make_config
    _sharded_tail_emulation(H.ring_slice(make_config("C2", max_views=24), 0, 24), world)


def test_sharded_tail_of_independent_rings_needs_no_foreign_records():
    """weak scaling as bench.py --gpus N runs it: N rings without a pair between them, one per rank.  No rank depends on
    another's records (dist.shard_needs gives empty sets): NO record slab travels, every chain covers its own pools only,
    and the result is still the single context's byte for byte."""
    from line3dpp_amd import dist
    from line3dpp_amd.scene import Scene
    views = []
    for r in range(3):
        sub_ = make_scene(7, 280, n_neighbors=4, seed=31 + r)
        off = len(views)
        for v in sub_.views:
            v.cam += off
            v.neighbors = [int(x) + off for x in v.neighbors]
            views.append(v)
    sc = Scene(views, "3rings")
    M = {v.cam: len(v.segs) for v in sc.views}
    pairs = sc.pair_tests()[1]
    plan = dist.plan_halo(pairs, M, 3)
    assert plan["needs"] == [[], [], []] and not any(plan["runs"]), plan
    _sharded_tail_emulation(sc, 3)


def _sharded_tail_emulation(sc, world):
    import torch
    from line3dpp_amd import dist
    ref = _gpu(sc)
    assert ref.matchImages() and ref.computeAffinity()
    dev = torch.device("cuda", 0)
    ctxs = [_gpu(sc) for _ in range(world)]
    for g in ctxs:
        assert g.matchBegin()
    pairs, slot_off = ctxs[0].pairs()
    plan = dist.plan_halo(pairs, ctxs[0]._M, world)
    vb, pb, runs, needs = plan["view_bounds"], plan["pair_bounds"], plan["runs"], plan["needs"]
    assert all(all(q < r for q in needs[r]) for r in range(world)), "a rank only depends on ranks below it"
    bufs = []
    for r, g in enumerate(ctxs):
        assert g.shardOptions(min(needs[r] + [r]), False)
        assert g.matchPairs(int(pb[r]), int(pb[r + 1] - pb[r]))
        for _, f, n in runs[r]:
            assert g.packSlotIndices(f, n)
        ptr, n_slots = g.slot_index_buffer()
        bufs.append(dist.device_tensor(ptr, n_slots * 4, dev))
    off = [int(o) for o in slot_off] + [int(n_slots)]
    for r in range(world):
        for q, f, n in runs[r]:
            bufs[q][4 * off[f]:4 * off[f + n]].copy_(bufs[r][4 * off[f]:4 * off[f + n]])
    torch.cuda.synchronize()
    for q, g in enumerate(ctxs):
        for r in range(world):
            for qq, f, n in runs[r]:
                if qq == q:
                    assert g.expandSlotIndices(f, n)
    for attempt in range(8):
        slabs = []
        for r, g in enumerate(ctxs):
            sl = g.listsShardViews(r, world, int(vb[r]), int(vb[r + 1]))
            assert sl is not None and len(sl) == 4
            slabs.append(sl)
        for k in range(4):
            sb = slabs[0][k][1]
            fulls = [dist.device_tensor(sl[k][2], sb * world, dev) for sl in slabs]
            for r in range(world):
                for q in range(world):
                    if q != r and (k == 3 or r in needs[q]):      # records: only to the ranks that depend on them
                        fulls[q][r * sb:(r + 1) * sb].copy_(fulls[r][r * sb:(r + 1) * sb])
                    elif q != r:
                        fulls[q][r * sb:(r + 1) * sb].fill_(0xA5)   # what never arrives must not be read either
        torch.cuda.synchronize()
        res = [g.tailShardCount() for g in ctxs]
        assert len({rc for rc, _, _ in res}) == 1, "every rank takes the same decision (all of them see all pool counters)"
        if res[0][0] == 0:
            break
        assert res[0][0] == -10, res
    else:
        raise AssertionError("the pools never became large enough")
    counts = [(n, h) for _, n, h in res]
    assert sum(n for n, _ in counts) == sum(len(ref.matches(v.cam)[0]) for v in sc.views) and all(n > 0 for n, _ in counts)
    layouts = [g.tailShardLayout(world, counts, [int(v) for v in vb]) for g in ctxs]
    assert all(l is not None and len(l) == 9 for l in layouts)
    moved = 0
    for k in range(9):
        elt, parts = layouts[0][k][1], layouts[0][k][2]
        assert all(l[k][1] == elt and l[k][2] == parts for l in layouts), "every rank derives the same layout"
        total = max(f + n for f, n in parts) * elt
        fulls = [dist.device_tensor(l[k][0], total, dev) for l in layouts]
        for r, (f, n) in enumerate(parts):
            for q in range(world):
                if q != r and n:
                    fulls[q][f * elt:(f + n) * elt].copy_(fulls[r][f * elt:(f + n) * elt])
                    moved += n * elt
    torch.cuda.synchronize()
    assert moved > 0
    assert [g.tailShardCommit() for g in ctxs] == [0] * world
    # the affinity fill sharded by the same views (l3d_affinity_shard_begin / _finish): every context computes the
    # similarities of ITS views' surviving matches, the float parts are copied where the exchange would put them
    aparts = [g.affinityShardBegin(r, world) for r, g in enumerate(ctxs)]
    assert all(a is not None and a[1] == 4 and a[2] == aparts[0][2] for a in aparts), "every rank derives the same parts"
    assert [n for _, n in aparts[0][2]] == [n for n, _ in counts]
    n_all = sum(n for _, n in aparts[0][2])
    sims = [dist.device_tensor(a[0], 4 * n_all, dev) for a in aparts]
    for q in range(world):                           # poison what a rank did not compute: a part that never arrives shows
        for r, (f, n) in enumerate(aparts[0][2]):
            if q != r and n:
                sims[q][4 * f:4 * (f + n)].fill_(0xFF)
    for r, (f, n) in enumerate(aparts[0][2]):
        for q in range(world):
            if q != r and n:
                sims[q][4 * f:4 * (f + n)].copy_(sims[r][4 * f:4 * (f + n)])
    torch.cuda.synchronize()
    assert all(g.affinityShardFinish() for g in ctxs)
    assert ctxs[0].affinityShardBegin(0, world + 1) is None and ctxs[0].affinityShardFinish() is False   # wrong world; no open shard
    # an open shard refuses a second begin and the unsharded fill (the views are translated: ADVICE round 5); abort closes it
    # without the bookkeeping pass, after which the replicated fill gives the same A_
    assert ctxs[0].affinityShardBegin(0, world) is not None
    assert ctxs[0].affinityShardBegin(0, world) is None and ctxs[0].computeAffinity() is False
    assert ctxs[0].affinityShardAbort() and ctxs[0].affinityShardAbort()
    assert ctxs[0].computeAffinity()
    for g in ctxs:
        for v in sc.views:
            a, ao = g.matches(v.cam); b, bo = ref.matches(v.cam)
            assert np.array_equal(ao, bo) and a.tobytes() == b.tobytes()
            assert g.view_info(v.cam) == ref.view_info(v.cam)
        for x, y in zip(g.best(), ref.best()):
            assert x.tobytes() == y.tobytes()
        ge, gl, _ = g.affinity(); re_, rl, _ = ref.affinity()
        assert ge.tobytes() == re_.tobytes() and gl.tobytes() == rl.tobytes()
    # out of order: a commit without a layout, a layout with another world size
    g = ctxs[0]
    assert g.matchBegin() and g.tailShardCommit() != 0
    assert g.matchImages()                           # the failed call left a clean context
    assert g.affinityShardBegin(0, world) is None    # (a single-context call has no sharded tail behind it)
    assert g.computeAffinity()


def test_txt_writer_matches_accessor_and_reference_writer(tmp_path):
    """l3d_save_3d_lines_txt (Line3D::save3DLinesAsTXT): file name and content against get3Dlines() and against
    the file the reference's own writer produces for the same scene (oracle/_ref)."""
    from line3dpp_amd.io import read_3d_lines_txt
    from oracle import oracle as O
    sc = make_scene(12, 500, n_neighbors=6, seed=61)
    g = _gpu(sc)
    assert g.matchImages() and g.reconstruct3Dlines(3)
    name = g.outputFilename()
    assert name == "Line3D++__W_FULL__N_10__sigmaP_2.5__sigmaA_10__epiOverlap_0.25__kNN_10__vis_3"
    assert g.save3DLinesAsTXT(tmp_path)
    mine = read_3d_lines_txt(tmp_path / (name + ".txt"))
    acc = g.get3Dlines()
    assert len(mine) == len(acc) > 10
    segs = {v.cam: v.segs for v in sc.views}
    for a, b in zip(mine, acc):
        ref = np.concatenate([b["collinear3Dsegments"]["P1"], b["collinear3Dsegments"]["P2"]], 1)
        assert np.allclose(a["segments"], ref, rtol=1e-5, atol=1e-6)          # 6 significant digits in the file
        assert np.array_equal(a["residuals"], np.stack([b["residuals"]["cam"], b["residuals"]["seg"]], 1))
        for (cam, seg), co in zip(a["residuals"], a["coords2D"]):
            assert np.allclose(co, segs[int(cam)][int(seg)], rtol=1e-5)
    assert np.array_equal(g.getSegmentCoords2D(3, 17), sc.views[3].segs[17])
    assert np.array_equal(g.getSegmentCoords2D(999, 0), np.zeros(4, np.float32))     # unknown camera -> zeros
    # OBJ / STL writers: the library's files equal the reference formats applied to the same lines
    from line3dpp_amd.io import format_obj, format_stl
    assert g.saveResultAsOBJ(tmp_path) and g.saveResultAsSTL(tmp_path)
    assert open(tmp_path / (name + ".obj")).read() == format_obj(mine)
    full = [dict(segments=np.concatenate([b["collinear3Dsegments"]["P1"], b["collinear3Dsegments"]["P2"]], 1)) for b in acc]
    assert open(tmp_path / (name + ".stl")).read() == format_stl(full)
    # BIN writer (Line3D::save3DLinesAsBIN, boost binary archive layout): lossless -- the parsed file equals
    # get3Dlines() bit for bit, cluster line, residuals and reference view included, and re-serialises identically
    from line3dpp_amd.io import format_3d_lines_bin, read_3d_lines_bin
    assert g.save3DLinesAsBIN(tmp_path)
    binl, version = read_3d_lines_bin(tmp_path / (name + ".bin"))
    assert version == 10 and len(binl) == len(acc)
    for a, b in zip(binl, acc):
        cs = b["collinear3Dsegments"]
        assert np.array_equal(a["segments"], np.concatenate([cs["P1"], cs["P2"], cs["dir"]], 1))
        assert np.array_equal(a["seg_length"], cs["length"]) and np.array_equal(a["seg_valid"], cs["valid"].astype(np.uint8))
        cl = b["cluster_line"]
        assert np.array_equal(a["cluster_line"], np.concatenate([cl["P1"], cl["P2"], cl["dir"]]))
        assert a["cluster_length"] == cl["length"] and a["reference_view"] == b["reference_view"]
        assert np.array_equal(a["residuals"], np.stack([b["residuals"]["cam"], b["residuals"]["seg"]], 1))
    assert format_3d_lines_bin(binl, version) == open(tmp_path / (name + ".bin"), "rb").read()
    assert O.have_reference(), "oracle/_ref is missing: no downgrade of the checker on the GPU box"
    if True:
        r = O.Oracle(threads=1, reference=True)
        r.add_scene(sc); r.match_images(); r.reconstruct(3)
        d = tmp_path / "ref"; d.mkdir()
        r.save_txt(d)
        files = list(d.iterdir())
        assert [f.name for f in files] == [name + ".txt"]                     # createOutputFilename
        theirs = read_3d_lines_txt(files[0])
        key = lambda L: tuple(sorted(map(tuple, L["residuals"].tolist())))
        tm = {key(L): L for L in theirs}
        assert len(theirs) == len(mine) and set(tm) == {key(L) for L in mine}
        for L in mine:
            R = tm[key(L)]
            assert len(L["segments"]) == len(R["segments"])
            for p, q in zip(L["segments"], R["segments"]):
                qs = np.concatenate([q[3:], q[:3]])
                assert min(np.abs(p - q).max(), np.abs(p - qs).max()) <= 30.0 * H.REL_TOL


def test_real_testdata_c0_parity_and_fixture_plausibility():
    """BASELINE config C0 (the reference's bundled testdata; cameras recovered from its result fixture,
    tests/golden/make_real_scene.py): full pipeline of the HIP path against the oracle on real LSD segments and
    real geometry; final 3D lines against the reference's own code; most of the fixture's published lines re-found."""
    from line3dpp_amd.scene import make_config, C0_FILE
    from oracle import oracle as O
    sc = make_config("C0")
    g = _gpu(sc)
    assert g.matchImages() and g.reconstruct3Dlines(3)
    assert g.timings()["culled_pairs"] > 0.8 * len(g.pairs()[0])
    o = _oracle(sc, threads=16)
    o.match_images(); o.compute_affinity()
    worst, n = 0.0, 0
    for v in sc.views:
        r = H.compare_matches(g.matches(v.cam)[0], o.matches(v.cam)[0])
        assert not r["missing"] and not r["extra"], (v.cam, r["missing"][:3], r["extra"][:3])
        worst = max(worst, r["max_rel"]); n += r["n_cpu"]
    assert n > 50000 and worst < H.REL_TOL
    ge, gl, _ = g.affinity(); oe, ol = o.affinity()
    assert len(ge) == len(oe) > 20000 and np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])
    assert np.max(np.abs(ge["w"] - oe["w"]) / oe["w"]) < H.REL_TOL
    lines = g.get3Dlines()
    mine = [frozenset(map(tuple, np.stack([L["residuals"]["cam"], L["residuals"]["seg"]], 1).tolist())) for L in lines]
    assert O.have_reference(), "oracle/_ref is missing: no downgrade of the checker on the GPU box"
    if True:
        r = O.Oracle(threads=1, reference=True)
        r.add_scene(sc); r.match_images(); r.reconstruct(3)
        rl = r.lines()
        theirs = [frozenset(map(tuple, np.asarray(L["residuals"]).reshape(-1, 2).tolist())) for L in rl]
        assert set(mine) == set(theirs) and len(mine) == len(theirs)
        # final 3D end points of EVERY line against the reference's own code on the real data: north_star's 1e-4,
        # relative to the scene extent (an end point is a position, its coordinates have no scale of their own)
        by_key = {k: L for k, L in zip(theirs, rl)}
        pts = np.concatenate([np.concatenate([L["collinear3Dsegments"]["P1"], L["collinear3Dsegments"]["P2"]], 0) for L in lines], 0)
        extent = float(np.linalg.norm(np.percentile(pts, 95, axis=0) - np.percentile(pts, 5, axis=0)))
        worst_ep, n_ep = 0.0, 0
        for L, key in zip(lines, mine):
            a = np.concatenate([L["collinear3Dsegments"]["P1"], L["collinear3Dsegments"]["P2"]], 1)
            b = np.asarray(by_key[key]["collinear3Dsegments"])[:, :6]
            assert a.shape == b.shape, "collinear 3D segments of a line"
            for p, q in zip(a, b):
                qs = np.concatenate([q[3:], q[:3]])
                worst_ep = max(worst_ep, min(np.abs(p - q).max(), np.abs(p - qs).max()) / extent)
                n_ep += 1
        assert n_ep > 1500 and worst_ep < H.REL_TOL, (n_ep, worst_ep, extent)
    d = np.load(C0_FILE)
    inv = {}
    for i, m in enumerate(mine):
        for x in m:
            inv.setdefault(x, set()).add(i)
    off, res = d["fixture_res_off"], d["fixture_res"]
    found = 0
    for k in range(len(off) - 1):
        f = [tuple(x) for x in res[off[k]:off[k + 1]].tolist()]
        cnt = {}
        for x in f:
            for i in inv.get(x, ()):
                cnt[i] = cnt.get(i, 0) + 1
        found += bool(cnt) and max(cnt.values()) >= 0.6 * len(f)
    assert len(lines) > 1500 and found > 0.6 * (len(off) - 1), (len(lines), found)
    # the lines whose cluster is exactly one of the fixture's: end points against the reference's PUBLISHED result
    ep = d["fixture_endpoints"].astype(np.float64)
    fix = {frozenset(map(tuple, res[off[k]:off[k + 1]].tolist())): ep[k] for k in range(len(off) - 1)}
    dist = []
    for L, key in zip(lines, mine):
        if key in fix:
            sg = L["collinear3Dsegments"]
            p = np.concatenate([sg["P1"][0], sg["P2"][-1]]); f = fix[key]
            dist.append(min(np.abs(p - f).max(), np.abs(p - np.concatenate([f[3:], f[:3]])).max()))
    # scene extent ~5 units: a median of 3e-4 is 6e-5 relative -- with cameras that were themselves estimated
    assert len(dist) > 900 and np.median(dist) < 1e-3 and np.percentile(dist, 90) < 5e-3, (len(dist), np.median(dist))


def test_seam_level_score_matches():
    """l3d_score_matches (replaces score_matches_GPU / Line3D::scoringGPU's kernel, line3D.cc:1297-1414) against the
    scores Line3D::scoringCPU gives the same lists in the oracle (scored lists = fresh + inverse matches of a view)."""
    from line3dpp_amd.api import score_matches
    sc = make_scene(8, 300, n_neighbors=4, seed=95)
    o = _oracle(sc, record_scored=True)
    o.match_images()
    ks = {v.cam: o.view_info(v.cam)["k"] for v in sc.views}
    Cs = {v.cam: -v.R.T @ v.t for v in sc.views}
    checked = total = 0
    for v in sc.views[1:]:                        # views with inverse matches in their lists
        m, off = o.scored(v.cam)
        A = v.R.T @ np.linalg.inv(v.K)
        rows, ranges, regs, want = [], [], [], []
        for s_ in range(len(v.segs)):
            seg = m[off[s_]:off[s_ + 1]]
            seg = seg[np.argsort(seg["tgt_cam"], kind="stable")]          # sortMatches: grouped by target camera
            ranges.append((len(rows), len(rows) + len(seg) - 1) if len(seg) else (-1, -1))
            for r in seg:
                p1 = np.array([v.segs[s_][0], v.segs[s_][1], 1.0]); p2 = np.array([v.segs[s_][2], v.segs[s_][3], 1.0])
                r1 = A @ p1; r1 /= np.linalg.norm(r1); r2 = A @ p2; r2 /= np.linalg.norm(r2)
                P1 = Cs[v.cam] + r1 * float(r["d_p1"]); P2 = Cs[v.cam] + r2 * float(r["d_p2"])
                t = int(r["tgt_cam"])
                regs.append((np.float32(np.linalg.norm(P1 - Cs[t]) * float(ks[t])), np.float32(np.linalg.norm(P2 - Cs[t]) * float(ks[t]))))
                rows.append((s_, t, r["d_p1"], r["d_p2"])); want.append(r["score3D"])
        got = score_matches(v.segs, rows, ranges, regs, A, Cs[v.cam], 200.0, ks[v.cam])
        want = np.array(want, np.float32)
        assert len(got) == len(want)
        total += len(got)
        assert np.array_equal(got > 0, want > 0)
        nz = want > 0
        assert np.max(np.abs(got[nz] - want[nz]) / want[nz]) < 1e-3       # reg_tgt recomputed in numpy (float32 inputs)
        checked += int(nz.sum())
    assert checked > 200 and total > 5000


def test_seam_level_find_collinear_segments():
    """l3d_find_collinear_segments (replaces View::findCollinGPU, view.cc:173-209) against View::findCollinCPU of
    the oracle, on real LSD segments (BASELINE C0) and on a split synthetic view."""
    from line3dpp_amd.api import find_collinear_segments
    from line3dpp_amd.scene import make_config
    for sc, t in ((make_config("C0"), 2.0), (H.split_scene(make_scene(3, 500, n_neighbors=2, seed=91)), 6.0)):
        o = _oracle(sc, threads=8)
        o.match_images(); o.set_collinearity(t); o.compute_affinity()
        total = 0
        for v in sc.views[:4]:
            off, idx = find_collinear_segments(v.segs, t)
            ooff, oidx = o.collinear(v.cam, len(v.segs))
            assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
            total += len(idx)
        assert total > 20
    off, idx = find_collinear_segments(sc.views[0].segs, -1.0)      # disabled: empty lists
    assert not off.any() and len(idx) == 0


def test_real_testdata_c0_collinearity_and_diffusion():
    """The widened rows on real data (BASELINE C0): collinear links (collinearity_t = 2 px, real fragmented LSD
    segments) and the matrix diffusion, both against the reference's own code."""
    from line3dpp_amd.api import diffuse_affinity
    from line3dpp_amd.scene import make_config
    from oracle import oracle as O
    sc = make_config("C0")
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    e0, l0, _ = g.affinity()
    assert O.have_cuda_path(), "oracle/_ref/libl3d_ref_cuda.so is missing"
    ref = O.rdd_reference(e0, len(l0))
    out = diffuse_affinity(e0, len(l0))
    assert np.array_equal(out["i"], ref["i"]) and np.array_equal(out["j"], ref["j"])
    assert np.max(np.abs(out["w"] - ref["w"]) / ref["w"]) < H.REL_TOL
    assert g.reconstruct3Dlines(3, False, 2.0)
    ge, gl, _ = g.affinity()
    assert len(ge) > len(e0)                                   # real collinear fragments add links
    o = _oracle(sc, threads=1)
    o.match_images(); o.set_collinearity(2.0); o.compute_affinity()
    oe, ol = o.affinity()
    assert len(ge) == len(oe) and np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])
    assert np.array_equal(np.stack([gl["cam"], gl["seg"]], 1), ol)
    assert O.have_reference(), "oracle/_ref is missing: no downgrade of the checker on the GPU box"
    if True:
        r = O.Oracle(threads=1, reference=True)
        r.add_scene(sc); r.match_images(); r.reconstruct(3, 2.0)
        key = lambda res: frozenset(map(tuple, np.asarray(res).reshape(-1, 2).tolist()))
        mine = {key(np.stack([L["residuals"]["cam"], L["residuals"]["seg"]], 1)) for L in g.get3Dlines()}
        assert mine == {key(L["residuals"]) for L in r.lines()}
