"""GPU parity at the BASELINE sizes (pytest -m gpu): the HIP path against the reference's own CPU code (oracle/_ref =
line3D.cc / view.cc / clustering.cc compiled in place by oracle/Makefile; its absence FAILS these tests -- there is no
downgrade to the restatement) on

  C1  full (64 views x 2000 segments, 10 neighbours) -- the configuration bench.py's headline number is quoted on
  C3  full (1024 views x 1000 segments, two rings)
  C2  a 24-view slice at the configured 4096 segments x 20 neighbours (the views in the middle of the slice have their
      full neighbour sets: the configured list lengths), all views through phase B + 32 directed pairs of phase A
  C4  an 11-view slice at 16384 segments x 10 neighbours, likewise

Everything is compared: surviving match lists (sets, order inside every list, overlap and depths bit-exact, score3D
within 1e-4), best hypotheses (same choice, 3D end points within 1e-4), A_ as a map with weights within 1e-4.  kNN rows
whose order differs from the reference's must be exact-overlap ties (libstdc++ priority_queue pop order of equal keys).
"""
import os
import time

import numpy as np
import pytest

from line3dpp_amd.scene import make_config
from tests import helpers as H

pytestmark = pytest.mark.gpu

THREADS = min(16, len(os.sched_getaffinity(0)))   # the reference's per-row containers stop scaling beyond 16 threads


def _context(scene):
    from line3dpp_amd.api import Line3D
    g = Line3D()
    g.add_scene(scene)
    assert g.matchImages() and g.computeAffinity()
    return g


def _reference(scene):
    from oracle import oracle as O
    assert O.have_reference(), "oracle/_ref/libl3d_ref.so (the reference's own code, built by oracle/Makefile where " \
                               "/root/reference exists) is missing: the at-size parity tests do not fall back"
    o = O.Oracle(threads=THREADS, reference=True)
    o.add_scene(scene)
    t0 = time.perf_counter()
    o.match_images(); o.compute_affinity()
    return o, time.perf_counter() - t0


def _assert_clean(r, min_surviving, min_edges):
    assert r["ok"], r
    assert r["set_diff"] == 0 and r["best_set_diff"] == 0 and r["affinity_set_diff"] == 0
    assert r["inexact_phase_a_fields"] == 0 and r["best_choice_diff"] == 0
    assert r["order_rows"] == r["tie_rows"], "list order differs from the reference outside exact-overlap ties"
    assert r["surviving"] >= min_surviving and r["affinity_entries"] >= min_edges, r
    assert max(r["max_rel_score3D"], r["max_rel_endpoints"], r["max_rel_affinity"]) <= H.REL_TOL


def _phase_a_sample(g, scene, n_pairs, seed=0):
    """phase-A slots of `n_pairs` sampled directed pairs against the restatement's matchingCPU, bit for bit"""
    from oracle import oracle as O
    o = O.Oracle(threads=THREADS)
    o.add_scene(scene)
    o.begin_match()
    pairs, _ = g.pairs()
    rng = np.random.default_rng(seed)
    total = ties = 0
    for pi in rng.choice(len(pairs), min(n_pairs, len(pairs)), replace=False):
        om, _ = o.match_pair(int(pairs[pi][0]), int(pairs[pi][1]))
        r = H.compare_pair_fast(g.pair_slots(int(pi)), om)
        assert r["set_diff"] == 0 and r["inexact_fields"] == 0, (int(pi), r)
        assert r["order_rows"] == r["tie_rows"], (int(pi), r)
        total += r["n_cpu"]; ties += r["tie_rows"]
    o.end_match()
    return total, ties


def test_full_c1_against_the_reference():
    sc = make_config("C1")
    g = _context(sc)
    assert g.pair_tests() == 1_280_000_000
    o, secs = _reference(sc)
    r = H.full_result_diff(g, o, sc)
    print("C1", r, f"reference CPU path {secs:.1f} s on {THREADS} threads")
    _assert_clean(r, 100_000, 100_000)
    n, ties = _phase_a_sample(g, sc, 32)
    assert n > 400_000


def test_full_c3_against_the_reference():
    sc = make_config("C3")
    g = _context(sc)
    o, secs = _reference(sc)
    r = H.full_result_diff(g, o, sc)
    print("C3", r, f"reference CPU path {secs:.1f} s on {THREADS} threads")
    _assert_clean(r, 10_000, 1_000)
    assert r["views"] == 1024
    _phase_a_sample(g, sc, 32)


@pytest.mark.parametrize("config,count,mid", [("C2", 24, 12), ("C4", 11, 5)])
def test_slice_of_c2_c4_at_configured_size(config, count, mid):
    full_nb = {"C2": 20, "C4": 10}[config]
    sc = H.ring_slice(make_config(config, max_views=count), 0, count)
    assert len(sc.views[mid].neighbors) == full_nb and len(sc.views[mid].segs) == {"C2": 4096, "C4": 16384}[config]
    g = _context(sc)
    o, secs = _reference(sc)
    r = H.full_result_diff(g, o, sc)
    print(config, "slice", r, f"reference CPU path {secs:.1f} s on {THREADS} threads")
    _assert_clean(r, 10_000, 1_000)
    n, ties = _phase_a_sample(g, sc, 32)
    assert n > 1_000_000


def test_keep_all_mode_at_c1_size_through_the_culled_walk(monkeypatch):
    """kNN <= 0 (every accepted match is kept, in ascending target order: line3D.cc:982-992).  Until round 3 this mode
    streamed every pair unculled in both of its passes; since round 4 the COUNT pass takes the epipolar-band walk (a
    count does not depend on the order; the fill pass keeps streaming: its arrival order is the reference's order).  Full
    C1 phase A: 32 sampled directed pairs bit for bit and in the reference's order against matchingCPU of the
    restatement (pinned byte for byte on the reference's own code), with and without the culled count pass
    (L3D_KEEPALL_NO_CULL=1) giving the same slots, the culled form faster; then the whole pipeline in this mode against
    the reference's own code on a quarter-size scene (its scoring is quadratic in the list length)."""
    from line3dpp_amd.api import Line3D
    from line3dpp_amd.scene import make_scene
    sc = make_config("C1")
    out = {}
    for name, env in (("culled", None), ("unculled", "1")):
        if env:
            monkeypatch.setenv("L3D_KEEPALL_NO_CULL", env)
        else:
            monkeypatch.delenv("L3D_KEEPALL_NO_CULL", raising=False)
        g = Line3D()
        g.add_scene(sc)
        assert g.setTimingLevel(2)                             # cull_prepare_ms is a profiling-level time
        n_pairs = None
        for rep in range(2):                                   # second call: warm pools, the time that counts
            assert g.matchBegin(kNN=0)
            n_pairs = len(g.pairs()[0])
            assert g.matchPairs(0, n_pairs)
            tm = g.timings()
            if rep == 0:
                assert g.matchAbort()
        # (GPU time of the two kernels -- count and fill -- and of the culling set-up; match_pairs_ms also spans the host's
        # sizing of the rows between the two passes)
        out[name] = (g, tm["match_kernel_ms"] + tm["cull_prepare_ms"], tm["culled_pairs"])
    (gc, ms_c, culled), (gu, ms_u, unculled) = out["culled"], out["unculled"]
    assert culled == n_pairs and unculled == 0            # (the count pass takes the culled walk, the fill pass streams)
    rng = np.random.default_rng(5)
    sample = rng.choice(n_pairs, 32, replace=False)
    for pi in sample:
        a, b = gc.pair_slots(int(pi)), gu.pair_slots(int(pi))
        assert a.shape == b.shape and np.array_equal(a["tgt_seg"], b["tgt_seg"]) and np.array_equal(a["overlap"], b["overlap"])
        assert np.array_equal(a["d_p1"], b["d_p1"]) and np.array_equal(a["d_q2"], b["d_q2"])
    from oracle import oracle as O
    o = O.Oracle(threads=THREADS)
    o.add_scene(sc)
    o.begin_match(kNN=0)
    pairs, _ = gc.pairs()
    total = 0
    for pi in sample:
        om, _ = o.match_pair(int(pairs[pi][0]), int(pairs[pi][1]))
        r = H.compare_pair_fast(gc.pair_slots(int(pi)), om)
        assert r["set_diff"] == 0 and r["inexact_fields"] == 0 and r["order_rows"] == 0, (int(pi), r)
        total += r["n_cpu"]
    o.end_match()
    assert total > 500_000
    print(f"keep-all C1 phase A: culled count pass {ms_c:.2f} ms, both passes unculled {ms_u:.2f} ms ({ms_u / ms_c:.2f} x), {total} matches in 32 sampled pairs identical")
    assert ms_u > 1.1 * ms_c, (ms_c, ms_u)
    gc.matchAbort(); gu.matchAbort(); gc.close(); gu.close()
    # the whole pipeline in this mode, against the reference's own code
    monkeypatch.delenv("L3D_KEEPALL_NO_CULL", raising=False)
    sq = make_scene(16, 1000, n_neighbors=6, seed=23)
    g = Line3D()
    g.add_scene(sq)
    assert g.matchImages(kNN=0) and g.computeAffinity()
    assert O.have_reference()
    oq = O.Oracle(threads=THREADS, reference=True)
    oq.add_scene(sq)
    oq.match_images(kNN=0); oq.compute_affinity()
    r = H.full_result_diff(g, oq, sq)
    assert r["ok"] and r["set_diff"] == 0 and r["order_rows"] == 0 and r["inexact_phase_a_fields"] == 0, r
    assert r["surviving"] > 10_000
