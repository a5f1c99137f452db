"""GPU tests (pytest -m gpu) of the parameter modes and the failure paths of the context layer THROUGH the HIP
library's own clamp / state code (l3d_match_begin, l3d_api.hip), against the reference's own code (oracle/_ref):

  * clamped arguments (epipolar_overlap = -1.7, sigma_angle = -200, num_neighbors = 0), line3D.cc:394-413
  * fixed 3D regulariser (sigma_p < 0) with and without const_regularization_depth, line3D.cc:426-433, view.h:124-127
  * a second matchImages with different parameters on the same context
  * failing calls (kNN beyond the build's limit, an abandoned matchBegin) leave the views untranslated and the context
    usable: the next matchImages gives the results of a fresh context
  * kNN beyond the per-row top-K tables in LDS (kNN = 1000): every row through the exact replay (k_match_tied_rows)
"""
import numpy as np
import pytest

from line3dpp_amd.scene import make_scene
from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _gpu(scene):
    from line3dpp_amd.api import Line3D
    g = Line3D()
    g.add_scene(scene)
    return g


def _ref(scene, calls):
    assert O.have_reference(), "oracle/_ref is missing: these tests compare with the reference's own code only"
    o = O.Oracle(threads=1, reference=True)
    o.add_scene(scene)
    for kw in calls:
        o.match_images(**kw)
    o.compute_affinity()
    return o


_O2G = dict(sigma_p="sigma_position", sigma_a="sigma_angle", num_neighbors="num_neighbors", epi_overlap="epipolar_overlap",
            kNN="kNN", const_reg_depth="const_regularization_depth")


def _g(kw):
    return {_O2G[k]: v for k, v in kw.items()}


def _assert_same(g, o, sc):
    r = H.full_result_diff(g, o, sc)
    assert r["ok"] and r["order_rows"] == 0, r
    # single-threaded reference: A_ ids and order, estimated_position3D_ order are deterministic -> identical
    ge, gl, _ = g.affinity(); oe, ol = o.affinity()
    assert np.array_equal(ge["i"], oe["i"]) and np.array_equal(ge["j"], oe["j"])
    assert np.array_equal(np.stack([gl["cam"], gl["seg"]], 1), ol)
    assert np.allclose(g.translation(), o.translation(), rtol=0, atol=0)
    return r


@pytest.mark.parametrize("params", [dict(epi_overlap=-1.7, sigma_a=-200.0, num_neighbors=0),
                                    dict(sigma_p=-0.05), dict(sigma_p=-0.05, const_reg_depth=20.0),
                                    dict(sigma_a=5.0, sigma_p=1.0), dict(kNN=3, epi_overlap=0.5),
                                    dict(sigma_p=0.01)])
def test_parameter_modes_through_the_hip_path(params):
    sc = make_scene(7, 260, n_neighbors=4, seed=11)
    g = _gpu(sc)
    assert g.matchImages(**_g(params)) and g.computeAffinity()
    r = _assert_same(g, _ref(sc, [params]), sc)
    assert r["surviving"] > 0 or params.get("epi_overlap", 0) < 0 or params.get("sigma_p") == 0.01   # overlap > 0.99 /
    # a 0.1 px regulariser leave next to nothing


def test_second_match_images_with_other_parameters():
    sc = make_scene(6, 200, n_neighbors=4, seed=7)
    g = _gpu(sc)
    calls = [dict(), dict(kNN=5, sigma_a=7.0, epi_overlap=0.4)]
    for kw in calls:
        assert g.matchImages(**_g(kw))
    assert g.computeAffinity()
    _assert_same(g, _ref(sc, calls), sc)


def test_failed_and_abandoned_calls_leave_a_clean_context():
    sc = make_scene(6, 240, n_neighbors=4, seed=23)
    fresh = _gpu(sc)
    assert fresh.matchImages() and fresh.computeAffinity()
    g = _gpu(sc)
    # (1) kNN beyond the limit of this build (4096): refused, nothing moved
    assert not g.matchImages(kNN=5000) and g.last_status == -9
    from line3dpp_amd import _lib
    assert "kNN" in _lib.last_error()
    # (2) a begin that is never finished, then another begin on top of it
    assert g.matchBegin()
    assert g.matchBegin(kNN=4)
    assert g.matchAbort() and g.matchAbort()          # idempotent
    assert not g.matchPairs(0, 1) and g.last_status == -7
    # (3) after all that the context behaves like a fresh one -- a view left translated would shift its camera
    # centre, the next translation would come out near zero and every 3D end point would move
    assert g.matchImages() and g.computeAffinity()
    assert np.array_equal(g.translation(), fresh.translation())
    a, b = g.best(), fresh.best()
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    ea, la, _ = g.affinity(); eb, lb, _ = fresh.affinity()
    assert ea.tobytes() == eb.tobytes() and la.tobytes() == lb.tobytes()
    assert g.reconstruct3Dlines() and fresh.reconstruct3Dlines()
    for x, y in zip(g.get3Dlines(), fresh.get3Dlines()):
        assert x["collinear3Dsegments"].tobytes() == y["collinear3Dsegments"].tobytes()


@pytest.mark.parametrize("kNN", [2000, 450])
def test_knn_beyond_the_lds_tables_equals_the_reference(kNN):
    """Line3D::matchingCPU accepts any kNN (line3D.cc:982-1007, commons.h:217-231).  The per-row top-K tables of
    k_match_pairs live in LDS (about 420 entries per row in the row form of the kernel, about 1 600 / 800 with 16 / 32 rows
    per work item in the tile form); beyond that every row takes the exact replay path (k_match_tied_rows: all accepted
    matches through the reference's heap, kNN pops).  With kNN = 2000 every form replays and nearly every row keeps ALL its
    matches, in the pop order of the reference's priority_queue; kNN = 450 replays in the row form only.  The threshold is
    lowered so that rows hold hundreds of matches."""
    from line3dpp_amd import _lib
    replays_before = _lib.load().l3d_debug_counter(b"knn_replay_calls")
    sc = make_scene(6, 700, n_neighbors=4, seed=41)
    g = _gpu(sc)
    kw = dict(kNN=kNN, epi_overlap=0.05)
    assert g.matchImages(**_g(kw)) and g.computeAffinity()
    o = _ref(sc, [kw])
    r = _assert_same(g, o, sc)
    assert r["surviving"] > 1000
    # phase A itself: the slots of a pair row by row against the reference's kept matches would need its scored lists;
    # what the reference keeps per row is visible in the surviving lists compared above.  Rows longer than the LDS limit
    # must exist for the test to mean anything:
    slots = g.pair_slots(0)
    from line3dpp_amd._lib import EMPTY
    longest = int((slots["tgt_seg"] != EMPTY).sum(1).max())
    assert longest > 100, longest
    replayed = _lib.load().l3d_debug_counter(b"knn_replay_calls") > replays_before
    assert replayed or kNN < 2000
    if replayed:
        assert g.timings()["tied_rows"] >= 700      # every row of every pair went through the replay


def test_repeated_calls_and_a_growing_scene_see_what_the_reference_sees():
    """l3d_match_begin keeps its pair list / fundamental matrices / culling set-up, and the device keeps its tables,
    while views, neighbour sets and kNN are those of the previous call (l3d_api.hip: begin_sig, l3d_host.h: upload_table).
    Whatever changes in between must be seen: (1) the same call three times, (2) two more views added to the same context,
    (3) another kNN, (4) the first parameters again -- after every step the context equals the reference's own code
    driven through the same sequence (its visual_neighbors_ persist across calls exactly like this context's)."""
    sc = make_scene(8, 220, n_neighbors=4, seed=31)
    first, rest = sc.views[:6], sc.views[6:]
    from line3dpp_amd.api import Line3D
    g = Line3D()
    assert O.have_reference()
    o = O.Oracle(threads=1, reference=True)

    def add(views):
        for v in views:
            g.addImage(v.cam, (v.width, v.height), v.K, v.R, v.t, v.median_depth, v.neighbors, v.segs)
            assert o.add_view(v.cam, v.segs, v.K, v.R, v.t, v.width, v.height, v.median_depth, v.neighbors) == 0

    class _Part:                 # what full_result_diff walks: the views present so far
        def __init__(self, views): self.views = views

    def step(kw, views):
        assert g.matchImages(**_g(kw)) and g.computeAffinity()
        o.match_images(**kw); o.compute_affinity()
        return _assert_same(g, o, _Part(views))

    add(first)
    r0 = step(dict(), first)
    for _ in range(2):           # (1) nothing changed: the kept lists and tables are used
        r = step(dict(), first)
        assert r["surviving"] == r0["surviving"]
    add(rest)                    # (2) the scene grows
    r1 = step(dict(), sc.views)
    assert r1["surviving"] > r0["surviving"]
    step(dict(kNN=4), sc.views)  # (3) other slot layout
    r2 = step(dict(), sc.views)  # (4) back
    assert r2["surviving"] == r1["surviving"]


def test_unscaled_division_and_sqrt_equal_the_compilers_expansions_on_the_device():
    """l3d_dev.h: the exact tests of a pair whose operands the host has range-checked (kPairFastMath) divide and take
    square roots without the operand scaling of the compiler's IEEE expansions (59 -> 29-46 and 86 -> 66 issue cycles,
    tools/valu_calib.hip).  On 4e8 operand sets across and beyond that range -- exact zeros, denominators at L3D_EPS,
    quotients next to 1, perfect squares -- every result must have the bits of `a / b` and `sqrt(x)` as hipcc compiles
    them for the same device."""
    import ctypes as C
    from line3dpp_amd import _lib
    L = _lib.load()
    counts = (C.c_uint64 * 3)()
    assert L.l3d_selftest_arith(0, 100_000_000, 20260925, counts) == 0
    assert list(counts) == [0, 0, 0], list(counts)


def test_cpp_rccl_driver_single_rank_equals_the_python_front_end(tmp_path):
    """tests/cpp/rccl_driver.cpp (C-ABI + RCCL, the halo sequence) with a one-rank communicator on this box's GPU: the
    plan, the view-sharded list pass, the in-place all-gather of the record slabs and the finish on the records alone
    give what l3d_match_images gives."""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rccl_driver")
    lib_dir = os.path.join(root, "line3dpp_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-w", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "rccl_driver.cpp"), "-o", exe, "-L" + lib_dir, "-ll3dpp_hip",
                           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + lib_dir])
    sc = make_scene(8, 300, n_neighbors=4, seed=1)
    path = str(tmp_path / "scene.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", sc.n_views))
        for v in sc.views:
            f.write(struct.pack("<5I", v.cam, len(v.segs), v.width, v.height, len(v.neighbors)))
            f.write(np.ascontiguousarray(v.K, np.float64).tobytes()); f.write(np.ascontiguousarray(v.R, np.float64).tobytes())
            f.write(np.ascontiguousarray(v.t, np.float64).tobytes()); f.write(struct.pack("<f", v.median_depth))
            f.write(np.asarray(v.neighbors, np.uint32).tobytes()); f.write(np.ascontiguousarray(v.segs, np.float32).tobytes())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, path, "0", "1", str(tmp_path / "nccl_id")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0]
    kv = dict(x.split("=") for x in line.split()[1:])
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    ms = [g.matches(v.cam)[0] for v in sc.views]
    assert int(kv["matches"]) == sum(len(m) for m in ms)
    assert abs(float(kv["score_sum"]) - sum(float(m["score3D"].astype(np.float64).sum()) for m in ms)) < 1e-3
    e, l2g, _ = g.affinity()
    assert int(kv["hypotheses"]) == len(g.best()[0]) and int(kv["edges"]) == len(e) and int(kv["rows"]) == len(l2g)
    assert abs(float(kv["wsum"]) - float(e["w"].astype(np.float64).sum())) < 1e-3


@pytest.mark.parametrize("num_neighbors", [2, 4])
def test_worldpoint_lists_instead_of_neighbour_lists(num_neighbors):
    """An instance constructed with neighbors_by_worldpoints=true (line3D.cc:216-219): addImage takes the view's SfM
    worldpoints, matchImages finds the visual neighbours from the worldpoint overlap (findVisualNeighborsFromWPs,
    line3D.cc:578-699 -> l3d_neighbors.hip, inside l3d_match_begin on the translated views) and matches what it found.
    Everything downstream -- pair list, matches, best hypotheses, affinity -- against the reference's own run in the same
    mode; a second call with another num_neighbors finds the neighbours anew (the reference resets the set each call)."""
    from line3dpp_amd.api import Line3D
    from line3dpp_amd.scene import add_worldpoints
    sc = make_scene(12, 220, n_neighbors=4, seed=21, rings=2)
    add_worldpoints(sc, n_points=2500, seed=5, keep=0.4)
    assert O.have_reference(), "oracle/_ref is missing: this test compares with the reference's own code only"
    o = O.Oracle(threads=1, reference=True, by_worldpoints=True)
    o.add_scene(sc)
    g = Line3D(neighbors_by_worldpoints=True)
    g.add_scene(sc)
    for nn in (num_neighbors, num_neighbors + 2):
        o.match_images(num_neighbors=nn)
        assert g.matchImages(num_neighbors=nn)
        sizes = []
        for v in sc.views:
            ref_nb = o.visual_neighbors(v.cam)
            assert np.array_equal(g.visualNeighbors(v.cam), ref_nb), (v.cam, g.visualNeighbors(v.cam), ref_nb)
            sizes.append(len(ref_nb))
        assert max(sizes) == min(nn, 5)       # (this scene offers five cameras within pi/2 of a view's axis)
    o.compute_affinity()
    assert g.computeAffinity()
    r = _assert_same(g, o, sc)
    assert r["surviving"] > 0
    # a view without worldpoints is refused like a view without neighbours (line3D.cc:153-166)
    v = sc.views[0]
    g.addImage(999, (v.width, v.height), v.K, v.R, v.t, v.median_depth, [], v.segs)
    assert g.last_status == -4


def test_inverse_hypotheses_sorted_with_global_cursors_equal_the_lds_form(monkeypatch):
    """k_pair_csr (round 4: the inverse hypotheses of a pair counting-sorted by target segment) keeps its cursors in
    LDS; views beyond the LDS capacity (more than 32 768 segments) keep them in the pair's own offset array.  The test
    hook L3D_CSR_GLOBAL=1 sends every pair through that second form: same result as the reference's own code, on a scene
    with ragged views and long lists (more neighbours than usual)."""
    sc = make_scene(9, 420, n_neighbors=6, seed=31)
    from line3dpp_amd import _lib
    L = _lib.load()
    monkeypatch.setenv("L3D_CSR_GLOBAL", "1")
    before = L.l3d_debug_counter(b"csr_global_launches")
    g = _gpu(sc)
    assert g.matchImages() and g.computeAffinity()
    # the hook is read per call (round 4 latched it at the first launch of the process and this test then ran the LDS form):
    # the global-cursor form must actually have been launched
    assert L.l3d_debug_counter(b"csr_global_launches") > before
    r = _assert_same(g, _ref(sc, [{}]), sc)
    assert r["surviving"] > 1000


def test_timing_levels_change_what_is_timed_and_nothing_else():
    """l3d_set_timing_level: 2 records every HIP event of a call (all of l3d_timings filled), 1 only the pair around the
    pair-matching kernel, 0 none -- the results are those of the reference at every level, and l3d_get_timings itself
    never waits for the GPU (tied_rows arrives with the call's one read-back, written by the last kernel of the tail)."""
    sc = make_scene(6, 260, n_neighbors=3, seed=41)
    o = _ref(sc, [dict()])
    g = _gpu(sc)
    seen = {}
    for level in (2, 1, 0, 2):
        assert g.setTimingLevel(level)
        assert g.matchImages() and g.computeAffinity()
        _assert_same(g, o, sc)
        seen[level] = g.timings()
    assert seen[2]["match_kernel_ms"] > 0 and seen[2]["finish_ms"] > 0 and seen[2]["begin_ms"] > 0 and seen[2]["affinity_ms"] > 0
    assert seen[2]["lists_ms"] > 0 and seen[2]["match_pairs_ms"] >= seen[2]["match_kernel_ms"]
    assert seen[1]["match_kernel_ms"] > 0 and seen[1]["match_kernel_launches"] == 1
    assert seen[1]["finish_ms"] == 0 and seen[1]["begin_ms"] == 0 and seen[1]["affinity_ms"] == 0 and seen[1]["match_pairs_ms"] == 0
    assert seen[0]["match_kernel_ms"] == 0 and seen[0]["finish_ms"] == 0
    # the counters that travel with the read-back do not depend on the level
    for level in (1, 0):
        for k in ("list_entries", "support_words", "chain_sweeps", "culled_pairs"):
            assert seen[level][k] == seen[2][k], (level, k)
    from line3dpp_amd import _lib
    assert not g.setTimingLevel(3) and "level" in _lib.last_error()


def test_forms_of_the_match_kernel_give_the_same_bytes(monkeypatch):
    """k_match_pairs has three forms for the bounded-kNN launch (k_match.hip): the row form on the padded width-class
    layout of the source rows (the default of round 5), the row form on the legacy layout (L3D_MATCH_CLASSES=0), and the
    tile form with 16 rows per work item and the LDS FIFO of target records (L3D_MATCH_TILE=16; since round 6 what launches
    of up to 2 048 row-form items take by themselves: the last pass below lets the library choose).  The switches are
    read at every l3d_match_begin, so ONE context runs all of them in turn: every slot of phase A, every surviving list,
    best hypothesis and affinity entry must be the same bytes, and equal the reference's own code.  The scene has ragged
    views (classes of very different sizes, views smaller than one work item) and a kNN that leaves rows unfilled."""
    sc = make_scene(9, 1100, n_neighbors=4, seed=97)
    rng = np.random.default_rng(5)
    for i, v in enumerate(sc.views):
        v.segs = v.segs[:max(5, int(len(v.segs) * (0.02 if i == 3 else rng.uniform(0.4, 1.0))))].copy()
    g = _gpu(sc)
    o = _ref(sc, [dict(kNN=7)])
    first = None
    for tile, classes in (("0", "1"), ("0", "0"), ("16", "1"), ("0", "1"), (None, "1")):
        if tile is None:
            monkeypatch.delenv("L3D_MATCH_TILE")
        else:
            monkeypatch.setenv("L3D_MATCH_TILE", tile)
        monkeypatch.setenv("L3D_MATCH_CLASSES", classes)
        assert g.matchImages(kNN=7) and g.computeAffinity()
        assert g.timings()["culled_pairs"] > 0
        _assert_same(g, o, sc)
        n_pairs = len(g.pairs()[0])
        snap = [g.pair_slots(p).tobytes() for p in range(n_pairs)] + [g.matches(v.cam)[0].tobytes() for v in sc.views] + \
               [x.tobytes() for x in g.best()] + [g.affinity()[0].tobytes()]
        if first is None:
            first = snap
        assert snap == first, (tile, classes)
