"""Pins the oracle restatement (oracle/l3d_oracle.cpp) against THE REFERENCE'S OWN CODE: oracle/_ref is
line3D.cc / view.cc / clustering.cc compiled in place from /root/reference against thin shim headers
(oracle/ref_shim: an Eigen subset with real arithmetic, Boost/OpenCV stand-ins for code the
explicit-segments path never reaches).  Byte-for-byte equality of matches_, estimated_position3D_, A_ and
the per-view scalars.  Skipped where oracle/_ref has not been built (it needs /root/reference)."""
import numpy as np
import pytest

from line3dpp_amd.scene import make_scene
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.have_reference(), reason="oracle/_ref not built (needs /root/reference)")


def run(scene, reference, threads=1, **params):
    o = O.Oracle(threads=threads, reference=reference)
    o.add_scene(scene)
    o.match_images(**params)
    o.compute_affinity()
    return o


def assert_identical(r, o, scene, ordered=True):
    n = 0
    for v in scene.views:
        rm, roff = r.matches(v.cam); om, ooff = o.matches(v.cam)
        assert np.array_equal(roff, ooff)
        assert rm.tobytes() == om.tobytes(), f"matches_ of view {v.cam}"
        ri, oi = r.view_info(v.cam), o.view_info(v.cam)
        assert ri["k"] == oi["k"] and ri["median_depth"] == oi["median_depth"]
        assert np.array_equal(ri["C"], oi["C"]) and np.array_equal(ri["t"], oi["t"]), "untranslate() restores the views"
        n += len(om)
    for a, b in zip(r.best(), o.best()):
        assert a.tobytes() == b.tobytes(), "estimated_position3D_"
    assert np.array_equal(r.translation(), o.translation())
    assert r.med_scene_depth_lines() == o.med_scene_depth_lines()
    re_, rl = r.affinity(); oe, ol = o.affinity()
    if ordered:   # single-threaded: A_ and the first-touch row ids are deterministic
        assert re_.tobytes() == oe.tobytes() and rl.tobytes() == ol.tobytes(), "A_ / local2global_"
    else:
        from tests.helpers import affinity_map
        assert affinity_map(re_, rl) == affinity_map(oe, ol)
    assert r.pair_tests() == o.pair_tests() == scene.pair_tests()[0]
    return n, len(oe)


@pytest.mark.parametrize("n_views,n_segs,nn,seed", [(8, 300, 4, 1), (12, 500, 6, 2), (6, 129, 2, 3), (10, 700, 8, 4)])
def test_restatement_equals_reference_code(n_views, n_segs, nn, seed):
    sc = make_scene(n_views, n_segs, n_neighbors=nn, seed=seed)
    n, ne = assert_identical(run(sc, True), run(sc, False), sc)
    assert n > 0


@pytest.mark.parametrize("params", [dict(kNN=0), dict(kNN=3, epi_overlap=0.5), dict(sigma_p=-0.05),
                                    dict(sigma_p=-0.05, const_reg_depth=20.0), dict(sigma_a=5.0, sigma_p=1.0),
                                    dict(epi_overlap=-1.7, sigma_a=-200.0, num_neighbors=0)])
def test_parameter_modes_equal_reference_code(params):
    sc = make_scene(7, 260, n_neighbors=4, seed=11)
    assert_identical(run(sc, True, **params), run(sc, False, **params), sc)


def test_release_build_of_the_reference_gives_identical_results():
    """oracle/_ref/libl3d_ref_release.so = the same reference sources at -O3 -DNDEBUG (the reference's own Release
    configuration, CMakeLists.txt:3), which bench.py times as cpu_baseline: byte-identical to the -O2 build."""
    if not O.have_release():
        pytest.skip("release build of oracle/_ref not present")
    sc = make_scene(10, 500, n_neighbors=6, seed=17)
    assert_identical(run(sc, "release"), run(sc, True), sc)
    assert_identical(run(sc, "release", threads=4), run(sc, True), sc, ordered=False)


def test_asymmetric_neighbours_equal_reference_code():
    sc = make_scene(7, 220, n_neighbors=4, seed=19)
    remap = {i: 10 + 7 * i for i in range(7)}
    for v in sc.views:
        v.neighbors = [remap[n] for n in v.neighbors if (v.cam + n) % 3 != 0] or [remap[(v.cam + 1) % 7]]
        v.cam = remap[v.cam]
    assert_identical(run(sc, True), run(sc, False), sc)


def test_reference_openmp_path_equals_restatement_as_sets():
    """the reference's OpenMP path (what bench.py times as cpu_baseline kind "reference"): entry order of
    estimated_position3D_ / A_ is thread-timing dependent there, the content is not"""
    sc = make_scene(8, 400, n_neighbors=4, seed=5)
    assert_identical(run(sc, True, threads=4), run(sc, False, threads=1), sc, ordered=False)


def test_second_match_images_call_equals_reference_code():
    sc = make_scene(6, 200, n_neighbors=4, seed=7)
    r = O.Oracle(threads=1, reference=True); o = O.Oracle(threads=1)
    for x in (r, o):
        x.add_scene(sc); x.match_images(); x.match_images(kNN=5); x.compute_affinity()
    assert_identical(r, o, sc)


@pytest.mark.parametrize("collin_t", [2.0, 6.0])
def test_collinearity_links_equal_reference_code(collin_t):
    """collinearity_t > 0: View::findCollinCPU lists and the extra affinity links of
    computingAffinityMatrix (line3D.cc:1904-1974), restatement vs the reference's own code, byte for byte."""
    from tests.helpers import split_scene
    sc = split_scene(make_scene(8, 400, n_neighbors=4, seed=9))
    out = []
    for reference in (True, False):
        o = O.Oracle(threads=1, reference=reference)
        o.add_scene(sc)
        o.match_images()
        o.set_collinearity(collin_t)
        o.compute_affinity()
        out.append(o)
    r, o = out
    total = 0
    for v in sc.views:
        ro, ri = r.collinear(v.cam, len(v.segs)); oo, oi = o.collinear(v.cam, len(v.segs))
        assert np.array_equal(ro, oo) and np.array_equal(ri, oi), f"collinear lists of view {v.cam}"
        total += len(oi)
    assert total > 1000
    re_, rl = r.affinity(); oe, ol = o.affinity()
    assert re_.tobytes() == oe.tobytes() and rl.tobytes() == ol.tobytes(), "A_ / local2global_ with collinear links"
    # the links add edges
    plain = run(sc, False)
    assert len(oe) > len(plain.affinity()[0]) + 200


def test_real_testdata_scene_equals_reference_code():
    """BASELINE config C0: the reference's bundled testdata (26 images) with the cameras recovered from the
    reference's own result fixture (tests/golden/make_real_scene.py) -- real LSD segments, real geometry.
    Restatement vs the reference's own code, byte for byte; the reconstruction re-finds the fixture's lines."""
    from line3dpp_amd.scene import make_config, C0_FILE
    sc = make_config("C0")
    r, o = run(sc, True, threads=8), run(sc, False, threads=8)
    n, ne = assert_identical(r, o, sc, ordered=False)    # 8 threads: A_ compared as a map
    assert n > 50000 and ne > 20000
    r1 = run(sc, True)                                    # single thread: the reference's deterministic order
    r1.reconstruct(3)
    lines = r1.lines()
    d = np.load(C0_FILE)
    mine = [frozenset(map(tuple, np.asarray(L["residuals"]).reshape(-1, 2).tolist())) for L in lines]
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
    # the inputs are only the segments that survived in the fixture and the cameras are estimates, still most of the
    # reference's published 3D lines come back
    assert len(lines) > 1500 and found > 0.6 * (len(off) - 1), (len(lines), found)


@pytest.mark.skipif(not O.have_cuda_path(), reason="oracle/_ref/libl3d_ref_cuda.so not built (needs /root/reference)")
def test_rdd_restatement_equals_the_reference_cuda_path_run_as_host_code():
    """the replicator-dynamics diffusion has no CPU path in the reference.  oracle/Makefile compiles its
    CUDA path in place -- line3D.cc (performRDD, :2026-2076), sparsematrix.cc, cudawrapper.cu (K_sparseMat_* :432-544,
    replicator_dynamics_diffusion_GPU :708-766) -- against a host stand-in for the CUDA runtime (oracle/ref_shim_cuda:
    device memory = host memory, a launch = a loop over the grid) and the restatement lo_rdd, which the product's
    k_rdd.hip used to be checked against, must reproduce it byte for byte: symmetric patterns (A_ always holds (i,j) and
    (j,i); with a one-sided entry the reference's kernels index row -1, undefined behaviour that cannot be pinned), rows
    without entries, duplicate-free random patterns, asymmetric weights, weights down to the L3D_EPS_GPU clamp."""
    rng = np.random.default_rng(11)
    total = 0
    for trial, (n, m, sym) in enumerate([(2, 4, True), (3, 6, True), (40, 120, True), (300, 1500, True), (300, 1500, False),
                                         (1000, 6000, True), (64, 2000, True)]):
        pairs = {(int(a), int(b)) for a, b in rng.integers(0, n, (m, 2)) if a != b}
        pairs |= {(b, a) for a, b in pairs}
        if not pairs:
            continue
        w = {}
        for a, b in sorted(pairs):     # sym: w(i,j) == w(j,i) as computingAffinityMatrix emits them; else independent weights
            w[(a, b)] = w.get((b, a), np.float32(rng.uniform(1e-6 if trial % 2 else 0.5, 1.0))) if sym else np.float32(rng.uniform(0.05, 1.0))
        e = np.array([(a, b, w[(a, b)]) for a, b in sorted(pairs)], dtype=O.CLEDGE_DTYPE)
        e = e[rng.permutation(len(e))]
        ref = O.rdd_reference(e, n)
        port = O.Oracle.rdd(e, n)
        assert ref.tobytes() == port.tobytes(), (trial, n, len(e))
        total += len(ref)
    assert total > 10_000
