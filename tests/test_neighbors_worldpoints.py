"""Visual neighbours from shared worldpoints (Line3D::findVisualNeighborsFromWPs, line3D.cc:578-699; processWPlist
:230-240) -- line3dpp_amd/csrc/l3d_neighbors.hip behind l3d_neighbors_from_worldpoints / l3d_add_view_worldpoints --
against the reference's OWN code (oracle/_ref: an instance constructed with neighbors_by_worldpoints=true, matchImages
run on the CPU, visual_neighbors_ read back).  Host code: no GPU needed for the context-free entry point."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from line3dpp_amd.scene import add_worldpoints, make_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def _reference_neighbors(scene, num_neighbors):
    assert orc.have_reference(), "oracle/_ref is missing: these tests pin against the reference's own code"
    o = orc.Oracle(reference=True, by_worldpoints=True, threads=4)
    o.add_scene(scene)
    o.match_images(num_neighbors=num_neighbors, kNN=2)
    return {v.cam: o.visual_neighbors(v.cam) for v in scene.views}


def _ours(scene, num_neighbors):
    from line3dpp_amd.api import neighbors_from_worldpoints
    vs = scene.views
    return neighbors_from_worldpoints([v.cam for v in vs], [v.K for v in vs], [v.R for v in vs], [v.t for v in vs],
                                      [v.worldpoints for v in vs], num_neighbors)


def _check(scene, num_neighbors):
    ref = _reference_neighbors(scene, num_neighbors)
    got = _ours(scene, num_neighbors)
    n_sets = 0
    for v in scene.views:
        assert np.array_equal(ref[v.cam], got[v.cam]), (v.cam, ref[v.cam], got[v.cam])
        n_sets += len(ref[v.cam]) > 0
    return ref, n_sets


@pytest.mark.parametrize("num_neighbors", [2, 4, 10, 50])
def test_neighbour_sets_equal_the_reference_on_a_ring(num_neighbors):
    """ring of 24 cameras, 4000 random worldpoints: more candidates than num_neighbors for the small values (the 80 %
    score cut, the re-sort by the distance score, the splice of the full list behind it), fewer for 50"""
    sc = make_scene(24, 40, n_neighbors=4, seed=5)
    add_worldpoints(sc, n_points=4000, seed=11)
    ref, n_sets = _check(sc, num_neighbors)
    assert n_sets == 24 and max(len(x) for x in ref.values()) == min(num_neighbors, max(len(x) for x in ref.values()))
    assert any(len(x) == num_neighbors for x in ref.values()) or num_neighbors == 50


def test_edge_cases_equal_the_reference():
    """few common worldpoints (<= 4: no neighbour), a camera looking the other way (axis angle >= pi/2), two cameras
    closer than the 0.1 baseline, a view whose points nobody shares (empty set), duplicate worldpoint ids in a list,
    camera ids that are not 0..n-1"""
    sc = make_scene(14, 30, n_neighbors=4, seed=9, rings=2)
    add_worldpoints(sc, n_points=2500, seed=3)
    v = sc.views
    for i, view in enumerate(v):
        view.cam = 100 + 7 * i                                           # sparse ids
    v[1].worldpoints = v[0].worldpoints[:4] + [900000 + i for i in range(50)]          # 4 in common with view 0 only
    C2 = -v[2].R.T @ v[2].t
    v[2].R = np.diag([-1.0, 1.0, -1.0]) @ v[2].R; v[2].t = -v[2].R @ C2                # same centre, looks the other way
    C3 = -v[3].R.T @ v[3].t
    v[4].R = v[3].R.copy(); v[4].t = -v[4].R @ (C3 + np.array([0.03, 0.02, 0.0]))      # 0.036 from view 3
    v[4].worldpoints = list(v[3].worldpoints)
    v[5].worldpoints = [800000 + i for i in range(40)]                                 # shares nothing
    v[6].worldpoints = v[6].worldpoints + v[6].worldpoints[:20]                        # duplicates
    for nn in (3, 8):
        ref, _ = _check(sc, nn)
        assert len(ref[v[5].cam]) == 0 and v[0].cam not in ref[v[1].cam]
        assert v[4].cam not in ref[v[3].cam] and v[3].cam not in ref[v[4].cam]
        # the camera that looks away keeps only neighbours whose axes point its way (the far side of the rings)
        assert v[1].cam not in ref[v[2].cam] and v[3].cam not in ref[v[2].cam] and v[2].cam not in ref[v[3].cam]


def test_argument_errors():
    from line3dpp_amd import _lib
    L = _lib.load()
    assert L.l3d_neighbors_from_worldpoints(0, None, None, None, None, None, None, 10, None, None, 0) != 0
    ids = np.array([1, 1], np.uint32); K = np.tile(np.eye(3), (2, 1)).reshape(2, 9); t = np.zeros((2, 3))
    off = np.array([0, 1, 2], np.uint64); w = np.array([0, 0], np.uint32); nb = np.zeros(3, np.uint64)
    rc = L.l3d_neighbors_from_worldpoints(2, _lib.ptr(ids), _lib.ptr(K), _lib.ptr(K), _lib.ptr(t), _lib.ptr(off), _lib.ptr(w), 10,
                                          _lib.ptr(nb), None, 0)
    assert rc == -3 and b"already in use" in L.l3d_last_error()


def test_random_scenes_with_score_ties_equal_the_reference():
    """40 random scenes with few worldpoints per view: equal scores and equal distance scores are common, so the two
    STABLE list sorts of the reference (std::list::sort by score, then by distance score) and the position of the 80 %
    cut decide the sets"""
    rng = np.random.default_rng(2024)
    ties = 0
    for k in range(40):
        nv = int(rng.integers(5, 31))
        sc = make_scene(nv, 24, n_neighbors=2, seed=1000 + k, rings=int(rng.integers(1, 3)) if nv >= 8 else 1)
        add_worldpoints(sc, n_points=int(rng.integers(150, 1500)), seed=k, keep=float(rng.uniform(0.05, 0.5)))
        if k % 5 == 0:                                    # quantised lists: many exact ties
            for v in sc.views:
                v.worldpoints = v.worldpoints[:12 + 6 * (v.cam % 3)]
        if any(len(v.worldpoints) == 0 for v in sc.views):
            for v in sc.views:
                if not v.worldpoints:
                    v.worldpoints = [10 ** 6 + v.cam]     # the reference refuses a view without worldpoints
        nn = int(rng.integers(2, 13))
        ref, _ = _check(sc, nn)
        ties += sum(len(x) for x in ref.values())
    assert ties > 500
