"""A fixed, seeded handful of the randomised stress cases of tests/stress/ inside pytest (the scripts there run hundreds on
the GPU box; their tallies are quoted in DESIGN.md):

  * full pipeline (matchImages + affinity) against the reference's own code on random geometries and parameters: metric
    regulariser, keep-all kNN, ragged views, asymmetric neighbour lists
  * phase A with epipolar-band culling + fp32 pre-filter against the brute-force path, slot for slot
"""
import numpy as np
import pytest

from line3dpp_amd._lib import EMPTY
from oracle import oracle as O
from tests import helpers as H
from tests.stress.cases import culling_case, pipeline_case

pytestmark = pytest.mark.gpu

_O2G = dict(sigma_p="sigma_position", sigma_a="sigma_angle", epi_overlap="epipolar_overlap", kNN="kNN")


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_random_scene_through_the_whole_pipeline_equals_the_reference(seed):
    from line3dpp_amd.api import Line3D
    assert O.have_reference(), "oracle/_ref is missing: this test compares with the reference's own code only"
    sc, kw = pipeline_case(np.random.default_rng(seed), max_views=10, max_segs=400)
    g = Line3D(); g.add_scene(sc)
    assert g.matchImages(**{_O2G[k]: v for k, v in kw.items()}) and g.computeAffinity()
    o = O.Oracle(threads=4, reference=True); o.add_scene(sc)
    o.match_images(**kw); o.compute_affinity()
    r = H.full_result_diff(g, o, sc)
    assert r["ok"], (kw, r)
    g.close()


@pytest.mark.parametrize("seed", [31, 32, 33, 34, 35, 36, 37, 38])
def test_random_scene_culled_equals_brute_force_slot_for_slot(seed):
    from line3dpp_amd.api import Line3D
    sc, knn, epi = culling_case(np.random.default_rng(seed), max_views=9, max_segs=1800)
    out = []
    for brute in (0, 1):
        g = Line3D(); g.add_scene(sc); g.set_brute_force(brute)
        n_pairs = len(g.pairs()[0]) if g.matchBegin(kNN=knn, epipolar_overlap=epi) else 0
        assert n_pairs and g.matchPairs(0, n_pairs)
        out.append([g.pair_slots(pi) for pi in range(n_pairs)])
        g.matchAbort(); g.close()
    n_matches = 0
    for pi, (a, b) in enumerate(zip(*out)):
        assert np.array_equal(a, b), (seed, pi)
        n_matches += int((b["tgt_seg"] != EMPTY).sum())
    assert n_matches > 0
