"""CPU checks of the vectorised whole-scene comparison used by the BASELINE-size GPU parity tests and by bench.py's
parity block (tests/helpers.py: full_result_diff, ring_slice): restatement against the reference's own code must
come out clean, and a planted difference must be reported."""
import numpy as np
import pytest

from line3dpp_amd._lib import SEGMENT2D_DTYPE, SEGMENT3D_DTYPE
from line3dpp_amd.scene import make_scene
from oracle import oracle as O
from tests import helpers as H


class AsContext:
    """presents an Oracle through the accessors of line3dpp_amd.Line3D that full_result_diff reads"""

    def __init__(self, o):
        self.o = o

    def matches(self, cam):
        return self.o.matches(cam)

    def view_info(self, cam):
        return self.o.view_info(cam)

    def best(self):
        cs, geo, ln, m = self.o.best()
        s2 = np.zeros(len(cs), SEGMENT2D_DTYPE); s2["cam"] = cs[:, 0]; s2["seg"] = cs[:, 1]
        s3 = np.zeros(len(cs), SEGMENT3D_DTYPE)
        s3["P1"] = geo[:, 0:3]; s3["P2"] = geo[:, 3:6]; s3["dir"] = geo[:, 6:9]
        return s2, s3, m

    def affinity(self):
        e, l = self.o.affinity()
        l2 = np.zeros(len(l), SEGMENT2D_DTYPE); l2["cam"] = l[:, 0]; l2["seg"] = l[:, 1]
        return e, l2, self.o.med_scene_depth_lines()


def _run(sc, reference, threads):
    o = O.Oracle(threads=threads, reference=reference)
    o.add_scene(sc); o.match_images(); o.compute_affinity()
    return o


@pytest.mark.skipif(not O.have_reference(), reason="oracle/_ref not built")
def test_full_result_diff_clean_and_sensitive():
    sc = make_scene(8, 300, n_neighbors=4, seed=5)
    port, ref = _run(sc, False, 1), _run(sc, True, 4)     # multi-threaded reference: maps, not global order
    r = H.full_result_diff(AsContext(port), ref, sc)
    assert r["ok"] and r["surviving"] > 300 and r["best"] > 100 and r["affinity_entries"] > 50, r
    assert r["set_diff"] == r["order_rows"] == r["best_set_diff"] == r["affinity_set_diff"] == 0

    class Tampered(AsContext):
        def matches(self, cam):
            m, off = self.o.matches(cam)
            if cam == sc.views[2].cam and len(m) > 3:
                m = m.copy(); m["tgt_seg"][3] += 1; m["score3D"][0] *= 1.001
            return m, off
    r2 = H.full_result_diff(Tampered(port), ref, sc)
    assert not r2["ok"] and r2["set_diff"] == 2 and r2["max_rel_score3D"] > 5e-4


def test_ring_slice_keeps_full_neighbour_sets_in_the_middle():
    sc = make_scene(20, 50, n_neighbors=6, seed=3)
    sl = H.ring_slice(sc, 5, 9)
    assert [v.cam for v in sl.views] == list(range(5, 14))
    mid = [v for v in sl.views if v.cam in (8, 9, 10)]
    assert all(len(v.neighbors) == 6 for v in mid)
    assert all(set(v.neighbors) <= set(range(5, 14)) for v in sl.views)
    assert len(sc.views[5].neighbors) == 6          # the source scene is untouched
