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


def test_committed_float_samples_belong_to_the_stored_records_and_catch_a_planted_error(monkeypatch):
    """tests/golden/full/<config>_floats_sample.npz: the compact form of the reference's float fields that the full-size
    checks fall back to where oracle/_ref/cache did not travel -- it must match the committed record of its
    configuration, sample every array, and a value off by 1 % at a sampled position must fail the comparison."""
    from tests import full_digest as FD
    monkeypatch.setenv("L3D_FLOATS_SAMPLE_ONLY", "1")
    for cfg in ("C1", "C2", "C4"):
        meta, ref = FD.load_reference(cfg)
        assert meta is not None and ref is not None and "_strides" in ref, cfg
        st = dict(zip(FD.FLOAT_KEYS, (int(x) for x in ref["_strides"])))
        assert len(ref["score3D"]) == -(-meta["exact"]["surviving"] // st["score3D"])
        assert len(ref["best_geo"]) == -(-meta["exact"]["best"] // st["best_geo"])
        # a candidate that IS the reference at the sampled positions (other positions never enter the comparison)
        cand = {k: np.repeat(np.asarray(ref[k]), st[k], axis=0) for k in FD.FLOAT_KEYS}
        r = FD.compare(meta["exact"], cand, meta["exact"], ref, H.REL_TOL)
        assert r["ok"] and r["floats_checked"] == "sample" and r["max_rel"] == 0.0
        bad = dict(cand); a = cand["affinity_w"].copy(); a[3 * st["affinity_w"]] *= 1.01; bad["affinity_w"] = a
        r = FD.compare(meta["exact"], bad, meta["exact"], ref, H.REL_TOL)
        assert not r["ok"] and r["max_rel"] > 5e-3
        # no floats at all: the comparison says so and callers (test_gpu_full_size, bench.py) fail on it
        r = FD.compare(meta["exact"], cand, meta["exact"], None, H.REL_TOL)
        assert r["floats_checked"] is False and r["max_rel"] is None
