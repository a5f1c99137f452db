"""GPU parity on the FULL BASELINE configurations C1, C2 (256 views x 4096 segments x 20 neighbours: 4.3e10 pair tests)
and C4 (128 x 16384 x 10: 1.7e11) against stored runs of the reference's own code (oracle/_ref, Release build).

Running the reference on C2 / C4 costs minutes of CPU each, so its result is stored once per reference build by
tools/ref_digest.py as a RESULT RECORD (tests/full_digest.py): SHA-256 digests of everything that must be identical --
per view the surviving match lists in order with overlap and depths bit for bit, the best-hypothesis keys and choices,
the pairs of A_ -- committed under tests/golden/full/, and the float fields (score3D, 3D end points, affinity weights,
median depths) as arrays beside oracle/_ref (git-ignored like it, shipped with the snapshot).  The HIP result of the
full scene is reduced to the same record and compared.  A missing record FAILS the test, and so do missing floats: when
the full arrays are not there the committed strided sample of them (tests/golden/full/<config>_floats_sample.npz) is
compared instead, and with neither the test fails -- the float half of the check cannot silently drop out.

The C2 run also goes through the pool-regrow path of phase B at full size: L3D_POOL_SCALE shrinks the initial record
pools so that the first list passes overflow and are repeated with larger ones (l3d_timings.pool_retries > 0), with the
same result required.
"""
import os

import pytest

from line3dpp_amd.scene import make_config
from tests import full_digest as FD
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(config, env=None):
    from line3dpp_amd.api import Line3D
    meta, floats = FD.load_reference(config)
    assert meta is not None, f"tests/golden/full/{config}.json is missing: run tools/ref_digest.py {config}"
    assert floats is not None, (f"neither oracle/_ref/cache/full_{config}.npz nor tests/golden/full/{config}_floats_sample.npz "
                                f"matches the stored record: the float half of the check would be skipped")
    sc = make_config(config)
    assert FD.scene_hash(sc) == meta["scene_sha256"], "the stored reference record is of another scene"
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        g = Line3D()
        g.add_scene(sc)
        assert g.matchImages() and g.computeAffinity()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    assert g.pair_tests() == meta["pair_tests"]
    tm = g.timings()
    exact, fl = FD.result_record(g, sc, False)
    r = FD.compare(exact, fl, meta["exact"], floats, H.REL_TOL)
    print(config, "full", {k: r[k] for k in ("surviving", "best", "affinity_entries", "exact_ok", "floats_checked", "max_rel")},
          "pool_retries", tm["pool_retries"], "reference:", meta["seconds"], "s on", meta["threads"], "threads")
    assert not r["differing_views"], f"surviving lists differ from the reference in views {r['differing_views'][:8]}"
    assert r["counts_equal"] and r["best_equal"] and r["affinity_pairs_equal"], r
    assert r["floats_checked"] in ("full", "sample") and r["max_rel"] is not None and r["max_rel"] <= H.REL_TOL, r
    g.close()
    return r, tm


def test_full_c1_against_the_stored_reference_record():
    r, _ = _run("C1")
    assert r["surviving"] == 138514


def test_full_c2_with_pool_regrowth_against_the_stored_reference_record():
    r, tm = _run("C2", env={"L3D_POOL_SCALE": "0.02"})
    assert tm["pool_retries"] >= 1, "the shrunken pools were meant to overflow at least once"
    assert r["surviving"] > 1_000_000


def test_full_c4_against_the_stored_reference_record():
    r, _ = _run("C4")
    assert r["surviving"] > 1_000_000


def test_full_c1_against_the_committed_float_sample_alone(monkeypatch):
    """the compact form on its own: what the check falls back to where oracle/_ref/cache did not travel"""
    monkeypatch.setenv("L3D_FLOATS_SAMPLE_ONLY", "1")
    r, _ = _run("C1")
    assert r["floats_checked"] == "sample" and r["max_rel"] <= H.REL_TOL
