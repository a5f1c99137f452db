"""Runs the reference's own code (oracle/_ref, Release build) on a FULL BASELINE configuration and stores its result
record (tests/full_digest.py): python tools/ref_digest.py C2 [threads]
Needs oracle/_ref (i.e. /root/reference at build time); minutes of CPU for C2 / C4."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from line3dpp_amd.scene import make_config  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import full_digest as FD  # noqa: E402

cfg = sys.argv[1]
threads = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
sc = make_config(cfg)
kind = "release" if O.have_release() else True
assert O.have_reference(), "oracle/_ref is missing"
o = O.Oracle(threads=threads, reference=kind)
o.add_scene(sc)
tests = sc.pair_tests()[0]
t0 = time.perf_counter()
o.match_images(); o.compute_affinity()
secs = time.perf_counter() - t0
exact, floats = FD.result_record(o, sc, True)
lib = os.path.join(ROOT, "oracle", "_ref", "libl3d_ref_release.so" if kind == "release" else "libl3d_ref.so")
_h = hashlib.md5()
for _f in ("line3D.o", "view.o", "clustering.o"):   # the reference's own translation units (the driver around them may change)
    _h.update(open(os.path.join(ROOT, "oracle", "_ref", "rel" if kind == "release" else "", _f), "rb").read())
objs_md5 = _h.hexdigest()
meta = {"config": cfg, "scene_sha256": FD.scene_hash(sc), "pair_tests": tests,
        "reference_library": os.path.basename(lib), "reference_objects_md5": objs_md5,
        "reference_objects": "oracle/_ref/rel/{line3D,view,clustering}.o (the reference's translation units, oracle/Makefile)",
        "threads": threads, "seconds": round(secs, 1), "M_pair_tests_per_s": round(tests / secs / 1e6, 1),
        "parameters": "matchImages defaults (sigma_p 2.5, sigma_a 10, kNN 10, epipolar_overlap 0.25) + computeAffinity",
        "exact": exact}
FD.store_reference(cfg, meta, floats)
print(cfg, {k: v for k, v in meta.items() if k != "exact"}, {k: exact[k] for k in ("surviving", "best", "affinity_entries")})
