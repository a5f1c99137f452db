"""The call a user makes ONCE per scene, in a warm process: python tools/second_scene_trace.py C3 [n_scenes]
(L3D_TRACE=1 is set here, before the library loads: host-side timeline of every matchImages on stderr.)
Scene A on one context (three calls), closed; then a FRESH context per further scene of the same configuration (other seeds):
first call and second call of each, wall clock of matchImages + affinity."""
import os, sys, time
os.environ["L3D_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config

cfg = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def call(g, tag):
    torch.cuda.synchronize()
    print(f"---- {tag}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    ok = g.matchImages(kNN=10)
    t1 = time.perf_counter()
    ok = ok and g.computeAffinity()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tm = g.timings()
    print(f"{cfg} {tag}: {1e3 * (t2 - t0):.3f} ms (matchImages {1e3 * (t1 - t0):.3f}, affinity {1e3 * (t2 - t1):.3f}) "
          f"pool_retries {tm['pool_retries']} chain_extra_rounds {tm['chain_extra_rounds']} chain_sweeps {tm['chain_sweeps']}", flush=True)
    assert ok


sc = make_config(cfg)
g = Line3D(); g.add_scene(sc)
for k in range(3):
    call(g, f"scene A call {k + 1}")
g.close()
for s in range(n):
    sc2 = make_config(cfg, seed=0x5EED0002 + s)
    t0 = time.perf_counter()
    g = Line3D(); g.add_scene(sc2)
    torch.cuda.synchronize()
    print(f"{cfg} scene {chr(66 + s)}: create + add_views {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    call(g, f"scene {chr(66 + s)} call 1 (fresh context)")
    call(g, f"scene {chr(66 + s)} call 2")
    g.close()
