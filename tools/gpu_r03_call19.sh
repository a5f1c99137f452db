#!/bin/bash
# round-3 GPU call 19: where the first call of a process spends its time (host-side trace of the smoke scene, twice)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
L3D_TRACE=1 python - > $O/trace.log 2>&1 <<'PY'
import time
t0=time.time()
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_scene
sc = make_scene(8, 300, n_neighbors=4, seed=1)
t1=time.time()
g = Line3D(); g.add_scene(sc)
t2=time.time()
print("create+add %.1f ms" % ((t2-t1)*1e3), flush=True)
for k in range(3):
    ta=time.time(); ok = g.matchImages() and g.computeAffinity(); tb=time.time()
    print("call %d: %.2f ms wall, timings %s" % (k, (tb-ta)*1e3, {a: round(b,3) for a,b in g.timings().items() if a.endswith('_ms')}), flush=True)
g2 = Line3D(); g2.add_scene(sc)
ta=time.time(); g2.matchImages(); g2.computeAffinity(); tb=time.time()
print("second context, first call: %.2f ms wall" % ((tb-ta)*1e3), flush=True)
PY
tail -80 $O/trace.log | cut -c1-220
