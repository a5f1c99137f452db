#!/bin/bash
# round-3 GPU call 21: host-side trace of the first call of a fresh process on C0 and C3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
for c in C0 C3; do
L3D_TRACE=1 python - $c > $O/trace_$c.log 2>&1 <<'PY'
import sys, time
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
sc = make_config(sys.argv[1])
g = Line3D(); g.add_scene(sc)
for k in range(2):
    ta=time.time(); ok = g.matchImages() and g.computeAffinity(); tb=time.time()
    print("call %d: %.2f ms wall, timings %s" % (k, (tb-ta)*1e3, {a: round(b,3) for a,b in g.timings().items() if a.endswith('_ms') or a in ('pool_retries','chain_extra_rounds','chain_sweeps')}), flush=True)
PY
echo "== $c"; grep -v "hyp_\|scan enq\|seg_write\|median" $O/trace_$c.log | head -24 | cut -c1-260
done
