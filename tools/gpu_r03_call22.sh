#!/bin/bash
# round-3 GPU call 22: which runtime call makes the first l3d_match_begin of a fresh process slow on C0 (HIP API trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
cat > /tmp/first_call.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[2])
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
sc = make_config(sys.argv[1])
g = Line3D(); g.add_scene(sc)
ta=time.time(); ok = g.matchImages() and g.computeAffinity(); tb=time.time()
print("first call %.2f ms" % ((tb-ta)*1e3), flush=True)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --output-format csv -d $O/hip -o p -- python /tmp/first_call.py C0 $R > $O/hip.log 2>&1
grep "first call" $O/hip.log
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/hip/**/p_hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# the calls of the last 15 ms of the process's HIP activity that took longer than 0.2 ms
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 200 and t_end - int(r["Start_Timestamp"]) < 40e6:
        print("%-32s %9.1f us   at -%.2f ms" % (r["Function"], d, (t_end - int(r["Start_Timestamp"])) / 1e6))
PY
