import os, sys
sys.path.insert(0, '/root/repo')
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
import time
g = Line3D(); g.add_scene(make_config("C1"))
for i in range(4):
    t=time.perf_counter(); ok = g.matchImages(); t1=time.perf_counter(); ok2 = g.computeAffinity(); t2=time.perf_counter()
    print("step", i, "matchImages %.3f ms affinity %.3f ms"%((t1-t)*1e3,(t2-t1)*1e3), g.timings(), file=sys.stderr)
