import sys, time, os
sys.path.insert(0, sys.argv[2])
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
sc = make_config(sys.argv[1])
g = Line3D(); g.add_scene(sc)
ta = time.time(); ok = g.matchImages() and g.computeAffinity(); tb = time.time()
print("%s first call %.2f ms, begin %.3f" % (sys.argv[1], (tb - ta) * 1e3, g.timings()["begin_ms"]), flush=True)
for name in ("second", "third", "fourth"):
    ta = time.time(); ok = g.matchImages() and g.computeAffinity(); tb = time.time()
    print("%s %s call %.2f ms" % (sys.argv[1], name, (tb - ta) * 1e3), flush=True)
